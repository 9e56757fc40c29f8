#!/usr/bin/env python3
"""bench.py -- GICP scan-pairs/s on MI355X (BASELINE.json metric), one process per GPU.

A "step" = one pass of the hot path (lh_gicp_align_batch: NN-index build + covariances-from-normals + 20 outer
GICP iterations with BFGS) over one batch of `--pairs` independent synthetic 100k-point scan pairs per GPU
(BASELINE configs[1] replicated as a queued stream, like configs[3]).  Inputs (xyz + k=20 normals) are resident
in HBM before the timed region.  value = pairs aligned by all ranks / max-over-ranks time.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from locus_amd import capi, synth  # noqa: E402
from locus_amd import dist as ldist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def make_pairs(ctx, n_pairs, rank, rings, az, scale):
    """Synthetic consecutive-scan pairs: distinct seeds per rank and pair.  Normals are
    computed on the GPU with the K3 kernel (k=20), like the NormalComputation nodelet upstream of GICP."""
    S, T, host = [], [], []
    for p in range(n_pairs):
        seed = 1000 * rank + 10 + 2 * p
        src, tgt, delta = synth.scan_pair(n_rings=rings, n_az=az, scale=scale, noise=0.02, seed=seed)
        cs, ct = capi.Cloud(ctx, src), capi.Cloud(ctx, tgt)
        cs.normals_knn(20)
        ct.normals_knn(20)
        ct.drop_index()
        S.append(cs)
        T.append(ct)
        host.append((src, tgt, delta))
    return S, T, host


def cpu_baseline(S, T, host, P, budget_s=20.0):
    """Reference-algorithm restatement (oracle) timed on the host cores: OMP on the two loops the reference
    parallelises (k-NN/NN), serial cost functor (gicp.hpp:291-402).  Bounded sample of the same workload.  The thread
    count matters a lot (the serial functor dominates and idle OMP teams get in its way), so one pair is timed at each of
    1 / 4 (LOCUS default, locus.launch:75-77) / 16 / 64 / all hardware threads and the rest of the budget goes to the
    fastest setting; `value` is that setting's rate, the others are listed in `by_threads`."""
    from oracle import oracle as O
    ncores = os.cpu_count() or 1

    def params(threads):
        return O.default_params(max_iterations=P.max_iterations, max_inner_iterations=P.max_inner_iterations, corr_dist=P.corr_dist,
                                transformation_epsilon=P.transformation_epsilon, rotation_epsilon=P.rotation_epsilon,
                                gicp_epsilon=P.gicp_epsilon, num_threads=threads)

    def host_pair(p):
        a, b = S[p].download(), T[p].download()
        return (O.xyz4(np.stack([a["x"], a["y"], a["z"]], 1)), O.nrm4(np.stack([a["normal_x"], a["normal_y"], a["normal_z"]], 1)),
                O.xyz4(np.stack([b["x"], b["y"], b["z"]], 1)), O.nrm4(np.stack([b["normal_x"], b["normal_y"], b["normal_z"]], 1)))

    poses, stats, t_total = {}, {}, 0.0
    cands = [t for t in (1, 4, 16, 64) if t < ncores] + [ncores]
    nxt = 0

    def run_one(th):
        nonlocal nxt, t_total
        src4, ns, tgt4, nt = host_pair(nxt)
        t0 = time.perf_counter()
        r = O.gicp_align(src4, ns, tgt4, nt, params(th), want_trace=False)
        dt = time.perf_counter() - t0
        poses[nxt] = r["T"]
        st = stats.setdefault(th, [0, 0.0])
        st[0] += 1
        st[1] += dt
        t_total += dt
        nxt += 1

    def rate(th):
        return stats[th][0] / stats[th][1]

    for th in cands:                        # round 1: one pair per setting
        if nxt < len(S):
            run_one(th)
    for th in sorted(stats, key=rate, reverse=True)[:3]:   # round 2: a second pair for the three fastest (single pairs are noisy)
        if nxt < len(S) and t_total < budget_s:
            run_one(th)
    best = max((th for th in stats if stats[th][0] >= min(2, max(v[0] for v in stats.values()))), key=rate)
    while t_total < budget_s and nxt < len(S) and stats[best][0] < 10:
        run_one(best)
    k, tt = stats[best]
    return {"value": k / tt, "unit": "scan-pairs/s", "cores": best, "kind": "port",
            "by_threads": {str(th): round(v[0] / v[1], 4) for th, v in stats.items()},
            "sample": "%d of the step's %d-pt pairs at %d OMP threads (fastest of %s on this %d-thread host; %.1f s of CPU work in all), "
                      "20 outer iterations, OMP on NN loops + serial cost functor like the reference"
                      % (k, len(S[0]), best, "/".join(str(c) for c in stats), ncores, t_total)}, poses


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=64, help="scan pairs per GPU per step")
    ap.add_argument("--in-flight", type=int, default=64)
    ap.add_argument("--rings", type=int, default=64)
    ap.add_argument("--azimuths", type=int, default=1563)  # 64 x 1563 = 100 032 points / scan
    ap.add_argument("--scale", type=float, default=2.0)
    ap.add_argument("--cost-mode", type=int, default=1, help="lh_gicp_params.cost_mode (1 = moments, 0 = per-evaluation passes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--single-latency", action="store_true", help="also time one-pair-at-a-time lh_gicp_align")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, the measured configuration) or gloo (functional check)")
    ap.add_argument("--same-gpu", action="store_true",
                    help="functional check of the N > 1 code path on a 1-GPU box: every rank uses cuda:0 (needs --dist-backend gloo); "
                         "the printed rate is meaningless")
    args = ap.parse_args()

    # safety net: a rank that stops making progress dumps every thread's Python stack and exits instead of hanging the box
    import faulthandler
    faulthandler.dump_traceback_later(float(os.environ.get("LH_BENCH_WATCHDOG_S", "1500")), exit=True)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.same_gpu:
        assert args.dist_backend == "gloo", "--same-gpu needs --dist-backend gloo (RCCL cannot put two ranks on one device)"
        local_rank = 0
    torch.cuda.set_device(local_rank)
    ddev = "cuda" if args.dist_backend == "nccl" else None   # where the tiny exchange tensors live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")
    assert world == args.gpus or world == 1, (world, args.gpus)

    ctx = capi.Context(local_rank)
    # forced 20 outer iterations: eps = 0 would divide by zero in the ratio, use a vanishing eps instead
    P = capi.default_params(max_iterations=20, max_inner_iterations=20, corr_dist=1.0, transformation_epsilon=1e-12,
                            rotation_epsilon=1e-12, cost_mode=args.cost_mode)
    S, T, host = make_pairs(ctx, args.pairs, rank, args.rings, args.azimuths, args.scale)
    n_pts = len(S[0])

    def step(in_flight=None, exchange=True):
        for t in T:
            t.drop_index()  # align() rebuilds the target index every scan, like pcl::Registration::initCompute
        out = capi.align_batch(ctx, P, S, T, max_in_flight=in_flight or args.in_flight)
        if world > 1 and exchange:  # result gather over RCCL/xGMI: 16 floats per pair; no data-path collective
            ldist.gather_poses(np.stack([o["T"] for o in out]), world, device=ddev)
        return out

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = ldist.max_over_ranks(elapsed, world, device=ddev)
    total_pairs = args.pairs * args.steps * world
    value = total_pairs / elapsed
    ok = all(o["status"] == 0 for o in out)
    iters = [int(o["iterations"]) for o in out]
    passes = float(np.mean([o["cost_passes"] for o in out]))

    # accuracy of the timed work vs the simulated motion (sanity, not the parity test)
    errs = []
    for o, (_, _, delta) in zip(out, host):
        Tm = np.asarray(o["T"], np.float64).reshape(4, 4).T
        errs.append(np.abs(Tm[:3, 3] - delta[:3, 3]).max())

    result = None
    if rank == 0:
        # ---- roofline leg: the same steps again with HIP-event timing of every launch on the library's stream ----
        # Profiling runs ONE scheduler group so kernels never overlap; to time launches of the same shape as the timed
        # region's (two half-batches of in_flight/2 pairs each) the leg runs with in_flight/2 pairs per launch.
        prof_in_flight = max(8, args.in_flight // 2) if args.in_flight >= 16 else args.in_flight
        ctx.profile(True)
        ctx.profile_reset()
        for _ in range(max(1, min(args.steps, 2))):
            step(prof_in_flight, exchange=False)  # rank 0 only: no collective may be called here
        stats = ctx.profile_get()
        ctx.profile(False)
        dom = max(stats.items(), key=lambda kv: kv[1]["ms"])
        name, st = dom
        achieved = st["bytes"] / 1e9 / (st["ms"] / 1e3) if st["ms"] > 0 else 0.0
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc_path):
            try:
                ent = json.load(open(pmc_path)).get(name, {})
                traffic = ent.get("hbm_bytes_per_launch")
                if traffic is not None and ent.get("jobs_per_launch"):  # PMC run used 32-job launches: scale to this leg's
                    traffic = traffic * prof_in_flight / ent["jobs_per_launch"]
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "avg_launch_us": round(1e3 * st["ms"] / max(1, st["launches"]), 2), "launches": st["launches"],
                    "jobs_per_launch": prof_in_flight,
                    "algorithmic_bytes_per_launch": round(st["bytes"] / max(1, st["launches"]), 1),
                    "kernels_ms": {k: round(v["ms"], 3) for k, v in sorted(stats.items(), key=lambda kv: -kv[1]["ms"])}}
        result = {
            "metric": "GICP scan-pairs/s (100k-pt clouds, 20 iters)", "value": round(value, 3), "unit": "scan-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 geometry / f64 cost",
            "data": "synthetic",
            "config": {"workload": "configs[1]: 100k-pt Velodyne-style scan-to-scan GICP, 20 outer iterations, odometry params "
                                   "(corr_dist 1.0, inner 20), covariances from k=20 normals; %d independent pairs per GPU per step, "
                                   "%d in flight" % (args.pairs, args.in_flight),
                       "points_per_scan": n_pts, "pairs_per_gpu_per_step": args.pairs, "parallelism": "pairs sharded over %d GPU(s)" % world},
            "all_ok": bool(ok), "outer_iterations_min_mean_max": [min(iters), float(np.mean(iters)), max(iters)],
            "cost_mode": args.cost_mode, "mean_cost_evaluations_per_pair": passes, "max_translation_err_vs_truth_m": float(np.max(errs)),
            "roofline": roofline,
        }
        if args.single_latency:
            g = capi.Gicp(ctx, P)
            g.set_source(S[0])
            g.set_target(T[0])
            T[0].drop_index()
            g.align(want_trace=False)
            t1 = time.perf_counter()
            for _ in range(5):
                T[0].drop_index()
                g.align(want_trace=False)
            result["single_pair_latency_ms"] = round(1e3 * (time.perf_counter() - t1) / 5, 3)
        if world == 1 and args.cost_mode == 1:
            # the same workload in the reference-arithmetic mode (one device pass per BFGS evaluation, float T*p): the strict
            # parity mode (<= 1e-4 m vs the CPU path); reported next to the headline, never as `value`
            P0 = capi.default_params(max_iterations=20, max_inner_iterations=20, corr_dist=1.0, transformation_epsilon=1e-12,
                                     rotation_epsilon=1e-12, cost_mode=0)

            def step0():
                for t in T:
                    t.drop_index()
                return capi.align_batch(ctx, P0, S, T, max_in_flight=args.in_flight)

            step0()
            ctx.synchronize()
            t1 = time.perf_counter()
            out0 = step0()
            ctx.synchronize()
            dt0 = time.perf_counter() - t1
            result["cost_mode0"] = {"value": round(args.pairs / dt0, 2), "unit": "scan-pairs/s",
                                    "max_abs_pose_diff_vs_mode1": float(max(np.abs(np.asarray(a["T"]) - np.asarray(b["T"])).max()
                                                                            for a, b in zip(out0, out)))}
        if world == 1:
            # SURVEY 8d "natural convergence" run: the same pairs with the production stopping rule (tf_eps 1e-3, rot_eps 2e-3,
            # gicp.h:119, parameters.yaml) instead of 20 forced iterations: iterations to converge, rate, distance to the forced result
            Pn = capi.default_params(max_iterations=20, max_inner_iterations=20, corr_dist=1.0, transformation_epsilon=1e-3,
                                     rotation_epsilon=2e-3, cost_mode=args.cost_mode)

            def stepn():
                for t in T:
                    t.drop_index()
                return capi.align_batch(ctx, Pn, S, T, max_in_flight=args.in_flight)

            stepn()
            ctx.synchronize()
            t1 = time.perf_counter()
            outn = stepn()
            ctx.synchronize()
            dtn = time.perf_counter() - t1
            itn = [int(o["iterations"]) for o in outn]
            result["natural_convergence"] = {
                "value": round(args.pairs / dtn, 2), "unit": "scan-pairs/s", "iterations_min_mean_max": [min(itn), float(np.mean(itn)), max(itn)],
                "all_converged": bool(all(o["converged"] == 1 for o in outn)),
                "max_abs_pose_diff_vs_20_forced_iterations": float(max(np.abs(np.asarray(a["T"]) - np.asarray(b["T"])).max() for a, b in zip(outn, out)))}
        if world == 1 and not args.no_cpu_baseline:
            cb, poses = cpu_baseline(S, T, host, P)
            # parity of the timed GPU work against the CPU path on the sampled pairs (reported, asserted in tests/)
            d = 0.0
            for k, To in poses.items():
                d = max(d, float(np.abs(np.asarray(out[k]["T"]) - np.asarray(To)).max()))
            cb["max_abs_pose_diff_vs_gpu"] = d
            result["cpu_baseline"] = cb
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
