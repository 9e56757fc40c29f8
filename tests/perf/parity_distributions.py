#!/usr/bin/env python3
"""Full-size parity DISTRIBUTIONS of the benched configuration under BOTH stopping rules (runs on the GPU box): for the first N bench pairs
(default 64; 100 032 points, odometry parameters) and for each regime

    forced20    -- 20 outer iterations whatever happens (tf_eps, rotation_eps -> 0: SURVEY 8d's throughput protocol, the headline's)
    production  -- the rule LOCUS runs with: tf_eps 1e-3 (point_cloud_odometry/config/parameters.yaml:12), rotation_epsilon 2e-3 (gicp.h:119),
                   gicp.hpp:566

the GPU result of cost_mode 1 (the benched default) and of cost_mode 0 (reference arithmetic) against the CPU restatement, next to the
distance between the restatement's own two legal builds (float T*p with / without FMA contraction, gicp.hpp:382):
    pose            |dt| (m), |dR| of the final transform
    fitness         relative error of getFitnessScore (the GPU's score of ITS pose vs the restatement's score of its own)
    per-iteration   the largest |dT| (max abs entry of the 4x4) over the outer iterations the two traces share
    iterations      outer-iteration counts
Quantiles per column + every pair.  One JSON object on stdout.

    python tests/perf/parity_distributions.py [n_pairs=64] > gpurun_out/parity_distributions.json"""
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from locus_amd import synth  # noqa: E402

REGIMES = {"forced20": dict(transformation_epsilon=1e-12, rotation_epsilon=1e-12),
           "production": dict(transformation_epsilon=1e-3, rotation_epsilon=2e-3)}
BASE = dict(max_iterations=20, max_inner_iterations=20, corr_dist=1.0)


def _gen(seed):
    return synth.scan_pair(n_rings=64, n_az=1563, scale=2.0, noise=0.02, seed=seed)


def q(v):
    v = np.asarray(v, np.float64)
    return {"median": float(np.median(v)), "p90": float(np.quantile(v, 0.9)), "max": float(v.max()), "within_1e-4": int((v <= 1e-4).sum()), "n": int(len(v))}


def main():
    n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    seeds = [10 + 2 * p for p in range(n_pairs)]   # bench.py gen_pairs_host(), rank 0
    import multiprocessing as mp
    with mp.get_context("fork").Pool(min(16, os.cpu_count() or 1)) as pool:   # before any GPU runtime exists in this process
        host = pool.map(_gen, seeds, chunksize=2)
    from locus_amd import capi
    from oracle import oracle as O
    cores = os.cpu_count() or 4
    omp = 4
    workers = max(1, min(n_pairs, cores // omp))
    ctx = capi.Context(0)
    L = O.lib()
    S = [capi.Cloud(ctx, src) for src, _, _ in host]
    T = [capi.Cloud(ctx, tgt) for _, tgt, _ in host]
    for o in range(0, n_pairs, 32):
        capi.normals_knn_batch(S[o:o + 32] + T[o:o + 32], 20)
    inputs = []
    for (src, tgt, _), cs, ct in zip(host, S, T):
        a, b = cs.download(), ct.download()
        inputs.append((O.xyz4(src), O.nrm4(np.stack([a["normal_x"], a["normal_y"], a["normal_z"]], 1)),
                       O.xyz4(tgt), O.nrm4(np.stack([b["normal_x"], b["normal_y"], b["normal_z"]], 1))))
    trees = [O.Tree(i[2]) for i in inputs]

    def pose_err(A16, B16):
        A, B = O.T_to_mat(A16), O.T_to_mat(B16)
        return float(np.abs(A[:3, 3] - B[:3, 3]).max()), float(np.abs(A[:3, :3] - B[:3, :3]).max())

    def iter_err(ta, tb):
        k = min(len(ta["T"]), len(tb["T"]))
        return float(np.abs(np.asarray(ta["T"][:k], np.float64) - np.asarray(tb["T"][:k], np.float64)).max()) if k else 0.0

    out = {"workload": "bench pairs (seeds 10, 12, ...), 100 032 points, odometry parameters, covariances from k = 20 normals; GPU: lh_gicp_align with its trace "
                       "(the host-driven loop: bit-identical to the device-driven loop of the batches, tests/test_gpu_solver.py); CPU: oracle, %d threads per pair" % omp,
           "n_pairs": n_pairs, "regimes": {}}
    for regime, eps in REGIMES.items():
        kw = dict(BASE, **eps)
        ref = []
        for v in (0, 1):   # the variant switch is process-wide: all pairs of one variant concurrently (ctypes releases the GIL), then the other
            L.lo_set_cost_variant(v)
            with ThreadPoolExecutor(workers) as ex:
                ref.append(list(ex.map(lambda a: O.gicp_align(a[0], a[1], a[2], a[3], O.default_params(num_threads=omp, **kw), want_trace=True), inputs)))
        L.lo_set_cost_variant(0)
        with ThreadPoolExecutor(workers) as ex:
            ref_fit = list(ex.map(lambda p: O.fitness(inputs[p][0], ref[0][p]["T"], trees[p], threads=omp), range(n_pairs)))
        rows = [dict(seed=s) for s in seeds]
        for mode in (1, 0):
            g = capi.Gicp(ctx, capi.default_params(cost_mode=mode, **kw))
            for p in range(n_pairs):
                g.set_source(S[p])
                g.set_target(T[p])
                T[p].drop_index()
                r = g.align(want_trace=True)
                fit = g.fitness()
                dt, dr = pose_err(r["T"], ref[0][p]["T"])
                rows[p]["mode%d" % mode] = {"dt": dt, "dR": dr, "fitness_rel": abs(fit - ref_fit[p]) / ref_fit[p], "iter_dT_max": iter_err(r["trace"], ref[0][p]["trace"]),
                                           "iterations": int(r["iterations"]), "status": int(r["status"])}
            g.close()
        for p in range(n_pairs):
            dt, dr = pose_err(ref[0][p]["T"], ref[1][p]["T"])
            with_fma = O.fitness(inputs[p][0], ref[1][p]["T"], trees[p], threads=omp)
            rows[p]["ref_vs_ref_fma"] = {"dt": dt, "dR": dr, "fitness_rel": abs(with_fma - ref_fit[p]) / ref_fit[p], "iter_dT_max": iter_err(ref[0][p]["trace"], ref[1][p]["trace"]),
                                         "iterations": [int(ref[0][p]["iterations"]), int(ref[1][p]["iterations"])]}
        summ = {}
        for col in ("mode1", "mode0", "ref_vs_ref_fma"):
            summ[col] = {k: q([r[col][k] for r in rows]) for k in ("dt", "dR", "fitness_rel", "iter_dT_max")}
        summ["iterations"] = {"reference_mean": float(np.mean([r["ref_vs_ref_fma"]["iterations"][0] for r in rows])), "mode1_mean": float(np.mean([r["mode1"]["iterations"] for r in rows])),
                              "mode0_mean": float(np.mean([r["mode0"]["iterations"] for r in rows])),
                              "mode1_same_count_as_reference": int(sum(r["mode1"]["iterations"] == r["ref_vs_ref_fma"]["iterations"][0] for r in rows)),
                              "mode0_same_count_as_reference": int(sum(r["mode0"]["iterations"] == r["ref_vs_ref_fma"]["iterations"][0] for r in rows)),
                              "reference_same_count_as_its_fma_build": int(sum(r["ref_vs_ref_fma"]["iterations"][0] == r["ref_vs_ref_fma"]["iterations"][1] for r in rows))}
        out["regimes"][regime] = {"params": kw, "summary": summ, "pairs": rows}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
    sys.stdout.flush()
    os._exit(0)   # (the result is out: see the end of bench.py)
