#!/usr/bin/env python3
"""three lidars x 128 rings x 2604 azimuths ~ 1.0 M points per frame, merged in the body frame
(PointCloudMerger.cc:158-159), voxel grid leaf 0.1 (K1) -> k=20 normals (K3) -> GICP against the previous frame.

    python tools/bench_merged1m.py                                  # 1 GPU, whole pair
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        tools/bench_merged1m.py                                     # N GPUs: source points sharded, RCCL all-reduce of 74 sums

With N > 1 every rank filters the whole frame (K1 / K3 are not sharded: 32 MB), keeps the whole previous frame +
index and a 1/N slice of the new frame; the only exchange is lh_set_allreduce's SUM of the 74 moment sums per outer
iteration (20 per pair).  Prints one JSON line (rank 0)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from locus_amd import capi, synth  # noqa: E402
from locus_amd import dist as ldist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--leaf", type=float, default=0.1)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--same-gpu", action="store_true", help="all ranks on cuda:0 (functional check on a 1-GPU box; use --backend gloo)")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dev = 0 if args.same_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend="gloo")
    ctx = capi.Context(dev)
    ctx.set_allreduce(ldist.make_sum_hook(world, device="cuda" if (world > 1 and args.backend == "nccl") else None) if world > 1 else None)
    ext = synth.husky_extrinsics()
    poses = [synth.pose_matrix(0.25 * i, -0.1 * i, 0.01 * i, 0.002 * i, -0.003 * i, 0.02 * i) for i in range(args.frames + 1)]
    raw = []  # per frame: three per-sensor device clouds already in the body frame (what the merger receives)
    for i, pose in enumerate(poses):
        raw.append([capi.Cloud(ctx, capi.make_pointxyzi(p)) for p in synth.multi_lidar_parts(pose, ext, seed=300 + 10 * i)])
    P = capi.default_params(max_iterations=20, max_inner_iterations=20, corr_dist=1.0, transformation_epsilon=1e-12, rotation_epsilon=1e-12)
    g = capi.Gicp(ctx, P)

    def preprocess(parts):
        merged = capi.Cloud.concat(parts)                                      # PointCloudMerger.cc:158-159
        merged = merged.crop_box([-0.6, -0.45, -0.3], [0.6, 0.45, 0.5], 0.0, True)  # BodyFilter: drop returns from the robot itself
        v = merged.voxel_grid(args.leaf, 2, -100.0, 100.0)                     # CustomVoxelGrid + z pass-through
        v.normals_knn(20)
        return merged, v

    _, prev = preprocess(raw[0])
    t_filter = t_align = 0.0
    n_raw = n_vox = 0
    errs = []
    for i in range(1, args.frames + 1):
        ctx.synchronize()
        t0 = time.perf_counter()
        merged, cur = preprocess(raw[i])
        ctx.synchronize()
        t1 = time.perf_counter()
        lo, hi = ldist.shard_range(len(cur), rank, world)
        src = cur if world == 1 else cur.slice(lo, hi - lo)
        prev.drop_index()
        g.set_target(prev)
        g.set_source(src)
        r = g.align(want_trace=False)
        ctx.synchronize()
        t2 = time.perf_counter()
        if i > 1:  # frame 1 is the warm-up
            t_filter += t1 - t0
            t_align += t2 - t1
        n_raw, n_vox = len(merged), len(cur)
        Tm = np.asarray(r["T"], np.float64).reshape(4, 4).T
        truth = np.linalg.inv(poses[i - 1]) @ poses[i]
        errs.append(float(np.abs(Tm[:3, 3] - truth[:3, 3]).max()))
        assert r["status"] == 0, r
        prev = cur
    k = max(1, args.frames - 1)
    t_filter = ldist.max_over_ranks(t_filter, world, device="cuda" if args.backend == "nccl" else None)
    t_align = ldist.max_over_ranks(t_align, world, device="cuda" if args.backend == "nccl" else None)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"workload": "configs[4] (SURVEY 8d config 5): 3 lidars -> merge -> body crop -> voxel %.2f -> k=20 normals -> GICP 20 iters vs previous frame" % args.leaf,
                          "n_gpus": world, "backend": args.backend if world > 1 else None, "raw_points": n_raw, "voxelised_points": n_vox,
                          "frames_timed": k, "ms_filter_per_frame": round(1e3 * t_filter / k, 3), "ms_gicp_per_frame": round(1e3 * t_align / k, 3),
                          "frames_per_s": round(k / (t_filter + t_align), 3), "max_translation_err_vs_truth_m": max(errs),
                          "source_sharding": "whole pair" if world == 1 else "source points / %d, all-reduce of 74 f64 per outer iteration" % world}))


if __name__ == "__main__":
    main()
