#!/usr/bin/env python3
"""BASELINE.json configs[2]: scan-to-submap localization, one 100k-point scan against a ~2M-point
local map on one MI355X.  Times, with everything resident in HBM:

  * the map's NN index build (once per map refresh, PointCloudMapper::InsertPoints / Refresh)
  * the LOCUS flow of Locus.cc:474-489: scan -> fixed frame -> ApproxNearestNeighbors (one map point per scan point) ->
    back to the sensor frame -> MeasurementUpdate (GICP, localization parameters) against those neighbours
  * MeasurementUpdate directly against the whole map (no neighbour extraction)

    python tools/bench_submap.py [--reps 5]

Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from locus_amd import capi, synth  # noqa: E402


def mat_to_T16(M):
    return np.ascontiguousarray(np.asarray(M, np.float32).T).reshape(16)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--scans", type=int, default=40)
    ap.add_argument("--map-points", type=int, default=2_000_000)
    args = ap.parse_args()
    ctx = capi.Context(0)
    # SURVEY 8d config 3 (the same construction as tests/test_gpu_configs.py::_submap_2M): the union of 40 scans along a 20 m path,
    # voxelised at 0.05 m, padded to EXACTLY 2 000 000 points with a seeded sample of the raw union
    parts = []
    for i in range(args.scans):
        pose = synth.pose_matrix(tx=-10.0 + 20.0 * i / max(1, args.scans - 1), ty=1.5 * np.sin(i / 4.0), yaw=0.05 * np.cos(i / 3.0))
        pts = synth.scan(pose, 64, 1563, (-25.0, 15.0), 2.0, 0.01, seed=200 + i)
        parts.append((pts.astype(np.float64) @ pose[:3, :3].T + pose[:3, 3]).astype(np.float32))
    allpts = np.concatenate(parts)
    vox, cnt = ctx.voxel_grid(capi.make_pointxyzi(allpts), 0.05, 2, -100.0, 100.0)
    vox = vox[:cnt, :3].copy()
    rng = np.random.default_rng(2_000_000)
    if vox.shape[0] >= args.map_points:
        mpts = vox[np.sort(rng.choice(vox.shape[0], args.map_points, replace=False))]
    else:
        mpts = np.concatenate([vox, allpts[rng.choice(allpts.shape[0], args.map_points - vox.shape[0], replace=False)]])
    mpts = np.ascontiguousarray(mpts, np.float32)
    cmap = capi.Cloud(ctx, mpts)
    cmap.normals_knn(20)
    true_pose = synth.pose_matrix(tx=0.7, ty=0.2, yaw=0.03)
    guess = synth.pose_matrix(tx=0.8, ty=0.15, yaw=0.04)
    q = synth.scan(true_pose, 64, 1563, (-25.0, 15.0), 2.0, 0.01, seed=777)
    cq = capi.Cloud(ctx, q)
    cq.normals_knn(20)
    G16, Ginv16 = mat_to_T16(guess), mat_to_T16(np.linalg.inv(guess))
    P = capi.default_params(max_iterations=20, max_inner_iterations=50, corr_dist=0.2, transformation_epsilon=1e-5)
    g = capi.Gicp(ctx, P)

    def timed(fn):
        fn()  # warm-up (buffers, pools)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            out = fn()
        ctx.synchronize()
        return (time.perf_counter() - t0) / args.reps, out

    def build():
        cmap.drop_index()
        cmap.build_index()

    t_index, _ = timed(build)

    def nn_only():
        return cmap.nearest_neighbors(cq.transform(G16, with_normals=True))

    t_nn, _ = timed(nn_only)

    def locus_flow():
        in_fixed = cq.transform(G16, with_normals=True)
        neigh = cmap.nearest_neighbors(in_fixed)
        neigh_s = neigh.transform(Ginv16, with_normals=True)
        g.set_source(cq)
        g.set_target(neigh_s)
        return g.align(want_trace=False)

    t_flow, r_flow = timed(locus_flow)

    def direct():
        g.set_source(cq)
        g.set_target(cmap)
        return g.align(guess=G16, want_trace=False)

    t_direct, r_direct = timed(direct)

    def err(r, compose):
        T = np.asarray(r["T"], np.float64).reshape(4, 4).T
        T = guess @ T if compose else T
        return float(np.abs(T[:3, 3] - true_pose[:3, 3]).max())

    print(json.dumps({
        "workload": "configs[2]: 100k-pt scan vs local map, localization parameters (corr_dist 0.2, inner 50, tf_eps 1e-5)",
        "map_points": int(len(cmap)), "scan_points": int(len(cq)), "reps": args.reps,
        "ms_map_index_build": round(1e3 * t_index, 3),
        "ms_transform_plus_nearest_neighbors": round(1e3 * t_nn, 3),
        "ms_locus_flow_neighbours_then_gicp": round(1e3 * t_flow, 3), "iterations_flow": int(r_flow["iterations"]),
        "translation_err_flow_m": err(r_flow, True),
        "ms_gicp_direct_vs_whole_map": round(1e3 * t_direct, 3), "iterations_direct": int(r_direct["iterations"]),
        "translation_err_direct_m": err(r_direct, False),
    }))


if __name__ == "__main__":
    main()
