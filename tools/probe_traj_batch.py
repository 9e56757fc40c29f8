"""GPU probe: the first trajectory pairs as a batch, by pairs in flight and loop flavour -- what makes the batch slow?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from locus_amd import capi
n = int(os.environ.get("N", "33"))
traj = bench.gen_trajectory_host(513, 64, 1563, 2.0)[:n]
ctx = capi.Context(0)
clouds = [capi.Cloud(ctx, p) for p in traj]
capi.normals_knn_batch(clouds, 20)
src, tgt = clouds[1:], clouds[:-1]
for solver in (1, 2):
    for inf in (1, 2, 4, 7, 8, 9, 16, 32):
        P = capi.default_params(max_iterations=20, max_inner_iterations=20, corr_dist=1.0, transformation_epsilon=1e-12, rotation_epsilon=1e-12, solver=solver)
        for rep in range(2):
            for t in tgt: t.drop_index()
            ctx.synchronize(); t0 = time.perf_counter()
            out = capi.align_batch(ctx, P, src, tgt, max_in_flight=inf)
            ctx.synchronize(); dt = time.perf_counter() - t0
        print("solver", solver, "in flight", inf, "pairs/s %.0f" % (len(src) / dt), "ms/pair %.3f" % (1e3 * dt / len(src)))
