"""why is the trajectory batch slower than the independent-pair batch?  (scheduler breakdown of both, LH_HOST_PROF=1)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["LH_HOST_PROF"] = "1"
import bench
from locus_amd import capi, synth
n = int(os.environ.get("N", "129"))
traj = bench.gen_trajectory_host(513, 64, 1563, 2.0)[:n]   # (the first n scans of the 513-scan drive: the step size depends on the total)
pairs = bench.gen_pairs_host(n - 1, 0, 64, 1563, 2.0)
ctx = capi.Context(0)
P = capi.default_params(max_iterations=20, max_inner_iterations=20, corr_dist=1.0, transformation_epsilon=1e-12, rotation_epsilon=1e-12)
clouds = []
for pts in traj:
    c = capi.Cloud(ctx, pts); c.normals_knn(20); c.drop_index(); clouds.append(c)
S, T, _ = bench.make_pairs(ctx, pairs)
for name, src, tgt in (("independent", S, T), ("trajectory", clouds[1:], clouds[:-1]), ("trajectory-copies", None, None)):
    if tgt is None:   # the same trajectory with every target an own copy of the scan (no cloud is source and target)
        tgt = []
        for pts in traj[:-1]:
            c = capi.Cloud(ctx, pts); c.normals_knn(20); c.drop_index(); tgt.append(c)
        src = clouds[1:]
    for rep in range(2):
        for t in tgt: t.drop_index()
        ctx.synchronize(); t0 = time.perf_counter()
        out = capi.align_batch(ctx, P, src, tgt, max_in_flight=128)
        ctx.synchronize(); dt = time.perf_counter() - t0
    it = [o["iterations"] for o in out]
    print(name, "pairs/s %.0f" % (len(src) / dt), "iterations", min(it), np.mean(it), max(it), "passes", np.mean([o["cost_passes"] for o in out]))
