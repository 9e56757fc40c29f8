"""GPU probe: one-at-a-time alignment latency of consecutive trajectory scans: which pairs are slow, and what do they look like?"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from locus_amd import capi
n = int(os.environ.get("N", "65"))
traj = bench.gen_trajectory_host(513, 64, 1563, 2.0)[:n]
ctx = capi.Context(0)
P = capi.default_params(max_iterations=20, max_inner_iterations=20, corr_dist=1.0, transformation_epsilon=1e-12, rotation_epsilon=1e-12)
clouds = [capi.Cloud(ctx, p) for p in traj]
capi.normals_knn_batch(clouds, 20)
g = capi.Gicp(ctx, P)
rows = []
for i in range(n - 1):
    g.set_source(clouds[i + 1]); g.set_target(clouds[i])
    clouds[i].drop_index(); g.align(want_trace=False)
    clouds[i].drop_index()
    t0 = time.perf_counter(); r = g.align(want_trace=True); dt = time.perf_counter() - t0
    tr = r.get("trace", {})
    rows.append((round(1e3 * dt, 2), r["iterations"], r["n_corr_last"], [int(x) for x in tr.get("n_corr", [])[:6]]))
ms = np.array([r[0] for r in rows])
print("pairs", len(rows), "latency ms: median %.2f p90 %.2f max %.2f" % (np.median(ms), np.quantile(ms, 0.9), ms.max()))
for i, r in enumerate(rows):
    if r[0] > 1.6 or i % 64 == 0:
        print(i, r)
ext = [np.abs(t).max(0) for t in traj[:8]]
print("extents", ext[:4])
