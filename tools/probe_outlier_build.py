"""index build time of a 100 k-point scan with and without far outliers (the start grid's tables: an outlier stretches the key grid, its
leaf spans many table cells, and ONE thread of k_nodex_b enters it in every one of them)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from locus_amd import capi, synth
ctx = capi.Context(0)
base = synth.scan(rings=64, azimuths=1563, scale=2.0, seed=3)[:, :3].astype(np.float32)
rng = np.random.default_rng(5)
cases = {
    "plain": base,
    "3 outliers at 4 km": np.concatenate([base, np.array([[4000.0, 0, 0], [0, -3000.0, 10.0], [-50.0, 60.0, 900.0]], np.float32)]),
    "50 outliers at 0.2-1 km": np.concatenate([base, (rng.normal(size=(50, 3)) * [600, 600, 100]).astype(np.float32)]),
    "2000 outliers at 100-300 m": np.concatenate([base, (rng.normal(size=(2000, 3)) * [200, 200, 30]).astype(np.float32)]),
}
for name, pts in cases.items():
    c = capi.Cloud(ctx, pts)
    c.build_index(); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        c.drop_index(); c.build_index()
    ctx.synchronize()
    print("%-28s n = %6d  index build %.3f ms" % (name, len(pts), 1e3 * (time.perf_counter() - t0) / 20))
