"""time of the one-launch index build of a ~3 k-point cloud (k_index_small), HIP events: python tools/probe_small_index.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from locus_amd import capi, synth
ctx = capi.Context(0)
pts = synth.scan(np.eye(4), 16, 1800, (-15.0, 15.0), 1.0, 0.02, seed=900)
raw = capi.Cloud(ctx, capi.make_pointxyzi(pts))
c = raw.voxel_grid(0.3068)
print("points", len(c))
for _ in range(5):
    c.drop_index(); c.build_index()
ctx.synchronize()
ctx.profile(True); ctx.profile_reset()
reps = 200
t0 = time.perf_counter()
for _ in range(reps):
    c.drop_index(); c.build_index()
ctx.synchronize()
wall = (time.perf_counter() - t0) / reps
st = ctx.profile_get(); ctx.profile(False)
print("wall us %.1f" % (1e6 * wall), {k: round(1e3 * v["ms"] / reps, 2) for k, v in st.items()})
