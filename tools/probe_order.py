"""GPU probe: does a batch get slow when an earlier batch of OTHER clouds used the same context?  (trajectory / independent / trajectory ...)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from locus_amd import capi
n = 33
traj = bench.gen_trajectory_host(513, 64, 1563, 2.0)[:n]
pairs = bench.gen_pairs_host(n - 1, 0, 64, 1563, 2.0)
ctx = capi.Context(0)
clouds = [capi.Cloud(ctx, p) for p in traj]
capi.normals_knn_batch(clouds, 20)
S, T, _ = bench.make_pairs(ctx, pairs)
P = capi.default_params(max_iterations=20, max_inner_iterations=20, corr_dist=1.0, transformation_epsilon=1e-12, rotation_epsilon=1e-12)
def run(name, src, tgt, inf=32):
    for t in tgt: t.drop_index()
    ctx.synchronize(); ctx.profile(True); ctx.profile_reset(); t0 = time.perf_counter()
    out = capi.align_batch(ctx, P, src, tgt, max_in_flight=inf)
    ctx.synchronize(); dt = time.perf_counter() - t0
    st = ctx.profile_get(); ctx.profile(False)
    print("%-12s pairs/s %6.0f  nn_sweep %.2f ms  solve %.2f ms  seed %.3f ms" % (name, len(src) / dt, st.get("nn_sweep", {}).get("ms", 0), st.get("bfgs_solve", {}).get("ms", 0), st.get("nn_seed", {}).get("ms", 0)), flush=True)
order = os.environ.get("ORDER", "t,t,i,i,t,t,i,t")
for o in order.split(","):
    if o == "t": run("trajectory", clouds[1:], clouds[:-1])
    elif o == "i": run("independent", S, T)
    elif o == "r": run("traj reversed", clouds[:-1], clouds[1:])
