"""GPU probe: per-kernel HIP-event times of a 64-pair batch (32 in flight, one scheduler group at a time) for the headline's independent pairs
and for consecutive scans of the trajectory -- where do the trajectory's pairs spend twice the time?"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from locus_amd import capi
n = int(os.environ.get("N", "65"))
traj = bench.gen_trajectory_host(513, 64, 1563, 2.0)[:n]   # (the first n scans of the 513-scan drive: the step size depends on the total)
pairs = bench.gen_pairs_host(n - 1, 0, 64, 1563, 2.0)
ctx = capi.Context(0)
P = capi.default_params(max_iterations=20, max_inner_iterations=20, corr_dist=1.0, transformation_epsilon=1e-12, rotation_epsilon=1e-12)
clouds = [capi.Cloud(ctx, p) for p in traj]
capi.normals_knn_batch(clouds, 20)
S, T, _ = bench.make_pairs(ctx, pairs)
out = {}
for name, src, tgt in (("independent", S, T), ("trajectory", clouds[1:], clouds[:-1])):
    for t in tgt: t.drop_index()
    capi.align_batch(ctx, P, src, tgt, max_in_flight=32)
    for t in tgt: t.drop_index()
    ctx.synchronize(); ctx.profile(True); ctx.profile_reset()
    t0 = time.perf_counter()
    res = capi.align_batch(ctx, P, src, tgt, max_in_flight=32)
    ctx.synchronize(); dt = time.perf_counter() - t0
    st = ctx.profile_get(); ctx.profile(False)
    out[name] = {"pairs_per_s_profiled": round(len(src) / dt), "iterations_mean": float(np.mean([o["iterations"] for o in res])),
                 "n_corr_last_mean": float(np.mean([o["n_corr_last"] for o in res])),
                 "kernels_ms": {k: [round(v["ms"], 2), v["launches"]] for k, v in sorted(st.items(), key=lambda kv: -kv[1]["ms"])}}
print(json.dumps(out, indent=1))
