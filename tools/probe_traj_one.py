"""GPU probe: kernel-level profile of single trajectory pairs (argv: pair indices)"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from locus_amd import capi
idx = [int(a) for a in sys.argv[1:]] or [0, 218]
traj = bench.gen_trajectory_host(513, 64, 1563, 2.0)
ctx = capi.Context(0)
P = capi.default_params(max_iterations=20, max_inner_iterations=20, corr_dist=1.0, transformation_epsilon=1e-12, rotation_epsilon=1e-12)
for i in idx:
    a, b = capi.Cloud(ctx, traj[i]), capi.Cloud(ctx, traj[i + 1])
    capi.normals_knn_batch([a, b], 20)
    g = capi.Gicp(ctx, P)
    g.set_source(b); g.set_target(a)
    a.drop_index(); g.align(want_trace=False)
    a.drop_index()
    ctx.profile(True); ctx.profile_reset()
    t0 = time.perf_counter(); r = g.align(want_trace=True); dt = time.perf_counter() - t0
    st = ctx.profile_get(); ctx.profile(False)
    tr = r["trace"]
    print("pair", i, "ms %.2f" % (1e3 * dt), "iters", r["iterations"], "passes", r["cost_passes"], "inner", [int(x) for x in tr["n_inner"]], "n_passes", [int(x) for x in tr["n_passes"]])
    print("   kernels", {k: (round(v["ms"], 3), v["launches"]) for k, v in sorted(st.items(), key=lambda kv: -kv[1]["ms"])[:6]})
    stt = a.traversal_stats(b, None) if hasattr(a, "traversal_stats") else None
    print("   cold traversal stats", stt)
