"""GPU probe: LOCUS's operating point -- ~3 000-point scans, one update at a time, production stopping: wall time per update and where it goes
(HIP-event kernel sums, launches), for solver = host loop / device loop."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from locus_amd import capi, synth
ctx = capi.Context(0)
scans = []
for i in range(12):
    pose = synth.pose_matrix(0.12 * i, 0.03 * np.sin(0.5 * i), 0.0, 0.0, 0.0, 0.01 * i)
    pts = synth.scan(pose, 16, 1800, (-15.0, 15.0), 1.0, 0.02, seed=900 + i)
    c = capi.Cloud(ctx, capi.make_pointxyzi(pts)).voxel_grid(0.3068)
    c.normals_knn(20)
    scans.append(c)
print("points", [len(c) for c in scans[:4]])
out = {}
for solver in (1, 2):
    P = capi.default_params(max_iterations=20, max_inner_iterations=20, corr_dist=1.0, transformation_epsilon=1e-3, rotation_epsilon=2e-3, solver=solver)
    g = capi.Gicp(ctx, P)
    for prof in (False, True):
        walls, iters = [], []
        if prof:
            ctx.profile(True); ctx.profile_reset()
        for rep in range(3):
            for i in range(len(scans) - 1):
                g.set_source(scans[i + 1]); g.set_target(scans[i])
                scans[i].drop_index()
                ctx.synchronize()
                t0 = time.perf_counter()
                r = g.align(want_trace=False)
                walls.append(time.perf_counter() - t0); iters.append(r["iterations"])
        if prof:
            st = ctx.profile_get(); ctx.profile(False)
            n = len(walls)
            out["solver%d_profiled" % solver] = {"wall_us_median": round(1e6 * float(np.median(walls)), 1),
                                                 "kernels_us_per_update": {k: [round(1e3 * v["ms"] / n, 2), round(v["launches"] / n, 1)] for k, v in sorted(st.items(), key=lambda kv: -kv[1]["ms"])},
                                                 "kernel_sum_us": round(1e3 * sum(v["ms"] for v in st.values()) / n, 1), "launches_per_update": round(sum(v["launches"] for v in st.values()) / n, 1)}
        else:
            out["solver%d" % solver] = {"wall_us_median": round(1e6 * float(np.median(walls)), 1), "wall_us_min": round(1e6 * float(np.min(walls)), 1), "iterations_mean": float(np.mean(iters))}
print(json.dumps(out, indent=1))
