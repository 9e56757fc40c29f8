"""GPU probe: filter stages (K1 voxel grid, K3 normals) at BASELINE config-5 size, HIP-event kernel times"""
import sys, os, json, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from locus_amd import capi, synth
ctx = capi.Context(0)
pts = np.concatenate([synth.scan(synth.pose_matrix(0.1 * k, 0, 0.1 * k), 128, 2604, (-25.0, 15.0), 2.0, 0.02, seed=300 + k) for k in range(3)])
print("points", pts.shape)
xyzi = capi.make_pointxyzi(pts)
out = {}
def timed(name, fn, reps=3):
    fn()
    ctx.profile(True); ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    wall = (time.perf_counter() - t0) / reps
    st = ctx.profile_get(); ctx.profile(False)
    out[name] = {"wall_ms": round(1e3 * wall, 2), **{k: round(v["ms"] / reps, 3) for k, v in st.items()}}
timed("voxel_grid_1M_leaf0.1", lambda: ctx.voxel_grid(xyzi, 0.1, 2, -100.0, 100.0))
c1m = capi.Cloud(ctx, pts)
timed("normals_knn20_1M", lambda: c1m.normals_knn(20))
v, cnt = ctx.voxel_grid(xyzi, 0.1, 2, -100.0, 100.0)
cv = capi.Cloud(ctx, v[:, :3].copy())
print("voxelised", cnt)
timed("normals_knn20_voxelised", lambda: cv.normals_knn(20))
c100 = capi.Cloud(ctx, pts[:100032].copy())
timed("normals_knn20_100k", lambda: c100.normals_knn(20))
timed("cov_knn20_100k", lambda: (c100.drop_index(), c100.cov_knn(20, 1e-3)))
print(json.dumps(out, indent=1))
