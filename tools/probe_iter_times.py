"""per-launch sweep times of one 32-pair group (profiling mode: one scheduler group, HIP events per launch) -> which outer
iterations cost what.  LH_PROF_LOG must point at a file."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
log = os.environ.setdefault("LH_PROF_LOG", "/tmp/lh_prof.log")
if os.path.exists(log):
    os.remove(log)
from locus_amd import capi, synth
ctx = capi.Context(0)
P = capi.default_params(max_iterations=20, max_inner_iterations=20, corr_dist=1.0, transformation_epsilon=1e-12, rotation_epsilon=1e-12,
                        solver=int(os.environ.get("LH_PROBE_SOLVER", "0")))
S, T = [], []


def morton_order(pts, bits=10):
    """permutation that sorts points along a 3-D Z-order curve (LH_PROBE_SORT=1: what would a spatially sorted SOURCE buy?)"""
    lo, hi = pts.min(0), pts.max(0)
    g = np.clip(((pts - lo) / max(float((hi - lo).max()), 1e-9) * ((1 << bits) - 1)).astype(np.uint64), 0, (1 << bits) - 1)
    key = np.zeros(len(pts), np.uint64)
    for b in range(bits):
        for a in range(3):
            key |= ((g[:, a] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + a)
    return np.argsort(key, kind="stable")


NP = int(os.environ.get("LH_PROBE_PAIRS", "32"))
for p in range(NP):
    src, tgt, _ = synth.scan_pair(n_rings=64, n_az=1563, scale=2.0, noise=0.02, seed=10 + 2 * p)
    if os.environ.get("LH_PROBE_TGT_STRIDE"):   # a sparser target (every k-th return): a smaller tree under the same 100 k queries
        tgt = np.ascontiguousarray(tgt[::int(os.environ["LH_PROBE_TGT_STRIDE"])])
    if os.environ.get("LH_PROBE_SORT") == "1":
        src = src[morton_order(src)]
    elif os.environ.get("LH_PROBE_SORT") == "2":
        src = src[np.random.default_rng(p).permutation(len(src))]
    cs, ct = capi.Cloud(ctx, src), capi.Cloud(ctx, tgt)
    cs.normals_knn(20); ct.normals_knn(20); ct.drop_index()
    S.append(cs); T.append(ct)
capi.align_batch(ctx, P, S, T, max_in_flight=NP)   # warm-up
for t in T:
    t.drop_index()
ctx.profile(True); ctx.profile_reset()
capi.align_batch(ctx, P, S, T, max_in_flight=NP)
st = ctx.profile_get()
ctx.profile(False)
rows = [l.split() for l in open(log)]
sw = [float(v) for k, v in rows if k == "nn_sweep"]
seed = [float(v) for k, v in rows if k == "nn_seed" or k == "seed"]
print("compact =", os.environ.get("LH_COMPACT", "1"), "launches", len(sw), "total ms %.3f" % sum(sw))
print("per-iteration sweep us:", " ".join("%.0f" % (1e3 * v) for v in sw))
print("other entries:", {k: round(v["ms"], 3) for k, v in st.items()})
