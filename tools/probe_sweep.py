"""GPU probe: where does the NN sweep time go?  (ablation by entry point, HIP-event timing from the library)"""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from locus_amd import capi, synth

ctx = capi.Context(0)
src, tgt, delta = synth.scan_pair(n_rings=64, n_az=1563, scale=2.0, noise=0.02, seed=10)
cs, ct = capi.Cloud(ctx, src), capi.Cloud(ctx, tgt)
cs.normals_knn(20); ct.normals_knn(20)
P = capi.default_params(max_iterations=20, corr_dist=1.0, transformation_epsilon=1e-12, rotation_epsilon=1e-12)
g = capi.Gicp(ctx, P); g.set_source(cs); g.set_target(ct)
I = np.eye(4, dtype=np.float32).T.reshape(16)
out = {}
def timed(name, fn, reps=5):
    fn()
    ctx.profile(True); ctx.profile_reset()
    for _ in range(reps): fn()
    st = ctx.profile_get(); ctx.profile(False)
    out[name] = {k: round(1e3 * v["ms"] / max(1, v["launches"]), 1) for k, v in st.items()}
timed("nn1_self", lambda: ct.nn1(ct))
timed("nn1_src_vs_tgt", lambda: ct.nn1(cs))
timed("knn20_self", lambda: ct.knn(ct, 20), reps=2)
timed("sweep_identity(warm after first)", lambda: g.debug_sweep(I, len(cs)))
Td = np.ascontiguousarray(delta.astype(np.float32).T).reshape(16)
timed("sweep_true_pose(warm)", lambda: g.debug_sweep(Td, len(cs)))
x = np.zeros(6)
timed("cost", lambda: g.debug_cost(x))
ct.drop_index()
timed("index_build", lambda: (ct.drop_index(), ct.build_index()))
# shuffled queries (no spatial coherence between neighbouring threads)
rng = np.random.default_rng(0)
perm = rng.permutation(len(src))
cs2 = capi.Cloud(ctx, src[perm])
timed("nn1_shuffled_src_vs_tgt", lambda: ct.nn1(cs2))
print(json.dumps(out, indent=1))
