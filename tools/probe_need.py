"""fraction of source points whose certificate fails (= tree walks) at every outer iteration of one config-2 pair"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from locus_amd import capi, synth
ctx = capi.Context(0)
src, tgt, delta = synth.scan_pair(n_rings=64, n_az=1563, scale=2.0, noise=0.02, seed=10)
cs, ct = capi.Cloud(ctx, src), capi.Cloud(ctx, tgt)
cs.normals_knn(20); ct.normals_knn(20)
P = capi.default_params(max_iterations=20, max_inner_iterations=20, corr_dist=1.0, transformation_epsilon=1e-12, rotation_epsilon=1e-12)
g = capi.Gicp(ctx, P); g.set_source(cs); g.set_target(ct)
r = g.align(want_trace=True)
Ts = r["trace"]["T"]
g2 = capi.Gicp(ctx, P); g2.set_source(cs); g2.set_target(ct)
T = np.eye(4, dtype=np.float32).T.reshape(16)
prev = None
out = []
for t in range(len(Ts)):
    idx, _ = g2.debug_sweep(T, len(cs))
    s, n = g2.debug_stats()
    moved = 0.0 if prev is None else float(np.abs(np.asarray(T) - prev).max())
    changed = -1 if t == 0 else int((idx != last_idx).sum())
    out.append("%d:need %.3f dT %.4f nn-changed %d" % (t, s / max(n, 1), moved, changed))
    prev = np.asarray(T).copy(); last_idx = idx
    T = Ts[t]
print("\n".join(out))
