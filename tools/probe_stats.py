import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from locus_amd import capi, synth
ctx = capi.Context(0)
src, tgt, delta = synth.scan_pair(n_rings=64, n_az=1563, scale=2.0, noise=0.02, seed=10)
cs, ct = capi.Cloud(ctx, src), capi.Cloud(ctx, tgt)
I = np.eye(4, dtype=np.float32).T.reshape(16)
Td = np.ascontiguousarray(delta.astype(np.float32).T).reshape(16)
print("cold, identity pose:", ct.traversal_stats(cs, I))
print("cold, true pose    :", ct.traversal_stats(cs, Td))
print("cold, self         :", ct.traversal_stats(ct, None))
rng = np.random.default_rng(0)
cs2 = capi.Cloud(ctx, src[rng.permutation(len(src))])
print("cold, shuffled     :", ct.traversal_stats(cs2, I))

# warm searches: candidates = exact NN at a slightly different pose (what sweep k hands to sweep k+1)
for frac in (0.0, 0.7, 0.95):
    Tm = np.eye(4)
    Tm[:3, :3] = np.eye(3) * (1 - frac) + delta[:3, :3] * frac
    Tm[:3, 3] = delta[:3, 3] * frac
    Tp = np.ascontiguousarray(Tm.astype(np.float32).T).reshape(16)
    idx, _ = ct.nn1(cs.transform(Tp))
    print("warm from pose %.2f -> true pose, no prescan  :" % frac, ct.traversal_stats(cs, Td, idx, False))
    print("warm from pose %.2f -> true pose, leaf prescan:" % frac, ct.traversal_stats(cs, Td, idx, True))
seed = np.repeat(ct.nn1(cs.transform(Td))[0][::8], 8)[: len(src)]
print("seeded (every 8th) no prescan  :", ct.traversal_stats(cs, Td, seed, False))
print("seeded (every 8th) leaf prescan:", ct.traversal_stats(cs, Td, seed, True))
