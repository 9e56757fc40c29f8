"""PMC probe: one batched alignment of 16 pairs (what bench.py times), nothing else."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from locus_amd import capi, synth
ctx = capi.Context(0)
S, T = [], []
for p in range(32):  # = one scheduler group of the default bench (64 in flight = 2 groups of 32)
    src, tgt, _ = synth.scan_pair(n_rings=64, n_az=1563, scale=2.0, noise=0.02, seed=10 + 2 * p)
    cs, ct = capi.Cloud(ctx, src), capi.Cloud(ctx, tgt)
    cs.normals_knn(20); ct.normals_knn(20); ct.drop_index()
    S.append(cs); T.append(ct)
P = capi.default_params(max_iterations=20, corr_dist=1.0, transformation_epsilon=1e-12, rotation_epsilon=1e-12)
ctx.profile(True)   # single scheduler group: kernels do not overlap
out = capi.align_batch(ctx, P, S, T, max_in_flight=32)
print([o["iterations"] for o in out])
