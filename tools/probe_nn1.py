"""minimal GPU probe for PMC runs: a few cold nn1 launches + warm sweeps on one 100k pair"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from locus_amd import capi, synth
ctx = capi.Context(0)
src, tgt, delta = synth.scan_pair(n_rings=64, n_az=1563, scale=2.0, noise=0.02, seed=10)
cs, ct = capi.Cloud(ctx, src), capi.Cloud(ctx, tgt)
cs.normals_knn(20); ct.normals_knn(20)
for _ in range(5):
    ct.nn1(cs)
P = capi.default_params(max_iterations=20, corr_dist=1.0, transformation_epsilon=1e-12, rotation_epsilon=1e-12)
g = capi.Gicp(ctx, P); g.set_source(cs); g.set_target(ct)
for _ in range(2):
    g.align(want_trace=False)
