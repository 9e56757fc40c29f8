"""GPU probe program for tools/pmc_filters.sh: the two filter kernels in front of GICP on BASELINE-sized inputs, a few launches each --
K3 = k = 20 normals of 32 x 100 032-point scans in one batch (index build + block k-NN), K1 = voxel grid (leaf 0.1) of a 1 M-point merged cloud."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from locus_amd import capi, synth  # noqa: E402

ctx = capi.Context(0)
scans = [synth.scan(synth.pose_matrix(0.05 * i, 0.02 * i, 0), 64, 1563, (-25.0, 15.0), 2.0, 0.02, seed=100 + i) for i in range(32)]
clouds = [capi.Cloud(ctx, s) for s in scans]
for rep in range(3):
    for c in clouds:
        c.drop_index()
    capi.normals_knn_batch(clouds, 20)
ctx.synchronize()
big = np.concatenate([synth.scan(synth.pose_matrix(0.1 * k, 0, 0.1 * k), 128, 2604, (-25.0, 15.0), 2.0, 0.02, seed=300 + k) for k in range(3)])
cb = capi.Cloud(ctx, big)
for rep in range(3):
    v = cb.voxel_grid(0.1)
ctx.synchronize()
print("K3: 32 x %d points; K1: %d -> %d points" % (len(scans[0]), len(big), len(v)))
