"""GPU probe: K3 (k = 20 normals / covariances) on 100 032-point scans, one cloud and a batch of 32 -- HIP-event times of the index
build and of the k-NN launch (block search + redo), per cloud; optional check against the oracle on one cloud.
  python tools/probe_knn.py [n_clouds=32] [--check] [--json out.json]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from locus_amd import capi, synth  # noqa: E402

n_clouds = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 32
ctx = capi.Context(0)
scans = [synth.scan(synth.pose_matrix(0.05 * i, 0.02 * i, 0), 64, 1563, (-25.0, 15.0), 2.0, 0.02, seed=100 + i) for i in range(n_clouds)]
clouds = [capi.Cloud(ctx, s) for s in scans]
out = {"points_per_cloud": len(scans[0]), "n_clouds": n_clouds}


def timed(name, fn, reps=5, per=1):
    fn()
    ctx.synchronize()
    ctx.profile(True)
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.synchronize()
    wall = (time.perf_counter() - t0) / reps
    st = ctx.profile_get()
    ctx.profile(False)
    out[name] = {"wall_us_per_cloud": round(1e6 * wall / per, 1), **{k + "_us_per_cloud": round(1e3 * v["ms"] / reps / per, 2) for k, v in st.items()}}


def drop(cs):
    for c in cs:
        c.drop_index()


timed("normals_single_cloud_with_build", lambda: (clouds[0].drop_index(), clouds[0].normals_knn(20)))
timed("normals_single_cloud_index_kept", lambda: clouds[0].normals_knn(20))
timed("normals_batch_with_build", lambda: (drop(clouds), capi.normals_knn_batch(clouds, 20)), per=n_clouds)
timed("normals_batch_index_kept", lambda: capi.normals_knn_batch(clouds, 20), per=n_clouds)
timed("cov_batch_index_kept", lambda: capi.cov_knn_batch(clouds, 20, 1e-3 + 1e-9 * np.random.rand()), per=n_clouds)
for k in (8, 32):
    timed("normals_batch_index_kept_k%d" % k, lambda: capi.normals_knn_batch(clouds, k), per=n_clouds)
if "--check" in sys.argv:
    from oracle import oracle as O
    idx, d2 = clouds[0].knn(clouds[0], 20)
    io, do = O.Tree(O.xyz4(scans[0])).knn(O.xyz4(scans[0]), 20, threads=16)
    out["knn_bit_exact_vs_oracle_100k"] = bool((idx == io).all() and (d2 == do).all())
print(json.dumps(out, indent=1))
if "--json" in sys.argv:
    json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
