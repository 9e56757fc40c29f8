#!/usr/bin/env python3
"""Generates tests/golden/config1_chain.json: BASELINE configs[0] (5 k-pt plumbing pair: voxel grid 0.25 m -> k = 20 normals ->
GICP with the odometry parameters) run on the CPU restatement (oracle/, "reference-algorithm restatement": the reference
itself cannot be built here).  The numbers pin the restatement against regressions and give the GPU chain its target."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from locus_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


def chain(threads=4):
    src, tgt, delta = synth.config1_pair()
    zi = np.zeros((src.shape[0], 1), np.float32)
    vs = O.voxel_grid(np.concatenate([src, zi], 1), 0.25)      # CustomVoxelGrid leaf 0.25 (lo_settings.yaml / SURVEY 8d)
    vt = O.voxel_grid(np.concatenate([tgt, zi[: tgt.shape[0]]], 1), 0.25)
    s4, t4 = O.xyz4(vs[:, :3]), O.xyz4(vt[:, :3])
    ns, nt = O.normals_knn(s4, 20, threads=threads), O.normals_knn(t4, 20, threads=threads)   # NormalComputation k = 20
    P = O.default_params(max_iterations=20, corr_dist=1.0, transformation_epsilon=1e-3, num_threads=threads)  # odometry yaml
    r = O.gicp_align(s4, ns, t4, nt, P)
    return dict(src=src, tgt=tgt, delta=delta, vs=vs, vt=vt, ns=ns, nt=nt, result=r)


if __name__ == "__main__":
    c = chain()
    r = c["result"]
    out = {"n_raw": [int(c["src"].shape[0]), int(c["tgt"].shape[0])], "n_voxel": [int(c["vs"].shape[0]), int(c["vt"].shape[0])],
           "T_colmajor": [float(x) for x in r["T"]], "iterations": int(r["iterations"]), "converged": int(r["converged"]),
           "n_corr_last": int(r["n_corr_last"]), "voxel_checksum": [float(c["vs"][:, :3].astype(np.float64).sum()), float(c["vt"][:, :3].astype(np.float64).sum())],
           "generator": "tests/golden/make_config1_golden.py (oracle/locus_oracle.c)"}
    json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "config1_chain.json"), "w"), indent=1)
    print(json.dumps(out))
