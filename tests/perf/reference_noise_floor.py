#!/usr/bin/env python3
"""The reference algorithm's own float noise floor AT THE BENCH CONFIGURATION (BASELINE configs[1]: 100 032-point scan pairs,
odometry parameters, 20 forced outer iterations).

The reference evaluates T*p in float inside the cost functor (gicp.hpp:382).  Compiling that same source with or without FMA
contraction (-march=native vs. plain x86-64: both legal builds of LOCUS) changes those roundings; the BFGS line search
compares cost values at their last bits, takes other branches, and the outer loop stalls (BFGS "no progress", gradient norm
< 1e-2) at a different point.  oracle/locus_oracle.c restates both builds (lo_set_cost_variant 0 / 1).  The distance between
their results is what "matches the reference to a float tolerance" can mean at this configuration: no bit-different
evaluation of the same cost (our cost_mode 1 included) can be held closer to ONE of the two builds than they are to each
other.

Writes one JSON object (committed as profiles/r02_reference_noise_floor.json); tests/test_gpu_align.py reads the tolerance
of the full-size parity test from the same constant (FLOOR_T / FLOOR_R there).  CPU only: ~3 s per pair and variant.

    python tests/perf/reference_noise_floor.py [n_pairs=16] > profiles/r02_reference_noise_floor.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from locus_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    threads = os.cpu_count() or 1
    L = O.lib()
    rows = []
    for p in range(n_pairs):
        seed = 10 + 2 * p   # bench.py make_pairs(), rank 0
        src, tgt, _ = synth.scan_pair(n_rings=64, n_az=1563, scale=2.0, noise=0.02, seed=seed)
        s4, t4 = O.xyz4(src), O.xyz4(tgt)
        ns, nt = O.normals_knn(s4, 20, threads=threads), O.normals_knn(t4, 20, threads=threads)
        P = O.default_params(max_iterations=20, max_inner_iterations=20, corr_dist=1.0, transformation_epsilon=1e-12,
                             rotation_epsilon=1e-12, num_threads=threads)
        res = []
        for v in (0, 1):
            L.lo_set_cost_variant(v)
            res.append(O.gicp_align(s4, ns, t4, nt, P))
        L.lo_set_cost_variant(0)
        A, B = O.T_to_mat(res[0]["T"]), O.T_to_mat(res[1]["T"])
        ta, tb = res[0]["trace"]["T"], res[1]["trace"]["T"]
        k = min(len(ta), len(tb))
        rows.append({"seed": seed, "dt": float(np.abs(A[:3, 3] - B[:3, 3]).max()), "dR": float(np.abs(A[:3, :3] - B[:3, :3]).max()),
                     "iterations": [int(res[0]["iterations"]), int(res[1]["iterations"])],
                     "per_iteration_max_dT": [float(x) for x in np.abs(ta[:k] - tb[:k]).max(1)]})
        sys.stderr.write("seed %d: |dt| %.2e |dR| %.2e\n" % (seed, rows[-1]["dt"], rows[-1]["dR"]))
    dts = np.array([r["dt"] for r in rows])
    drs = np.array([r["dR"] for r in rows])
    print(json.dumps({
        "what": "reference restatement built without vs with FMA contraction in the cost functor's float T*p (gicp.hpp:382)",
        "workload": "bench pairs of rank 0 (seeds 10, 12, ...): 100 032-pt scans, odometry params, 20 forced outer iterations, k=20 normals (oracle)",
        "n_pairs": n_pairs, "max_dt_m": float(dts.max()), "median_dt_m": float(np.median(dts)), "max_dR": float(drs.max()),
        "median_dR": float(np.median(drs)), "pairs": rows}, indent=1))


if __name__ == "__main__":
    main()
