#!/usr/bin/env python3
"""Full-size parity table for the benched configuration (runs on the GPU box): for the first N bench pairs, the pose of
lh_gicp_align_batch in cost_mode 0 and 1 against both builds of the reference restatement (oracle variant 0 = no FMA,
1 = FMA-contracted float T*p), next to the distance between those two builds (the reference's own noise floor).
Prints one JSON object.

    python tests/perf/fullsize_parity.py [n_pairs=8] > gpurun_out/fullsize_parity.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from locus_amd import capi, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


def err(A16, B16):
    A, B = O.T_to_mat(A16), O.T_to_mat(B16)
    return float(np.abs(A[:3, 3] - B[:3, 3]).max()), float(np.abs(A[:3, :3] - B[:3, :3]).max())


def main():
    n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    threads = os.cpu_count() or 4
    ctx = capi.Context(0)
    L = O.lib()
    kw = dict(max_iterations=20, max_inner_iterations=20, corr_dist=1.0, transformation_epsilon=1e-12, rotation_epsilon=1e-12)
    S, T, ref = [], [], []
    for p in range(n_pairs):
        seed = 10 + 2 * p
        src, tgt, _ = synth.scan_pair(n_rings=64, n_az=1563, scale=2.0, noise=0.02, seed=seed)
        cs, ct = capi.Cloud(ctx, src), capi.Cloud(ctx, tgt)
        cs.normals_knn(20)
        ct.normals_knn(20)
        a, b = cs.download(), ct.download()
        ns = O.nrm4(np.stack([a["normal_x"], a["normal_y"], a["normal_z"]], 1))
        nt = O.nrm4(np.stack([b["normal_x"], b["normal_y"], b["normal_z"]], 1))
        rr = []
        for v in (0, 1):
            L.lo_set_cost_variant(v)
            rr.append(O.gicp_align(O.xyz4(src), ns, O.xyz4(tgt), nt, O.default_params(num_threads=threads, **kw), want_trace=False))
        L.lo_set_cost_variant(0)
        S.append(cs)
        T.append(ct)
        ref.append(rr)
    rows = [dict(seed=10 + 2 * p) for p in range(n_pairs)]
    for mode in (0, 1):
        out = capi.align_batch(ctx, capi.default_params(cost_mode=mode, **kw), S, T, max_in_flight=min(n_pairs, 32))
        for p in range(n_pairs):
            rows[p]["mode%d_vs_ref" % mode] = err(out[p]["T"], ref[p][0]["T"])
            rows[p]["mode%d_vs_ref_fma" % mode] = err(out[p]["T"], ref[p][1]["T"])
            rows[p]["mode%d_iterations" % mode] = int(out[p]["iterations"])
    for p in range(n_pairs):
        rows[p]["ref_vs_ref_fma"] = err(ref[p][0]["T"], ref[p][1]["T"])
        rows[p]["ref_iterations"] = [int(ref[p][0]["iterations"]), int(ref[p][1]["iterations"])]

    def col(key):
        return [r[key][0] for r in rows]
    summary = {k: {"median_dt": float(np.median(col(k))), "max_dt": float(np.max(col(k))), "max_dR": float(max(r[k][1] for r in rows))}
               for k in ("mode0_vs_ref", "mode1_vs_ref", "mode1_vs_ref_fma", "ref_vs_ref_fma")}
    print(json.dumps({"workload": "bench pairs (seeds 10, 12, ...), 100 032 pts, 20 forced iterations, odometry params; (|dt| m, |dR|) pairs",
                      "n_pairs": n_pairs, "summary": summary, "pairs": rows}, indent=1))


if __name__ == "__main__":
    main()
