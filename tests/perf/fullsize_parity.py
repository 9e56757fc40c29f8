#!/usr/bin/env python3
"""Full-size parity DISTRIBUTION for the benched configuration (runs on the GPU box): for the first N bench pairs (default 64),
the pose of lh_gicp_align_batch -- cost_mode 0 (host loop) and cost_mode 1 through the DEVICE-DRIVEN loop with 32 pairs in
flight (k_solve, group admission, two streams: the path bench.py times) -- against both builds of the reference restatement
(oracle variant 0 = no FMA, 1 = FMA-contracted float T*p, gicp.hpp:382), next to the distance between those two builds (the
reference's own noise floor).  Quantiles (median, p90, max) per column, and every pair beyond 2.5e-4 m named with the
reference-vs-reference distance OF THAT PAIR.  Prints one JSON object.

    python tests/perf/fullsize_parity.py [n_pairs=64] > gpurun_out/fullsize_parity.json
"""
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from locus_amd import synth  # noqa: E402


def _gen(seed):
    return synth.scan_pair(n_rings=64, n_az=1563, scale=2.0, noise=0.02, seed=seed)


def main():
    n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    seeds = [10 + 2 * p for p in range(n_pairs)]   # bench.py gen_pairs_host(), rank 0
    import multiprocessing as mp
    with mp.get_context("fork").Pool(min(16, os.cpu_count() or 1)) as pool:   # before any GPU runtime exists in this process
        host = pool.map(_gen, seeds, chunksize=2)
    from locus_amd import capi
    from oracle import oracle as O

    def err(A16, B16):
        A, B = O.T_to_mat(A16), O.T_to_mat(B16)
        return float(np.abs(A[:3, 3] - B[:3, 3]).max()), float(np.abs(A[:3, :3] - B[:3, :3]).max())

    cores = os.cpu_count() or 4
    omp = 4
    workers = max(1, min(n_pairs, cores // omp))
    ctx = capi.Context(0)
    L = O.lib()
    kw = dict(max_iterations=20, max_inner_iterations=20, corr_dist=1.0, transformation_epsilon=1e-12, rotation_epsilon=1e-12)
    S, T, inputs = [], [], []
    for src, tgt, _ in host:
        cs, ct = capi.Cloud(ctx, src), capi.Cloud(ctx, tgt)
        cs.normals_knn(20)
        ct.normals_knn(20)
        a, b = cs.download(), ct.download()
        inputs.append((O.xyz4(src), O.nrm4(np.stack([a["normal_x"], a["normal_y"], a["normal_z"]], 1)),
                       O.xyz4(tgt), O.nrm4(np.stack([b["normal_x"], b["normal_y"], b["normal_z"]], 1))))
        S.append(cs)
        T.append(ct)
    ref = []
    for v in (0, 1):   # the variant switch is process-wide: all pairs of one variant concurrently (ctypes releases the GIL), then the other
        L.lo_set_cost_variant(v)
        with ThreadPoolExecutor(workers) as ex:
            ref.append(list(ex.map(lambda a: O.gicp_align(a[0], a[1], a[2], a[3], O.default_params(num_threads=omp, **kw), want_trace=False), inputs)))
    L.lo_set_cost_variant(0)
    rows = [dict(seed=s) for s in seeds]
    for mode in (0, 1):
        out = capi.align_batch(ctx, capi.default_params(cost_mode=mode, **kw), S, T, max_in_flight=min(n_pairs, 32))
        for p in range(n_pairs):
            rows[p]["mode%d_vs_ref" % mode] = err(out[p]["T"], ref[0][p]["T"])
            rows[p]["mode%d_vs_ref_fma" % mode] = err(out[p]["T"], ref[1][p]["T"])
            rows[p]["mode%d_iterations" % mode] = int(out[p]["iterations"])
    for p in range(n_pairs):
        rows[p]["ref_vs_ref_fma"] = err(ref[0][p]["T"], ref[1][p]["T"])
        rows[p]["ref_iterations"] = [int(ref[0][p]["iterations"]), int(ref[1][p]["iterations"])]

    def stats(key):
        dt = np.array([r[key][0] for r in rows])
        dr = np.array([r[key][1] for r in rows])
        return {"median_dt": float(np.median(dt)), "p90_dt": float(np.quantile(dt, 0.9)), "max_dt": float(dt.max()),
                "median_dR": float(np.median(dr)), "p90_dR": float(np.quantile(dr, 0.9)), "max_dR": float(dr.max()),
                "pairs_within_1e-4": int((dt <= 1e-4).sum()), "pairs_within_2.5e-4": int((dt <= 2.5e-4).sum())}
    summary = {k: stats(k) for k in ("mode0_vs_ref", "mode1_vs_ref", "mode1_vs_ref_fma", "ref_vs_ref_fma")}
    outliers = [{"seed": r["seed"], "mode1_vs_ref_dt": r["mode1_vs_ref"][0], "ref_vs_ref_fma_dt": r["ref_vs_ref_fma"][0],
                 "ref_iterations": r["ref_iterations"], "mode1_iterations": r["mode1_iterations"]}
                for r in rows if r["mode1_vs_ref"][0] > 2.5e-4]
    print(json.dumps({"workload": "bench pairs (seeds 10, 12, ...), 100 032 pts, 20 forced iterations, odometry params; (|dt| m, |dR|) per pair; "
                                  "cost_mode 1 through the device-driven loop, %d pairs in flight" % min(n_pairs, 32),
                      "n_pairs": n_pairs, "summary": summary, "mode1_pairs_beyond_2.5e-4": outliers, "pairs": rows}, indent=1))


if __name__ == "__main__":
    main()
