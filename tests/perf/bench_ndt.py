#!/usr/bin/env python3
"""registration_method "ndt" on a config-2 style pair (100 032-point scans, small inter-scan motion): HIP path vs the CPU
restatement (oracle/locus_oracle_ndt.c, single-threaded; the reference parallelises computeDerivatives over points with OMP).
Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from locus_amd import capi, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    ctx = capi.Context(0)
    delta = synth.pose_matrix(0.04, -0.03, 0.01, 0.002, -0.001, 0.006)
    src, tgt, delta = synth.scan_pair(n_rings=64, n_az=1563, scale=2.0, noise=0.02, seed=10, delta=delta)
    out = {"workload": "NDT (KDTREE search), %d-pt scan pair" % len(src), "rows": []}
    for res in (1.0, 2.0):
        P = capi.default_ndt_params(resolution=res, transformation_epsilon=1e-3, max_iterations=30)
        ndt = capi.Ndt(ctx, P)
        cs, ct = capi.Cloud(ctx, src), capi.Cloud(ctx, tgt)
        ndt.set_source(cs)
        ndt.set_target(ct)
        r = ndt.align()                      # warm-up (allocations, target grid)
        ctx.profile(True); ctx.profile_reset()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            ndt.set_target(ct)               # setInputTarget re-initialises the voxel structure every scan
            r = ndt.align()
        dt = (time.perf_counter() - t0) / reps
        st = ctx.profile_get(); ctx.profile(False)
        po = O.ndt_default_params(resolution=res, transformation_epsilon=1e-3, max_iterations=30)
        t0 = time.perf_counter()
        ro = O.ndt_align(O.xyz4(src), O.xyz4(tgt), po)
        dto = time.perf_counter() - t0
        T, To = O.T_to_mat(r["T"]), O.T_to_mat(ro["T"])
        out["rows"].append({"resolution": res, "cells": r["n_cells"], "iterations": r["iterations"], "evaluations": r["evaluations"],
                            "gpu_ms_per_align": round(1e3 * dt, 3), "cpu_oracle_1_thread_s": round(dto, 3), "speedup": round(dto / dt, 1),
                            "max_abs_pose_diff_gpu_vs_cpu": float(np.abs(T - To).max()),
                            "translation_err_vs_truth_m": float(np.abs(T[:3, 3] - delta[:3, 3]).max()),
                            "kernel_ms_per_align": {k: round(v["ms"] / reps, 3) for k, v in sorted(st.items(), key=lambda kv: -kv[1]["ms"])[:6]}})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
