#!/usr/bin/env python3
"""SURVEY.md 8d CPU baseline table: the reference-algorithm restatement (oracle; NOT PCL) on config-2 pairs at
OMP threads 1, 4 (LOCUS Husky default, locus.launch:75-77) and all hardware threads, with the reference's parallelisation
(OMP on the NN loops, serial cost functor) and the "fully parallel" variant (cost functor with an OMP reduction).
Prints one JSON object; run it on the GPU box so the numbers sit beside bench.py's."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from locus_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    ncores = os.cpu_count() or 1
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 25.0
    pairs = []
    for p in range(3):
        src, tgt, _ = synth.scan_pair(n_rings=64, n_az=1563, scale=2.0, noise=0.02, seed=10 + 2 * p)
        s4, t4 = O.xyz4(src), O.xyz4(tgt)
        pairs.append((s4, O.normals_knn(s4, 20, threads=ncores), t4, O.normals_knn(t4, 20, threads=ncores)))
    rows = []
    for threads in sorted({1, 4, ncores}):
        for parallel_cost in (0, 1):
            if threads == 1 and parallel_cost:
                continue
            po = O.default_params(max_iterations=20, max_inner_iterations=20, corr_dist=1.0, transformation_epsilon=1e-12,
                                  rotation_epsilon=1e-12, num_threads=threads, parallel_cost=parallel_cost)
            done, used = 0, 0.0
            while used < budget / 5 and done < len(pairs):
                s4, ns, t4, nt = pairs[done]
                t0 = time.perf_counter()
                r = O.gicp_align(s4, ns, t4, nt, po, want_trace=False)
                used += time.perf_counter() - t0
                done += 1
            rows.append({"threads": threads, "cost_functor": "omp-reduction" if parallel_cost else "serial (reference)",
                         "pairs": done, "seconds": round(used, 2), "pairs_per_s": round(done / used, 4), "iterations": int(r["iterations"])})
    print(json.dumps({"workload": "config 2: 100 032-pt scan pair, 20 forced outer iterations, odometry params", "host_threads": ncores,
                      "kind": "port (oracle/locus_oracle.c)", "rows": rows}))


if __name__ == "__main__":
    main()
