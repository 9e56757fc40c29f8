#!/usr/bin/env python3
"""Average duration of the bench's profile-leg launches in a rocprofv3 kernel trace.

usage: trace_agreement.py <kernel_trace.csv> <warmup> <steps> [groups_per_step=2] [iters=20]

bench.py launches k_sweep_fused (and k_moments_final) `iters` times per group of 32 pairs: first
(warmup+steps) x groups overlapped on two streams, then the serial profile leg (groups x 2 launches of
each kind per iteration are not overlapped there), then the natural-convergence leg.  The profile leg is
what roofline.avg_launch_us is measured on with HIP events, so its slice of the trace must agree.
"""
import csv
import sys

path, warmup, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
groups = int(sys.argv[4]) if len(sys.argv) > 4 else 2
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
rows = list(csv.DictReader(open(path)))
for name in ("k_sweep_fused", "k_moments_final"):
    r = sorted((int(x["Start_Timestamp"]), int(x["End_Timestamp"])) for x in rows if name in x["Kernel_Name"])
    d = [(e - s) / 1e3 for s, e in r]
    lo = (warmup + steps) * groups * iters
    hi = lo + 2 * groups * iters
    leg = d[lo:hi]
    print(f"{name}: {len(d)} launches; timed region [0:{lo}] avg {sum(d[:lo]) / max(lo, 1):.1f} us; "
          f"profile leg [{lo}:{hi}] avg {sum(leg) / max(len(leg), 1):.1f} us")
