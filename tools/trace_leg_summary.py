#!/usr/bin/env python3
"""Per-kernel duration summary of one roctx range of a rocprofv3 run of bench.py.

usage: trace_leg_summary.py <kernel_trace.csv> <marker_api_trace.csv> [range_name=profile_leg]

bench.py brackets its legs with roctx ranges ("timed_region", "profile_leg").  The profile leg runs ONE scheduler group
with HIP events around every launch (no overlap), and roofline.avg_launch_us is measured there, so the mean duration of
the dominant kernel inside that range in the rocprof trace must agree with it.  The whole-command kernel_stats.csv mixes in
the timed region's launches, which overlap on four streams and are individually longer."""
import csv
import collections
import sys

ktrace, mtrace = sys.argv[1], sys.argv[2]
rname = sys.argv[3] if len(sys.argv) > 3 else "profile_leg"
lo = hi = None
for r in csv.DictReader(open(mtrace)):
    fn = r.get("Function", "")
    if fn == rname or r.get("Message", "") == rname:   # (exact: "profile_leg" must not pick "profile_leg_natural")
        lo, hi = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
if lo is None:
    sys.exit("range %s not found in %s" % (rname, mtrace))
d = collections.defaultdict(list)
for r in csv.DictReader(open(ktrace)):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s >= lo and e <= hi:
        d[r["Kernel_Name"].split("(")[0].replace("void ", "")].append((e - s) / 1e3)
print("range %s: %.3f ms, %d kernel launches" % (rname, (hi - lo) / 1e6, sum(len(v) for v in d.values())))
print("%-44s %7s %10s %10s %10s %12s" % ("kernel", "calls", "mean_us", "min_us", "max_us", "total_us"))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print("%-44s %7d %10.2f %10.2f %10.2f %12.1f" % (k[:44], len(v), sum(v) / len(v), min(v), max(v), sum(v)))
