"""per-kernel duration summary of a rocprofv3 --kernel-trace csv: mean over all dispatches, and mean over the LAST k dispatches
(the certificate-only iterations when the traced program is tools/probe_iter_times.py)"""
import csv, sys, collections
path = sys.argv[1]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
pats = sys.argv[3].split(",") if len(sys.argv) > 3 else None
d = collections.defaultdict(list)
with open(path) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        d[name].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for name, v in sorted(d.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
    if pats and not any(p in name for p in pats):
        continue
    v.sort()
    dur = [x[1] for x in v]
    print("%-40s calls %4d  mean %8.1f us  last-%d mean %8.1f us  first %8.1f us  total %9.1f us" %
          (name[:40], len(dur), sum(dur) / len(dur) / 1e3, k, sum(dur[-k:]) / len(dur[-k:]) / 1e3, dur[0] / 1e3, sum(dur) / 1e3))
