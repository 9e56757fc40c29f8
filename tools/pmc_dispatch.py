"""print the counters of chosen dispatches (by ordinal among the kernels matching a pattern) of a rocprofv3 --pmc csv"""
import csv, sys, collections
path, pat = sys.argv[1], sys.argv[2]
which = [int(x) for x in sys.argv[3].split(",")]
rows = collections.defaultdict(lambda: collections.defaultdict(float))
with open(path) as f:
    for r in csv.DictReader(f):
        if pat in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
order = sorted(rows)
print("kernel ~", pat, "dispatches", len(order))
names = sorted({c for d in rows.values() for c in d})
for c in names:
    print("  %-30s " % c + " ".join("#%d %14.0f" % (w, rows[order[w]][c]) for w in which if -len(order) <= w < len(order)))
