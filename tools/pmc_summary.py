"""summarise a rocprofv3 --pmc counter_collection.csv: per kernel, mean counter value of the first and of the last k dispatches"""
import csv, sys, collections
path, pat = sys.argv[1], sys.argv[2]
k = int(sys.argv[3]) if len(sys.argv) > 3 else 8
rows = collections.defaultdict(lambda: collections.defaultdict(float))  # dispatch -> counter -> value
order = []
with open(path) as f:
    for r in csv.DictReader(f):
        if pat not in r["Kernel_Name"]:
            continue
        d = int(r["Dispatch_Id"])
        if d not in rows:
            order.append(d)
        rows[d][r["Counter_Name"]] += float(r["Counter_Value"])
order.sort()
def mean(ds):
    out = collections.defaultdict(float)
    for d in ds:
        for c, v in rows[d].items():
            out[c] += v / len(ds)
    return out
if not order:
    print("no dispatches match", pat); sys.exit()
first, last, allm = mean(order[:1]), mean(order[-k:]), mean(order)
print("kernel ~", pat, "dispatches", len(order))
for c in sorted(last):
    print("  %-28s first %14.0f   last-%d mean %14.0f   all mean %14.0f" % (c, first[c], k, last[c], allm[c]))
