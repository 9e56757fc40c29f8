# EXPERIMENT: shadow launches of k_late with parts removed (LH_EXP_LATE codes; bit 0: no Mahalanobis / moment math, bit 1: no LDS staging + MFMA,
# bit 2: no row store) next to the real one; per-dispatch durations from the kernel trace, in launch order per iteration
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kt
LH_EXP_LATE=$1 timeout 200 rocprofv3 --kernel-trace -d /tmp/kt -o run --output-format csv -- python $R/tools/probe_iter_times.py > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python - "$f" "$1" <<'PY'
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    if "k_late" in r["Kernel_Name"]:
        rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
rows.sort()
codes = sys.argv[2].split(",") + ["real"]
k = len(codes)
d = [x[1] for x in rows]
n_it = len(d) // k
print("iterations", n_it)
for j, c in enumerate(codes):
    per = d[j::k]
    last = per[-20:]   # the profiled group's 20 iterations (the warm-up group's come first)
    print("variant %-5s per-iteration us:" % c, " ".join("%.0f" % v for v in last))
PY
