# rocprofv3 kernel trace of one 32-pair group (tools/probe_iter_times.py, host-driven loop): durations of the sweep kernels per launch,
# in launch order (the last profiled alignment batch).  usage (GPU box): bash tools/trace_sweeps.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kt
LH_PROBE_SOLVER=1 timeout 200 rocprofv3 --kernel-trace -d /tmp/kt -o run --output-format csv -- python $R/tools/probe_iter_times.py > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("lh::", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
for pat in ("k_sweep_coop", "k_sweep_fused", "k_seed", "k_late", "k_walk_coop", "k_walk", "k_moments_final"):
    v = [(s, e) for s, e, n in rows if (n == pat if pat == "k_walk" else n.startswith(pat))]
    v = v[-20:] if pat in ("k_sweep_coop", "k_sweep_fused", "k_moments_final", "k_seed") else v
    half = v[len(v) // 2:] if pat in ("k_late", "k_walk", "k_walk_coop") else v   # the second (profiled) batch
    print("%-18s n=%3d  us:" % (pat, len(half)), " ".join("%.0f" % ((e - s) / 1e3) for s, e in half))
# gaps: time from the end of k_late to the start of k_walk, and k_walk end -> final start (last batch)
late = [(s, e) for s, e, n in rows if n.startswith("k_late")]
walk = [(s, e) for s, e, n in rows if n.startswith("k_walk")]
fin = [(s, e) for s, e, n in rows if n.startswith("k_moments_final")]
if late and walk:
    g1 = [(w[0] - l[1]) / 1e3 for l, w in zip(late, walk)]
    print("gap k_late -> k_walk us (last 10):", " ".join("%.1f" % x for x in g1[-10:]))
PY
