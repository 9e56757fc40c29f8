# what a k_solve launch executes: vector / scalar / LDS instructions and cycles per wave, per dispatch of the profiled
# alignment.  One wave per pair, so cycles / instruction is the dependent-issue latency of a lone wave.  Program: tools/probe_iter_times.py.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
rm -rf /tmp/pmcs
GPU_MAX_HW_QUEUES=4 timeout 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU -d /tmp/pmcs -o run --output-format csv -- python $R/tools/probe_iter_times.py > /tmp/pmcs.log 2>&1
f=$(find /tmp/pmcs -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY' | tee $R/gpurun_out/pmc/solve.txt
import csv, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    if "k_solve" in r["Kernel_Name"]:
        rows[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
v = [rows[k] for k in sorted(rows)]
v = v[len(v) // 2:]
print("k_solve dispatches (profiled alignment):", len(v))
for name in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU"):
    print("   %-22s per wave:" % name, " ".join("%.0f" % (d[name] / max(d["SQ_WAVES"], 1.0)) for d in v))
print("   wave cycles per VALU instruction:", " ".join("%.1f" % (d["SQ_WAVE_CYCLES"] / max(d["SQ_INSTS_VALU"], 1.0)) for d in v))
PY
