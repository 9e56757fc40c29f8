# lane utilisation of the walking kernels: SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU per dispatch (the average number of
# active lanes of a vector instruction) for the fused sweep and for k_walk, plus VALU busy.  Program: tools/probe_iter_times.py.
cd /tmp && export TMPDIR=/tmp
# (counter passes run with the runtime's default of four hardware queues: the program keeps one scheduler group in flight, so the queue count
# does not enter what is measured, and it is the configuration these passes have always been collected in)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
rm -rf /tmp/pmcl
GPU_MAX_HW_QUEUES=4 timeout 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES -d /tmp/pmcl -o run --output-format csv -- python $R/tools/probe_iter_times.py > /tmp/pmcl.log 2>&1
f=$(find /tmp/pmcl -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY' | tee $R/gpurun_out/pmc/lanes.txt
import csv, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(float)); tag = {}
for r in csv.DictReader(open(sys.argv[1])):
    for p in ("k_sweep_coop", "k_sweep_fused", "k_late", "k_walk", "k_seed"):
        if p in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"]); tag[int(r["Dispatch_Id"])] = p
per = collections.defaultdict(list)
for k in sorted(rows):
    per[tag[k]].append(rows[k])
for p, v in per.items():
    v = v[len(v) // 2:]      # the profiled alignment (second half of the program's launches)
    print(p, "dispatches (profiled alignment):", len(v))
    print("   active lanes per VALU instruction:", " ".join("%.1f" % (d["SQ_THREAD_CYCLES_VALU"] / max(d["SQ_INSTS_VALU"], 1.0)) for d in v))
    print("   VALU instructions per wave       :", " ".join("%.0f" % (d["SQ_INSTS_VALU"] / max(d["SQ_WAVES"], 1.0)) for d in v))
    print("   wave cycles (x4) per wave        :", " ".join("%.0f" % (d["SQ_WAVE_CYCLES"] / max(d["SQ_WAVES"], 1.0)) for d in v))
    print("   VALU-active cycles / busy cycles  :", " ".join("%.2f" % (d["SQ_ACTIVE_INST_VALU"] / max(d["SQ_BUSY_CYCLES"], 1.0)) for d in v))
PY
