# memory-system counters of the walking kernels (round 3: what does a wave step of the all-walk sweeps wait for?): vector-L1 (TCP)
# accesses / misses / requests to L2 and their latency, stall cycles of the address and data paths, translation misses, and how long the
# waves wait.  One rocprofv3 pass per counter group (gfx950 collects few TCP counters at a time).  Program: tools/probe_iter_times.py.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
OUT=$R/gpurun_out/pmc/memsys.txt
: > $OUT
pass() {
  name=$1; shift
  rm -rf /tmp/pmcm_$name
  timeout 200 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmcm_$name -o run --output-format csv -- python $R/tools/probe_iter_times.py > /tmp/pmcm_$name.log 2>&1
  f=$(find /tmp/pmcm_$name -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then echo "pass $name: no counter file"; tail -5 /tmp/pmcm_$name.log; return; fi
  python - "$f" "$name" <<'PY' | tee -a $OUT
import csv, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(float)); tag = {}
for r in csv.DictReader(open(sys.argv[1])):
    for p in ("k_sweep_fused", "k_late", "k_walk", "k_seed"):
        if p in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"]); tag[int(r["Dispatch_Id"])] = p
per = collections.defaultdict(list)
for k in sorted(rows):
    per[tag[k]].append(rows[k])
print("== pass", sys.argv[2])
for p in ("k_seed", "k_sweep_fused", "k_late", "k_walk"):
    v = per.get(p, [])
    v = v[len(v) // 2:][:5]      # the profiled alignment's first dispatches of that kernel (fused: its three all-walk sweeps)
    if not v: continue
    for name in sorted(v[0]):
        print("%-14s %-34s %s" % (p, name, " ".join("%.4g" % d[name] for d in v)))
PY
}
pass sq    SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
pass sq2   SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY SQ_INST_CYCLES_SALU
# (the three TCP / TA groups below did not finish inside their 200 s on the round-3 boxes -- the derived *_sum counters of the vector L1
#  serialise the run -- and each cost its whole timeout: run them one at a time, with a larger limit, only when the question needs them)
if [ -n "$PMC_MEMSYS_L1" ]; then
pass tcp1  TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum
pass tcp2  TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
pass ta    TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum
fi
pass tlb   TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TD_TD_BUSY_sum TD_TC_STALL_sum
echo "== source order A/B (per-iteration sweep us of one 32-pair group)" | tee -a $OUT
for s in 0 1 2; do echo "LH_PROBE_SORT=$s" | tee -a $OUT; LH_PROBE_SORT=$s python $R/tools/probe_iter_times.py 2>&1 | grep "per-iteration" | tee -a $OUT; done
