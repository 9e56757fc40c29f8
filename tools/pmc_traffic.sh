# HBM-side traffic of the sweep kernel (roofline.traffic), as MI355X_MICROARCH.md's HBM section prescribes: separate rocprofv3 --pmc
# passes with --kernel-trace only; the SIZED fabric request counters of the L2 (TCC_EA0_RDREQ_{32,64,128}B, WRREQ / WRREQ_64B) instead of
# FETCH_SIZE, which tallies 128-B requests at 64 B on gfx950.  Program: tools/probe_iter_times.py (32 pairs; the profiled alignment runs in
# one scheduler group = launches of 32 jobs, the warm-up in four groups of 8).  Writes gpurun_out/pmc/traffic.json + the raw per-dispatch values.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmct_$i
  timeout 250 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmct_$i -o run --output-format csv -- python $R/tools/probe_iter_times.py > /tmp/pmct_$i.log 2>&1
  cp $(find /tmp/pmct_$i -name "*counter_collection.csv" | head -1) $R/gpurun_out/pmc/traffic_pass$i.csv
done
python - <<'PY'
import csv, json, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
def load(path, pat):
    rows = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        if pat in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    return [rows[k] for k in sorted(rows)]
out = {}
for pat, name in (("k_sweep_fused", "nn_sweep"), ("k_seed", "nn_seed"), ("k_moments_final", "moments_final")):
    rd, wr = load(R + "/gpurun_out/pmc/traffic_pass1.csv", pat), load(R + "/gpurun_out/pmc/traffic_pass2.csv", pat)
    def rbytes(d):
        other = d["TCC_EA0_RDREQ_sum"] - d["TCC_EA0_RDREQ_32B_sum"] - d["TCC_EA0_RDREQ_64B_sum"] - d["TCC_EA0_RDREQ_128B_sum"]
        return 128 * d["TCC_EA0_RDREQ_128B_sum"] + 64 * d["TCC_EA0_RDREQ_64B_sum"] + 32 * d["TCC_EA0_RDREQ_32B_sum"] + 64 * max(other, 0.0)
    def wbytes(d):
        return 64 * d["TCC_EA0_WRREQ_64B_sum"] + 32 * (d["TCC_EA0_WRREQ_sum"] - d["TCC_EA0_WRREQ_64B_sum"])
    n = min(len(rd), len(wr))
    if n == 0:
        continue
    per = [rbytes(rd[k]) + wbytes(wr[k]) for k in range(n)]
    prof = per[-20:] if name == "nn_sweep" else per[-1:]      # the profiled alignment's launches (32 jobs each)
    ent = {"dispatches": n, "jobs_per_launch": 32, "hbm_bytes_per_launch": sum(prof) / len(prof),
           "per_launch_bytes_profiled_alignment": [round(v) for v in prof]}
    if name == "nn_sweep":
        late = per[-10:]
        ent["late_sweep_bytes"] = sum(late) / len(late)
        ent["late_read_bytes"] = sum(rbytes(d) for d in rd[-10:]) / 10
        ent["late_write_bytes"] = sum(wbytes(d) for d in wr[-10:]) / 10
        hit = sum(d["TCC_HIT_sum"] for d in wr[-20:]); miss = sum(d["TCC_MISS_sum"] for d in wr[-20:])
        ent["l2_hit_rate_profiled_alignment"] = hit / max(hit + miss, 1.0)
    out[name] = ent
out["_method"] = ("rocprofv3 --kernel-trace --output-format csv --pmc <set> (two separate passes, no sys/hip traces) on tools/probe_iter_times.py; bytes = "
                  "128*RDREQ_128B + 64*RDREQ_64B + 32*RDREQ_32B + 64*(other reads) + 64*WRREQ_64B + 32*(WRREQ - WRREQ_64B), TCC_EA0 counters summed over the XCDs; "
                  "hbm_bytes_per_launch = mean over the 20 sweep launches (32 jobs each) of the profiled alignment")
json.dump(out, open(R + "/gpurun_out/pmc/traffic.json", "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "per_launch_bytes_profiled_alignment"} if isinstance(v, dict) else v for k, v in out.items()}, indent=1)[:2500])
PY
