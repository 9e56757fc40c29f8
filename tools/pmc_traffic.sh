# HBM-side traffic of the sweep kernel (roofline.traffic), as MI355X_MICROARCH.md's HBM section prescribes: separate rocprofv3 --pmc
# passes with --kernel-trace only; the SIZED fabric request counters of the L2 (TCC_EA0_RDREQ_{32,64,128}B, WRREQ / WRREQ_64B) instead of
# FETCH_SIZE, which tallies 128-B requests at 64 B on gfx950.  Program: tools/probe_iter_times.py (32 pairs; the profiled alignment runs in
# one scheduler group = launches of 32 jobs, the warm-up in four groups of 8).  Writes gpurun_out/pmc/traffic.json + the raw per-dispatch values.
cd /tmp && export TMPDIR=/tmp
# (counter passes run with the runtime's default of four hardware queues: the program keeps one scheduler group in flight, so the queue count
# does not enter what is measured, and it is the configuration these passes have always been collected in)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GRAFT_REPO_ROOT=$R
mkdir -p $R/gpurun_out/pmc
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmct_$i
  GPU_MAX_HW_QUEUES=4 timeout 150 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmct_$i -o run --output-format csv -- python $R/tools/probe_iter_times.py > /tmp/pmct_$i.log 2>&1
  cp $(find /tmp/pmct_$i -name "*counter_collection.csv" | head -1) $R/gpurun_out/pmc/traffic_pass$i.csv
done
python - <<'PY'
import csv, json, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
def load(path, pats):
    """per dispatch (in launch order) of the kernels matching any pattern: (kernel tag, counters)"""
    rows = collections.defaultdict(lambda: collections.defaultdict(float))
    tag = {}
    for r in csv.DictReader(open(path)):
        for p in pats:
            if p in r["Kernel_Name"]:
                rows[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
                tag[int(r["Dispatch_Id"])] = p
    return [(tag[k], rows[k]) for k in sorted(rows)]
def rbytes(d):
    other = d["TCC_EA0_RDREQ_sum"] - d["TCC_EA0_RDREQ_32B_sum"] - d["TCC_EA0_RDREQ_64B_sum"] - d["TCC_EA0_RDREQ_128B_sum"]
    return 128 * d["TCC_EA0_RDREQ_128B_sum"] + 64 * d["TCC_EA0_RDREQ_64B_sum"] + 32 * d["TCC_EA0_RDREQ_32B_sum"] + 64 * max(other, 0.0)
def wbytes(d):
    return 64 * d["TCC_EA0_WRREQ_64B_sum"] + 32 * (d["TCC_EA0_WRREQ_sum"] - d["TCC_EA0_WRREQ_64B_sum"])
out = {}
# ---- the sweep ("nn_sweep" in bench.py's profile): k_sweep_fused for the first sweeps of a pair, k_late + k_walk afterwards.
# The profiled alignment is the LAST of the program: its 20 sweeps = the last F fused launches + the last L (k_late, k_walk) pairs.
SW = ("k_sweep_fused", "k_late", "k_walk")
rd, wr = load(R + "/gpurun_out/pmc/traffic_pass1.csv", SW), load(R + "/gpurun_out/pmc/traffic_pass2.csv", SW)
if rd and wr and len(rd) == len(wr):
    # walk the launches backwards until 20 sweeps are collected (a sweep = one k_sweep_fused launch, or one k_late launch with the k_walk behind it)
    seq = [(t, rbytes(a), wbytes(b), b["TCC_HIT_sum"], b["TCC_MISS_sum"]) for (t, a), (_, b) in zip(rd, wr)]
    prof = collections.defaultdict(list)
    sweeps = 0
    for t, r, w, h, m in reversed(seq):
        if sweeps == 20 and t != "k_walk":
            break
        prof[t].insert(0, (r, w, h, m))
        if t in ("k_sweep_fused", "k_late"):
            sweeps += 1
        if sweeps == 20 and t in ("k_sweep_fused", "k_late"):
            break
    F, L = len(prof["k_sweep_fused"]), len(prof["k_late"])
    prof["k_walk"] = prof["k_walk"][-L:] if L else []
    tot = sum(r + w for v in prof.values() for r, w, _, _ in v)
    late = [(prof["k_late"][i][0] + prof["k_late"][i][1]) + (prof["k_walk"][i][0] + prof["k_walk"][i][1]) for i in range(max(0, L - 10), L)] if L else []
    hit = sum(h for v in prof.values() for _, _, h, _ in v); miss = sum(m for v in prof.values() for _, _, _, m in v)
    out["nn_sweep"] = {"jobs_per_launch": 32, "sweeps_profiled_alignment": sweeps, "fused_launches": F, "late_walk_launch_pairs": L,
                       "hbm_bytes_per_launch": tot / max(sweeps, 1),
                       "fused_bytes_per_launch": [round(r + w) for r, w, _, _ in prof.get("k_sweep_fused", [])],
                       "k_late_bytes_per_launch": [round(r + w) for r, w, _, _ in prof.get("k_late", [])],
                       "k_walk_bytes_per_launch": [round(r + w) for r, w, _, _ in prof.get("k_walk", [])],
                       "late_sweep_bytes": (sum(late) / len(late)) if late else None,
                       "l2_hit_rate_profiled_alignment": hit / max(hit + miss, 1.0)}
for pat, name in (("k_seed", "nn_seed"), ("k_moments_final", "moments_final")):
    rd, wr = load(R + "/gpurun_out/pmc/traffic_pass1.csv", (pat,)), load(R + "/gpurun_out/pmc/traffic_pass2.csv", (pat,))
    n = min(len(rd), len(wr))
    if n == 0:
        continue
    per = [rbytes(rd[k][1]) + wbytes(wr[k][1]) for k in range(n)]
    out[name] = {"dispatches": n, "jobs_per_launch": 32, "hbm_bytes_per_launch": per[-1]}
out["_method"] = ("rocprofv3 --kernel-trace --output-format csv --pmc <set> (two separate passes, no sys/hip traces) on tools/probe_iter_times.py (32 pairs, one scheduler group); bytes = "
                  "128*RDREQ_128B + 64*RDREQ_64B + 32*RDREQ_32B + 64*(other reads) + 64*WRREQ_64B + 32*(WRREQ - WRREQ_64B), TCC_EA0 counters summed over the XCDs; "
                  "nn_sweep.hbm_bytes_per_launch = all bytes of k_sweep_fused + k_late + k_walk of the profiled alignment / its 20 sweeps (32 jobs each)")
import hashlib
out["_lib_sha256"] = hashlib.sha256(open(os.environ.get("LH_LIB") or (R + "/locus_amd/csrc/liblocus_hip.so"), "rb").read()).hexdigest()   # which binary these counters describe
json.dump(out, open(R + "/gpurun_out/pmc/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
