# K3 (block k-NN normals) and K1 (voxel grid) under rocprofv3: kernel durations (--kernel-trace --stats), instruction counters and the
# HBM-side traffic (sized TCC_EA0 request counters: MI355X_MICROARCH.md's HBM section), each in its own pass, no sys/hip traces.
# Program: tools/probe_k3_k1.py.  Writes gpurun_out/pmc_filters/{kernel_stats.csv, counters.txt, traffic.json}.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_filters
mkdir -p $O
rm -rf /tmp/pf_* $O/pass*.csv
timeout 280 rocprofv3 --kernel-trace --stats -d /tmp/pf_stats -o run --output-format csv -- python $R/tools/probe_k3_k1.py > /tmp/pf_stats.log 2>&1
cp $(find /tmp/pf_stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
i=0
PASSES=${PMC_PASSES:-4}
for set in "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); [ $i -gt $PASSES ] && break
  timeout 280 rocprofv3 --kernel-trace --pmc $set -d /tmp/pf_$i -o run --output-format csv -- python $R/tools/probe_k3_k1.py > /tmp/pf_$i.log 2>&1
  cp $(find /tmp/pf_$i -name "*counter_collection.csv" | head -1) $O/pass$i.csv
done
python - <<'PY'
import csv, json, os, collections
R = os.environ["GRAFT_REPO_ROOT"]; O = R + "/gpurun_out/pmc_filters"
KER = ("k_knn_block", "k_knn_redo", "k_voxel", "k_rs_", "k_leaves_b", "k_radix_b", "k_nodex_b", "k_key_b", "k_bbox_b", "k_leafcell_b", "k_scan", "k_sort", "k_hist", "k_scatter")
def load(path):
    rows = collections.defaultdict(lambda: collections.defaultdict(float)); name = {}
    for r in csv.DictReader(open(path)):
        rows[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"]); name[int(r["Dispatch_Id"])] = r["Kernel_Name"].split("(")[0]
    per = collections.defaultdict(list)
    for k in sorted(rows):
        per[name[k]].append(rows[k])
    return per
txt = []
a, b = load(O + "/pass1.csv"), load(O + "/pass2.csv")
for kn in sorted(a):
    v, w = a[kn], b.get(kn, [])
    d = v[-1]; e = w[-1] if w else collections.defaultdict(float)
    waves = max(d["SQ_WAVES"], 1.0)
    txt.append("%-90s launches %3d | waves %8.0f  VALU/wave %7.0f  lanes/VALU %5.1f  SALU/wave %6.0f  SMEM/wave %5.0f  LDS/wave %5.0f  VMEM rd/wr per wave %5.0f/%4.0f  VALU busy %4.1f%% of SQ busy"
               % (kn[:90], len(v), waves, d["SQ_INSTS_VALU"] / waves, d["SQ_THREAD_CYCLES_VALU"] / max(d["SQ_INSTS_VALU"], 1.0), e["SQ_INSTS_SALU"] / waves, e["SQ_INSTS_SMEM"] / waves,
                  e["SQ_INSTS_LDS"] / waves, e["SQ_INSTS_VMEM_RD"] / waves, e["SQ_INSTS_VMEM_WR"] / waves, 100.0 * d["SQ_ACTIVE_INST_VALU"] * 4 / max(d["SQ_BUSY_CYCLES"], 1.0)))
open(O + "/counters.txt", "w").write("\n".join(txt) + "\n")
print("\n".join(txt))
if not os.path.exists(O + "/pass4.csv"): raise SystemExit
rd, wr = load(O + "/pass3.csv"), load(O + "/pass4.csv")
def rbytes(d):
    other = d["TCC_EA0_RDREQ_sum"] - d["TCC_EA0_RDREQ_32B_sum"] - d["TCC_EA0_RDREQ_64B_sum"] - d["TCC_EA0_RDREQ_128B_sum"]
    return 128 * d["TCC_EA0_RDREQ_128B_sum"] + 64 * d["TCC_EA0_RDREQ_64B_sum"] + 32 * d["TCC_EA0_RDREQ_32B_sum"] + 64 * max(other, 0.0)
def wbytes(d):
    return 64 * d["TCC_EA0_WRREQ_64B_sum"] + 32 * (d["TCC_EA0_WRREQ_sum"] - d["TCC_EA0_WRREQ_64B_sum"])
out = {}
for kn in sorted(rd):
    if kn in wr and len(wr[kn]) == len(rd[kn]):
        r, w = rd[kn][-1], wr[kn][-1]
        out[kn] = {"launches": len(rd[kn]), "read_bytes_last_launch": rbytes(r), "write_bytes_last_launch": wbytes(w), "l2_hit_rate": w["TCC_HIT_sum"] / max(w["TCC_HIT_sum"] + w["TCC_MISS_sum"], 1.0)}
out["_method"] = "rocprofv3 --kernel-trace --pmc, separate passes; bytes = 128*RDREQ_128B + 64*RDREQ_64B + 32*RDREQ_32B + 64*other + 64*WRREQ_64B + 32*(WRREQ - WRREQ_64B), TCC_EA0 summed over XCDs; last launch of each kernel"
json.dump(out, open(O + "/traffic.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if "knn" in k or "voxel" in k}, indent=1))
PY
