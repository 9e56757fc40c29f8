# Where LOCUS's whole per-scan work (locus_amd/host/locus_stream: odometry + map neighbours + MeasurementUpdate + keyframes; host and device-resident surfaces) goes:
# rocprofv3 kernel + HIP API trace of the stream, summed per API / kernel name and divided by the number of updates.
# usage (GPU box): bash tools/trace_locus_stream.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python - <<PY
import sys, numpy as np
sys.path.insert(0, "$R")
from locus_amd import capi, synth
import bench
ctx = capi.Context(0)
leaf, scans = 0.35, []
for i in range(40):
    pose = synth.pose_matrix(0.12 * i, 0.03 * np.sin(0.5 * i), 0.0, 0.0, 0.0, 0.01 * i)
    pts = synth.scan(pose, 16, 1800, (-15.0, 15.0), 1.0, 0.02, seed=900 + i)
    raw = capi.Cloud(ctx, capi.make_pointxyzi(pts))
    c = raw.voxel_grid(leaf)
    for _ in range(6 if i == 0 else 0):
        if 2800 <= len(c) <= 3200: break
        leaf *= (len(c) / 3000.0) ** 0.6
        c = raw.voxel_grid(leaf)
    c.normals_knn(20)
    scans.append(bench.host_pointf(c))
with open("/tmp/scans.bin", "wb") as f:
    f.write(np.int32(len(scans)).tobytes())
    for a in scans:
        f.write(np.int32(len(a)).tobytes()); f.write(a.tobytes())
print("scans written", len(scans))
PY
rm -rf /tmp/ptrace
timeout 120 rocprofv3 --kernel-trace --hip-runtime-trace -d /tmp/ptrace -o run --output-format csv -- $R/locus_amd/host/locus_stream /tmp/scans.bin 3 > /tmp/ptrace.out 2>/tmp/ptrace.err
tail -1 /tmp/ptrace.out | cut -c1-260
python - <<'PY'
import csv, glob, collections
kt = glob.glob("/tmp/ptrace/**/*kernel_trace.csv", recursive=True)
ht = glob.glob("/tmp/ptrace/**/*hip_api_trace.csv", recursive=True)
U = 2 * 39 + 2 * 7   # scans of the two timed passes (host surface, device-resident) + the two warm-up passes
if kt:
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(kt[0])):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("lh::", "")
        d[n][0] += 1; d[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("kernels per update (us, launches):")
    for n, (c, t) in sorted(d.items(), key=lambda kv: -kv[1][1])[:24]:
        print("  %-34s %7.1f us  %5.1f launches" % (n, t / U, c / U))
    print("  total kernel time per update %.1f us" % (sum(v[1] for v in d.values()) / U))
if ht:
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(ht[0])):
        d[r["Function"]][0] += 1; d[r["Function"]][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("HIP API per update (us, calls):")
    for n, (c, t) in sorted(d.items(), key=lambda kv: -kv[1][1])[:24]:
        print("  %-34s %7.1f us  %5.1f calls" % (n, t / U, c / U))
PY
