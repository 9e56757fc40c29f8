mkdir -p gpurun_out/r6g
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_measurement_update.py tests/test_gpu_host_mirror.py tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -4
python - <<'PY' 2>&1 | tail -40
import os, sys, json
sys.path.insert(0, os.getcwd())
from locus_amd import capi
import bench
ctx = capi.Context(0)
r = bench.production_leg(ctx)
print(json.dumps(r["locus_per_scan"], indent=1))
print(r["gpu_promote"]["ms_per_update_median"], r["cpu_4_threads_ms_per_update_median"])
PY
bash tools/trace_index.sh 2>&1 | grep -E "k_boxes|k_leaves|k_leafcell"
