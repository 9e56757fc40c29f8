mkdir -p gpurun_out/r6c
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_measurement_update.py tests/test_gpu_host_mirror.py -m gpu -x -q > gpurun_out/r6c/pytest_mu.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/r6c/pytest_mu.txt
tail -25 gpurun_out/r6c/pytest_mu.txt
python - <<'PY' > gpurun_out/r6c/locus_stream.txt 2>&1
import os, sys, json, subprocess, tempfile, numpy as np
sys.path.insert(0, os.getcwd())
from locus_amd import capi, synth
import bench
ctx = capi.Context(0)
r = bench.production_leg(ctx)
print(json.dumps(r, indent=1))
PY
tail -60 gpurun_out/r6c/locus_stream.txt
