mkdir -p gpurun_out/r6e
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q > gpurun_out/r6e/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/r6e/pytest_gpu.txt
tail -15 gpurun_out/r6e/pytest_gpu.txt
grep -n "^E  \|FAILED" gpurun_out/r6e/pytest_gpu.txt | head -30
