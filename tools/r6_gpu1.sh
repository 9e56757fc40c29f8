mkdir -p gpurun_out/r6a
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_align.py -m gpu -x -q > gpurun_out/r6a/pytest_kernels_align.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/r6a/pytest_kernels_align.txt
for c in 1 0; do
  LH_SWEEP_COOP=$c timeout 300 bash tools/trace_sweeps.sh > gpurun_out/r6a/trace_sweeps_coop$c.txt 2>&1
  cd $GRAFT_REPO_ROOT
  LH_SWEEP_COOP=$c timeout 400 python bench.py --quick > gpurun_out/r6a/bench_quick_coop$c.json 2> gpurun_out/r6a/bench_quick_coop$c.err
done
tail -5 gpurun_out/r6a/pytest_kernels_align.txt
cat gpurun_out/r6a/trace_sweeps_coop1.txt gpurun_out/r6a/trace_sweeps_coop0.txt
cat gpurun_out/r6a/bench_quick_coop1.json gpurun_out/r6a/bench_quick_coop0.json | cut -c1-400
