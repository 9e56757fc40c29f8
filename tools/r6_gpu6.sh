mkdir -p gpurun_out/r6f
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_align.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r6f/pytest.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/r6f/pytest.txt
tail -6 gpurun_out/r6f/pytest.txt
bash tools/trace_index.sh > $GRAFT_REPO_ROOT/gpurun_out/r6f/index_trace.txt 2>&1
cd $GRAFT_REPO_ROOT; cat gpurun_out/r6f/index_trace.txt
python bench.py --quick > gpurun_out/r6f/bench_quick.json 2> gpurun_out/r6f/bench_quick.err; cat gpurun_out/r6f/bench_quick.json | cut -c1-300
bash tools/trace_locus_stream.sh > $GRAFT_REPO_ROOT/gpurun_out/r6f/locus_trace.txt 2>&1
cd $GRAFT_REPO_ROOT; cat gpurun_out/r6f/locus_trace.txt
