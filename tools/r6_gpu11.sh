export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { python bench.py --quick 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1', r['value'], r['natural'], r['ms_per_step'], r['iters'])"; }
for v in grid6 grid7; do
  LH_LIB=$R/ab_libs/liblocus_hip_$v.so timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_align.py -m gpu -x -q 2>&1 | tail -2
done
run base
LH_LIB=$R/ab_libs/liblocus_hip_grid6.so run grid6
LH_LIB=$R/ab_libs/liblocus_hip_grid7.so run grid7
run base
LH_LIB=$R/ab_libs/liblocus_hip_grid6.so run grid6
LH_LIB=$R/ab_libs/liblocus_hip_grid7.so run grid7
LH_LIB=$R/ab_libs/liblocus_hip_grid6.so bash tools/trace_sweeps.sh 2>&1 | grep -E "k_walk |k_late|k_sweep_fused"
cd $R; LH_LIB=$R/ab_libs/liblocus_hip_grid7.so bash tools/trace_sweeps.sh 2>&1 | grep -E "k_walk |k_late|k_sweep_fused"
cd $R; LH_LIB=$R/ab_libs/liblocus_hip_grid6.so bash tools/trace_index.sh 2>&1 | grep -E "k_key_b|k_nodex_b"
cd $R; LH_LIB=$R/ab_libs/liblocus_hip_grid7.so bash tools/trace_index.sh 2>&1 | grep -E "k_key_b|k_nodex_b"
