export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
LH_LIB=$R/ab_libs/liblocus_hip_grid6.so timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "sweep or nn1 or grid or index" 2>&1 | grep -E "^E |FAILED|Error|assert" | head -30
