export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_align.py -m gpu -x -q 2>&1 | tail -3
run() { python bench.py --quick 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1', r['value'], r['natural'], r['ms_per_step'])"; }
LH_WALK_FEW_FROM=99 run base
LH_WALK_FEW_FROM=7 LH_WALK_FEW_WGS=8 run from7_wgs8
LH_WALK_FEW_FROM=7 LH_WALK_FEW_WGS=16 run from7_wgs16
LH_WALK_FEW_FROM=6 LH_WALK_FEW_WGS=16 run from6_wgs16
LH_WALK_FEW_FROM=7 LH_WALK_FEW_WGS=4 run from7_wgs4
LH_WALK_FEW_FROM=99 run base
LH_WALK_FEW_FROM=7 LH_WALK_FEW_WGS=8 run from7_wgs8
