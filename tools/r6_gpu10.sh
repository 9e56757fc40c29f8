export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_align.py -m gpu -x -q 2>&1 | tail -5
run() { python bench.py --quick 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1', r['value'], r['natural'], r['ms_per_step'], r['iters'])"; }
LH_WALK_COOP_UNTIL=0 run until0
LH_WALK_COOP_UNTIL=7 run until7
LH_WALK_COOP_UNTIL=5 run until5
LH_WALK_COOP_UNTIL=9 run until9
LH_WALK_COOP_UNTIL=99 run until99
LH_WALK_COOP_UNTIL=0 run until0
LH_WALK_COOP_UNTIL=7 run until7
LH_WALK_COOP_UNTIL=7 bash tools/trace_sweeps.sh 2>&1 | grep -E "k_walk|k_late|k_sweep"
