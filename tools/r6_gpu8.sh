export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_align.py -m gpu -x -q 2>&1 | tail -4
for f in 99 7 5 7 99; do LH_WALK_FEW_FROM=$f python bench.py --quick 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('few_from $f', r['value'], r['natural'], r['iters'])"; done
for w in 1 4 8; do LH_WALK_FEW_WGS=$w python bench.py --quick 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('few_wgs $w', r['value'], r['natural'])"; done
