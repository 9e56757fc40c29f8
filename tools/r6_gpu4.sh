mkdir -p gpurun_out/r6d
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r6d/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/r6d/pytest_gpu.txt
tail -15 gpurun_out/r6d/pytest_gpu.txt
timeout 1500 python bench.py > gpurun_out/r6d/bench_full.json 2> gpurun_out/r6d/bench_full.err
echo "bench rc=$?"
tail -c 1500 gpurun_out/r6d/bench_full.err
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r6d/bench_full.json') if x.startswith('{')]
if l:
    r=json.loads(l[-1])
    print(json.dumps({k:r.get(k) for k in ('value','ms_per_step','all_ok')}))
    print(json.dumps(r['roofline'].get('per_kernel')), r['roofline']['frac'], r['roofline']['design_compulsory']['frac'])
    print(json.dumps(r.get('cpu_baseline',{}).get('sample')))
    print(json.dumps(r.get('production',{}).get('locus_per_scan'))[:900])
    print(json.dumps(r.get('config3_submap'))[:1500])
    print(r.get('extras_error'))
PY
