E=gpurun_out/evidence2; mkdir -p $E
python bench.py --gpus 2 --same-gpu --dist-backend gloo --quick --pairs 64 --in-flight 64 --steps 2 --warmup 1 > $E/bench_self_launched_two_ranks.json 2> $E/bench_self_launched_two_ranks.err
python bench.py --gpus 2 --same-gpu --dist-backend gloo --pairs 32 --in-flight 32 --steps 2 --warmup 1 --no-trajectory --no-configs > $E/bench_two_ranks_full_line.json 2> $E/bench_two_ranks_full_line.err
tail -3 $E/bench_two_ranks_full_line.err
python - <<'PY'
import json
for f in ("bench_self_launched_two_ranks", "bench_two_ranks_full_line"):
    l=[x for x in open("gpurun_out/evidence2/%s.json" % f) if x.startswith("{")]
    r=json.loads(l[-1]); sp=r.get("config5_sharded_pair") or {}
    print(f, r["n_gpus"], r["value"], "cpu_baseline" in r, sp.get("status"), sp.get("all_ranks_same_transform_bit_for_bit"), sp.get("ms_per_alignment"), sp.get("error"))
PY
