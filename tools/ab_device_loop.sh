# A/B of the scheduler settings on one box (tools: not part of the product): bench.py --quick lines into gpurun_out/ab.log
mkdir -p gpurun_out
: > gpurun_out/ab.log
for cfg in "$@"; do
  name=${cfg%%:*}; rest=${cfg#*:}
  envs=""; args=""
  for w in $rest; do case $w in LH_*) envs="$envs $w";; *) args="$args $w";; esac; done
  echo "== $name" >> gpurun_out/ab.log
  env $envs timeout 150 python bench.py --quick --steps 4 --warmup 2 $args >> gpurun_out/ab.log 2>/dev/null
done
cat gpurun_out/ab.log
