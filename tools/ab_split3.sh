#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export LH_PROBE_SOLVER=1
for from in 99 3; do
  echo "=== probe LH_SPLIT_FROM=$from"
  LH_SPLIT_FROM=$from timeout 300 python tools/probe_iter_times.py 2>&1 | tail -3 | head -2
done
unset LH_PROBE_SOLVER
for cfg in "99 512" "3 512" "2 512" "4 512" "3 256" "99 512" "3 512"; do
  set -- $cfg
  echo "=== bench LH_SPLIT_FROM=$1 LH_WALK_SPAN=$2"
  LH_SPLIT_FROM=$1 LH_WALK_SPAN=$2 timeout 300 python bench.py --quick --steps 4 --warmup 1 2>&1 | tail -1 | cut -c1-80
done
