#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
for dk in 0 1; do
  for from in 99 3; do
    echo "=== probe HIP_FORCE_DEV_KERNARG=$dk LH_SPLIT_FROM=$from"
    HIP_FORCE_DEV_KERNARG=$dk LH_PROBE_SOLVER=1 LH_SPLIT_FROM=$from timeout 60 python tools/probe_iter_times.py 2>&1 | tail -3 | head -2
  done
done
for dk in 0 1 0 1; do
  for from in 99 3; do
    echo "=== bench HIP_FORCE_DEV_KERNARG=$dk LH_SPLIT_FROM=$from"
    HIP_FORCE_DEV_KERNARG=$dk LH_SPLIT_FROM=$from timeout 120 python bench.py --quick --steps 4 --warmup 1 2>&1 | tail -1 | cut -c1-80
  done
done
