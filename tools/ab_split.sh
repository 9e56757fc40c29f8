#!/bin/bash
# A/B of the two-launch sweep (k_late + k_walk) against the fused sweep: per-iteration sweep times of one 32-pair group
# (host-driven loop, one scheduler group, HIP events per launch) and the bench's quick throughput line.
# usage (GPU box): bash tools/ab_split.sh > gpurun_out/ab_split.log 2>&1
cd "$(dirname "$0")/.." || exit 1
export LH_PROBE_SOLVER=1
for from in 99 0 1 2 3; do
  for span in ${SPANS:-512}; do
    echo "=== LH_SPLIT_FROM=$from LH_WALK_SPAN=$span"
    LH_SPLIT_FROM=$from LH_WALK_SPAN=$span timeout 300 python tools/probe_iter_times.py 2>&1 | tail -3
  done
done
for from in 99 0 2; do
  echo "=== bench --quick LH_SPLIT_FROM=$from"
  LH_SPLIT_FROM=$from timeout 300 python bench.py --quick --steps 3 --warmup 1 2>&1 | tail -1
done
