#!/bin/bash
# A/B of the sweep settings: per-iteration sweep times of one 32-pair group (host-driven loop, one scheduler group, HIP events per
# launch), the walkers per sweep, and the bench's quick throughput line.   usage (GPU box): bash tools/ab_split.sh
cd "$(dirname "$0")/.." || exit 1
for from in ${FROMS:-99 3}; do
  echo "=== probe LH_SPLIT_FROM=$from"
  LH_PROBE_SOLVER=1 LH_SPLIT_FROM=$from timeout 100 python tools/probe_iter_times.py 2>&1 | tail -3 | head -2
done
echo "=== walkers per sweep (4 pairs)"
LH_WALK_LOG=1 LH_PROBE_PAIRS=4 LH_PROBE_SOLVER=1 timeout 60 python tools/probe_iter_times.py 2>&1 | grep "lh walks" | tail -80 | awk '{w[$4]=w[$4]" "$8} END {for (s in w) print "slot", s, ":", w[s]}'
for from in ${FROMS:-99 3} ${FROMS:-99 3}; do
  echo "=== bench --quick LH_SPLIT_FROM=$from"
  LH_SPLIT_FROM=$from timeout 200 python bench.py --quick --steps 4 --warmup 1 2>&1 | tail -1 | cut -c1-90
done
