#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
for np in 32 16 8 4; do
  echo "=== probe pairs=$np (host loop, split from 3)"
  LH_PROBE_PAIRS=$np LH_PROBE_SOLVER=1 LH_SPLIT_FROM=3 timeout 60 python tools/probe_iter_times.py 2>&1 | tail -3 | head -2
done
