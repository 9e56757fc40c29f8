#!/bin/bash
# bench --quick matrix over the split-sweep settings (overlapped, four scheduler groups: the number that counts)
cd "$(dirname "$0")/.." || exit 1
for cfg in "99 512 16" "2 512 16" "3 512 16" "4 512 16" "2 256 16" "2 1024 16" "3 1024 16" "4 1024 16" "2 512 8" "2 512 32" "99 512 16"; do
  set -- $cfg
  echo "=== LH_SPLIT_FROM=$1 LH_WALK_SPAN=$2 LH_WALK_REFILL=$3"
  LH_SPLIT_FROM=$1 LH_WALK_SPAN=$2 LH_WALK_REFILL=$3 timeout 300 python bench.py --quick --steps 4 --warmup 1 2>&1 | tail -1 | cut -c1-80
done
