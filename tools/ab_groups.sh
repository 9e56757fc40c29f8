# scheduler groups (= HIP streams) against pairs in flight: the quick line's rate for LH_DEVICE_GROUPS settings at 32 / 64 / 128 pairs in flight
# usage (GPU box): bash tools/ab_groups.sh
cd $GRAFT_REPO_ROOT
for spec in "32:0 2 4 8" "64:0 4 6 8 12 16" "128:0 8 12 16"; do
  inf=${spec%%:*}
  for g in ${spec#*:}; do
    echo "== in flight $inf, LH_DEVICE_GROUPS=$g (0 = the built-in rule)"
    LH_DEVICE_GROUPS=$g BENCH_FAST_EXIT=1 python bench.py --quick --pairs 256 --in-flight $inf --steps 3 --warmup 1 2>/dev/null | tail -1 | cut -c1-220
  done
done
