# A/B of library variants (ab_libs/liblocus_hip_<name>.so, `make -C locus_amd/csrc variant NAME=.. DEFS=..`) against the product build:
# per-iteration sweep times of one 32-pair group and the bench's quick line.   usage (GPU box): bash tools/ab_libs.sh name1 name2 ...
cd $GRAFT_REPO_ROOT
for v in base "$@" base "$@"; do
  lib=$GRAFT_REPO_ROOT/ab_libs/liblocus_hip_$v.so
  [ "$v" = base ] && lib=$GRAFT_REPO_ROOT/locus_amd/csrc/liblocus_hip.so
  echo "== $v"
  LH_LIB=$lib LH_PROBE_SOLVER=1 python tools/probe_iter_times.py 2>&1 | grep -E "per-iteration|other"
  LH_LIB=$lib python bench.py --quick --steps 4 --warmup 1 2>&1 | tail -1 | cut -c1-100
  [ -n "$AB_NATURAL" ] && LH_LIB=$lib python tools/bench_natural.py 2>&1 | tail -1
done
