# timing experiment: the walking sweeps against the size of the tree they walk (same 100 k queries per pair) and the number of jobs per launch
cd $GRAFT_REPO_ROOT
for st in 1 2 4 8 16; do
  echo "== target stride $st"
  LH_PROBE_TGT_STRIDE=$st LH_PROBE_SOLVER=1 python tools/probe_iter_times.py 2>&1 | grep -E "per-iteration"
done
for np in 8 16; do
  echo "== pairs $np"
  LH_PROBE_PAIRS=$np LH_PROBE_SOLVER=1 python tools/probe_iter_times.py 2>&1 | grep -E "per-iteration"
done
