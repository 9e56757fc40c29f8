# timing experiment: a sweep's time vs the number of lanes that walk (LH_EXP_LANES, results are wrong)
cd $GRAFT_REPO_ROOT
for e in 0 1 2; do
  echo "== LH_EXP_LANES=$e"
  LH_EXP_LANES=$e LH_PROBE_SOLVER=1 python tools/probe_iter_times.py 2>&1 | grep -E "per-iteration|other"
done
