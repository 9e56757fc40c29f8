# A/B of environment settings of the product library: per-iteration sweep times of one 32-pair group + the bench's quick line (+ natural convergence)
#   usage (GPU box): bash tools/ab_env.sh "LH_PERM=0" "LH_PERM=1" ...
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for e in "$@"; do
  echo "== $e"
  env $e LH_PROBE_SOLVER=1 python tools/probe_iter_times.py 2>&1 | grep -E "per-iteration"
  env $e python bench.py --quick --steps 4 --warmup 1 2>&1 | tail -1 | cut -c1-100
done
done
