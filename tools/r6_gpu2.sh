mkdir -p gpurun_out/r6b
export TMPDIR=/tmp
for g in 0 1 2 3; do
  LH_COOP_GREEDY=$g timeout 300 bash tools/trace_sweeps.sh 2>&1 | grep -E "k_sweep_coop|k_seed" > $GRAFT_REPO_ROOT/gpurun_out/r6b/trace_greedy$g.txt
  cd $GRAFT_REPO_ROOT
  echo "greedy $g"; cat gpurun_out/r6b/trace_greedy$g.txt
done
LH_SWEEP_COOP=0 timeout 300 bash tools/trace_sweeps.sh 2>&1 | grep -E "k_sweep_fused|k_seed"
