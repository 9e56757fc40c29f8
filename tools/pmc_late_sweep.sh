# rocprofv3 --pmc passes over tools/probe_iter_times.py (one 32-pair group, profiling mode) -> counters of chosen k_sweep_fused dispatches
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES SQ_LEVEL_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 250 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_$i -o run --output-format csv -- python $R/tools/probe_iter_times.py > /tmp/pmc_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  python $R/tools/pmc_dispatch.py $f k_sweep_fused 40,41,44,59 >> $R/gpurun_out/pmc/late_sweep.txt 2>&1
done
cat $R/gpurun_out/pmc/late_sweep.txt
