# Everything profiles/rNN_* is made from, in one GPU call:  bash tools/collect_evidence.sh   -> gpurun_out/evidence/
cd $GRAFT_REPO_ROOT
E=gpurun_out/evidence
rm -rf $E; mkdir -p $E
python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -5 > $E/pytest_gpu.txt
python bench.py > $E/bench_driver_style.json 2> $E/bench_driver_style.err
python bench.py --steps 20 --warmup 5 > $E/bench_final.json 2> $E/bench_final.err
cp gpurun_out/pmc/traffic.json $E/pmc_traffic.json          # (the counter passes of the bench run above: measured by the command itself)
bash tools/profile_bench.sh > $E/profile_bench.log 2>&1
cp gpurun_out/prof/kernel_stats.csv $E/bench_kernel_stats.csv
cp gpurun_out/prof/profile_leg_kernels.txt $E/bench_profile_leg_kernels.txt
cp gpurun_out/prof/timed_region_kernels.txt $E/bench_timed_region_kernels.txt
cp gpurun_out/prof/profile_leg_natural_kernels.txt $E/bench_profile_leg_natural_kernels.txt
cp gpurun_out/prof/bench_line.json $E/bench_line_under_rocprof.json
bash tools/pmc_lanes.sh > $E/pmc_lanes.txt 2>&1
bash tools/pmc_solve.sh > $E/pmc_solve.txt 2>&1
bash tools/trace_sweeps.sh > $E/sweep_kernel_trace.txt 2>&1
bash tools/trace_index.sh > $E/index_kernel_trace.txt 2>&1
# round 6: the cooperative all-walk sweep (opt-in) beside the default, same box
cd $GRAFT_REPO_ROOT; LH_SWEEP_COOP=1 bash tools/pmc_lanes.sh > $E/coop_pmc_lanes.txt 2>&1
cd $GRAFT_REPO_ROOT; LH_SWEEP_COOP=1 bash tools/trace_sweeps.sh > $E/coop_sweep_kernel_trace.txt 2>&1
cd $GRAFT_REPO_ROOT; LH_SWEEP_COOP=1 python bench.py --quick > $E/coop_bench_quick.json 2> $E/coop_bench_quick.err
cd $GRAFT_REPO_ROOT; python bench.py --quick > $E/default_bench_quick.json 2> $E/default_bench_quick.err
cd $GRAFT_REPO_ROOT
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --same-gpu --dist-backend gloo --quick --pairs 64 --in-flight 64 --steps 2 --warmup 1 > $E/bench_two_ranks_one_gpu.json 2> $E/bench_two_ranks_one_gpu.err
python bench.py --gpus 2 --same-gpu --dist-backend gloo --quick --pairs 64 --in-flight 64 --steps 2 --warmup 1 > $E/bench_self_launched_two_ranks.json 2> $E/bench_self_launched_two_ranks.err
python bench.py --gpus 2 --same-gpu --dist-backend gloo --pairs 32 --in-flight 32 --steps 2 --warmup 1 --no-trajectory --no-configs > $E/bench_two_ranks_full_line.json 2> $E/bench_two_ranks_full_line.err
bash tools/trace_production.sh > $E/production_update_trace.txt 2>&1
cd $GRAFT_REPO_ROOT; bash tools/trace_locus_stream.sh > $E/locus_stream_trace.txt 2>&1
cd $GRAFT_REPO_ROOT
python tests/perf/bench_ndt.py > $E/ndt.json 2> $E/ndt.err
ls -la $E
