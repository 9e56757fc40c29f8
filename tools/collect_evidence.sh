# Everything profiles/rNN_* is made from, in one GPU call:  bash tools/collect_evidence.sh   -> gpurun_out/evidence/
cd $GRAFT_REPO_ROOT
E=gpurun_out/evidence
mkdir -p $E
python bench.py > $E/bench_driver_style.json 2> $E/bench_driver_style.err
python bench.py --steps 20 --warmup 5 --no-trajectory > $E/bench_final.json 2> $E/bench_final.err
bash tools/profile_bench.sh > $E/profile_bench.log 2>&1
cp gpurun_out/prof/kernel_stats.csv $E/bench_kernel_stats.csv
cp gpurun_out/prof/profile_leg_kernels.txt $E/bench_profile_leg_kernels.txt
cp gpurun_out/prof/timed_region_kernels.txt $E/bench_timed_region_kernels.txt
cp gpurun_out/prof/profile_leg_natural_kernels.txt $E/bench_profile_leg_natural_kernels.txt
cp gpurun_out/prof/bench_line.json $E/bench_line_under_rocprof.json
bash tools/pmc_traffic.sh > $E/pmc_traffic.log 2>&1
cp gpurun_out/pmc/traffic.json $E/pmc_traffic.json
bash tools/pmc_lanes.sh > $E/pmc_lanes.txt 2>&1
bash tools/trace_sweeps.sh > $E/sweep_kernel_trace.txt 2>&1
bash tools/trace_index.sh > $E/index_kernel_trace.txt 2>&1
python tools/bench_submap.py > $E/config3_submap.json 2> $E/config3.err
python tools/bench_merged1m.py > $E/config5_merged1m.json 2> $E/config5.err
python tests/perf/bench_ndt.py > $E/ndt.json 2> $E/ndt.err
ls -la $E
