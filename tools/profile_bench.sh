# rocprofv3 --kernel-trace --stats (+ marker trace for the leg slices) of the bench command -> gpurun_out/prof/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
rm -rf /tmp/bprof
(cd $R && timeout 200 rocprofv3 --kernel-trace --marker-trace --stats -d /tmp/bprof -o run --output-format csv -- python bench.py --no-cpu-baseline --no-trajectory --no-pmc --no-configs > $R/gpurun_out/prof/bench_line.json 2> $R/gpurun_out/prof/bench.err)
ls /tmp/bprof/* | head -20 > $R/gpurun_out/prof/files.txt
ks=$(find /tmp/bprof -name "*kernel_stats.csv" | head -1); kt=$(find /tmp/bprof -name "*kernel_trace.csv" | head -1); mt=$(find /tmp/bprof -name "*marker_api_trace.csv" | head -1)
cp $ks $R/gpurun_out/prof/kernel_stats.csv
head -3 $mt > $R/gpurun_out/prof/marker_head.txt
python $R/tools/trace_leg_summary.py $kt $mt profile_leg > $R/gpurun_out/prof/profile_leg_kernels.txt 2>&1
python $R/tools/trace_leg_summary.py $kt $mt timed_region > $R/gpurun_out/prof/timed_region_kernels.txt 2>&1
python $R/tools/trace_leg_summary.py $kt $mt profile_leg_natural > $R/gpurun_out/prof/profile_leg_natural_kernels.txt 2>&1
cat $R/gpurun_out/prof/profile_leg_kernels.txt | head -12; head -6 $R/gpurun_out/prof/timed_region_kernels.txt
