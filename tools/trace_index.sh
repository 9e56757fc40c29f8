cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kt
timeout 200 rocprofv3 --kernel-trace -d /tmp/kt -o run --output-format csv -- python $R/tools/probe_iter_times.py > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_summary.py $f 2 k_rs,k_bbox,k_key,k_leaf,k_leaves,k_tile,k_radix,k_chunk,k_boxes,k_nodex,k_seed
