# Does a full bench.py run hang in teardown AFTER printing its line (seen once in round 4)?  Runs the bench a few times (it tears down normally
# since round 5, bounded by tools/exitguard; BENCH_TEARDOWN_LIMIT_S=600 here so that the guard does not end a hang before it is looked at);
# a run still alive 40 s after its line gets its threads' names, states and kernel wait channels dumped.
# usage (GPU box): bash tools/probe_exit_hang.sh [runs=3]
cd $GRAFT_REPO_ROOT
for r in $(seq 1 ${1:-3}); do
  rm -f /tmp/b_$r.json
  BENCH_TEARDOWN_LIMIT_S=600 python bench.py > /tmp/b_$r.json 2> /tmp/b_$r.err &
  pid=$!
  for i in $(seq 1 240); do sleep 1; [ -s /tmp/b_$r.json ] && break; kill -0 $pid 2>/dev/null || break; done
  t0=$(date +%s)
  for i in $(seq 1 40); do kill -0 $pid 2>/dev/null || break; sleep 1; done
  if kill -0 $pid 2>/dev/null; then
    echo "run $r: still alive 40 s after its line (pid $pid)"
    for t in /proc/$pid/task/*; do
      echo "  thread $(basename $t) $(cat $t/comm 2>/dev/null) wchan=$(cat $t/wchan 2>/dev/null) $(grep State $t/status 2>/dev/null)"
      head -6 $t/stack 2>/dev/null | sed 's/^/      /'
    done
    kill -9 $pid
  else
    echo "run $r: exited $(( $(date +%s) - t0 )) s after its line"
  fi
done
