#!/bin/bash
# Where does a profiled run stop at process exit?  tools/exit_hang_probe.sh <tag> [MODE]
# Runs tools/exit_probe.py under ONE rocprofv3 PMC pass in the background, waits for the script's "end of script" marker,
# gives the process 25 s to leave, and if it is still there asks it for a native backtrace (SIGUSR1 -> exitprobe.so), tries
# rocgdb for all threads, and kills it.  Output: gpurun_out/<tag>.log
tag=${1:-exit_probe}
export MODE=${2:-}
R=$GRAFT_REPO_ROOT
gcc -shared -fPIC -O1 -o $R/tools/micro/exitprobe.so $R/tools/micro/exitprobe.c || exit 1
cd /tmp && export TMPDIR=/tmp
log=$R/gpurun_out/$tag.log
: > $log
d=$R/gpurun_out/${tag}_pmc; rm -rf $d
export ORBSLAM_AMD_TRACE_EXIT=1
rocprofv3 --kernel-trace --pmc ${PMC:-SQ_INSTS_VALU SQ_WAVES} --output-format csv -d $d -- python3 $R/tools/exit_probe.py >> $log 2>&1 &
pid=$!
for i in $(seq 1 120); do
  grep -q "end of script" $log && break
  kill -0 $pid 2>/dev/null || break
  sleep 1
done
echo "[probe.sh] script end seen after ${i}s" >> $log
for i in $(seq 1 25); do kill -0 $pid 2>/dev/null || break; sleep 1; done
if kill -0 $pid 2>/dev/null; then
  echo "[probe.sh] STILL ALIVE 25 s after the end of the script: hang reproduced" >> $log
  ps -o pid,ppid,stat,wchan:32,cmd --ppid $pid -p $pid >> $log 2>&1
  # the python process may be a child of the rocprofv3 launcher
  for p in $pid $(pgrep -P $pid); do
    echo "[probe.sh] threads of $p:" >> $log
    for t in /proc/$p/task/*; do echo "  $(basename $t) $(cat $t/comm 2>/dev/null) wchan=$(cat $t/wchan 2>/dev/null) state=$(grep State $t/status 2>/dev/null)" >> $log; done
    kill -USR1 $p 2>/dev/null
  done
  sleep 3
  echo "[probe.sh] ptrace_scope=$(cat /proc/sys/kernel/yama/ptrace_scope 2>/dev/null)" >> $log
  for p in $(pgrep -P $pid) $pid; do
    timeout 60 rocgdb -p $p -batch -ex "thread apply all bt 25" >> $log 2>&1 && break
  done
  for p in $(pgrep -P $pid) $pid; do kill -9 $p 2>/dev/null; done
  echo "[probe.sh] killed" >> $log
else
  wait $pid
  echo "[probe.sh] exited by itself, rc $?" >> $log
fi
tail -60 $log
