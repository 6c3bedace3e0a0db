#!/bin/bash
# tools/exit_hang_series.sh <n> <tag> [MODE]: n bounded runs of tools/exit_hang_probe.sh, one summary line each
n=$1; tag=$2; mode=$3
for i in $(seq 1 $n); do
  timeout -k 10 200 $GRAFT_REPO_ROOT/tools/exit_hang_probe.sh ${tag}_$i $mode > /dev/null 2>&1
  f=$GRAFT_REPO_ROOT/gpurun_out/${tag}_$i.log
  echo "$tag run $i: hang=$(grep -c 'threads of' $f) segv=$(grep -c SIGSEGV $f) before=$(grep -c 'handler registered at the first create' $f) after_modules=$(grep -c 'code objects unregistered' $f) probe_atexit=$(grep -c 'exitprobe. atexit handler' $f) rc=$(grep -o 'exited by itself, rc [0-9]*' $f)"
done
