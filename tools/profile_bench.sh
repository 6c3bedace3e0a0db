#!/bin/bash
# the round's profile set for the default bench line: tools/profile_bench.sh <tag>
#   kernel trace + stats, FETCH_SIZE and WRITE_SIZE passes (separate runs), summaries under gpurun_out/<tag>_*
# Two different exit-time failures of profiled runs exist (profiles/r03_profiler_exit_hang.md): (A) after a COOPERATIVE
# launch the process segfaults inside libamdhip64's __hipUnregisterFatBinary at exit under rocprofv3 -- teams are launched
# non-cooperatively here; (B) a --pmc pass races its own worker thread when the process exits right after a burst of
# dispatches -- orbslam_in_practice_amd/_lib.py drains 2 s at exit under rocprofv3.  The timeouts are bounds, not fixes.
export ORBSLAM_AMD_BA_NO_COOP=1
tag=${1:-prof}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="--steps 5 --warmup 2 --cpu-frames 0 --ba-reps 2"
d=$R/gpurun_out/${tag}_stats; rm -rf $d
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/bench.py $ARGS > $R/gpurun_out/${tag}_bench_under_rocprof.json 2> $d.err
echo "stats rc $?"
for c in FETCH_SIZE WRITE_SIZE; do
  d=$R/gpurun_out/${tag}_pmc_$c; rm -rf $d
  timeout -k 5 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 $R/bench.py $ARGS > $d.json 2> $d.err
  echo "$c rc $?"
done
cp $(ls $R/gpurun_out/${tag}_stats/*/*kernel_stats.csv | head -1) $R/gpurun_out/${tag}_kernel_stats.csv
python3 $R/tools/collect_traffic.py $R/gpurun_out/${tag}_pmc_FETCH_SIZE $R/gpurun_out/${tag}_pmc_WRITE_SIZE $R/gpurun_out/${tag}_traffic.json 256 $R/gpurun_out/${tag}_sq.json
