#!/bin/bash
# the round's profile set for the default bench line: tools/profile_bench.sh <tag>
#   kernel trace + stats, FETCH_SIZE and WRITE_SIZE passes (separate runs), summaries under gpurun_out/<tag>_*
# rocprofv3 (ROCm 7.2) crashes or hangs at process exit after a cooperative launch / after its output is written:
# teams are launched non-cooperatively here and every pass runs under a timeout.
tag=${1:-prof}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export ORBSLAM_AMD_BA_NO_COOP=1
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
python3 $R/tools/collect_traffic.py $R/gpurun_out/${tag}_pmc_FETCH_SIZE $R/gpurun_out/${tag}_pmc_WRITE_SIZE $R/gpurun_out/${tag}_traffic.json 256 $R/profiles/r02_pmc_sq_frontend.json
