#!/bin/bash
# The round's profile set, one call: tools/round_profiles.sh <tag>      (outputs under gpurun_out/<tag>_*; copy what is
# to be judged into profiles/).  Passes, each a separate rocprofv3 run (PMC never beside anything but --kernel-trace):
#   1. kernel trace + stats of the default bench line            -> <tag>_kernel_stats.csv, <tag>_kernel_trace_by_launch_shape.json
#   2. SQ counter groups over the front-end (tools/frontend_pmc.sh) -> <tag>_sq.json
#   3. FETCH_SIZE / WRITE_SIZE over the default bench line       -> <tag>_traffic.json (stamped with the kernel source hashes)
#   4. SQ counter groups over the BA phase kernels (tools/ba_pmc.sh) -> pmc_<tag>.json
tag=${1:-r03}
R=$GRAFT_REPO_ROOT
$R/tools/frontend_pmc.sh $tag || exit 1
$R/tools/profile_bench.sh $tag || exit 1
python3 $R/tools/trace_by_shape.py $(ls $R/gpurun_out/${tag}_stats/*/*kernel_trace.csv | head -1) $R/gpurun_out/${tag}_kernel_trace_by_launch_shape.json
$R/tools/ba_pmc.sh $tag
ls -la $R/gpurun_out/${tag}_*.json $R/gpurun_out/${tag}_*.csv $R/gpurun_out/pmc_${tag}.json
