#!/bin/bash
# SQ counter passes over the front-end kernels (default bench workload, a few steps): tools/frontend_pmc.sh <tag>
# separate rocprofv3 runs per counter group, summarised by tools/collect_sq.py into gpurun_out/<tag>_sq.json
tag=${1:-fe}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="--steps 3 --warmup 1 --cpu-frames 0 --ba-windows 0 --sequence-frames 0"
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES GRBM_GUI_ACTIVE"; do
  d=$R/gpurun_out/${tag}_$(echo $grp | cut -d' ' -f1)
  rm -rf $d
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -- python3 $R/bench.py $ARGS > $d.json 2> $d.err
  echo "pass $(echo $grp | cut -d' ' -f1): rc $?"
done
python3 $R/tools/collect_sq.py $R/gpurun_out/${tag}_sq.json $R/gpurun_out/${tag}_SQ_INSTS_VALU $R/gpurun_out/${tag}_SQ_WAVE_CYCLES
