#!/bin/bash
# PMC passes over the BA phase kernels (one 256-window batch, two solves): tools/ba_pmc.sh <tag>
# separate rocprofv3 runs per counter group (kernel-trace only beside --pmc), summarised by tools/collect_sq.py.
# A process that exits right after a burst of dispatches races rocprofv3's counter-collection worker (crash or hang inside
# the profiler, profiles/r03_profiler_exit_hang.md); orbslam_in_practice_amd/_lib.py drains 2 s at exit under rocprofv3, the
# timeout below is only a bound.
tag=${1:-ba}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
  d=$R/gpurun_out/pmc_${tag}_$(echo $grp | cut -d' ' -f1)
  rm -rf $d
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -- python3 $R/tools/ba_probe.py 32 > $d.log 2>&1
  echo "pass $(echo $grp | cut -d' ' -f1): rc $? $(ls $d/*/*counter_collection.csv 2>/dev/null | wc -l) csv"
done
python3 $R/tools/collect_sq.py $R/gpurun_out/pmc_${tag}.json $R/gpurun_out/pmc_${tag}_*/
