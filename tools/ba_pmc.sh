#!/bin/bash
# PMC passes over the BA phase kernels (one 256-window batch, two solves): tools/ba_pmc.sh <tag>
# separate rocprofv3 runs per counter group (kernel-trace only beside --pmc), summarised by tools/collect_sq.py
set -e
tag=${1:-ba}
export ORBSLAM_AMD_BA_SPLIT=1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  d=$R/gpurun_out/pmc_${tag}_$(echo $grp | cut -d' ' -f1)
  rm -rf $d
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -- python3 $R/tools/ba_probe.py 32 > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
done
python3 $R/tools/collect_sq.py $R/gpurun_out/pmc_${tag}.json $R/gpurun_out/pmc_${tag}_*/ 
