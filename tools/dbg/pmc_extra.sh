#!/bin/bash
# extra SQ counter passes for the compositing kernels (usage: bash tools/dbg/pmc_extra.sh <tag>)
set -o pipefail
TAG=${1:-x1}
R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp
for P in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_LEVEL_WAVES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_BRANCH" "SQ_INSTS_VALU_TRANS_F32 SQ_IFETCH SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU"; do
  D=$R/gpurun_out/${TAG}_pmc_$(echo $P | cut -d" " -f1)
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $D -- python3 $R/tools/prof_step.py > $D.log 2>&1 || exit 1
done
echo "pmc_extra done"
