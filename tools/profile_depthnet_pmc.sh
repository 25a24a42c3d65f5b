#!/bin/bash
# MFMA / LDS counters of the depth network's GEMM and attention kernels at the shapes of
# tools/bench_gemm.py (two passes, counters only with --kernel-trace). Digest in the build container:
#   python tools/digest_depthnet_pmc.py <tag>
# Usage (GPU box, repo root): bash tools/profile_depthnet_pmc.sh <tag>
set -o pipefail
TAG=${1:-r02}
R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp
I=0
for P in "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE"; do
  D=$R/gpurun_out/${TAG}_dnpmc_$I
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $D -- python3 $R/tools/bench_gemm.py > $D.log 2>&1 || exit 1
  echo "pmc pass $I done"
  I=$((I+1))
done
