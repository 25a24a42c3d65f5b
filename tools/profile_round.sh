#!/bin/bash
# Collect the round's evidence on the GPU box: rocprofv3 kernel stats of bench.py (and of the depth
# network) and the PMC passes (separate passes, counters only with --kernel-trace, as
# MI355X_MICROARCH.md prescribes). Digest afterwards in the build container:
#   python tools/digest_profiles.py <tag>
# Usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
set -o pipefail
TAG=${1:-r02}
R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/${TAG}_stats.log 2>&1 || exit 1
echo "stats done"
# (tools/prof_step.py runs the HEADLINE step -- Adam fused into the projection backward -- and the gradients-written
# step, so every pass has project_bwd_kernel<true> AND <false>)
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" "TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  D=$R/gpurun_out/${TAG}_pmc_$(echo $P | cut -d" " -f1)
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $D -- python3 $R/tools/prof_step.py > $D.log 2>&1 || { echo "pmc $P FAILED (see $D.log)"; continue; }
  echo "pmc $P done"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_dn_stats -- python3 $R/tools/bench_depthnet.py --backbones vitl --iters 3 > $R/gpurun_out/${TAG}_dn_stats.log 2>&1 || exit 1
echo "profile_round done"
