#!/bin/bash
# PMC counters of the two SSIM kernels:  bash tools/prof_ssim.sh   -> gpurun_out/ssim_pmc.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
: > $R/gpurun_out/ssim_pmc.txt
for P in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  D=/tmp/ssim_pmc_$(echo $P | cut -d" " -f1)
  rm -rf $D
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $D -o s -- python3 $R/tools/prof_ssim.py > $D.log 2>&1 || { echo "pmc $P failed"; tail -3 $D.log; continue; }
  python3 - $D/s_counter_collection.csv >> $R/gpurun_out/ssim_pmc.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    if "ssim" in k:
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in cs.items()})
PY
done
cat $R/gpurun_out/ssim_pmc.txt
