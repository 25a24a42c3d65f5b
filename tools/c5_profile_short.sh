#!/bin/bash
# kernel-by-kernel profile of a SHORT c5 rehearsal (reference schedule x 0.1: 3 000 steps on the 2 M-seed scene).
#   bash tools/c5_profile_short.sh <tag>     -> gpurun_out/<tag>_c5_short_kernel_stats.csv, gpurun_out/<tag>_c5_short.json
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$root/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c5
rocprofv3 --kernel-trace --stats -d /tmp/prof_c5 -o c5 --output-format csv -- python3 "$root/tools/c5_rehearsal.py" --steps-scaler 0.1 \
  > "$root/gpurun_out/${tag}_c5_short.log" 2>&1 || { tail -20 "$root/gpurun_out/${tag}_c5_short.log"; exit 1; }
cp /tmp/prof_c5/c5_kernel_stats.csv "$root/gpurun_out/${tag}_c5_short_kernel_stats.csv"
grep -a '"what"' "$root/gpurun_out/${tag}_c5_short.log" > "$root/gpurun_out/${tag}_c5_short.json"
head -12 "$root/gpurun_out/${tag}_c5_short_kernel_stats.csv" | cut -c1-160
