#!/bin/bash
# Build and run the long-body VALU issue-rate microbenchmark on the GPU box; writes
# gpurun_out/valu_rate_long.jsonl + valu_rate_long_isa.txt (copy to profiles/r03_valu_rate_long.*).
#   gpurun -- bash tools/ubench/run_valu_rate_long.sh
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
OUT="$ROOT/gpurun_out"; mkdir -p "$OUT/ubench_long"
cd "$OUT/ubench_long"
hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize --save-temps "$ROOT/tools/ubench/valu_rate_long.hip" -o valu_rate_long 2> build.log
S=valu_rate_long-hip-amdgcn-amd-amdhsa-gfx950.s
python3 "$ROOT/tools/isa_hist.py" "$S" | grep -E "^==|innermost" > "$OUT/valu_rate_long_isa.txt"
./valu_rate_long | tee "$OUT/valu_rate_long.jsonl"
