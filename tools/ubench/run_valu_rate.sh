#!/bin/bash
# Build and run the VALU issue-rate microbenchmark on the GPU box; writes
# gpurun_out/valu_rate.jsonl (copy to profiles/r<NN>_valu_rate.jsonl to keep it).
#   gpurun -- bash tools/ubench/run_valu_rate.sh
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
OUT="$ROOT/gpurun_out"; mkdir -p "$OUT/ubench"
cd "$OUT/ubench"
hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize --save-temps "$ROOT/tools/ubench/valu_rate.hip" -o valu_rate 2> build.log
S=valu_rate-hip-amdgcn-amd-amdhsa-gfx950.s
# VALU instructions in the loop bodies of the two composite modes, from the ISA
C1=$(python3 "$ROOT/tools/isa_hist.py" "$S" --kernel Li6E --json | python3 -c 'import json,sys; d=json.load(sys.stdin); k=next(iter(d.values())); print(max(l["valu_total"] for l in k["loops"] if l["innermost"]))')
C2=$(python3 "$ROOT/tools/isa_hist.py" "$S" --kernel Li7E --json | python3 -c 'import json,sys; d=json.load(sys.stdin); k=next(iter(d.values())); print(max(l["valu_total"] for l in k["loops"] if l["innermost"]))')
echo "composite bodies: $C1 / $C2 VALU instructions per trip" >&2
./valu_rate "$C1" "$C2" | tee "$OUT/valu_rate.jsonl"
python3 "$ROOT/tools/isa_hist.py" "$S" > "$OUT/valu_rate_isa.txt"
