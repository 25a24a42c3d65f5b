#!/bin/bash
# Build and run the VALU issue-rate microbenchmark on the GPU box; writes
# gpurun_out/valu_rate.jsonl + valu_rate_isa.txt (copy to profiles/r<NN>_valu_rate.* to keep).
#   gpurun -- bash tools/ubench/run_valu_rate.sh
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
OUT="$ROOT/gpurun_out"; mkdir -p "$OUT/ubench"
cd "$OUT/ubench"
hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize --save-temps "$ROOT/tools/ubench/valu_rate.hip" -o valu_rate 2> build.log
S=valu_rate-hip-amdgcn-amd-amdhsa-gfx950.s
# VALU instructions in the innermost loop of every mode, from the ISA
ARGS=""
for M in 0 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15; do
  C=$(python3 "$ROOT/tools/isa_hist.py" "$S" --kernel "Li${M}E" --json | python3 -c 'import json,sys; d=json.load(sys.stdin); k=next(iter(d.values())); print(max(l["valu_total"] for l in k["loops"] if l["innermost"]))')
  ARGS="$ARGS $C"
done
echo "VALU instructions per trip, modes 0..15:$ARGS" >&2
./valu_rate $ARGS | tee "$OUT/valu_rate.jsonl"
python3 "$ROOT/tools/isa_hist.py" "$S" > "$OUT/valu_rate_isa.txt"
