#!/bin/bash
# A/B of library variants (tools/build_variants.sh) inside ONE gpurun call, each twice, interleaved:
#   bash tools/ab_libs.sh head v1 v2 ...        ("head" = the in-tree libgsrast.so)
set -uo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
for REP in 1 2; do
for NAME in "$@"; do
  LIB="$ROOT/3dgs_monocular_depth_init_amd/lib/variants/libgsrast_$NAME.so"
  [ "$NAME" = head ] && LIB="$ROOT/3dgs_monocular_depth_init_amd/lib/libgsrast.so"
  GSRAST_LIB="$LIB" python3 "$ROOT/bench.py" --steps 30 --warmup 5 --no-cpu-baseline > "$ROOT/gpurun_out/ab_$NAME.json" 2> "$ROOT/gpurun_out/ab_$NAME.err" || echo "$NAME failed"
  python3 - "$ROOT/gpurun_out/ab_$NAME.json" "$NAME" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "ms/step %.4f"%d["ms_per_step"], " ".join("%s=%.4f"%(k.replace("gsr_",""),v) for k,v in d["kernel_ms"].items()), flush=True)
PY
done
done
