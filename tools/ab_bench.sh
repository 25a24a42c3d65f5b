#!/bin/bash
# A/B bench of the variant libraries inside ONE gpurun call (box-to-box variation is ~3 %).
#   bash tools/ab_bench.sh v0 v1 v2   -> gpurun_out/ab_<name>.json
set -uo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
for NAME in "$@"; do
  GSRAST_LIB="$ROOT/3dgs_monocular_depth_init_amd/lib/variants/libgsrast_$NAME.so" python3 "$ROOT/bench.py" --steps 30 --warmup 5 --no-cpu-baseline > "$ROOT/gpurun_out/ab_$NAME.json" 2> "$ROOT/gpurun_out/ab_$NAME.err" || echo "$NAME failed"
  python3 - "$ROOT/gpurun_out/ab_$NAME.json" "$NAME" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d["kernel_ms"]
print(sys.argv[2], "ms/step %.4f"%d["ms_per_step"], "fwd %.4f bwd %.4f"%(k["gsr_rasterize_fwd"],k["gsr_rasterize_bwd"]), flush=True)
PY
done
