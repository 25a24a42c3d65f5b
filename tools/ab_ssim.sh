#!/bin/bash
# SSIM kernel times for library variants (bench.py's full-loss leg), one gpurun call:  bash tools/ab_ssim.sh head v1 ...
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
for REP in 1 2; do
for NAME in "$@"; do
  LIB="$ROOT/3dgs_monocular_depth_init_amd/lib/variants/libgsrast_$NAME.so"
  [ "$NAME" = head ] && LIB="$ROOT/3dgs_monocular_depth_init_amd/lib/libgsrast.so"
  GSRAST_LIB="$LIB" python3 "$ROOT/bench.py" --steps 20 --warmup 3 --no-cpu-baseline > "$ROOT/gpurun_out/ab_$NAME.json" 2> "$ROOT/gpurun_out/ab_$NAME.err" || echo "$NAME failed"
  python3 - "$ROOT/gpurun_out/ab_$NAME.json" "$NAME" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
f=d["metric_full_loss_step"]
print(sys.argv[2], "full-loss ms/step %.4f"%f["ms_per_step"], f["kernel_ms"], flush=True)
PY
done
done
