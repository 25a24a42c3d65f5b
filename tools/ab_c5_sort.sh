#!/bin/bash
# sort kernel time in the dense (c5-like) scene for library variants, one gpurun call:  bash tools/ab_c5_sort.sh head v1 ...
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
for NAME in "$@"; do
  LIB="$ROOT/3dgs_monocular_depth_init_amd/lib/variants/libgsrast_$NAME.so"
  [ "$NAME" = head ] && LIB="$ROOT/3dgs_monocular_depth_init_amd/lib/libgsrast.so"
  export GSRAST_LIB="$LIB"
  bash "$ROOT/tools/c5_profile_short.sh" "ab_$NAME" > /dev/null || echo "$NAME failed"
  python3 - "$ROOT/gpurun_out/ab_${NAME}_c5_short_kernel_stats.csv" "$NAME" <<'PY'
import csv, sys
rows = {r["Name"].split("(")[0].replace("void ", "").replace("gsr::", ""): float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(sys.argv[1]))}
print(sys.argv[2], " ".join("%s=%.1f" % (k, v) for k, v in rows.items() if k.startswith(("bucket_", "tile_order"))), flush=True)
PY
done
