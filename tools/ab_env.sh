#!/bin/bash
# (the LDS-pad / GEMM-core variables are read only by a library built with -DGSR_EXPERIMENT_KNOBS=1:
#  bash tools/build_variants.sh knobs "-DGSR_EXPERIMENT_KNOBS=1"; GSRAST_LIB=.../libgsrast_knobs.so)
# A/B of environment switches inside ONE gpurun call (box-to-box variation is ~3-5 %):
#   bash tools/ab_env.sh "name1:VAR=1 VAR2=x" "name2:" ...      (each twice, interleaved)
set -uo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
for REP in 1 2; do
for SPEC in "$@"; do
  NAME="${SPEC%%:*}"; ENVS="${SPEC#*:}"
  env $ENVS python3 "$ROOT/bench.py" --steps 30 --warmup 5 --no-cpu-baseline > "$ROOT/gpurun_out/ab_$NAME.json" 2> "$ROOT/gpurun_out/ab_$NAME.err" || echo "$NAME failed"
  python3 - "$ROOT/gpurun_out/ab_$NAME.json" "$NAME" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "ms/step %.4f"%d["ms_per_step"], " ".join("%s=%.4f"%(k.replace("gsr_",""),v) for k,v in d["kernel_ms"].items()), flush=True)
PY
done
done
