#!/bin/bash
# Build A/B variants of libgsrast.so (extra -D flags) into 3dgs_monocular_depth_init_amd/lib/variants/
# for in-one-call comparisons on the GPU box:  bash tools/build_variants.sh name "-DX=1" name2 "-DX=2" ...
# Goes through build.build(extra_flags=, out=): per-file objects, only what the flags touch is recompiled.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
O="$ROOT/3dgs_monocular_depth_init_amd/lib/variants"; mkdir -p "$O"
while [ $# -ge 2 ]; do
  NAME=$1; FLAGS=$2; shift 2
  ( cd "$ROOT" && python - "$NAME" "$FLAGS" <<'PY'
import importlib, shlex, sys
pkg = importlib.import_module("3dgs_monocular_depth_init_amd")
name, flags = sys.argv[1], shlex.split(sys.argv[2])
print(pkg.build(extra_flags=flags, out=pkg.build.__globals__["LIB_DIR"] / "variants" / f"libgsrast_{name}.so", jobs=4))
PY
  )
done
ls -la "$O"
