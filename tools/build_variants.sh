#!/bin/bash
# Build A/B variants of libgsrast.so (extra -D flags) into 3dgs_monocular_depth_init_amd/lib/variants/
# for in-one-call comparisons on the GPU box:  bash tools/build_variants.sh name "-DX=1" name2 "-DX=2" ...
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
C="$ROOT/3dgs_monocular_depth_init_amd/csrc"; O="$ROOT/3dgs_monocular_depth_init_amd/lib/variants"; mkdir -p "$O"
SRCS="api.hip project.hip isect.hip isect_bucket.hip raster_fwd.hip raster_bwd.hip init_depth.hip train_ops.hip ssim.hip knn.hip"
for f in depthnet.hip pointcloud.hip rbf.hip; do [ -f "$C/$f" ] && SRCS="$SRCS $f"; done
while [ $# -ge 2 ]; do
  NAME=$1; FLAGS=$2; shift 2
  ( cd "$C" && hipcc -O3 -std=c++17 -shared -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -fno-slp-vectorize -Wno-unused-function $FLAGS $SRCS -o "$O/libgsrast_$NAME.so" ) &
done
wait
ls -la "$O"
