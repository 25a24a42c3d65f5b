#!/bin/bash
# A/B of variant libraries (tools/build_variants.sh) on the two-launch path (projection backward + gsr_adam_step):
#   bash tools/ab_separate_adam.sh v0 v1 ...   (inside one gpurun call; prints ms/step and the projection / Adam kernel times)
ROOT=$PWD
for NAME in "$@"; do
  GSRAST_LIB="$ROOT/3dgs_monocular_depth_init_amd/lib/variants/libgsrast_$NAME.so" python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --separate-adam 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']; print('$NAME', round(d['ms_per_step'],4), {x:k[x] for x in k if 'adam' in x or 'project' in x})"
done
