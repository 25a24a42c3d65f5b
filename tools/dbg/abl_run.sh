#!/bin/bash
# time the compositing kernels with each ablation build (tools/dbg/abl/lib_N.so)
for a in "$@"; do
  GSRAST_LIB=$PWD/tools/dbg/abl/lib_$a.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms']
print('abl $a', 'bwd', k['gsr_rasterize_bwd'], 'fwd', k['gsr_rasterize_fwd'], 'step', d['ms_per_step'])" || exit 1
done
