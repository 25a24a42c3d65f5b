#!/bin/bash
# time the step with each build variant (tools/dbg/abl/lib_NAME.so)
for a in "$@"; do
  GSRAST_LIB=$PWD/tools/dbg/abl/lib_$a.so python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms']
print('abl $a', 'fwd', k['gsr_rasterize_fwd'], 'bwd', k['gsr_rasterize_bwd'], 'sort', k['gsr_bucket_sort'], 'step', round(d['ms_per_step'],4))" || exit 1
done
