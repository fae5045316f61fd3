#!/bin/bash
# A/B the GEMM kernel variants on the GPU box: builds each variant into its own .so and runs the bench.
# usage (on the GPU box): bash tools/ab_gemm.sh "<flags A>" "<flags B>" ...
set -e
cd "$(dirname "$0")/.."
i=0
for flags in "$@"; do
  LIST_HIPCC_FLAGS="$flags" python learning-implicitly-from-spatial-transformers-network_amd/build.py --force > /dev/null 2>&1
  for rep in 1 2; do
    python bench.py --steps 10 --warmup 2 --precision ${PREC:-fp16} --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('variant [$flags] rep $rep:', round(d['value']/1e6,2),'Mpts/s', {k:round(v,3) for k,v in d['stage_ms'].items() if k.startswith('fc') or k.startswith('gather')})"
  done
  i=$((i+1))
done
