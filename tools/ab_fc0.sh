#!/bin/bash
# usage: bash ab_fc0.sh "<flags>" ...   prints fc_0 ms per variant
cd "$(dirname "$0")/.."
for rep in 1 2; do
for flags in "$@"; do
  LIST_HIPCC_FLAGS="$flags" python learning-implicitly-from-spatial-transformers-network_amd/build.py --force > /dev/null 2>&1
  python bench.py --steps 10 --warmup 2 --precision ${PREC:-fp16} --no-cpu-baseline --no-train-step --no-channels-last-alt 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('[$flags] rep $rep: fc_0', round(d['kernel_ms']['fc_0'],3), 'ms; step', round(d['ms_per_step'],3))"
done
done
