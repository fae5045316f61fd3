#!/bin/bash
# A/B compile-time variants on the GPU box: bash tools/ab_build.sh "<flags A>" "<flags B>" ...
# prints the per-kernel ms of each variant (2 repetitions, interleaved order A B A B).
set -e
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for flags in "$@"; do
    LIST_HIPCC_FLAGS="$flags" python learning-implicitly-from-spatial-transformers-network_amd/build.py --force > /dev/null 2>&1
    # (the flags are part of the library's build fingerprint: bench.py must see them too or it rebuilds without them)
    LIST_HIPCC_FLAGS="$flags" python bench.py --steps 10 --warmup 2 --precision ${PREC:-fp16} --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('[$flags] rep $rep:', round(d['value']/1e6,2),'Mpts/s', {k:round(v,3) for k,v in d['kernel_ms'].items()})"
  done
done
