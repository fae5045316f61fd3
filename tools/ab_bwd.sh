#!/bin/bash
# A/B compile-time variants of the backward on the GPU box: bash tools/ab_bwd.sh "<flags A>" "<flags B>" ...
# prints the per-stage ms of tools/bwd_bench.py for each variant (LIST_BWD_OVERLAP=0: stages in line).
set -e
cd "$(dirname "$0")/.."
for flags in "$@"; do
  LIST_HIPCC_FLAGS="$flags" python learning-implicitly-from-spatial-transformers-network_amd/build.py --force > /dev/null 2>&1
  echo "=== [$flags] overlap=${LIST_BWD_OVERLAP:-1}"
  timeout -k 10 200 python tools/bwd_bench.py ${PREC:-fp16} 3 2>/dev/null | grep -v amdgpu
done
