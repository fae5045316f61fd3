#!/bin/bash
# per-stage backward timing of prebuilt variants: bash tools/ab_bwd_variants.sh name1 name2 ... (LIST_BWD_OVERLAP=0: stages in line)
cd "$(dirname "$0")/.."
for rep in 1 2; do
for name in "$@"; do
  echo "=== [$name] rep $rep overlap=${LIST_BWD_OVERLAP:-1} prec=${PREC:-fp16}"
  LIST_HIP_LIB=$PWD/variants/$name.so timeout -k 10 200 python tools/bwd_bench.py ${PREC:-fp16} 5 2>/dev/null | grep -v amdgpu
done
done
