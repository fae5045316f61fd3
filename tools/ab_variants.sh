#!/bin/bash
# A/B the prebuilt variants/<name>.so on one box: bash tools/ab_variants.sh name1 name2 ...  (2 interleaved repetitions)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for rep in 1 2; do
for name in "$@"; do
  LIST_HIP_LIB=$PWD/variants/$name.so timeout -k 10 200 python bench.py --steps ${STEPS:-20} --warmup 3 --precision ${PREC:-fp16} --no-cpu-baseline --no-train-step --no-channels-last-alt 2>gpurun_out/ab_$name.err | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('[$name] rep $rep: fc_0', round(d['kernel_ms']['fc_0'],4), 'ms; step', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms'].items()} if '${FULL:-0}'=='1' else '')"
done
done
