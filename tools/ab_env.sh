#!/bin/bash
# A/B one environment switch of the library on one box: bash tools/ab_env.sh NAME v1 v2 ...   (2 interleaved repetitions)
# e.g. bash tools/ab_env.sh LIST_LAUNCH_IN_ORDER 1 0
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
var=$1; shift
for rep in 1 2; do
for v in "$@"; do
  env $var=$v timeout -k 10 200 python bench.py --steps ${STEPS:-20} --warmup 3 --precision ${PREC:-fp16} --no-cpu-baseline --no-train-step --no-channels-last-alt 2>gpurun_out/ab_env.err | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernel_ms'];print('[$var=$v] rep $rep: step', round(d['ms_per_step'],3), 'gathers', round(k.get('gathers_back_to_back',0),3), 'fc_0', round(k['fc_0'],3))"
done
done
