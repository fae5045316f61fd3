#!/bin/bash
# A/B the prebuilt variants/<name>.so on one box, printing the per-gather kernel times: bash tools/ab_gather.sh name1 name2 ...
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for rep in 1 2; do
for name in "$@"; do
  lib=$name; box=1; if [ $name = old ]; then lib=base; box=0; fi
  LIST_GATHER_BOX=$box LIST_HIP_LIB=$PWD/variants/$lib.so timeout -k 10 200 python bench.py --steps ${STEPS:-20} --warmup 3 --precision ${PREC:-fp16} --no-cpu-baseline --no-train-step --no-channels-last-alt 2>gpurun_out/ab_$name.err | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernel_ms'];print('[$name] rep $rep: step', round(d['ms_per_step'],3), 'l4', round(k['gather_vox_l4'],4), 'l5', round(k['gather_vox_l5'],4), 'group', round(k['gathers_back_to_back'],4), 'fc_0', round(k['fc_0'],4))"
done
done
