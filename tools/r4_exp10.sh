#!/bin/bash
# training step (bench.py's run_train_step through tools/r4_pileup_train.py) of prebuilt variants, interleaved on one box
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
for name in "$@"; do
  LIST_HIP_LIB=$PWD/variants/$name.so timeout -k 10 300 python tools/r4_pileup_train.py ${PREC:-fp16} 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());a=d['synthetic_camera'];b=d['piled_on_clamp'];print('[$name] rep $rep: step', a['ms_per_step'], 'bwd', a['backward_ms'], '| piled: step', b['ms_per_step'], 'bwd', b['backward_ms'])"
done
done
