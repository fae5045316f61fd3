#!/bin/bash
# training step with the window levels' adjoint on the matrix cores (k_scatter_vox_box): LIST_SCATTER_BOX = 0 neither level,
# 1 the 8^3 level, 2 both, 3 the 16^3 level, unset (-1 here) = the default (16^3; 8^3 too when the backward is not forked)
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
for box in ${MODES:-0 -1 2}; do
  [ "$box" = "-1" ] && unset LIST_SCATTER_BOX || export LIST_SCATTER_BOX=$box
  timeout -k 10 300 python tools/r4_pileup_train.py ${PREC:-fp16} 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());a=d['synthetic_camera'];b=d['piled_on_clamp'];print('[LIST_SCATTER_BOX = $box] rep $rep: step', a['ms_per_step'], 'bwd', a['backward_ms'], '| piled: step', b['ms_per_step'], 'bwd', b['backward_ms'])"
done
done
