#!/bin/bash
# what each stage of the forked backward costs the STEP: the stage left out (diagnostic build -DLIST_BWD_KNOCKOUT, wrong
# gradients), fp16 training step of config 2.  LIST_BWD_SKIP bits: 1 dW0, 2 direct-atomic levels, 4 16^3 window level,
# 8 8^3 window level, 16 voxel-side gather (32^3), 32 image gradient (map gather, trans_mat, adjoint resize)
cd "$(dirname "$0")/.."
for rep in 1 2; do
for skip in 0 1 2 4 8 16 32 48 63; do
  LIST_BWD_SKIP=$skip LIST_HIP_LIB=$PWD/variants/knock.so timeout -k 10 300 python tools/r4_pileup_train.py ${PREC:-fp16} 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());a=d['synthetic_camera'];b=d['piled_on_clamp'];print('[skip = $skip] rep $rep: step', a['ms_per_step'], 'bwd', a['backward_ms'], '| piled: step', b['ms_per_step'], 'bwd', b['backward_ms'])"
done
done
