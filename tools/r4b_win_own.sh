#!/bin/bash
# training step with the 8^3 window level on the third auxiliary stream (ABI 9, default) or behind the gathers on the
# caller's stream (LIST_BWD_WIN2_OWN=0), interleaved on one box
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
for own in 0 1; do
  LIST_BWD_WIN2_OWN=$own timeout -k 10 300 python tools/r4_pileup_train.py ${PREC:-fp16} 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());a=d['synthetic_camera'];b=d['piled_on_clamp'];print('[aux_streams[2] = $own] rep $rep: step', a['ms_per_step'], 'bwd', a['backward_ms'], '| piled: step', b['ms_per_step'], 'bwd', b['backward_ms'])"
done
done
