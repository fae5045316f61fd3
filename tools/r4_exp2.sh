#!/bin/bash
# Round 4, experiment 2 (one box): fc_0 with the perceptual block produced on chip (default) vs the unfused path
# (LIST_FUSED_FC0=0) vs the 128 x 512 tile with every K-tile from X (LIST_FUSED_FC0=x).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() {   # name env...
  local name=$1; shift 1
  env "$@" timeout -k 10 200 python bench.py --steps ${STEPS:-20} --warmup 3 --precision ${PREC:-fp16} --no-cpu-baseline --no-train-step --no-channels-last-alt --sustained-steps 0 2>gpurun_out/exp2_$name.err | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernel_ms'];print('[$name]: step', round(d['ms_per_step'],4), 'ev_med', round(d['step_events_ms']['median'],4), 'prep', round(k['prep_img_resize_nhwc']+k['prep_vox_ndhwc'],4), 'sort', round(k['sort_points'],4), 'group', round(k['gathers_back_to_back'],4), 'fc_0', round(k['fc_0'],4), 'tail', round(k['fc_2_out'],4), 'img', round(k['gather_img'],4))" | tee -a gpurun_out/exp2.log
}
for rep in 1 2 3; do
  run unfused LIST_FUSED_FC0=0
  run fused LIST_FUSED_FC0=1
  run tile LIST_FUSED_FC0=x
done
