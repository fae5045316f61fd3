#!/bin/bash
# fused fc_0 on / off, interleaved pairs on one box (metric workload and config 5)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() {   # name workload env...
  local name=$1 wl=$2; shift 2
  env "$@" timeout -k 10 300 python bench.py --workload $wl --steps ${STEPS:-40} --warmup 5 --precision fp16 --no-cpu-baseline --no-train-step --no-channels-last-alt --sustained-steps 0 2>gpurun_out/exp5_$name.err | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernel_ms'];print('[$name $wl]: step', round(d['ms_per_step'],4), 'ev_med', round(d['step_events_ms']['median'],4), 'group', round(k['gathers_back_to_back'],4), 'fc_0', round(k['fc_0'],4))" | tee -a gpurun_out/exp5.log
}
for rep in 1 2 3 4 5; do
  run unfused list_im2sdf_b8_n20k_224 LIST_FUSED_FC0=0
  run fused list_im2sdf_b8_n20k_224 LIST_FUSED_FC0=1
done
for rep in 1 2; do
  run unfused list_im2sdf_b8_n50k_512 LIST_FUSED_FC0=0
  run fused list_im2sdf_b8_n50k_512 LIST_FUSED_FC0=1
done
