#!/bin/bash
# scalar level riding with the fine level's gather (default) vs k_gather_tail (LIST_TAIL_RIDE=0), interleaved on one box
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() {   # name prec env...
  local name=$1 prec=$2; shift 2
  env "$@" timeout -k 10 300 python bench.py --steps ${STEPS:-40} --warmup 5 --precision $prec --no-cpu-baseline --no-train-step --no-channels-last-alt --sustained-steps 0 2>gpurun_out/exp6_$name.err | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernel_ms'];print('[$name $prec]: step', round(d['ms_per_step'],4), 'ev_med', round(d['step_events_ms']['median'],4), 'group', round(k['gathers_back_to_back'],4), 'l1', round(k['gather_vox_l1'],4), 'tail', round(k['gather_tail'],4), 'fc_0', round(k['fc_0'],4))" | tee -a gpurun_out/exp6.log
}
for rep in 1 2 3; do
  run tailkernel fp16 LIST_TAIL_RIDE=0
  run ride fp16 LIST_TAIL_RIDE=1
done
run tailkernel bf16x3 LIST_TAIL_RIDE=0
run ride bf16x3 LIST_TAIL_RIDE=1
