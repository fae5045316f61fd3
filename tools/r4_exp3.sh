#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() {   # name lib env...
  local name=$1 lib=$2; shift 2
  env "$@" LIST_HIP_LIB=$PWD/variants/$lib.so timeout -k 10 200 python bench.py --steps ${STEPS:-20} --warmup 3 --precision ${PREC:-fp16} --no-cpu-baseline --no-train-step --no-channels-last-alt --sustained-steps 0 2>gpurun_out/exp3_$name.err | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernel_ms'];print('[$name]: step', round(d['ms_per_step'],4), 'ev_med', round(d['step_events_ms']['median'],4), 'group', round(k['gathers_back_to_back'],4), 'fc_0', round(k['fc_0'],4), 'g+f', round(k['gathers_back_to_back']+k['fc_0'],4))" | tee -a gpurun_out/exp3.log
}
for rep in 1 2; do
  for v in "$@"; do
    run $v $v LIST_FUSED_FC0=1
  done
  run unfused $1 LIST_FUSED_FC0=0
done
