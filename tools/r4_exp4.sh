#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() {   # name prec env...
  local name=$1 prec=$2; shift 2
  env "$@" timeout -k 10 200 python bench.py --steps ${STEPS:-20} --warmup 3 --precision $prec --no-cpu-baseline --no-train-step --no-channels-last-alt --sustained-steps 0 2>gpurun_out/exp4_$name.err | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernel_ms'];print('[$name $prec]: step', round(d['ms_per_step'],4), 'Mpts', round(d['value']/1e6,2), 'group', round(k['gathers_back_to_back'],4), 'fc_0', round(k['fc_0'],4), 'g+f', round(k['gathers_back_to_back']+k['fc_0'],4), 'img', round(k['gather_img'],4))" | tee -a gpurun_out/exp4.log
}
for rep in 1 2; do
  for prec in bf16x3 bf16; do
    run unfused $prec LIST_FUSED_FC0=0
    run fused $prec LIST_FUSED_FC0=1
  done
done
