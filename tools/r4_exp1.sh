#!/bin/bash
# Round 4, experiment 1 (one box): gathers on several queues (LIST_FWD_FORK), the 2-D resize beside the 3-D hand-off
# (LIST_BENCH_PREP_FORK), fc_0 with its A operand resident in L2 (stage (a) of the fused-kernel question), non-temporal X
# stores in the 2-D gather.  Prebuilt variants/{base,aresident,imgnt}.so (tools/variants.sh).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() {   # name lib env...
  local name=$1 lib=$2; shift 2
  env "$@" LIST_HIP_LIB=$PWD/variants/$lib.so timeout -k 10 200 python bench.py --steps ${STEPS:-20} --warmup 3 --precision ${PREC:-fp16} --no-cpu-baseline --no-train-step --no-channels-last-alt --sustained-steps 0 2>gpurun_out/exp1_$name.err | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernel_ms'];print('[$name]: step', round(d['ms_per_step'],4), 'ev_med', round(d['step_events_ms']['median'],4), 'prep', round(k['prep_img_resize_nhwc']+k['prep_vox_ndhwc'],4), 'sort', round(k['sort_points'],4), 'group', round(k['gathers_back_to_back'],4), 'fc_0', round(k['fc_0'],4), 'tail', round(k['fc_2_out'],4), 'img', round(k['gather_img'],4))" | tee -a gpurun_out/exp1.log
}
for rep in 1 2; do
  run base base
  run aresident aresident
  run imgnt imgnt
  run fork0 base LIST_FWD_FORK=00000000
  run forkA base LIST_FWD_FORK=00000011
  run forkB base LIST_FWD_FORK=00011122
  run forkC base LIST_FWD_FORK=00220112
  run forkD base LIST_FWD_FORK=00123321
  run prepfork base LIST_BENCH_PREP_FORK=1
  run prepforkC base LIST_BENCH_PREP_FORK=1 LIST_FWD_FORK=00220112
done
