#!/bin/bash
# A/B: launch order of the gathers (compile-time LIST_GATHER_SEQ) on the whole default bench line: every mode and the
# training step, two interleaved repetitions on one box
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for seq in 45I123T 12345IT 123I45T; do
    flags="-DLIST_GATHER_SEQ=$seq"
    LIST_HIPCC_FLAGS="$flags" python learning-implicitly-from-spatial-transformers-network_amd/build.py --force > /dev/null 2>&1
    LIST_HIPCC_FLAGS="$flags" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustained-steps 0 --no-channels-last-alt 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());s=d['summary'];print('[$seq] rep $rep:', 'fp16', s['fp16']['ms'], 'bf16x3', s['bf16x3']['ms'], 'bf16', s['bf16']['ms'], 'train', s['train_ms'], 'unfused', s['unfused'][0])"
  done
done
python learning-implicitly-from-spatial-transformers-network_amd/build.py --force > /dev/null 2>&1
