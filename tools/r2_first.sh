#!/bin/bash
# round 2, first GPU call: GPU test suite, then A/B of the new resize kernel and the packed fp16 conversion
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2_gputest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2_gputest.log
tail -5 gpurun_out/r2_gputest.log
timeout -k 10 600 bash tools/ab_build.sh "" "-DLIST_PREP_IMG_NO_ROWS" "-DLIST_HALF4_SCALAR" > gpurun_out/r2_ab1.log 2>&1
cat gpurun_out/r2_ab1.log
