#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q > gpurun_out/r2_gputest3.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2_gputest3.log
tail -4 gpurun_out/r2_gputest3.log
timeout -k 10 600 bash tools/ab_build.sh "" "-DLIST_HALF4_SCALAR" > gpurun_out/r2_ab3.log 2>&1
cat gpurun_out/r2_ab3.log
