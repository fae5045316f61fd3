#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_hip_backward.py -x -q -m gpu > gpurun_out/exp7_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -n 25 gpurun_out/exp7_pytest.log
[ $rc -eq 0 ] || exit $rc
PREC=bf16x3 STEPS=10 bash tools/ab_variants.sh base planar
PREC=bf16x3 LIST_BWD_OVERLAP=0 bash tools/ab_bwd_variants.sh base planar > gpurun_out/ab_x3i_bwd.log 2>&1; grep -E "===|wgrad_fc0|forward|wall|backward" gpurun_out/ab_x3i_bwd.log
