#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu > gpurun_out/exp3_pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 gpurun_out/exp3_pytest.log
STEPS=10 bash tools/ab_variants.sh base shape32
