#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_backward.py -x -q -m gpu > gpurun_out/exp6_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -n 15 gpurun_out/exp6_pytest.log
[ $rc -eq 0 ] || exit $rc
LIST_BWD_OVERLAP=0 bash tools/ab_bwd_variants.sh base oldadj > gpurun_out/ab_adj_inline.log 2>&1; grep -E "===|wall|backward" gpurun_out/ab_adj_inline.log
LIST_BWD_OVERLAP=1 bash tools/ab_bwd_variants.sh base oldadj > gpurun_out/ab_adj_forked.log 2>&1; grep -E "===|wall|backward" gpurun_out/ab_adj_forked.log
