#!/bin/bash
# full GPU test suite + the default bench line
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/full_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -n 5 gpurun_out/full_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python bench.py > gpurun_out/full_bench.json 2> gpurun_out/full_bench.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/full_bench.json
