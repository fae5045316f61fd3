#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 bash tools/ab_build.sh "" "-DLIST_NO_NAN_PROBE" "-DLIST_NO_NAN_PROBE -DLIST_SAT_H_MINMAX -DLIST_HALF4_SCALAR" > gpurun_out/r2_ab2.log 2>&1
cat gpurun_out/r2_ab2.log
