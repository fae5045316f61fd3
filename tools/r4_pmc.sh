#!/bin/bash
# PMC traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the forward kernels, fused fc_0 on (default) or off
# (LIST_FUSED_FC0=0 in the environment): bash tools/r4_pmc.sh <tag>
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
T=${1:-r04}
export TMPDIR=/tmp
rm -rf gpurun_out/${T}prof
A="--precision fp16 --no-cpu-baseline --steps 5 --warmup 1 --sustained-steps 0 --no-train-step --no-channels-last-alt"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${T}prof/fetch -- python3 bench.py $A > /dev/null 2> gpurun_out/${T}prof_fetch.err; echo "fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${T}prof/write -- python3 bench.py $A > /dev/null 2> gpurun_out/${T}prof_write.err; echo "write rc=$?"
python tools/pmc_traffic.py gpurun_out/${T}prof/fetch gpurun_out/${T}prof/write fp16 gpurun_out/${T}_pmc_traffic.json > gpurun_out/${T}_pmc_traffic.txt
cat gpurun_out/${T}_pmc_traffic.txt
rm -rf gpurun_out/${T}prof
