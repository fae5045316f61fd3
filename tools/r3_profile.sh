#!/bin/bash
# round-3 evidence for profiles/: kernel trace + stats of the default bench command, the two PMC passes
# (FETCH_SIZE / WRITE_SIZE, separately, with --kernel-trace only), MFMA utilisation counters of fc_0.
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
T=${TAG:-r03}
# the library travels with the snapshot; rebuild only if a source is newer, and stop if that fails (a stale .so would be profiled silently)
python learning-implicitly-from-spatial-transformers-network_amd/build.py > gpurun_out/${T}prof_build.log 2>&1 || { echo "build failed"; tail -5 gpurun_out/${T}prof_build.log; exit 1; }
export TMPDIR=/tmp
rm -rf gpurun_out/${T}prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}prof/kt -- python3 bench.py --no-cpu-baseline --sustained-steps 0 > gpurun_out/${T}prof_kt.json 2> gpurun_out/${T}prof_kt.err; echo "kt rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${T}prof/fetch -- python3 bench.py --precision fp16 --no-cpu-baseline --steps 5 --warmup 1 --sustained-steps 0 --no-train-step --no-channels-last-alt > /dev/null 2> gpurun_out/${T}prof_fetch.err; echo "fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${T}prof/write -- python3 bench.py --precision fp16 --no-cpu-baseline --steps 5 --warmup 1 --sustained-steps 0 --no-train-step --no-channels-last-alt > /dev/null 2> gpurun_out/${T}prof_write.err; echo "write rc=$?"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${T}prof/mfma -- python3 bench.py --precision fp16 --no-cpu-baseline --steps 5 --warmup 1 --sustained-steps 0 --no-train-step --no-channels-last-alt > /dev/null 2> gpurun_out/${T}prof_mfma.err; echo "mfma rc=$?"
python tools/summarize_prof.py gpurun_out/${T}prof/kt > gpurun_out/${T}_rocprof_summary.txt
python tools/summarize_prof.py gpurun_out/${T}prof/mfma > gpurun_out/${T}_pmc_mfma_utilisation.txt
python tools/pmc_traffic.py gpurun_out/${T}prof/fetch gpurun_out/${T}prof/write fp16 gpurun_out/${T}_pmc_traffic.json > gpurun_out/${T}_pmc_traffic.txt
cp gpurun_out/${T}prof/kt/*/*kernel_stats.csv gpurun_out/${T}_kernel_stats.csv 2>/dev/null
head -30 gpurun_out/${T}_rocprof_summary.txt
cat gpurun_out/${T}_pmc_traffic.txt
rm -rf gpurun_out/${T}prof/fetch gpurun_out/${T}prof/write gpurun_out/${T}prof/mfma gpurun_out/${T}prof/kt
