#!/bin/bash
# Round-4 evidence for profiles/: the plain bench FIRST (its fc_0 HIP-event time is what the bench line's roofline
# uses), then -- same box, same call -- the kernel trace + stats of the same command, the two PMC passes (FETCH_SIZE /
# WRITE_SIZE, separately, --kernel-trace only), and the MFMA utilisation counters.  profiles/README.md quotes both the
# event time and the traced average of fc_0 from this one box.
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
T=${TAG:-r04}
python learning-implicitly-from-spatial-transformers-network_amd/build.py > gpurun_out/${T}prof_build.log 2>&1 || { echo "build failed"; tail -5 gpurun_out/${T}prof_build.log; exit 1; }
export TMPDIR=/tmp
rm -rf gpurun_out/${T}prof
A="--precision fp16 --no-cpu-baseline --steps 5 --warmup 1 --sustained-steps 0 --no-train-step --no-channels-last-alt"
python3 bench.py --no-cpu-baseline --sustained-steps 0 > gpurun_out/${T}prof_plain.json 2> gpurun_out/${T}prof_plain.err; echo "plain rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}prof/kt -- python3 bench.py --no-cpu-baseline --sustained-steps 0 > gpurun_out/${T}prof_kt.json 2> gpurun_out/${T}prof_kt.err; echo "kt rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${T}prof/fetch -- python3 bench.py $A > /dev/null 2> gpurun_out/${T}prof_fetch.err; echo "fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${T}prof/write -- python3 bench.py $A > /dev/null 2> gpurun_out/${T}prof_write.err; echo "write rc=$?"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${T}prof/mfma -- python3 bench.py $A > /dev/null 2> gpurun_out/${T}prof_mfma.err; echo "mfma rc=$?"
python tools/summarize_prof.py gpurun_out/${T}prof/kt > gpurun_out/${T}_rocprof_summary.txt
python tools/summarize_prof.py gpurun_out/${T}prof/mfma > gpurun_out/${T}_pmc_mfma_utilisation.txt
python tools/pmc_traffic.py gpurun_out/${T}prof/fetch gpurun_out/${T}prof/write fp16 gpurun_out/${T}_pmc_traffic.json > gpurun_out/${T}_pmc_traffic.txt
cp gpurun_out/${T}prof/kt/*/*kernel_stats.csv gpurun_out/${T}_kernel_stats.csv 2>/dev/null
python - <<PY
import json
p=json.loads(open("gpurun_out/${T}prof_plain.json").read().strip().split("\n")[-1])
k=json.loads(open("gpurun_out/${T}prof_kt.json").read().strip().split("\n")[-1])
print("fc_0 by HIP events: plain run", round(p["kernel_ms"]["fc_0"],4), "ms (frac", round(p["roofline"]["frac"],3), "), under the tracer", round(k["kernel_ms"]["fc_0"],4), "ms; step", round(p["ms_per_step"],4), "/", round(k["ms_per_step"],4))
PY
grep -n "k_fc0_fused\|k_gemm_nt_pp" gpurun_out/${T}_rocprof_summary.txt | head -5
cat gpurun_out/${T}_pmc_traffic.txt
rm -rf gpurun_out/${T}prof/fetch gpurun_out/${T}prof/write gpurun_out/${T}prof/mfma gpurun_out/${T}prof/kt
