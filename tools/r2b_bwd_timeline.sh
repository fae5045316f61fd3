#!/bin/bash
# kernel trace of the training step (forked backward) -> per-kernel summary + timeline of one step
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/bwdprof
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/bwdprof -- python3 tools/bwd_bench.py ${PREC:-fp16} 5 > gpurun_out/bwdprof.log 2>&1; echo "rc=$?"
python tools/summarize_prof.py gpurun_out/bwdprof > gpurun_out/${TAG:-r02b}_bwd_rocprof_summary.txt
echo "== timeline of one step (k_head to k_head)" >> gpurun_out/${TAG:-r02b}_bwd_rocprof_summary.txt
python tools/timeline.py gpurun_out/bwdprof >> gpurun_out/${TAG:-r02b}_bwd_rocprof_summary.txt
tail -n 4 gpurun_out/bwdprof.log
rm -rf gpurun_out/bwdprof
