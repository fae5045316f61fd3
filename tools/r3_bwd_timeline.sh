#!/bin/bash
# kernel trace of the training step (forked backward) of ONE library build -> timeline of one step + the forked phase's
# span against the sum of its kernel durations:  LIB=variants/x.so NAME=x bash tools/r3_bwd_timeline.sh
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
N=${NAME:-head}
rm -rf gpurun_out/bwdprof_$N
[ -n "$LIB" ] && export LIST_HIP_LIB=$PWD/$LIB
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/bwdprof_$N -- python3 tools/bwd_bench.py ${PREC:-fp16} 5 > gpurun_out/bwdprof_$N.log 2>&1; rc=$?; echo "[$N] rc=$rc"
[ $rc -eq 0 ] || { tail -5 gpurun_out/bwdprof_$N.log; exit $rc; }
python tools/timeline.py gpurun_out/bwdprof_$N > gpurun_out/${TAG:-r03}_bwd_timeline_$N.txt
tail -n 12 gpurun_out/${TAG:-r03}_bwd_timeline_$N.txt
rm -rf gpurun_out/bwdprof_$N
