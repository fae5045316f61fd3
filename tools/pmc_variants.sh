#!/bin/bash
# MFMA-busy cycles and the clock fc_0 holds, per prebuilt variant: bash tools/pmc_variants.sh name1 name2 ...
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
for name in "$@"; do
  rm -rf gpurun_out/pmcv_$name
  LIST_HIP_LIB=$PWD/variants/$name.so rocprofv3 --kernel-trace --pmc ${PMC:-SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE} --output-format csv -d gpurun_out/pmcv_$name -- python3 bench.py --precision fp16 --no-cpu-baseline --no-train-step --no-channels-last-alt --steps 5 --warmup 1 > /dev/null 2> gpurun_out/pmcv_$name.err; echo "$name rc=$?"
  python tools/summarize_prof.py gpurun_out/pmcv_$name > gpurun_out/pmcv_$name.txt
  grep -A3 "k_gemm_nt_pp<0, 1>\|k_gemm_nt_pp<0,1>" gpurun_out/pmcv_$name.txt | head -12
  rm -rf gpurun_out/pmcv_$name
done
