#!/bin/bash
# round-2 (second session) experiment 1: GEMM regimes + row-chunk sweep of the metric workload
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 400 python tools/gemm_regimes.py > gpurun_out/gemm_regimes.jsonl 2> gpurun_out/gemm_regimes.err; echo "regimes rc=$?"
cat gpurun_out/gemm_regimes.jsonl
for R in 0 81920 40960 32768; do
  LIST_WS_ROWS=$R timeout -k 10 300 python bench.py --no-cpu-baseline --no-channels-last-alt --no-train-step --steps 20 --warmup 3 > gpurun_out/chunk_${R}.json 2> gpurun_out/chunk_${R}.err; echo "chunk $R rc=$?"
  python - gpurun_out/chunk_${R}.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print(sys.argv[1], round(d["value"]/1e6,2),"Mpts/s", round(d["ms_per_step"],3),"ms", {k: round(v["ms"],3) for k,v in d.get("per_kernel",{}).items()} if isinstance(d.get("per_kernel"),dict) else "")
PY
done
tail -n 3 gpurun_out/*.err
