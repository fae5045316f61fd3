#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "pingpong or gemm" > gpurun_out/exp2_pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 gpurun_out/exp2_pytest.log
timeout -k 10 400 python tools/gemm_regimes.py > gpurun_out/gemm_regimes2.jsonl 2> gpurun_out/gemm_regimes2.err; echo "regimes rc=$?"
cat gpurun_out/gemm_regimes2.jsonl
timeout -k 10 300 python bench.py --no-cpu-baseline --no-channels-last-alt --no-train-step --steps 20 --warmup 3 > gpurun_out/exp2_bench.json 2> gpurun_out/exp2_bench.err; echo "bench rc=$?"
python - gpurun_out/exp2_bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print(sys.argv[1], round(d["value"]/1e6,2),"Mpts/s", round(d["ms_per_step"],3),"ms", {k: round(v["ms"],3) for k,v in d.get("per_kernel",{}).items()})
PY
tail -n 3 gpurun_out/exp2*.err gpurun_out/gemm_regimes2.err
