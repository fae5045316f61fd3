#!/bin/bash
# rehearsal of the N > 1 bench path on the one-GPU box: 2 gloo ranks sharing the GPU (RCCL refuses duplicate devices)
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python learning-implicitly-from-spatial-transformers-network_amd/build.py > /dev/null 2>&1
LIST_BENCH_BACKEND=gloo timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/r02_bench_2rank_gloo.json 2> gpurun_out/r02_bench_2rank_gloo.err
echo "2-rank rc=$?"
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r02_bench_2rank_gloo.json").read().strip().split("\n")[-1])
    print("n_gpus", d["n_gpus"], "value", round(d["value"] / 1e6, 2), "Mpts/s", "ms", round(d["ms_per_step"], 3), d["config"]["parallelism"], "roof", round(d["roofline"]["frac"], 3))
except Exception as e:
    print("unreadable:", e)
PY
tail -5 gpurun_out/r02_bench_2rank_gloo.err
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r2_gputest7.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r2_gputest7.log
