#!/bin/bash
# rehearsal of the N > 1 bench path on the one-GPU box: 2 gloo ranks sharing the GPU (RCCL refuses duplicate devices),
# in BOTH scaling modes (weak: B = 8 per rank; strong: BASELINE config 3's B = 64 split over the ranks -- here a
# global batch of 5 so that the shards are ragged, 3 + 2).  Control flow and fields only -- never a scaling number.
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
T=${TAG:-r03}
python learning-implicitly-from-spatial-transformers-network_amd/build.py > gpurun_out/${T}_multi_build.log 2>&1 || { echo "build failed"; tail -5 gpurun_out/${T}_multi_build.log; exit 1; }
for mode in weak strong; do
  extra=""; [ $mode = strong ] && extra="--global-batch 5"
  LIST_BENCH_BACKEND=gloo timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --sustained-steps 50 --no-cpu-baseline --scaling $mode $extra > gpurun_out/${T}_bench_2rank_gloo_$mode.json 2> gpurun_out/${T}_bench_2rank_gloo_$mode.err
  echo "2-rank $mode rc=$?"
  python - "$T" "$mode" <<'PY'
import json, sys
try:
    d = json.loads(open(f"gpurun_out/{sys.argv[1]}_bench_2rank_gloo_{sys.argv[2]}.json").read().strip().split("\n")[-1])
    print("n_gpus", d["n_gpus"], "scaling", d["scaling"], "value", round(d["value"] / 1e6, 2), "Mpts/s", "ms", round(d["ms_per_step"], 3),
          d["config"]["parallelism"], "global_batch", d["config"]["global_batch"], "ranks", d["ranks"])
except Exception as e:
    print("unreadable:", e)
PY
  tail -3 gpurun_out/${T}_bench_2rank_gloo_$mode.err
done
