#!/bin/bash
# Round 4: the single-GPU BASELINE configs through bench.py (JSON lines under gpurun_out/), whole model included.
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
T=${TAG:-r04}
timeout -k 10 500 python bench.py > gpurun_out/${T}_bench_n1.json 2> gpurun_out/${T}_bench_n1.err; echo "metric rc=$?"
timeout -k 10 500 python bench.py --workload list_grid256_b1 --steps 3 --warmup 1 --no-channels-last-alt > gpurun_out/${T}_bench_config4_grid256.json 2> gpurun_out/${T}_bench_config4.err; echo "c4 rc=$?"
timeout -k 10 500 python bench.py --workload list_im2sdf_b8_n50k_512 --steps 10 --warmup 2 --no-channels-last-alt > gpurun_out/${T}_bench_config5_b8_n50k_512.json 2> gpurun_out/${T}_bench_config5.err; echo "c5 rc=$?"
timeout -k 10 500 python bench.py --whole-model --precision fp16 --no-cpu-baseline --no-train-step --no-channels-last-alt --sustained-steps 0 > gpurun_out/${T}_bench_whole_model.json 2> gpurun_out/${T}_bench_whole.err; echo "whole rc=$?"
for f in gpurun_out/${T}_bench_n1.json gpurun_out/${T}_bench_config4_grid256.json gpurun_out/${T}_bench_config5_b8_n50k_512.json; do
python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    print(sys.argv[1], json.dumps(d["summary"]))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
python - gpurun_out/${T}_bench_whole_model.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); w=d["whole_model"]
print("whole model: forward", round(w["ms_per_forward"],2), "ms, query_sdf", round(w["query_sdf_ms"],3), "ms; fp16 vox encoder:", round(w["vox_encoder_fp16"]["ms_per_forward"],2), round(w["vox_encoder_fp16"]["query_sdf_ms"],3))
PY
for f in gpurun_out/${T}_bench_*.err; do tail -n 1 "$f"; done
