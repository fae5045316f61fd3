#!/bin/bash
# the three single-GPU BASELINE configs through bench.py (JSON lines under gpurun_out/)
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
T=${TAG:-r03}
timeout -k 10 500 python bench.py > gpurun_out/${T}_bench_n1.json 2> gpurun_out/${T}_bench_n1.err; echo "metric rc=$?"
timeout -k 10 500 python bench.py --workload list_grid256_b1 --steps 3 --warmup 1 --no-channels-last-alt > gpurun_out/${T}_bench_config4_grid256.json 2> gpurun_out/${T}_bench_config4.err; echo "c4 rc=$?"
timeout -k 10 500 python bench.py --workload list_im2sdf_b8_n50k_512 --steps 10 --warmup 2 --no-channels-last-alt > gpurun_out/${T}_bench_config5_b8_n50k_512.json 2> gpurun_out/${T}_bench_config5.err; echo "c5 rc=$?"
for f in gpurun_out/${T}_bench_n1.json gpurun_out/${T}_bench_config4_grid256.json gpurun_out/${T}_bench_config5_b8_n50k_512.json; do
python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    print(sys.argv[1], round(d["value"]/1e6,2),"Mpts/s", round(d["ms_per_step"],3),"ms", "roof", d["roofline"]["kernel"], round(d["roofline"]["frac"],3), "path", round(d["path_rates"]["whole_path_frac_of_binding_roof"],3), "parity", d["parity_max_abs_err_vs_cpu"], "cpu", d["cpu_baseline"] and round(d["cpu_baseline"]["value"]), "alt", d["alt"] and (d["alt"]["precision"], round(d["alt"]["value"]/1e6,2)), "train", d["train_step"] and (round(d["train_step"]["ms_per_step"],2), round(d["train_step"]["fp32_grade"]["ms_per_step"],2)))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
for f in gpurun_out/${T}_bench_*.err; do tail -n 2 "$f"; done
