#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 1100 bash tools/ab_build.sh "$@" > gpurun_out/r2_abx.log 2>&1
python - <<'PY'
import re
for line in open("gpurun_out/r2_abx.log"):
    m = re.match(r"\[(.*?)\] rep (\d): ([\d.]+) Mpts/s (\{.*\})", line)
    if m:
        d = eval(m.group(4))
        print(f"{m.group(1):46s} rep {m.group(2)}: {m.group(3)} Mpts/s " + " ".join(f"{k.replace('gather_','g_').replace('prep_','p_')[:7]} {v:.3f}" for k, v in d.items() if not k.startswith('_') and k not in ('prep_weights','exact_redo','sort_points','fc_1','fc_2_out')))
PY
