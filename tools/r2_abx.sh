#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python learning-implicitly-from-spatial-transformers-network_amd/build.py --force > /dev/null 2>&1
timeout -k 10 300 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "prep or fused or percep or odd or saturates or non_finite" 2>&1 | tail -3
timeout -k 10 1000 bash tools/ab_build.sh "$@" > gpurun_out/r2_abx.log 2>&1
python - <<'PY'
import re
for line in open("gpurun_out/r2_abx.log"):
    m = re.match(r"\[(.*?)\] rep (\d): ([\d.]+) Mpts/s (\{.*\})", line)
    if m:
        d = eval(m.group(4))
        print(f"{m.group(1):26s} rep {m.group(2)}: {m.group(3)} Mpts/s " + " ".join(f"{k.replace('gather_','g_').replace('prep_','p_')[:9]} {v:.3f}" for k, v in d.items() if not k.startswith('_')))
PY
