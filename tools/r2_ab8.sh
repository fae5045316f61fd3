#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python learning-implicitly-from-spatial-transformers-network_amd/build.py --force > /dev/null 2>&1
timeout -k 10 300 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "prep or fused or percep or odd" 2>&1 | tail -3
timeout -k 10 1000 bash tools/ab_build.sh "" "-DLIST_PREP_NO_XCD" > gpurun_out/r2_ab8.log 2>&1
python - <<'PY'
import re
for line in open("gpurun_out/r2_ab8.log"):
    m = re.match(r"\[(.*?)\] rep (\d): ([\d.]+) Mpts/s (\{.*\})", line)
    if m:
        d = eval(m.group(4))
        print(f"{m.group(1):34s} rep {m.group(2)}: {m.group(3)} Mpts/s  prep_img {d['prep_img_resize_nhwc']:.3f} prep_vox {d['prep_vox_ndhwc']:.3f} fc_0 {d['fc_0']:.3f} gather_img {d['gather_img']:.3f} gathers {sum(v for k,v in d.items() if k.startswith('gather')):.3f}")
PY
