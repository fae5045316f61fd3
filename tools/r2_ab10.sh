#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python learning-implicitly-from-spatial-transformers-network_amd/build.py --force > /dev/null 2>&1
timeout -k 10 400 python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -3
timeout -k 10 1000 bash tools/ab_build.sh "" "-DLIST_NEAR_NO_BOX" ${EXTRA_VARIANTS} > gpurun_out/r2_ab10.log 2>&1
python - <<'PY'
import re
for line in open("gpurun_out/r2_ab10.log"):
    m = re.match(r"\[(.*?)\] rep (\d): ([\d.]+) Mpts/s (\{.*\})", line)
    if m:
        d = eval(m.group(4))
        print(f"{m.group(1):28s} rep {m.group(2)}: {m.group(3)} Mpts/s  l3 {d['gather_vox_l3']:.3f} l4 {d['gather_vox_l4']:.3f} l5 {d['gather_vox_l5']:.3f} img {d['gather_img']:.3f} prep_img {d['prep_img_resize_nhwc']:.3f} fc_0 {d['fc_0']:.3f}")
PY
