#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q > gpurun_out/r2_gputest6.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2_gputest6.log
tail -4 gpurun_out/r2_gputest6.log
timeout -k 10 900 bash tools/ab_build.sh "" "-DLIST_PREP_LEVEL_MAJOR" "-DLIST_TRANSPOSE_NO_FUSE" "-DLIST_PREP_RY=32" "-DLIST_PREP_RY=8" > gpurun_out/r2_ab5.log 2>&1
python - <<'PY'
import re
for line in open("gpurun_out/r2_ab5.log"):
    m = re.match(r"\[(.*?)\] rep (\d): ([\d.]+) Mpts/s (\{.*\})", line)
    if m:
        d = eval(m.group(4))
        print(f"{m.group(1):28s} rep {m.group(2)}: {m.group(3)} Mpts/s  prep_img {d['prep_img_resize_nhwc']:.3f} prep_vox {d['prep_vox_ndhwc']:.3f} fc_0 {d['fc_0']:.3f} gathers {sum(v for k,v in d.items() if k.startswith('gather')):.3f}")
PY
