#!/bin/bash
cd "$(dirname "$0")/.."
LIST_HIPCC_FLAGS="-DLIST_BOX_DEBUG" python learning-implicitly-from-spatial-transformers-network_amd/build.py --force > /dev/null 2>&1
python bench.py --steps 1 --warmup 0 --precision fp16 --no-cpu-baseline > gpurun_out/boxdbg.log 2>&1
python - <<'PY'
import re, collections
ok = collections.Counter(); dims = collections.defaultdict(list)
for line in open("gpurun_out/boxdbg.log"):
    m = re.match(r"BOX (\d+) D=(\d+) ok=(\d) n=(-?\d+) (-?\d+) (-?\d+)", line)
    if m:
        D = int(m.group(2)); ok[(D, int(m.group(3)))] += 1
        dims[D].append(int(m.group(4)) * int(m.group(5)) * int(m.group(6)))
print(ok)
for D, v in dims.items():
    v.sort(); print(D, "voxels in box: median", v[len(v)//2], "p90", v[int(len(v)*0.9)], "max", v[-1])
PY
