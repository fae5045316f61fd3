#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
# build HERE first: bash tools/variants.sh perlevel="-DLIST_PREP_PER_LEVEL"
export LIST_HIP_LIB=$PWD/variants/${VARIANT:-perlevel}.so
export TMPDIR=/tmp
rm -rf gpurun_out/prof_prep
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_prep -- python3 tools/prof_prep.py > gpurun_out/prof_prep.log 2>&1
python - <<'PY'
import csv, glob, collections
rows = []
for f in glob.glob("gpurun_out/prof_prep/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
by = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "k_prep_img_rows" in n or "transpose" in n:
        by[(n[:60], r.get("Grid_Size_X", r.get("Grid_Size")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in by.items():
    v = sorted(v)
    print(f"{k[0]:62s} grid {k[1]:>10s}  n {len(v):3d}  median {v[len(v)//2]:8.1f} us  min {v[0]:8.1f}")
PY
