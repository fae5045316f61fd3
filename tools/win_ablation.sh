#!/bin/bash
# Ablations of the LDS-window adjoint (k_scatter_vox_win): kernel durations from a rocprofv3 kernel trace of the in-line
# backward, one prebuilt variant per pass.  bash tools/win_ablation.sh name1 name2 ...   (variants/<name>.so)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
export LIST_BWD_OVERLAP=0
for name in "$@"; do
  export LIST_HIP_LIB=$PWD/variants/$name.so
  rm -rf gpurun_out/winprof_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/winprof_$name -- python3 tools/bwd_bench.py fp16 5 > /dev/null 2> gpurun_out/winprof_$name.err
  python3 - "$name" <<'PY'
import csv, glob, sys
name = sys.argv[1]
for f in glob.glob(f"gpurun_out/winprof_{name}/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "scatter_vox_win" in r["Name"]:
            lvl = "16^3" if "18432" in r["Name"] else "8^3"
            print(f"[{name}] window level {lvl}: avg {float(r['AverageNs'])/1e3:8.1f} us over {r['Calls']} calls")
PY
  rm -rf gpurun_out/winprof_$name
done
