#!/bin/bash
# A/B: the fused + projected fc_0 with and without the epilogue's tap loads (what the exposed sampling costs)
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for flags in "" "-DLIST_FUSED_PROJ_NO_SAMPLE"; do
    LIST_HIPCC_FLAGS="$flags" python learning-implicitly-from-spatial-transformers-network_amd/build.py --force > /dev/null 2>&1
    LIST_HIPCC_FLAGS="$flags" timeout -k 10 200 python tools/imgproj_stages.py fp16 2>&1 | grep "proj-fused" | tail -1 | sed "s/^/[$flags] rep $rep: /"
  done
done
python learning-implicitly-from-spatial-transformers-network_amd/build.py --force > /dev/null 2>&1
