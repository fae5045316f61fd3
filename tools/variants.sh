#!/bin/bash
# Build compile-time variants of the library HERE (no GPU needed) for an A/B run on the GPU box:
#   bash tools/variants.sh name1="<flags>" name2="<flags>" ...   ->  variants/<name>.so   (default build restored at the end)
# On the box: LIST_HIP_LIB=variants/<name>.so python bench.py ...   (tools/ab_variants.sh)
set -e
cd "$(dirname "$0")/.."
mkdir -p variants
for spec in "$@"; do
  name="${spec%%=*}"; flags="${spec#*=}"
  LIST_HIPCC_FLAGS="$flags" python learning-implicitly-from-spatial-transformers-network_amd/build.py --force > /dev/null 2>&1
  cp learning-implicitly-from-spatial-transformers-network_amd/csrc/liblist_hip.so variants/$name.so
  echo "built variants/$name.so  [$flags]"
done
python learning-implicitly-from-spatial-transformers-network_amd/build.py --force > /dev/null 2>&1
