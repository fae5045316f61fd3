#!/bin/bash
# ablations of k_scatter_vox_box (wrong results; kernel time in the in-line backward, rocprofv3 kernel trace): both window levels on it
cd "$(dirname "$0")/.."
for n in "$@"; do
  LIST_SCATTER_BOX=2 LIST_BWD_OVERLAP=0 TAG=r04b LIB=variants/adj_$n.so NAME=adj_$n bash tools/r3_bwd_timeline.sh > /dev/null 2>&1
  echo "[$n] $(grep scatter_vox_box gpurun_out/r04b_bwd_timeline_adj_$n.txt | awk '{print $(NF-3)}' | tr '\n' ' ') us (16^3, 8^3)"
done
