#!/bin/bash
# Host-only AddressSanitizer + UBSan build of the C ABI's host side ("-fsanitize=address host build of the C ABI"):
# list_capi.hip compiled for the host alone, every kernel launcher replaced by an aborting stub (tests/csrc/), linked
# with the driver tests/csrc/capi_host_asan.cpp.  Usage: bash tools/build_capi_host_asan.sh <output binary>
set -e
cd "$(dirname "$0")/.."
OUT=${1:-/tmp/capi_host_asan}
TMP=$(mktemp -d)
CS=learning-implicitly-from-spatial-transformers-network_amd/csrc
H=${HIPCC:-/opt/rocm/bin/hipcc}
F="--cuda-host-only -x hip -O1 -g -std=c++17 -fsanitize=address,undefined -fno-gpu-sanitize -fno-sanitize-recover=undefined -Wno-unused-value -I include -I $CS"
$H $F -c $CS/list_capi.hip -o $TMP/capi.o
$H $F -c tests/csrc/capi_host_stubs.cpp -o $TMP/stubs.o
$H $F -c tests/csrc/capi_host_asan.cpp -o $TMP/drv.o
$H -fsanitize=address,undefined $TMP/capi.o $TMP/stubs.o $TMP/drv.o -o $OUT
rm -rf $TMP
echo $OUT
