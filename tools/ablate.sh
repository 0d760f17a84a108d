#!/bin/bash
# Build timing-only variants of libdiffab_hip.so (results may be WRONG; never shipped): tools/ablate.sh name "-DFLAG ..." ...
# Each lands in diffab-pytorch_amd/build_abl/<name>/libdiffab_hip.so; select one with DIFFAB_HIP_LIB=<path>.
set -e
cd "$(dirname "$0")/../diffab-pytorch_amd/csrc"
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  make -s EXPERIMENTAL=1 OBJ=../build_abl/$name OUT=../build_abl/$name EXTRA="$flags" >/dev/null
  echo "built build_abl/$name ($flags)"
done
