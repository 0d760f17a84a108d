#!/bin/bash
# Build timing-only variants of the EXPERIMENTAL library (results may be WRONG; never shipped): experiments/tools/ablate.sh name "-DFLAG ..." ...
# Each lands in experiments/build_abl/<name>/libdiffab_hip.so; select one with DIFFAB_HIP_LIB=<path>.  The -D hooks (AT_ABL_*, RG_ABL_*,
# PJ_ABL_*, SPA_/SPB_/TN_ABL_*, AT_STAGGER_*, HF_*) are part of patches/product_to_experimental.patch and patches/r04_heads_finish_*.patch.
set -e
here="$(cd "$(dirname "$0")/.." && pwd)"
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  EXTRA="$flags" bash "$here/build.sh" >/dev/null
  mkdir -p "$here/build_abl/$name" && cp "$here/build/libdiffab_hip.so" "$here/build_abl/$name/"
  echo "built experiments/build_abl/$name ($flags)"
done
