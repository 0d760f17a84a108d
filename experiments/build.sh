#!/bin/bash
# Builds experiments/build/libdiffab_hip.so: the product sources + the unadopted kernel variants (csrc/*.hip here) + the hooks of
# patches/product_to_experimental.patch (EXPERIMENTAL build: environment switches DIFFAB_*, the QUEUE / operand-plane / external-logits
# forms of the attention kernel, the *_ABL_* timing ablations, flags 2u / 4u / 8u).  The patch was cut against the product sources of
# the commit that introduced this directory (round 4); when it stops applying, check that commit out.
set -e
here="$(cd "$(dirname "$0")" && pwd)"; repo="$(dirname "$here")"
rm -rf "$here/build" && mkdir -p "$here/build/src/csrc" "$here/build/include"
cp "$repo"/diffab-pytorch_amd/csrc/* "$here"/csrc/*.hip "$here/build/src/csrc/"
cp "$repo"/include/diffab_hip.h "$here"/include/diffab_hip_experimental.h "$here/build/src/csrc/"
(cd "$here/build/src" && mv csrc new_csrc && patch -p0 -s < "$here/patches/product_to_experimental.patch" && mv new_csrc csrc)
mkdir -p "$here/build/include" && mv "$here/build/src/csrc/diffab_hip.h" "$here/build/src/csrc/diffab_hip_experimental.h" "$here/build/include/"
sed -i 's|../../include/|../../include/|g' "$here"/build/src/csrc/*.h "$here"/build/src/csrc/*.hip
make -C "$here/build/src/csrc" -j8 EXPERIMENTAL=1 OUT=../../ OBJ=../obj ${EXTRA:+EXTRA="$EXTRA"}
echo "built $here/build/libdiffab_hip.so  (select it with DIFFAB_HIP_LIB)"
