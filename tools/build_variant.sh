#!/bin/bash
# Build a variant of the library HERE (hipcc cross-compiles gfx950) into variants/<name>.so, with #define values edited in a scratch copy of the sources; the .so files travel
# to the GPU box with the snapshot, so a same-box A/B (tools/ab_prebuilt.sh) spends no GPU time compiling.
# usage: bash tools/build_variant.sh <name> [FILE:MACRO=value ...]      (FILE relative to repaq_amd/csrc)
set -e
cd "$(dirname "$0")/.."; name=$1; shift; T=$(mktemp -d); mkdir -p variants $T/repaq_amd; cp -r repaq_amd/csrc $T/repaq_amd/; cp -r include $T/
for e in "$@"; do f=${e%%:*}; kv=${e#*:}; m=${kv%%=*}; v=${kv#*=}; grep -qE "^#define $m " $T/repaq_amd/csrc/$f || { echo "no #define $m in $f"; exit 1; }; sed -i -E "s/^(#define $m )[^ ]+/\1$v/" $T/repaq_amd/csrc/$f; done
C=$T/repaq_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -Wl,--version-script=repaq_amd/csrc/exports.map -Wno-unused-result -I$T/include $C/rfq_api.hip $C/rfq_encode.hip $C/rfq_decode.hip -o variants/$name.so
rm -rf $T; ls -la variants/$name.so
