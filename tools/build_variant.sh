#!/bin/bash
# build a variant of the library with extra -D flags:  tools/build_variant.sh out.so -DRV_GROUP_M=8
#   ONLY="attention attention_w64" tools/build_variant.sh ...   recompiles just those sources; the other objects come from the default build
set -e
cd "$(dirname "$0")/../radvlm_amd/csrc"
OUT=$1; shift
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -I../../include -Wno-unused-result $@"
ALL="gemm_bf16 attention attention_w64 ops"
mkdir -p build/var
for f in ${ONLY:-$ALL}; do
  EXTRA=""; [ $f = attention ] && [ -z "${NO_ATTN_FLAGS:-}" ] && EXTRA="-fno-slp-vectorize"       # as build.sh
  hipcc $FLAGS $EXTRA -c $f.hip -o build/var/$f.o &
done; wait
OBJS=""
for f in $ALL; do
  if [[ " ${ONLY:-$ALL} " == *" $f "* ]]; then OBJS="$OBJS build/var/$f.o"; else OBJS="$OBJS build/$f.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o ../$OUT
