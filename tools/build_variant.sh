#!/bin/bash
# build a variant of the library with extra -D flags:  tools/build_variant.sh out.so -DRV_GROUP_M=8
set -e
cd "$(dirname "$0")/../radvlm_amd/csrc"
OUT=$1; shift
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -I../../include -Wno-unused-result $@"
mkdir -p build/var
for f in gemm_bf16 attention ops; do hipcc $FLAGS -c $f.hip -o build/var/$f.o & done; wait
hipcc --offload-arch=gfx950 -shared -fPIC build/var/gemm_bf16.o build/var/attention.o build/var/ops.o -o ../$OUT
