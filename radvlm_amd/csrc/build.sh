#!/bin/bash
# Build libradvlm_hip.so (gfx950) in-tree. hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
OUT=../libradvlm_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -I../../include -Wno-unused-result"
mkdir -p build
pids=()
for f in gemm_bf16 attention ops; do
  ( hipcc $FLAGS -c $f.hip -o build/$f.o ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC build/gemm_bf16.o build/attention.o build/ops.o -o $OUT
echo "built $OUT"
