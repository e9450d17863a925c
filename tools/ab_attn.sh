#!/bin/bash
# Same-box A/B of two builds of the library on the attention micro-benchmark (boxes of the pool differ by up to 8 %):
#   radvlm_amd/lib_A.so vs radvlm_amd/lib_B.so, interleaved A B A B.   Usage (through gpurun): bash tools/ab_attn.sh
for r in 1 2; do
  for k in A B; do
    echo "== $k (round $r)"; RADVLM_HIP_LIB=$PWD/radvlm_amd/lib_$k.so timeout -k 10 200 python tools/attn_bench.py 2>&1 | grep "natural-layout" || exit 1
  done
done
