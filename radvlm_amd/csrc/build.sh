#!/bin/bash
# Build libradvlm_hip.so (gfx950) in-tree. hipcc cross-compiles without a GPU.
# The compiler's per-kernel resource report is kept (build/*.res) and the build FAILS if a hot kernel (GEMM, attention) touches
# scratch memory: a rolled epilogue loop once turned the GEMM accumulators into a scratch array and cost 18 % unnoticed.
set -e
cd "$(dirname "$0")"
OUT=../libradvlm_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -I../../include -Wno-unused-result -Rpass-analysis=kernel-resource-usage"
mkdir -p build
pids=()
# attention.hip: no SLP vectorisation -- hipcc packs adjacent fp32 adds / multiplies of the softmax and dS arithmetic into v_pk_*_f32, which is slower
# than the two scalar instructions beside MFMAs (guide: "an anti-lever beside MFMAs"; same-box A/B profiles/r04_ab_attn_no_slp_merged_waits.txt)
for f in gemm_bf16 attention attention_w64 ops; do
  EXTRA=""; [ $f = attention ] && EXTRA="-fno-slp-vectorize"
  ( hipcc $FLAGS $EXTRA -c $f.hip -o build/$f.o 2> build/$f.res || { cat build/$f.res >&2; exit 1; } ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
grep -h "error\|warning" build/*.res | grep -v "Rpass" | head -20 || true
if grep -h -B8 "ScratchSize \[bytes/lane\]: [1-9]" build/gemm_bf16.res build/attention.res build/attention_w64.res | grep "Function Name"; then
  echo "ERROR: the kernels above use scratch memory (see build/*.res)" >&2
  exit 1
fi
# attention_w64.hip names its accumulator registers by hand (guide 5.7 item 4): the compiler must not touch the AGPR file there
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I. -I../../include -Wno-unused-result -S --cuda-device-only attention_w64.hip -o build/attention_w64.s 2>/dev/null
python3 audit_w64.py build/attention_w64.s
hipcc --offload-arch=gfx950 -shared -fPIC build/gemm_bf16.o build/attention.o build/attention_w64.o build/ops.o -o $OUT
echo "built $OUT"
