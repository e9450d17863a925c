// Common device helpers for the gfx950 (CDNA4) kernels of the LLaVA hot path.
// Wave = 64 lanes; MFMA shape used throughout: v_mfma_f32_16x16x32_bf16.
//   A operand: lane l holds A[row = l&15][k = 8*(l>>4) + j], j = 0..7   (16 contiguous bytes along k)
//   B operand: lane l holds B[k = 8*(l>>4) + j][col = l&15]
//   C/D      : lane l, reg r holds D[row = 4*(l>>4) + r][col = l&15]
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define RV_OK 0
#define RV_ERR_ARG -1
#define RV_ERR_LAUNCH -2

#define DEVINL __device__ __forceinline__

DEVINL float bf2f(bf16 x) { return (float)x; }
DEVINL bf16 f2bf(float x) { return (bf16)x; }

DEVINL f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

DEVINL int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
DEVINL int lane_id() { return (int)(threadIdx.x & 63); }

// 16-byte async global -> LDS copy (LDS-DMA). LDS destination = wave-uniform base + lane*16.
DEVINL void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

DEVINL float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
DEVINL float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide sum for blocks of NW waves; `red` is NW floats of LDS. All threads get the result.
template <int NW>
DEVINL float block_sum(float v, float* red) {
    v = wave_sum(v);
    if (NW == 1) return v;
    __syncthreads();
    if (lane_id() == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) t += red[i];
    return t;
}

// XCD-aware bijective remap of a 1-D block id: blocks that share an XCD (b % 8 equal) get a contiguous chunk.
DEVINL int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// Counter-based dropout mask (regenerable from (seed, element index): backward re-creates the forward's mask).  One splitmix64 value
// serves FOUR consecutive elements, 16 bits each: element i is KEPT when bits [16 (i & 3), +16) of hash(seed, i >> 2) are >= round(p * 2^16).
// (A hash per element made the mask generation, not HBM, the bound of every kernel that applies it inside another pass.)
DEVINL unsigned long long rv_hash64(unsigned long long seed, unsigned long long b) {
    unsigned long long z = (b + seed * 0x9E3779B97F4A7C15ull) + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// keep flags of the 8 consecutive elements e0 .. e0 + 7 (e0 % 8 == 0) as the low 8 bits
DEVINL unsigned rv_keep8(unsigned long long seed, unsigned long long e0, unsigned thr16) {
    const unsigned long long h0 = rv_hash64(seed, e0 >> 2), h1 = rv_hash64(seed, (e0 >> 2) + 1);
    unsigned m = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        m |= (((unsigned)(h0 >> (16 * j)) & 0xFFFFu) >= thr16 ? 1u : 0u) << j;
        m |= (((unsigned)(h1 >> (16 * j)) & 0xFFFFu) >= thr16 ? 1u : 0u) << (4 + j);
    }
    return m;
}
static inline unsigned rv_dropout_thr16(float p) { return (unsigned)((double)p * 65536.0 + 0.5); }

static inline int rv_check_launch() { return hipGetLastError() == hipSuccess ? RV_OK : RV_ERR_LAUNCH; }
