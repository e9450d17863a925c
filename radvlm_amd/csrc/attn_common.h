// Shared pieces of the attention kernels (attention.hip: head_dim 64 + the two-waves-per-SIMD natural-layout kernels;
// attention_w64.hip: the one-wave-per-SIMD head_dim 128 kernels).  Included inside each translation unit's anonymous namespace.
#pragma once
#include <type_traits>
#include <stdlib.h>
#include "common.h"
#include "radvlm_hip.h"

struct AttnParams {
    const bf16 *q, *k, *v, *o, *dout;   // natural (token-major) operands
    const bf16 *qT, *kT, *vT, *doT;     // transposed copies [b,h,HD,S_pad]
    bf16 *out, *dq, *dk, *dv;
    float *lse, *delta;
    const int* lens;
    const int* cu;      // packed (varlen) batches: sample b owns token rows [cu[b], cu[b+1]) of q/k/v/out; NULL = b*S padded layout
    const bf16* zeros;
    long ld_q, ld_k, ld_v, ld_o, ld_do, ld_dq, ld_dk, ld_dv;
    int B, H, S, S_pad;
    int kdiv, qrep;  // dK/dV pass: block head h reads k/v head h / kdiv and walks query heads [h*qrep, (h+1)*qrep)
    int Hkv, nrep;   // grouped-query attention: query head h reads key/value head h / nrep (repeat_kv, modeling_llama.py:201-210)
    float scale;
    // backward only: adjoint of the rotary embedding applied to dQ / dK in the epilogue (q, k are stored rotated; the projection
    // weights see un-rotated gradients).  cs = fp32 [positions, HD/2, 2]; position of row r of sample b: rope_pos[rb + r] or r.
    const float* rope_cs; const int* rope_pos;
    int paired;       // forward, causal: a block runs the query-block pair (nq - 1 - x, x) instead of one query block (attn_fwd_nat_kernel)
    int rope_dk;      // the dK/dV pass rotates dK itself (0 when per-query-head partials are summed first: group_sum_heads_kernel rotates)
};

namespace {


// Inverse rotation of one lane's accumulator column set: v[db] holds dimensions 16 db + 4 g + r of one row, partners are db and
// db + DB/2.  Rounds through bf16 first (the unfused path stored the gradient before rotating it: same rounding points).
template <int DB>
DEVINL void unrope(f32x4 (&v)[DB], const float* cs_row, int g) {
#pragma unroll
    for (int db = 0; db < DB / 2; ++db) {
        const float* c = cs_row + (16 * db + 4 * g) * 2;
        const f32x4 t0 = *(const f32x4*)c, t1 = *(const f32x4*)(c + 4);
        const float co[4] = {t0[0], t0[2], t1[0], t1[2]}, si[4] = {t0[1], t0[3], t1[1], t1[3]};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = bf2f(f2bf(v[db][r])), b = bf2f(f2bf(v[db + DB / 2][r]));
            v[db][r] = a * co[r] + b * si[r];
            v[db + DB / 2][r] = b * co[r] - a * si[r];
        }
    }
}

template <int LEFT> DEVINL void lds_wait4(bf16x8& a, bf16x8& b, bf16x8& c, bf16x8& d) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "i"(LEFT) : "memory");
}
template <int LEFT> DEVINL void lds_wait2(bf16x8& a, bf16x8& b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "i"(LEFT) : "memory");
}
// wait for one batch of N (2 or 4) fragments
template <int N, int LEFT> DEVINL void lds_wait(bf16x8 (&f)[N]) {
    if constexpr (N == 4) lds_wait4<LEFT>(f[0], f[1], f[2], f[3]);
    else lds_wait2<LEFT>(f[0], f[1]);
}
DEVINL bf16x8 pack8(f32x4 a, f32x4 b) {
    return bf16x8{f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3]), f2bf(b[0]), f2bf(b[1]), f2bf(b[2]), f2bf(b[3])};
}
DEVINL float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

// ================================================================================================ natural-layout kernels (HD = 128)
// No transposed operand copies: every tile is staged as it lies in memory ([64 token rows][128 head dims], 256-byte rows) into ONE
// LDS image that serves both kinds of MFMA operand reads --
//   row reads   (contraction over the head dimension): ds_read_b128 of 8 consecutive dims of one row;
//   column reads (contraction over the tile's token rows): ds_read_b64_tr_b16 pairs, which hand lane (g, i) dimension 16 db + i of the
//                rows 32 p + 4 g + {0..3} and 32 p + 16 + 4 g + {0..3} -- exactly the contraction order kappa(g, j) in which a score
//                tile sits in the accumulators, so P / dS feed the next MFMA chain straight from registers as before.
// Swizzle: 16-byte chunk ch of row r is stored at chunk ch ^ ((r & 7) << 1).  Row reads: a 16-lane service group sees 8 rows x 2
// chunk parities -> 16 distinct 16-byte slots; column reads: the 8 rows of a 32-lane half land on 8 distinct 32-byte granules.
typedef __attribute__((ext_vector_type(4))) short s16x4n;
struct TFrag { s16x4n t0, t1; };
DEVINL int nswz(int r) { return (r & 7) << 1; }
DEVINL unsigned lds_off(const char* a) { return (unsigned)(unsigned long)(__attribute__((address_space(3))) const char*)a; }

template <int NROWS, int NW>
DEVINL void stage_nat(const bf16* src, long row_stride, int valid_rows, const bf16* zeros, char* lds, int wid, int lane) {
    constexpr int NI = NROWS * 16 / 64;
    static_assert(NI % NW == 0, "tile must split evenly over the block's waves");
#pragma unroll
    for (int i = 0; i < NI / NW; ++i) {
        const int it = i * NW + wid;
        const int c = it * 64 + lane;
        const int r = c >> 4, pch = c & 15;
        const int lc = pch ^ nswz(r);
        const bf16* g = (r < valid_rows) ? (src + (long)r * row_stride + lc * 8) : zeros;
        glds16(g, lds + it * 1024);
    }
}
// Buffer-addressed staging of a natural tile: resource = this sample's rows of the operand (rows past its end read as zero in
// hardware, also through the scalar offset), per-lane byte offsets computed once per block, scalar offset = first row of the tile
// (+ the head's column offset): an M0 write and a `buffer_load_dwordx4 ... lds` per piece, no per-piece address arithmetic or selects.
// A wave's piece i covers rows (i NW + wid) 4 + (lane >> 4): consecutive pieces lie 4 NW rows apart (a multiple of 8: same swizzle term), so
// ONE per-lane offset serves all of them and the piece index goes into the scalar offset.
template <int NROWS, int NW>
struct NatPlan {
    int v0, step;
    DEVINL void init(long row_stride, int wid, int lane) {
        const int c = wid * 64 + lane;
        const int r = c >> 4, pch = c & 15;
        v0 = (int)(((long)r * row_stride + (pch ^ nswz(r)) * 8) * 2);
        step = (int)(4 * NW * row_stride * 2);
    }
    DEVINL void stage(__amdgpu_buffer_rsrc_t rsrc, int soff, char* lds, int wid) const {
#pragma unroll
        for (int i = 0; i < NROWS * 16 / 64 / NW; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + (i * NW + wid) * 1024), 16, v0, soff + i * step, 0, 0);
    }
};
DEVINL __amdgpu_buffer_rsrc_t rows_rsrc(const bf16* base, long rows, long row_stride) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (unsigned)(rows * row_stride * 2), 0x00020000);
}

DEVINL void rdrow_asm(bf16x8& dst, const char* tile, int row, int chunk) {
    const unsigned o = lds_off(tile + row * 256 + ((chunk ^ nswz(row)) << 4));
    asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(o) : "memory");
}
// column-read fragment of dims [16 db, 16 db + 16) over rows kappa(g, 0..7) of the 32-row step p
DEVINL void rdcol_asm(TFrag& f, const char* tile, int p, int db, int lane) {
    const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
    const int r1 = 32 * p + 4 * g + qq;
    const unsigned o = lds_off(tile + r1 * 256 + (((2 * db + (pp >> 1)) ^ nswz(r1)) << 4) + ((pp & 1) << 3));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.t0) : "v"(o) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(f.t1) : "v"(o) : "memory");     // rows + 16: same swizzle term
}
DEVINL bf16x8 tf_get(const TFrag& f) {
    typedef __attribute__((ext_vector_type(8))) short s16x8n;
    const s16x8n r = {f.t0[0], f.t0[1], f.t0[2], f.t0[3], f.t1[0], f.t1[1], f.t1[2], f.t1[3]};
    return __builtin_bit_cast(bf16x8, r);
}
// one wait for two batches that were issued together (two s_waitcnt with the same count are one instruction too many per batch)
template <int LEFT> DEVINL void lds_wait8(bf16x8 (&a)[4], bf16x8 (&b)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "i"(LEFT) : "memory");
}
template <int LEFT> DEVINL void tf_wait8(TFrag (&f)[4], TFrag (&h)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%16)" : "+v"(f[0].t0), "+v"(f[0].t1), "+v"(f[1].t0), "+v"(f[1].t1), "+v"(f[2].t0), "+v"(f[2].t1), "+v"(f[3].t0), "+v"(f[3].t1),
                   "+v"(h[0].t0), "+v"(h[0].t1), "+v"(h[1].t0), "+v"(h[1].t1), "+v"(h[2].t0), "+v"(h[2].t1), "+v"(h[3].t0), "+v"(h[3].t1) : "i"(LEFT) : "memory");
}
template <int LEFT> DEVINL void tf_wait4(TFrag (&f)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(f[0].t0), "+v"(f[0].t1), "+v"(f[1].t0), "+v"(f[1].t1), "+v"(f[2].t0), "+v"(f[2].t1), "+v"(f[3].t0), "+v"(f[3].t1)
                 : "i"(LEFT) : "memory");
}

// Block -> (x block, head, sample).  Workgroups go to the 8 XCDs round-robin in linear order, and every XCD has its own 4-MiB L2: the
// remap hands each XCD a CONTIGUOUS range of (sample, head) pairs, so the blocks that re-read one head's K / V (Q / dO) tiles share an L2
// (with the plain grid order every L2 streamed every head: 5 TB/s of L2 misses at S = 3056).  HEAVY_LAST_FIRST: under a causal mask the
// work of a query block grows with its index -- the longest blocks of a head are started first, so the launch does not end on them.
template <bool HEAVY_LAST_FIRST>
DEVINL void block_coords(int& xb, int& h, int& b) {
    const int gx = gridDim.x, gy = gridDim.y;
    const int lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const int pid = xcd_remap(lin, gx * gy * (int)gridDim.z);
    const int hh = pid / gx;
    xb = pid - hh * gx;
    if (HEAVY_LAST_FIRST) xb = gx - 1 - xb;
    h = hh % gy;
    b = hh / gy;
}

// Compile-time loop (the index is a constant expression inside the body: immediates of the asm reads below).
template <int N, class F>
DEVINL void sfor(F&& f) {
    if constexpr (N > 0) {
        sfor<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}
// Loop-invariant read addresses of a natural tile.  Everything that depends on the lane sits in 4 registers for the row reads (one
// per 32-dim k-step: the swizzle term depends on the row's low three bits = the lane's) and 8 for the column reads (one per 16-dim
// block); the 16-row block, the 32-row step, the +16-row partner and the operand image inside a stage are IMMEDIATES of the read.
// Per tile the kernels pay one add per register (stage toggle) instead of an address computation per read (was 43-69 VALU / tile).
struct NatAddr {
    unsigned row[4], col[8];
    DEVINL void init(const char* image, int lane) {
        const int g = lane >> 4, c = lane & 15, r1 = (lane >> 2) & 15, pp = lane & 3;
        const unsigned b = lds_off(image);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) row[ks] = b + c * 256 + (((ks * 4 + g) ^ nswz(c)) << 4);
#pragma unroll
        for (int db = 0; db < 8; ++db) col[db] = b + r1 * 256 + (((2 * db + (pp >> 1)) ^ nswz(r1)) << 4) + ((pp & 1) << 3);
    }
    DEVINL void shift(int delta) {
#pragma unroll
        for (int i = 0; i < 4; ++i) row[i] += delta;
#pragma unroll
        for (int i = 0; i < 8; ++i) col[i] += delta;
    }
};
// row fragment: rows 16 kb + (lane & 15) of the image at byte offset IMG within the stage: IMM = IMG + kb * 4096
template <int IMM> DEVINL void rdrow_imm(bf16x8& dst, unsigned a) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(a), "i"(IMM) : "memory");
}
// column fragment of the 32-row step p: IMM = IMG + p * 8192 (the partner rows + 16 sit 4096 bytes further, same swizzle term)
template <int IMM> DEVINL void rdcol_imm(TFrag& f, unsigned a) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.t0) : "v"(a), "i"(IMM) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.t1) : "v"(a), "i"(IMM + 4096) : "memory");
}
// v_max3_f32 / v_max_f32 without the canonicalising self-maximum hipcc puts in front of every fmaxf operand that comes out of an
// MFMA (IEEE mode: 32 extra VALU per tile).  volatile: they stay behind the volatile LDS reads issued after the MFMA chain, which
// provide the wait states between an MFMA result and its first VALU read (the hazard recogniser does not look into inline asm).
DEVINL float max3_asm(float a, float b, float c) { float r; asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

}  // namespace
