// bf16 MFMA GEMM for gfx950:  C[M,N] = act(A[M,K] * B[N,K]^T + bias[N]) + R[M,N]
// ("NT": both operands contraction-contiguous = torch.nn.Linear's  y = x W^T  with W stored [out,in]).
//
// Replaces (reference call sites): every nn.Linear / F.linear of the hot path -- HF Llama q/k/v/o/gate/up/down
// (finetuning/llava/model/language_model/modeling_llama.py:332-338,377,226), lm_head (:1323), CLIP q/k/v/out/fc1/fc2
// (HF:models/clip/modeling_clip.py:298-350), patch-embed conv as GEMM (:209), mm_projector
// (multimodal_projector/builder.py:41-48) and their autograd dgrad / wgrad (the contraction-major operand forms of the 256x256 kernel
// below read activations and weights in place: no transposed copies).
//
// Structure (v1): 128x128x64 block tile, 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32 tiles,
// LDS-DMA staging (global_load_lds 16 B) into a 2-stage ring, XOR-swizzled 128-B rows so that the
// ds_read_b128 fragment reads are bank-conflict free, XCD-aware + grouped block->tile map.
// Edge handling: out-of-range rows / k-chunks are sourced from a 16-byte zero page (per-lane source select),
// so any M, N and any K % 8 == 0 work; stores are masked.
// Compile-time switches that remain (measurement only; every other A/B variant of rounds 1-3 -- phase placement of the staging pieces,
// priorities, non-persistent blocks, the uncompacted epilogue -- was measured, recorded under profiles/ and removed from the source):
//   RV_GROUP_M=<n>  tile rows per group of the block -> tile map (tools/tile_order_probe.py; default 4)
//   RV_STAMPS       diagnostic build with s_memtime stamps in the 256x256 kernel (tools/gemm_stamps.py)
#include <type_traits>
#include "common.h"
#include "radvlm_hip.h"
#include <stdlib.h>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;  // 32 KiB
constexpr int NSTAGE = 2;

struct GemmParams {
    const bf16* A; const bf16* B; void* C; const bf16* bias; const void* R; const bf16* zeros;
    long lda, ldb, ldc, ldr;
    int M, N, K, act, out_f32, res_f32, tiles_m, tiles_n;
    float alpha;
    // MODE 1 (fused second operand pair, e.g. the LoRA path  [x | t] [W | B]^T):  C += A2[M,K2] * B2[N,K2]^T
    const bf16* A2; const bf16* B2; long lda2, ldb2; int K2;
    // MODE 2 (split-K for outputs with few tiles): block = (tile, slice); raw fp32 partials go to ws[slice][M][N]
    float* ws; int splits;
    // MODE 3 (tail split): blocks [0, n_full) compute whole tiles; the remaining tiles of the last, partly filled round are
    // each split over `splits` K-slices (fp32 partial tiles in ws, combined by tail_reduce_kernel)
    int n_full;
    int pgrid;     // persistent form: blocks [0, pgrid) walk the whole tiles with stride pgrid; MODE 3: blocks >= pgrid are the K-slices of the tail tiles
    // fused epilogues of the 256x256 kernel (EPI template parameter):
    //   EPI_ROPE        C = rope(A B^T + bias) on columns < rope_cols (q heads then k heads), table cs[pos][hd/2][cos,sin]
    //   EPI_SWIGLU_FWD  B = [gate; up] rows ([2F, K]); C = gate|up [M, 2F], C2 = silu(gate) * up [M, F]
    //   EPI_SWIGLU_BWD  acc = d(act) [M, F]; G = gate|up [M, 2F]; C = d(gate|up) [M, 2F]
    const float* rope_cs; const int* rope_pos; int rope_S, rope_cols, rope_hd;
    int F; bf16* C2; long ldc2; const bf16* G; long ldg;
    unsigned bytesA, bytesB;     // BUF kernels: extent of each operand = its buffer resource's num_records
    unsigned bytesA2, bytesB2;   // BUF + MODE 1: the second operand pair's extents
    // dropout on the product (general epilogue of the 256x256 kernel): C = residual + keep(seed, m * N + n) ? alpha * acc / (1 - p) : 0 -- the
    // adapter branch of a LoRA layer's input gradient (rv_gemm_dropout_add_bf16); drop_thr = 0: off
    unsigned drop_thr; float drop_scale; unsigned long long drop_seed;
};
enum { EPI_NONE = 0, EPI_ROPE = 1, EPI_SWIGLU_FWD = 2, EPI_SWIGLU_BWD = 3 };

// Epilogue activations, kept to a few straight-line instructions each: the epilogue is unrolled 16 x 8 times and inlining libm's
// erff / tanhf there pushed it past LLVM's pragma-unroll threshold -- the loops over the accumulator tile were then left
// rolled, the 128 accumulator registers became a dynamically indexed scratch array (528 B/lane) and every GEMM ran ~18 % slower
// (build.sh now fails on any kernel that uses scratch).
//   erf GELU  : erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16 / fp32-accumulate noise of the output)
//   tanh GELU : 0.5 x (1 + tanh u) == x * sigmoid(2u)  (exact identity), u = sqrt(2/pi) (x + 0.044715 x^3)
DEVINL float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __frcp_rn(1.f + 0.3275911f * ax);
    const float p = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float r = 1.f - p * __expf(-ax * ax);
    return copysignf(r, x);
}
DEVINL float apply_act(float x, int act) {
    if (act == RV_ACT_QUICK_GELU) return x / (1.f + __expf(-1.702f * x));
    if (act == RV_ACT_GELU) return 0.5f * x * (1.f + erf_as(x * 0.70710678118654752f));
    if (act == RV_ACT_GELU_TANH) return x / (1.f + __expf(-1.5957691216057308f * (x + 0.044715f * x * x * x)));
    return x;
}

// Stage one [ROWS=128][BK=64] bf16 tile (rows of 128 B) of `src` (row stride ld) into LDS at `lds` (16 KiB).
// Physical 16-B chunk p of row r holds logical chunk p ^ ((r >> 1) & 7).
DEVINL void stage_tile(const bf16* __restrict__ src, long ld, int row0, int nrows_total, int k0, int K,
                       const bf16* zeros, char* lds, int wid, int lane) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = (it * 4 + wid) * 64 + lane;   // chunk id in tile, 0..1023
        const int r = c >> 3, p = c & 7;
        const int lc = p ^ ((r >> 1) & 7);
        const int gr = row0 + r, gk = k0 + lc * 8;
        const bf16* g = (gr < nrows_total && gk < K) ? (src + (long)gr * ld + gk) : zeros;
        glds16(g, lds + (it * 4 + wid) * 1024);
    }
}

DEVINL bf16x8 read_frag(const char* tile, int row, int kchunk) {
    const int p = kchunk ^ ((row >> 1) & 7);
    return *(const bf16x8*)(tile + row * 128 + p * 16);
}

__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wid = wave_id(), lane = lane_id();
    const int wr = wid >> 1, wc = wid & 1;

    // block -> tile: XCD-contiguous chunks, then grouped (GROUP_M tile-rows share B panels in L2)
    const int nwg = P.tiles_m * P.tiles_n;
    int pid = xcd_remap(blockIdx.x, nwg);
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * P.tiles_n;
    const int group = pid / per_group;
    const int first_m = group * GROUP_M;
    const int gsz = min(P.tiles_m - first_m, GROUP_M);
    const int tm = first_m + (pid % per_group) % gsz;
    const int tn = (pid % per_group) / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nt = (P.K + BK - 1) / BK;
    stage_tile(P.A, P.lda, m0, P.M, 0, P.K, P.zeros, smem, wid, lane);
    stage_tile(P.B, P.ldb, n0, P.N, 0, P.K, P.zeros, smem + BM * BK * 2, wid, lane);

    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        char* cur = smem + (t & 1) * STAGE_BYTES;
        if (t + 1 < nt) {
            char* nxt = smem + ((t + 1) & 1) * STAGE_BYTES;
            stage_tile(P.A, P.lda, m0, P.M, (t + 1) * BK, P.K, P.zeros, nxt, wid, lane);
            stage_tile(P.B, P.ldb, n0, P.N, (t + 1) * BK, P.K, P.zeros, nxt + BM * BK * 2, wid, lane);
        }
        const char* At = cur;
        const char* Bt = cur + BM * BK * 2;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 a[4], b[4];
            const int kc = kk * 4 + (lane >> 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = read_frag(At, wr * 64 + i * 16 + (lane & 15), kc);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = read_frag(Bt, wc * 64 + j * 16 + (lane & 15), kc);
            // D[n][m]: the B-matrix fragment is the MFMA "A" operand so each lane ends up with 4 consecutive n
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(b[j], a[i], acc[i][j]);
        }
    }

    // epilogue: lane holds C[m = m0+wr*64+i*16+(lane&15)][n = n0+wc*64+j*16+4*(lane>>4)+r], r = 0..3
    const bool n_vec_ok = (P.N % 4 == 0) && (P.ldc % 4 == 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wr * 64 + i * 16 + (lane & 15);
        if (m >= P.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wc * 64 + j * 16 + 4 * (lane >> 4);
            if (n >= P.N) continue;
            float v[4] = {acc[i][j][0] * P.alpha, acc[i][j][1] * P.alpha, acc[i][j][2] * P.alpha, acc[i][j][3] * P.alpha};
            const int nv = min(4, P.N - n);
            if (P.bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (r < nv) v[r] += bf2f(P.bias[n + r]);
            }
            if (P.act != RV_ACT_NONE) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], P.act);
            }
            if (P.R) {
                if (P.res_f32) {
                    const float* rp = (const float*)P.R + (long)m * P.ldr + n;
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (r < nv) v[r] += rp[r];
                } else {
                    const bf16* rp = (const bf16*)P.R + (long)m * P.ldr + n;
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (r < nv) v[r] += bf2f(rp[r]);
                }
            }
            if (P.out_f32) {
                float* cp = (float*)P.C + (long)m * P.ldc + n;
                if (nv == 4 && n_vec_ok) *(f32x4*)cp = f32x4{v[0], v[1], v[2], v[3]};
                else for (int r = 0; r < nv; ++r) cp[r] = v[r];
            } else {
                bf16* cp = (bf16*)P.C + (long)m * P.ldc + n;
                if (nv == 4 && n_vec_ok) *(bf16x4*)cp = bf16x4{f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
                else for (int r = 0; r < nv; ++r) cp[r] = f2bf(v[r]);
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// LoRA down-projection with the adapter's input dropout applied to the operand fragments:
//     T[M, R] = alpha * dropout_p(X)[M, K] * A[R, K]^T            (peft LoraLayer: lora_A(lora_dropout(x)), train/train.py:1515-1532)
// The unfused sequence wrote dropout(X) to HBM (read + write of X) and read it back in an R-wide GEMM -- three passes over X for a
// product whose arithmetic is negligible (R = 64); here X is staged once (LDS-DMA), each lane masks the 8 elements of its A-operand
// fragment with the SAME counter-based mask rv_dropout_bf16 uses (element index m * K + k: backward regenerates it with ops.dropout) and
// the 1 / (1 - p) scale rides alpha.  HBM-bound: one pass over X.  64-row x 64-column x 64-deep tiles, 4 waves x 16 rows, 3-stage ring.
constexpr int LD_BM = 64, LD_BN = 64, LD_STAGE = (LD_BM + LD_BN) * BK * 2, LD_NSTAGE = 3;
struct LoraDownParams {
    const bf16* X; const bf16* A; bf16* T; const bf16* zeros;
    long ldx, lda, ldt;
    int M, R, K;
    float alpha;
    unsigned thr; unsigned long long seed;      // 16-bit keep threshold of the shared mask (common.h, rv_keep8); thr = 0: no dropout
};
DEVINL void stage_rows64(const bf16* __restrict__ src, long ld, int row0, int nrows_total, int k0, const bf16* zeros, char* lds, int wid, int lane) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int c = (it * 4 + wid) * 64 + lane;   // 16-B chunk id in the 64 x 64 tile, 0..511
        const int r = c >> 3, p = c & 7;
        const int lc = p ^ ((r >> 1) & 7);
        const int gr = row0 + r;
        const bf16* g = gr < nrows_total ? (src + (long)gr * ld + k0 + lc * 8) : zeros;
        glds16(g, lds + (it * 4 + wid) * 1024);
    }
}
__global__ __launch_bounds__(256, 3) void lora_down_kernel(LoraDownParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wid = wave_id(), lane = lane_id();
    const int m0 = blockIdx.x * LD_BM;
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nt = P.K / BK;
    auto stage = [&](int t) {
        char* dst = smem + (t % LD_NSTAGE) * LD_STAGE;
        stage_rows64(P.X, P.ldx, m0, P.M, t * BK, P.zeros, dst, wid, lane);
        stage_rows64(P.A, P.lda, 0, P.R, t * BK, P.zeros, dst + LD_BM * BK * 2, wid, lane);
    };
    stage(0);
    if (nt > 1) stage(1);
    const int row = wid * 16 + (lane & 15);                  // this lane's X row within the tile
    const unsigned long long ebase = (unsigned long long)(m0 + row) * (unsigned long long)P.K + (unsigned long long)((lane >> 4) * 8);
    for (int t = 0; t < nt; ++t) {
        // tile t was issued two steps ago: all but the 4 pieces of tile t + 1 must have landed
        if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 2 < nt) stage(t + 2);                        // its slot was read in step t - 1 (every wave is past that barrier)
        const char* At = smem + (t % LD_NSTAGE) * LD_STAGE;
        const char* Bt = At + LD_BM * BK * 2;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int kc = kk * 4 + (lane >> 4);
            bf16x8 a = read_frag(At, row, kc);
            if (P.thr) {
                const unsigned keep = rv_keep8(P.seed, ebase + (unsigned long long)(t * BK + kk * 32), P.thr);
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = (keep >> j) & 1 ? a[j] : (bf16)0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = mfma16(read_frag(Bt, j * 16 + (lane & 15), kc), a, acc[j]);
        }
    }
    // lane holds T[m0 + 16 wid + (lane & 15)][16 j + 4 (lane >> 4) + r]
    const int m = m0 + row;
    if (m >= P.M) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = j * 16 + 4 * (lane >> 4);
        if (n >= P.R) continue;
        bf16* tp = P.T + (long)m * P.ldt + n;
        *(bf16x4*)tp = bf16x4{f2bf(acc[j][0] * P.alpha), f2bf(acc[j][1] * P.alpha), f2bf(acc[j][2] * P.alpha), f2bf(acc[j][3] * P.alpha)};
    }
}

// ---------------------------------------------------------------------------------------------------------------
// v2: 256x256x64 block tile, 8 waves (2 M x 4 N), 128x64 per wave, 1 block/CU (all 160 KiB of LDS, <=256 VGPR).
// Each K-tile is staged as four 16-KiB half-tiles (A rows 0-127 / 128-255, B rows 0-127 / 128-255), ONE half-tile per
// phase, into a 3-deep A ring + 2-deep B ring (10 slots); a K-tile is computed in four phases of 16 MFMAs (one 64x32
// quadrant each):
//   ph1: stage A0(t+2) | read A(mh0), B(nh0) | MFMA (mh0,nh0)
//   ph2: stage A1(t+2) | read B(nh1)         | MFMA (mh0,nh1)      -- barrier: all B reads of tile t retired
//   ph3: stage B0(t+2) | read A(mh1)         | MFMA (mh1,nh1)
//   ph4: stage B1(t+2) |                     | MFMA (mh1,nh0)      -- vmcnt(8): tile t+1 landed, ALL of tile t+2 in flight
// Both B sub-tiles stay in registers, so the B slot of tile t is free after ph2 (reused by B(t+2) in ph3/ph4); the A slot
// of tile t-1 is free after the end barrier (reused by A(t+2)).  Every load is issued >= one K-tile (~2k cycles) before
// its first use, stays in flight across both barriers (counted vmcnt + raw s_barrier, never vmcnt(0) in the loop).
#ifndef RV_GROUP_M
#define RV_GROUP_M 4      // tile rows per group of the block -> tile map (L2 locality; profiles/r02_gemm_tile_order_l2.json; 8: -1.2 %, 16: -2.7 %)
#endif
constexpr int BM2 = 256, BN2 = 256;
constexpr int HALF_BYTES = 128 * BK * 2;      // 16 KiB
constexpr int LDS_BYTES2 = 10 * HALF_BYTES;   // A ring: 3 tiles x 2 halves, B ring: 2 tiles x 2 halves
constexpr int B_RING_OFF = 6 * HALF_BYTES;

// Normal operand (T = false): half-tile = [128 feature rows][64 contraction] (128-B rows), as in v1.
// Transposed operand (T = true): the matrix is stored [contraction][features]; half-tile = [64 contraction rows]
// [128 features] (256-B rows), read back with ds_read_b64_tr_b16 (hardware transpose: each 16-lane group gets a
// 4 x 16 block column-major).  32-B granule g of row m is stored at granule g ^ f(m), f(m) = (m&3) | ((m>>3)&1)<<2,
// which makes the 8 rows a 32-lane half touches land on 8 different granules (conflict-free).
DEVINL int tswz(int m) { return ((m & 3) | (((m >> 3) & 1) << 2)) << 1; }

// PERM (B operand of the 256x256 kernel): LDS feature row 32a + 16j + c holds feature 32a + 8(c>>2) + 4j + (c&3), so the
// two 16-row MFMA tiles j = 0,1 of a 32-row group interleave in units of 4 features and a lane's accumulators for the
// pair are 8 CONSECUTIVE output columns (one 16-byte store per lane, 64-byte runs per row).  For transposed operands the
// same interleave is a different column start of the tr read, no staging change.
DEVINL int perm32f(int r) { return (r & ~31) + (((r & 15) >> 2) << 3) + (((r >> 4) & 1) << 2) + (r & 3); }

// BMAP (B operand of the fused-epilogue kernels): which weight row a tile column holds, chosen so that the two values an epilogue
// combines sit in the SAME lane (accumulator groups a = 0 and a = 1 of a lane hold tile columns 64 wc + 8 q + j and + 32):
//   BMAP 1 (RoPE, head_dim 128): a tile = two heads; within a head, column 64 wcl + 32 a + 8 q + j holds head dimension
//           64 a + 32 wcl + 8 q + j, i.e. bits 5 and 6 swapped -> a lane's two groups are the rotation partners e and e + 64;
//   BMAP 2 (SwiGLU): a tile = 128 features; column 64 wc + 32 a + 8 q + j holds row a * F + 128 tn + 32 wc + 8 q + j of the
//           stacked [gate; up] weight -> group 0 is gate(f), group 1 is up(f) for the same 8 features.
template <bool T, bool PERM = false, int BMAP = 0>
DEVINL void stage_half(const bf16* __restrict__ src, long ld, int feat0, int nfeat, int k0, int K, const bf16* zeros,
                       char* lds, int wid, int lane, int h = 0, int F = 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int it = i * 8 + wid;           // wave-instruction index 0..15
        const int c = it * 64 + lane;         // 16-B chunk id 0..1023
        const bf16* g;
        if (!T) {
            const int r = c >> 3, p = c & 7;
            const int lc = p ^ ((r >> 1) & 7);
            int gr = feat0 + (PERM ? perm32f(r) : r);
            const int gk = k0 + lc * 8;
            bool ok = gr < nfeat;
            if (BMAP == 1) {          // feat0 = first row of this half (one head of 128)
                const int f = perm32f(r);
                gr = feat0 + ((f & 31) | ((f & 32) << 1) | ((f & 64) >> 1));
                ok = gr < nfeat;
            } else if (BMAP == 2) {   // feat0 = 128 * tn (first feature of the tile); h = which half of the tile's 256 columns
                const int cc = h * 128 + perm32f(r);
                const int f = feat0 + 32 * (cc >> 6) + (cc & 31);
                gr = ((cc >> 5) & 1) * F + f;
                ok = f < F;
            }
            g = (ok && gk < K) ? (src + (long)gr * ld + gk) : zeros;
        } else {
            const int r = c >> 4, p = c & 15;
            const int lc = p ^ tswz(r);
            const int gk = k0 + r, gf = feat0 + lc * 8;
            g = (gk < K && gf < nfeat) ? (src + (long)gk * ld + gf) : zeros;
        }
        glds16(g, lds + it * 1024);
    }
}

// Buffer-addressed staging (BUF kernels): the LDS-DMA is a `buffer_load_dwordx4 ... offen lds` whose address is
//   resource base + per-lane byte offset (computed ONCE per block, loop invariant) + scalar K offset (one s_add per K-tile),
// so a staging piece costs an M0 write and the load itself -- the flat form spent a 64-bit vector add, a range compare and two selects
// per piece (the zero-page source select) in front of every one of the 8 pieces a wave issues per K-tile.  Rows past the end of the
// operand (tile edges, K tails of contraction-major operands) are out of the resource's range and read as zero in hardware.
// Requirements checked on the host: operand extent < 2 GiB; K % 64 == 0 unless BOTH operands are contraction-major (a K tail inside
// the rows of a row-major operand is not a resource boundary), otherwise the flat path runs.
struct HalfPlan { int v[2]; };
template <bool T, bool PERM, int BMAP>
DEVINL HalfPlan plan_half(long ld, int feat0, int nfeat, int wid, int lane, int h, int F, unsigned limit) {
    HalfPlan pl;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int it = i * 8 + wid;
        const int c = it * 64 + lane;
        long off;
        bool ok = true;
        if (!T) {
            const int r = c >> 3, p = c & 7;
            const int lc = p ^ ((r >> 1) & 7);
            long gr = feat0 + (PERM ? perm32f(r) : r);
            if (BMAP == 1) { const int f = perm32f(r); gr = feat0 + ((f & 31) | ((f & 32) << 1) | ((f & 64) >> 1)); }
            else if (BMAP == 2) {
                const int cc = h * 128 + perm32f(r);
                const int f = feat0 + 32 * (cc >> 6) + (cc & 31);
                gr = (long)((cc >> 5) & 1) * F + f;
                ok = f < F;
            }
            off = (gr * ld + lc * 8) * 2;
        } else {
            const int r = c >> 4, p = c & 15;
            const int lc = p ^ tswz(r);
            off = ((long)r * ld + feat0 + lc * 8) * 2;
            ok = feat0 + lc * 8 < nfeat;
        }
        pl.v[i] = (ok && off < (long)limit) ? (int)off : (int)limit;      // `limit` = the resource size: always out of range -> zero
    }
    return pl;
}
DEVINL void stage_half_buf(__amdgpu_buffer_rsrc_t rsrc, const HalfPlan& pl, int soff, char* lds, int wid) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + (i * 8 + wid) * 1024), 16, pl.v[i], soff, 0, 0);
}

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

// A fragment of 16 features (fbase .. fbase+15, fbase % 16 == 0) x 32 contraction (k-step kk).
// Transposed operands are read with ds_read_b64_tr_b16 issued from inline asm: hipcc's waitcnt pass cannot see that
// the builtin form does not alias the in-flight LDS-DMA stores and would drain vmcnt(0) before every read group
// (measured: -25 % on the dgrad/wgrad forms).  The asm reads are retired by frag_wait4 (s_waitcnt lgkmcnt(0) that
// names every destination, so no consumer is scheduled above it) -- guide section 5.7 form (ii).
struct Frag { bf16x8 n; s16x4 t0, t1; };

// Transposed (contraction-major) operand fragment: two ds_read_b64_tr_b16 whose addresses are ONE per-lane base + immediates.  The base
// covers everything that depends on the lane, the tile's stage and the 32-feature group (the XOR swizzle mixes the group's chunk bits with
// the lane's row bits: not an additive constant); the k-step (+32 rows = 8192 bytes: tswz sees the same (m & 3) and ((m >> 3) & 1)), the
// second read's rows + 4 (1024 bytes, same swizzle term) and -- PERM -- the 16-feature tile inside its 32-feature group (8 bytes) are
// immediates of the read.  (Round 3 passed every address in a register: 41 v_add_u32 + 8 v_add3_u32 per K-tile and wave next to 48 reads
// and 64 MFMAs -- the weight-gradient forms were instruction-ISSUE bound, 2 x 1160 issue cycles per SIMD against 2048 MFMA cycles.)
// rowbase must be a multiple of 32 (0 for A, the wave's 64-column half for B).
template <bool PERM, int FOFF>
DEVINL unsigned tfrag_base(const char* tile, int rowbase, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int fb = rowbase + (PERM ? (FOFF & ~31) : FOFF);
    // the lane supplies the address of 4 consecutive features: fbase + 4p, or (PERM) 32a + 8p + 4j for tile j of group a
    const int c16 = PERM ? (fb >> 3) + p : (fb >> 3) + (p >> 1);
    const int half = PERM ? 0 : (p & 1);
    const int m = 8 * g + q;
    const char* a = tile + m * 256 + ((c16 ^ tswz(m)) << 4) + (half << 3);
    return (unsigned)(unsigned long)(__attribute__((address_space(3))) const char*)a;
}
template <bool PERM, int FOFF, int KK>
DEVINL void frag_issue_t(Frag& f, unsigned base) {
    constexpr int IMM = KK * 8192 + (PERM ? ((FOFF >> 4) & 1) * 8 : 0);
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.t0) : "v"(base), "i"(IMM) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.t1) : "v"(base), "i"(IMM + 1024) : "memory");
}
// LEFT = number of younger LDS operations that may stay in flight (LDS ops retire in order; compiler-issued reads in
// between only make the wait stricter, never weaker).
template <bool T, int LEFT = 0>
DEVINL void frag_wait4(Frag& a, Frag& b, Frag& c, Frag& d) {
    if (!T) return;
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(a.t0), "+v"(a.t1), "+v"(b.t0), "+v"(b.t1), "+v"(c.t0), "+v"(c.t1), "+v"(d.t0), "+v"(d.t1) : "i"(LEFT) : "memory");
}
template <bool T>
DEVINL bf16x8 frag_get(const Frag& f) {
    if (!T) return f.n;
    const s16x8 r = {f.t0[0], f.t0[1], f.t0[2], f.t0[3], f.t1[0], f.t1[1], f.t1[2], f.t1[3]};
    return __builtin_bit_cast(bf16x8, r);
}

// Compile-time loop: the body is instantiated once per index, so accumulator registers are always addressed statically (a
// `#pragma unroll` loop is only a request: when the optimiser declined it, acc[i][..] became a scratch array).
template <int N, class F>
DEVINL void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// Row-major (non-transposed) fragments issued from inline asm with an immediate offset, for the phase whose reads must be
// retired in two steps: hipcc's own waitcnt insertion put one `s_waitcnt lgkmcnt(0)` in front of phase 1's first MFMA, i.e.
// waited for all 16 reads of the phase (incl. the B(nh1) prefetch) -- ~500 LDS cycles with every wave reading at once.
DEVINL unsigned frag_base(const char* tile, int row0, int kk, int lane) {
    const int r = lane & 15;                       // row0 % 16 == 0: the swizzle term depends on the lane only
    const int p = (kk * 4 + (lane >> 4)) ^ ((r >> 1) & 7);
    const char* a = tile + (row0 + r) * 128 + p * 16;
    return (unsigned)(unsigned long)(__attribute__((address_space(3))) const char*)a;
}
template <int IMM>
DEVINL void frag_issue_imm(Frag& f, unsigned base) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.n) : "v"(base), "i"(IMM) : "memory");
}
// Counted waits that name the destinations of the asm-issued reads they retire (so no consumer is scheduled above them);
// LEFT = younger LDS operations allowed to stay in flight (LDS operations return in order).
template <bool T, int LEFT>
DEVINL void fwait1(Frag& a) {
    if constexpr (T) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a.t0), "+v"(a.t1) : "i"(LEFT) : "memory");
    else asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a.n) : "i"(LEFT) : "memory");
}
template <bool T, int LEFT>
DEVINL void fwait2(Frag& a, Frag& b) {
    if constexpr (T) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a.t0), "+v"(a.t1), "+v"(b.t0), "+v"(b.t1) : "i"(LEFT) : "memory");
    else asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a.n), "+v"(b.n) : "i"(LEFT) : "memory");
}
template <bool T, int LEFT>
DEVINL void fwait4(Frag& a, Frag& b, Frag& c, Frag& d) {
    if constexpr (T) asm volatile("s_waitcnt lgkmcnt(%8)"
                                  : "+v"(a.t0), "+v"(a.t1), "+v"(b.t0), "+v"(b.t1), "+v"(c.t0), "+v"(c.t1), "+v"(d.t0), "+v"(d.t1) : "i"(LEFT) : "memory");
    else asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a.n), "+v"(b.n), "+v"(c.n), "+v"(d.n) : "i"(LEFT) : "memory");
}
// One fragment read, either operand form: FOFF = feature-row offset (compile time) from `rowbase`; row-major operands use the
// per-k-step base address + an immediate, transposed ones the tr-read pair of frag_issue.
template <bool T, bool PERM, int FOFF, int KK>
DEVINL void fissue(Frag& f, const char* tile, unsigned base_rowmajor, int rowbase, int lane) {
    if constexpr (T) frag_issue_t<PERM, FOFF, KK>(f, tfrag_base<PERM, FOFF>(tile, rowbase, lane));
    else frag_issue_imm<FOFF * 128>(f, base_rowmajor);
}

#define BAR_LGKM() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// No wave-priority instructions: per-phase s_setprio 1 / 0 flips around the MFMA clusters COST 3.3 % with the buffer-addressed, branch-free K
// loop (same-box A/B over the decoder's shapes, round 2), a static s_setprio 1 for waves 4-7 cost 2.9 %; both switches are gone from the source.
#ifdef RV_STAMPS
// diagnostic build: cycles wave 0 spends parked at the mid-tile barrier [0], the end-of-tile vmcnt wait [1] and barrier [2]
#define RV_ACC_BEGIN() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); dbg[3] = __builtin_readcyclecounter(); } while (0)
#define RV_ACC_END(k) do { asm volatile("" ::: "memory"); dbg[k] += __builtin_readcyclecounter() - dbg[3]; } while (0)
#else
#define RV_ACC_BEGIN() do { } while (0)
#define RV_ACC_END(k) do { } while (0)
#endif

// One K-tile step of the 256x256 kernel (4 phases x 16 MFMAs per wave); ph1..ph4 issue this step's LDS-DMA staging.
template <bool TA, bool TB, class F1, class F2, class F3, class F4>
DEVINL void ktile_256(f32x4 (&acc)[8][4], const char* At, const char* Bt, int brow0, int lane, F1 ph1, F2 ph2, F3 ph3, F4 ph4,
                      long long (&dbg)[4]) {
    Frag fa[4][2], fb[2][2][2];
    bf16x8 a[4][2], b[2][2][2];
    constexpr int OA = TA ? 2 : 1, OB = TB ? 2 : 1;      // LDS operations per fragment (tr reads come in pairs)
    constexpr int G = 2 * OB + 4 * OA, B1_OPS = 4 * OB;   // one k-step group of phase 1 (B(nh0) x2 + A(mh0) x4); the B(nh1) prefetch
    constexpr int W0 = G + B1_OPS > 15 ? 15 : G + B1_OPS;   // lgkmcnt is a 4-bit count: a larger allowance is clamped (waits for a little more)
    const unsigned ab[2] = {TA ? 0u : frag_base(At, 0, 0, lane), TA ? 0u : frag_base(At, 0, 1, lane)};
    const unsigned bb[2] = {TB ? 0u : frag_base(Bt, brow0, 0, lane), TB ? 0u : frag_base(Bt, brow0, 1, lane)};

    // ---- phase 1: read A(mh0), B(nh0) and, ahead of time, B(nh1).  Every read is issued from asm and retired by counted
    // waits: the k-step-0 MFMAs start as soon as THEIR six fragments are back, the k-step-1 MFMAs after the next six, and
    // the B(nh1) prefetch stays in flight behind both (hipcc's own bookkeeping waited for all 16 reads before the first MFMA).
    // issue order inside a k-step group: B0, A0, B1, A1, A2, A3 -- the first MFMA needs only the first two fragments
    static_for<2>([&](auto kt) {
        constexpr int kk = decltype(kt)::value;
        fissue<TB, true, 0, kk>(fb[0][0][kk], Bt, bb[kk], brow0, lane);
        fissue<TA, false, 0, kk>(fa[0][kk], At, ab[kk], 0, lane);
        fissue<TB, true, 16, kk>(fb[0][1][kk], Bt, bb[kk], brow0, lane);
        fissue<TA, false, 16, kk>(fa[1][kk], At, ab[kk], 0, lane);
        fissue<TA, false, 32, kk>(fa[2][kk], At, ab[kk], 0, lane);
        fissue<TA, false, 48, kk>(fa[3][kk], At, ab[kk], 0, lane);
    });
    static_for<2>([&](auto kt) {
        constexpr int kk = decltype(kt)::value;
        static_for<2>([&](auto j) { fissue<TB, true, 32 + decltype(j)::value * 16, kk>(fb[1][decltype(j)::value][kk], Bt, bb[kk], brow0, lane); });
    });
    ph1();   // this phase's two staging pieces go out behind its 16 fragment reads (not in front: +0.8 % over the decoder shapes)
    // k-step 0 streams: each wait retires one more fragment (younger reads stay in flight) and releases the MFMAs it completes
    constexpr int R = G + B1_OPS;
#define RV_C15(x) ((x) > 15 ? 15 : (x))
    fwait1<TB, RV_C15(OB + 3 * OA + R)>(fb[0][0][0]);
    fwait1<TA, RV_C15(OB + 3 * OA + R)>(fa[0][0]);
    __builtin_amdgcn_sched_barrier(0);
    b[0][0][0] = frag_get<TB>(fb[0][0][0]); a[0][0] = frag_get<TA>(fa[0][0]);
    acc[0][0] = mfma16(b[0][0][0], a[0][0], acc[0][0]);
    __builtin_amdgcn_sched_barrier(0);
    fwait1<TB, RV_C15(3 * OA + R)>(fb[0][1][0]);
    __builtin_amdgcn_sched_barrier(0);
    b[0][1][0] = frag_get<TB>(fb[0][1][0]);
    acc[0][1] = mfma16(b[0][1][0], a[0][0], acc[0][1]);
    __builtin_amdgcn_sched_barrier(0);
    fwait1<TA, RV_C15(2 * OA + R)>(fa[1][0]);
    __builtin_amdgcn_sched_barrier(0);
    a[1][0] = frag_get<TA>(fa[1][0]);
    acc[1][0] = mfma16(b[0][0][0], a[1][0], acc[1][0]); acc[1][1] = mfma16(b[0][1][0], a[1][0], acc[1][1]);
    __builtin_amdgcn_sched_barrier(0);
    fwait1<TA, RV_C15(OA + R)>(fa[2][0]);
    __builtin_amdgcn_sched_barrier(0);
    a[2][0] = frag_get<TA>(fa[2][0]);
    acc[2][0] = mfma16(b[0][0][0], a[2][0], acc[2][0]); acc[2][1] = mfma16(b[0][1][0], a[2][0], acc[2][1]);
    __builtin_amdgcn_sched_barrier(0);
    fwait1<TA, RV_C15(R)>(fa[3][0]);
    __builtin_amdgcn_sched_barrier(0);
    a[3][0] = frag_get<TA>(fa[3][0]);
    acc[3][0] = mfma16(b[0][0][0], a[3][0], acc[3][0]); acc[3][1] = mfma16(b[0][1][0], a[3][0], acc[3][1]);
#undef RV_C15
    __builtin_amdgcn_sched_barrier(0);   // keep the k-step-0 MFMAs above the second wait
    fwait2<TB, B1_OPS>(fb[0][0][1], fb[0][1][1]);
    fwait4<TA, B1_OPS>(fa[0][1], fa[1][1], fa[2][1], fa[3][1]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 2; ++j) b[0][j][1] = frag_get<TB>(fb[0][j][1]);
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i][1] = frag_get<TA>(fa[i][1]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(b[0][j][1], a[i][1], acc[i][j]);

    // ---- phase 2: B(nh1) has landed behind phase 1's MFMAs; A(mh1) is read k-step by k-step behind this phase's MFMAs
    ph2();
    RV_ACC_BEGIN();
    BAR_LGKM();   // every wave's B reads of tile t are complete -> the B slot of tile t may be restaged
    RV_ACC_END(0);
    fwait4<TB, 0>(fb[1][0][0], fb[1][1][0], fb[1][0][1], fb[1][1][1]);   // already retired by the barrier's wait: pins the consumers below it
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < 2; ++j) b[1][j][kk] = frag_get<TB>(fb[1][j][kk]);
    static_for<2>([&](auto kt) {
        constexpr int kk = decltype(kt)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][2 + j] = mfma16(b[1][j][kk], a[i][kk], acc[i][2 + j]);
        // a[.][kk] is dead now: fetch A(mh1) for this k-step while the other k-step's MFMAs run
        static_for<4>([&](auto i) { fissue<TA, false, 64 + decltype(i)::value * 16, kk>(fa[decltype(i)::value][kk], At, ab[kk], 0, lane); });
    });

    // ---- phase 3
    ph3();
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        if (kk == 0) fwait4<TA, 4 * OA>(fa[0][0], fa[1][0], fa[2][0], fa[3][0]);   // the k-step-1 reads stay in flight
        else {
            fwait4<TA, 0>(fa[0][1], fa[1][1], fa[2][1], fa[3][1]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i][kk] = frag_get<TA>(fa[i][kk]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[4 + i][2 + j] = mfma16(b[1][j][kk], a[i][kk], acc[4 + i][2 + j]);
    }

    // ---- phase 4
    ph4();
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[4 + i][j] = mfma16(b[0][j][kk], a[i][kk], acc[4 + i][j]);
    }
}

template <int MODE>
DEVINL void epilogue_256(const f32x4 (&acc)[8][4], const GemmParams& P, int m0, int n0, int wr, int wc, int lane, int kslice) {
    // epilogue: with the interleaved B rows a lane holds, for m = m0 + wr*128 + i*16 + (lane&15), the 8 consecutive columns
    // n = n0 + wc*64 + 32a + 8*(lane>>4) + {0..7}: acc[i][2a][0..3] then acc[i][2a+1][0..3]
    const bool n_vec_ok = (P.N % 8 == 0) && (P.ldc % 8 == 0) && (!P.R || P.ldr % 8 == 0) &&
                          ((((uintptr_t)P.C) | ((uintptr_t)P.R) | ((uintptr_t)P.bias)) & 15) == 0;
    // the common case (plain bf16 product: every dgrad / wgrad-free forward GEMM of the decoder) gets a compact instruction stream:
    // the general path below is ~20k instructions of mostly untaken branches per kernel, fetched once per tile
    if (MODE == 0 && !P.bias && P.act == RV_ACT_NONE && !P.R && !P.out_f32 && P.alpha == 1.f && n_vec_ok && !P.drop_thr) {
        static_for<16>([&](auto ia) {
            constexpr int i = decltype(ia)::value >> 1, a = decltype(ia)::value & 1;
            const int m = m0 + wr * 128 + i * 16 + (lane & 15);
            const int n = n0 + wc * 64 + 32 * a + 8 * (lane >> 4);
            if (m >= P.M || n >= P.N) return;
            const f32x4 lo = acc[i][2 * a], hi = acc[i][2 * a + 1];
            *(bf16x8*)((bf16*)P.C + (long)m * P.ldc + n) =
                bf16x8{f2bf(lo[0]), f2bf(lo[1]), f2bf(lo[2]), f2bf(lo[3]), f2bf(hi[0]), f2bf(hi[1]), f2bf(hi[2]), f2bf(hi[3])};
        });
        return;
    }
    static_for<16>([&](auto ia) {
        constexpr int i = decltype(ia)::value >> 1, a = decltype(ia)::value & 1;
        const int m = m0 + wr * 128 + i * 16 + (lane & 15);
        const int n = n0 + wc * 64 + 32 * a + 8 * (lane >> 4);
        if (m >= P.M || n >= P.N) return;
        const f32x4 lo = acc[i][2 * a], hi = acc[i][2 * a + 1];
        if (MODE == 2) {   // raw partial sums; alpha / bias / act / residual are applied by the reduce kernel
            float* wp = P.ws + ((long)kslice * P.M + m) * P.N + n;
#pragma unroll
            for (int r = 0; r < 4; ++r) { if (n + r < P.N) wp[r] = lo[r]; if (n + 4 + r < P.N) wp[4 + r] = hi[r]; }
            return;
        }
        float v[8] = {lo[0] * P.alpha, lo[1] * P.alpha, lo[2] * P.alpha, lo[3] * P.alpha,
                      hi[0] * P.alpha, hi[1] * P.alpha, hi[2] * P.alpha, hi[3] * P.alpha};
        const int nv = min(8, P.N - n);
        const bool full = (nv == 8) && n_vec_ok;
        if (P.bias) {
            if (full) {
                const bf16x8 bb = *(const bf16x8*)(P.bias + n);
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] += bf2f(bb[r]);
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r) if (r < nv) v[r] += bf2f(P.bias[n + r]);
            }
        }
        if (P.act != RV_ACT_NONE) {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = apply_act(v[r], P.act);
        }
        if (P.drop_thr) {        // N % 8 == 0 (checked on the host): the 8 columns are one mask word
            const unsigned keep = rv_keep8(P.drop_seed, (unsigned long long)m * (unsigned long long)P.N + (unsigned long long)n, P.drop_thr);
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = (keep >> r) & 1 ? v[r] * P.drop_scale : 0.f;
        }
        if (P.R) {
            if (P.res_f32) {
                const float* rp = (const float*)P.R + (long)m * P.ldr + n;
                if (full) {
                    const f32x4 r0 = *(const f32x4*)rp, r1 = *(const f32x4*)(rp + 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v[r] += r0[r]; v[4 + r] += r1[r]; }
                } else {
#pragma unroll
                    for (int r = 0; r < 8; ++r) if (r < nv) v[r] += rp[r];
                }
            } else {
                const bf16* rp = (const bf16*)P.R + (long)m * P.ldr + n;
                if (full) {
                    const bf16x8 rr = *(const bf16x8*)rp;
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] += bf2f(rr[r]);
                } else {
#pragma unroll
                    for (int r = 0; r < 8; ++r) if (r < nv) v[r] += bf2f(rp[r]);
                }
            }
        }
        if (P.out_f32) {
            float* cp = (float*)P.C + (long)m * P.ldc + n;
            if (full) { *(f32x4*)cp = f32x4{v[0], v[1], v[2], v[3]}; *(f32x4*)(cp + 4) = f32x4{v[4], v[5], v[6], v[7]}; }
            else {
#pragma unroll
                for (int r = 0; r < 8; ++r) if (r < nv) cp[r] = v[r];
            }
        } else {
            bf16* cp = (bf16*)P.C + (long)m * P.ldc + n;
            if (full) *(bf16x8*)cp = bf16x8{f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3]), f2bf(v[4]), f2bf(v[5]), f2bf(v[6]), f2bf(v[7])};
            else {
#pragma unroll
                for (int r = 0; r < 8; ++r) if (r < nv) cp[r] = f2bf(v[r]);
            }
        }
    });
}

DEVINL float rbf16(float x) { return bf2f(f2bf(x)); }
DEVINL void store8(bf16* p, const float (&v)[8]) {
    *(bf16x8*)p = bf16x8{f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3]), f2bf(v[4]), f2bf(v[5]), f2bf(v[6]), f2bf(v[7])};
}

// Fused epilogues: a lane holds, for row m = m0 + wr*128 + i*16 + (lane&15), tile columns 64 wc + 8 q + {0..7} (group a = 0:
// acc[i][0], acc[i][1]) and the same + 32 (group a = 1: acc[i][2], acc[i][3]), q = lane >> 4.  The B-row maps above make the pair
// of groups exactly what the epilogue combines, so everything is lane-local; all stores are 16-byte vectors.
template <int EPI>
DEVINL void epilogue_fused(const f32x4 (&acc)[8][4], const GemmParams& P, int m0, int tn, int wr, int wc, int lane) {
    const int q = lane >> 4;
    static_for<8>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        const int m = m0 + wr * 128 + i * 16 + (lane & 15);
        if (m >= P.M) return;
        float lo[8] = {acc[i][0][0], acc[i][0][1], acc[i][0][2], acc[i][0][3], acc[i][1][0], acc[i][1][1], acc[i][1][2], acc[i][1][3]};
        float hi[8] = {acc[i][2][0], acc[i][2][1], acc[i][2][2], acc[i][2][3], acc[i][3][0], acc[i][3][1], acc[i][3][2], acc[i][3][3]};
        if (EPI == EPI_ROPE) {
            // q/k projection + rotary embedding (modeling_llama.py:332-338 then :167-198): column e of a head pairs with e + hd/2
            const int half = P.rope_hd >> 1;
            int nlo, e0;
            if (P.rope_hd == 128) { e0 = 32 * (wc & 1) + 8 * q; nlo = tn * BN2 + (wc >> 1) * 128 + e0; }
            else { e0 = 8 * q; nlo = tn * BN2 + wc * 64 + e0; }              // head_dim 64: partners 32 apart, natural order
            const int nhi = nlo + half;
            if (nlo >= P.N) return;
            if (P.bias) {
                const bf16x8 b0 = *(const bf16x8*)(P.bias + nlo), b1 = *(const bf16x8*)(P.bias + nhi);
#pragma unroll
                for (int r = 0; r < 8; ++r) { lo[r] += bf2f(b0[r]); hi[r] += bf2f(b1[r]); }
            }
            if (nlo < P.rope_cols) {
                const int pos = P.rope_pos ? P.rope_pos[m] : (m % P.rope_S);
                const float* cs = P.rope_cs + ((long)pos * half + e0) * 2;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const f32x4 t = *(const f32x4*)(cs + 4 * r4);            // cos, sin of two consecutive dimensions
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int r = 2 * r4 + u;
                        const float co = t[2 * u], si = t[2 * u + 1];
                        // the unfused path stored the projection in bf16 before rotating it: keep that rounding point
                        const float a = rbf16(lo[r]), b = rbf16(hi[r]);
                        lo[r] = a * co - b * si;
                        hi[r] = b * co + a * si;
                    }
                }
            }
            bf16* cp = (bf16*)P.C + (long)m * P.ldc;
            store8(cp + nlo, lo);
            store8(cp + nhi, hi);
        } else if (EPI == EPI_SWIGLU_FWD) {
            // LlamaMLP (modeling_llama.py:226): lo = gate(f .. f+7), hi = up(f .. f+7)
            const int f = tn * 128 + 32 * wc + 8 * q;
            if (f >= P.F) return;
            float a[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float g = rbf16(lo[r]), u = rbf16(hi[r]);               // gate|up are stored in bf16; the activation reads them back
                a[r] = rbf16(g / (1.f + __expf(-g))) * u;
            }
            bf16* gp = (bf16*)P.C + (long)m * P.ldc;
            store8(gp + f, lo);
            store8(gp + P.F + f, hi);
            store8(P.C2 + (long)m * P.ldc2 + f, a);
        }
    });
    if (EPI == EPI_SWIGLU_BWD) {
        // acc = d(act) of columns n .. n+7 (natural column order); d gate = da * up * silu'(gate), d up = da * silu(gate)
        auto dswiglu = [&](const f32x4& x0, const f32x4& x1, const bf16x8& gv, const bf16x8& uv, bf16* cp_g, bf16* cp_u) {
            const float da[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
            float dg[8], du[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float g = bf2f(gv[r]), u = bf2f(uv[r]), d = rbf16(da[r]);   // d(act) was a bf16 tensor in the unfused path
                const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-g));   // v_rcp_f32 (1 ulp), as swiglu_bwd_kernel: an IEEE division is 10 instructions, 128 per lane here
                du[r] = d * (g * sg);
                dg[r] = d * u * (sg * (1.f + g * (1.f - sg)));
            }
            store8(cp_g, dg);
            store8(cp_u, du);
        };
        if (m0 + BM2 <= P.M && tn * BN2 + BN2 <= P.F) {
            // interior tile: the 32 gate / up vectors of a lane are fetched in four batches of eight, the next batch in flight behind the
            // arithmetic of the current one (one load -> use round trip per 16-row block made this epilogue latency-bound: ~25 us per tile)
            const bf16* gp = P.G + (long)(m0 + wr * 128 + (lane & 15)) * P.ldg + tn * BN2 + wc * 64 + 8 * q;
            bf16* cp = (bf16*)P.C + (long)(m0 + wr * 128 + (lane & 15)) * P.ldc + tn * BN2 + wc * 64 + 8 * q;
            bf16x8 gq[2][4], uq[2][4];
            auto fetch = [&](int bi, bf16x8 (&gd)[4], bf16x8 (&ud)[4]) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int ia = bi * 4 + u, i = ia >> 1, a = ia & 1;
                    const bf16* p = gp + (long)i * 16 * P.ldg + 32 * a;
                    gd[u] = *(const bf16x8*)p;
                    ud[u] = *(const bf16x8*)(p + P.F);
                }
            };
            fetch(0, gq[0], uq[0]);
            static_for<4>([&](auto bt) {
                constexpr int bi = decltype(bt)::value;
                if constexpr (bi + 1 < 4) fetch(bi + 1, gq[(bi + 1) & 1], uq[(bi + 1) & 1]);
                static_for<4>([&](auto ut) {
                    constexpr int u = decltype(ut)::value, ia = bi * 4 + u, i = ia >> 1, a = ia & 1;
                    bf16* c = cp + (long)i * 16 * P.ldc + 32 * a;
                    dswiglu(acc[i][2 * a], acc[i][2 * a + 1], gq[bi & 1][u], uq[bi & 1][u], c, c + P.F);
                });
            });
            return;
        }
        static_for<16>([&](auto ia) {
            constexpr int i = decltype(ia)::value >> 1, a = decltype(ia)::value & 1;
            const int m = m0 + wr * 128 + i * 16 + (lane & 15);
            const int n = tn * BN2 + wc * 64 + 32 * a + 8 * q;
            if (m >= P.M || n >= P.F) return;
            const bf16x8 gv = *(const bf16x8*)(P.G + (long)m * P.ldg + n), uv = *(const bf16x8*)(P.G + (long)m * P.ldg + P.F + n);
            bf16* cp = (bf16*)P.C + (long)m * P.ldc;
            dswiglu(acc[i][2 * a], acc[i][2 * a + 1], gv, uv, cp + n, cp + P.F + n);
        });
    }
}

#ifdef RV_STAMPS
// Diagnostic build only (tools/gemm_stamps.py): wall-clock stamps (s_memrealtime, 100 MHz) of wave 0 of every block at kernel
// entry, first MFMA-ready barrier, end of the K loop and end of the epilogue (after its stores are acknowledged) -> a buffer
// no product code reads.  The buffer pointer rides the kernarg segment (P.ws, unused by MODE 0): no vector load, no extra wait.
static long long* g_stamp_host = nullptr;
extern "C" int rv_debug_set_stamp_buffer(void* p) { g_stamp_host = (long long*)p; return 0; }
#define RV_STAMP(i) do { if (MODE == 0 && P.ws && threadIdx.x == 0) ((long long*)P.ws)[(long)blockIdx.x * 4 + (i)] = wall_clock64(); } while (0)
#else
#define RV_STAMP(i) do { } while (0)
#endif

template <bool TA, bool TB, int MODE, int EPI = EPI_NONE, bool BUF = false>
__global__ __launch_bounds__(512, 1) void gemm_kernel_256(GemmParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wid = wave_id(), lane = lane_id();
    const int wr = wid >> 2, wc = wid & 3;
    RV_STAMP(0);

    const int nwg = P.tiles_m * P.tiles_n;
    // PERSIST (plain whole-tile launches with buffer-addressed staging): a block walks the tiles blockIdx.x, + gridDim.x, ... (grid =
    // min(tiles, CUs); same tile order per XCD as the one-tile-per-block launch) and issues the first K-tile of its NEXT output tile
    // before the epilogue of the current one, so the fill of the staging pipeline and the launch gap hide behind the epilogue's stores.
    constexpr bool PERSIST = (MODE == 0 || MODE == 1 || MODE == 3) && BUF;
    int vtile = blockIdx.x, kslice = 0;
    bool sliced = MODE == 2;
    if (MODE == 2) { vtile = (int)blockIdx.x / P.splits; kslice = (int)blockIdx.x % P.splits; }
    if (MODE == 3 && (int)blockIdx.x >= (PERSIST ? P.pgrid : P.n_full)) {
        const int r = (int)blockIdx.x - (PERSIST ? P.pgrid : P.n_full);
        vtile = P.n_full + r / P.splits; kslice = r % P.splits; sliced = true;
    }
    int m0, n0, tn;
    auto coords = [&](int vt) {
        const int pid = xcd_remap(vt, nwg);
        constexpr int GROUP_M = RV_GROUP_M;
        const int per_group = GROUP_M * P.tiles_n;
        const int group = pid / per_group;
        const int first_m = group * GROUP_M;
        const int gsz = min(P.tiles_m - first_m, GROUP_M);
        tn = (pid % per_group) / gsz;
        m0 = (first_m + (pid % per_group) % gsz) * BM2;
        n0 = tn * BN2;
    };
    coords(vtile);

    f32x4 acc[8][4];

    // K-tile steps of this block: the whole K (MODE 0), K then K2 (MODE 1), or this block's slice of K (MODE 2)
    const int nt1 = (P.K + BK - 1) / BK;
    const int t0 = sliced ? (int)((long)kslice * nt1 / P.splits) : 0;
    const int nt = sliced ? (int)((long)(kslice + 1) * nt1 / P.splits) - t0 : (MODE == 1 ? nt1 + (P.K2 + BK - 1) / BK : nt1);
    // BUF: resources + per-lane offsets of this block's staging pieces (loop invariant)
    __amdgpu_buffer_rsrc_t rsA, rsB;
    HalfPlan plA[2], plB[2];
    if constexpr (BUF) {
        rsA = __builtin_amdgcn_make_buffer_rsrc((void*)P.A, 0, P.bytesA, 0x00020000);
        rsB = __builtin_amdgcn_make_buffer_rsrc((void*)P.B, 0, P.bytesB, 0x00020000);
    }
    auto make_plans = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            plA[h] = plan_half<TA, false, 0>(P.lda, m0 + h * 128, P.M, wid, lane, h, 0, P.bytesA);
            if (EPI == EPI_ROPE) plB[h] = P.rope_hd == 128 ? plan_half<TB, true, 1>(P.ldb, n0 + h * 128, P.N, wid, lane, h, 0, P.bytesB)
                                                            : plan_half<TB, true, 0>(P.ldb, n0 + h * 128, P.N, wid, lane, h, 0, P.bytesB);
            else if (EPI == EPI_SWIGLU_FWD) plB[h] = plan_half<TB, true, 2>(P.ldb, tn * 128, P.N, wid, lane, h, P.F, P.bytesB);
            else plB[h] = plan_half<TB, true, 0>(P.ldb, n0 + h * 128, P.N, wid, lane, h, 0, P.bytesB);
        }
    };
    // MODE 1 (second operand pair, K-tiles nt1 .. nt-1): its own resources and per-lane offsets (other leading dimensions); which pair a
    // staging piece belongs to is a compile-time tag in the steady-state loops (PAIR 0 / 1) and a run-time test only in the prologue
    __amdgpu_buffer_rsrc_t rsA2, rsB2;
    HalfPlan plA2[2], plB2[2];
    if constexpr (BUF && MODE == 1) {
        rsA2 = __builtin_amdgcn_make_buffer_rsrc((void*)P.A2, 0, P.bytesA2, 0x00020000);
        rsB2 = __builtin_amdgcn_make_buffer_rsrc((void*)P.B2, 0, P.bytesB2, 0x00020000);
    }
    auto make_plans2 = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            plA2[h] = plan_half<TA, false, 0>(P.lda2, m0 + h * 128, P.M, wid, lane, h, 0, P.bytesA2);
            plB2[h] = plan_half<TB, true, 0>(P.ldb2, n0 + h * 128, P.N, wid, lane, h, 0, P.bytesB2);
        }
    };
    if constexpr (BUF) make_plans();
    if constexpr (BUF && MODE == 1) make_plans2();
    const int kstepA = TA ? (int)(P.lda * BK * 2) : BK * 2, kstepB = TB ? (int)(P.ldb * BK * 2) : BK * 2;    // bytes per K-tile
    const int kstepA2 = TA ? (int)(P.lda2 * BK * 2) : BK * 2, kstepB2 = TB ? (int)(P.ldb2 * BK * 2) : BK * 2;
    // PAIR: 0 = first operand pair, 1 = second, 2 = decide at run time (t >= nt1); only MODE 1 has a second pair
    auto stageA = [&](int t, int slot, int h, auto pair_tag) {
        constexpr int PAIR = decltype(pair_tag)::value;
        char* dst = smem + (slot * 2 + h) * HALF_BYTES;
        if constexpr (BUF) {
            if (MODE == 1 && (PAIR == 1 || (PAIR == 2 && t >= nt1))) stage_half_buf(rsA2, plA2[h], (t - nt1) * kstepA2, dst, wid);
            else stage_half_buf(rsA, plA[h], (t0 + t) * kstepA, dst, wid);
            return;
        }
        if (MODE == 1 && t >= nt1) stage_half<TA>(P.A2, P.lda2, m0 + h * 128, P.M, (t - nt1) * BK, P.K2, P.zeros, dst, wid, lane);
        else stage_half<TA>(P.A, P.lda, m0 + h * 128, P.M, (t0 + t) * BK, P.K, P.zeros, dst, wid, lane);
    };
    auto stageB = [&](int t, int h, auto pair_tag) {
        constexpr int PAIR = decltype(pair_tag)::value;
        char* dst = smem + B_RING_OFF + ((t & 1) * 2 + h) * HALF_BYTES;
        if constexpr (BUF) {
            if (MODE == 1 && (PAIR == 1 || (PAIR == 2 && t >= nt1))) stage_half_buf(rsB2, plB2[h], (t - nt1) * kstepB2, dst, wid);
            else stage_half_buf(rsB, plB[h], (t0 + t) * kstepB, dst, wid);
            return;
        }
        if (MODE == 1 && t >= nt1) stage_half<TB, true>(P.B2, P.ldb2, n0 + h * 128, P.N, (t - nt1) * BK, P.K2, P.zeros, dst, wid, lane);
        else if (EPI == EPI_ROPE) {
            if (P.rope_hd == 128) stage_half<TB, true, 1>(P.B, P.ldb, n0 + h * 128, P.N, (t0 + t) * BK, P.K, P.zeros, dst, wid, lane);
            else stage_half<TB, true>(P.B, P.ldb, n0 + h * 128, P.N, (t0 + t) * BK, P.K, P.zeros, dst, wid, lane);
        } else if (EPI == EPI_SWIGLU_FWD) stage_half<TB, true, 2>(P.B, P.ldb, tn * 128, P.N, (t0 + t) * BK, P.K, P.zeros, dst, wid, lane, h, P.F);
        else stage_half<TB, true>(P.B, P.ldb, n0 + h * 128, P.N, (t0 + t) * BK, P.K, P.zeros, dst, wid, lane);
    };

    // prologue: tiles 0 and 1 (tile 1 stays in flight)
    constexpr std::integral_constant<int, 0> P0{};
    constexpr std::integral_constant<int, 1> P1{};
    constexpr std::integral_constant<int, 2> PR{};
    stageA(0, 0, 0, P0); stageA(0, 0, 1, P0); stageB(0, 0, P0); stageB(0, 1, P0);
    int vt0 = blockIdx.x;
    do {    // one trip unless PERSIST
    if (nt > 1) { stageA(1, 1, 0, PR); stageA(1, 1, 1, PR); stageB(1, 0, PR); stageB(1, 1, PR); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    RV_STAMP(1);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    int aslot = 0;          // A ring slot of tile t (t % 3)
    long long dbg[4] = {0, 0, 0, 0};   // diagnostic build only (RV_STAMPS): parked-cycle accumulators; dead code otherwise
    // One K-tile step.  STAGE = this step issues the staging of tile t + 2 (all steps but the last two): the steady-state loop is
    // straight-line code without the four `t + 2 < nt` branches that used to cut it into basic blocks (hipcc schedules within one).
    auto step = [&](int t, auto stage_tag, auto pair_tag) {
        constexpr int STAGE = decltype(stage_tag)::value;      // 1 = stage tile t + 2, 0 = do not, 2 = decide at run time (t + 2 < nt)
        const char* At = smem + (aslot * 2 + wr) * HALF_BYTES;
        const char* Bt = smem + B_RING_OFF + ((t & 1) * 2 + (wc >> 1)) * HALF_BYTES;
        const int aslot2 = aslot == 0 ? 2 : aslot - 1;   // (t + 2) % 3: the slot tile t-1 just vacated
        const int brow0 = (wc & 1) * 64;
        ktile_256<TA, TB>(acc, At, Bt, brow0, lane,
                          [&]() { if (STAGE == 1 || (STAGE == 2 && t + 2 < nt)) stageA(t + 2, aslot2, 0, pair_tag); },
                          [&]() { if (STAGE == 1 || (STAGE == 2 && t + 2 < nt)) stageA(t + 2, aslot2, 1, pair_tag); },
                          [&]() { if (STAGE == 1 || (STAGE == 2 && t + 2 < nt)) stageB(t + 2, 0, pair_tag); },
                          [&]() { if (STAGE == 1 || (STAGE == 2 && t + 2 < nt)) stageB(t + 2, 1, pair_tag); }, dbg);
        // tile t+1 (issued during tile t-1) must have landed; the 8 loads of tile t+2 stay in flight
        RV_ACC_BEGIN();
        if (STAGE == 1 || (STAGE == 2 && t + 2 < nt)) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        RV_ACC_END(1);
        aslot = aslot == 2 ? 0 : aslot + 1;
        RV_ACC_BEGIN();
        BAR_LGKM();   // also: every wave's A reads of tile t are complete -> the A halves of this slot may be restaged
        RV_ACC_END(2);
    };
    if constexpr ((MODE == 2 || MODE == 1) && !BUF) {
        // the flat-addressed split-K and two-pair forms keep the single loop with run-time staging decisions (shapes the buffer-addressed
        // kernel does not take: not the step's bulk)
        for (int t = 0; t < nt; ++t) step(t, std::integral_constant<int, 2>{}, PR);
    } else if constexpr (MODE == 1) {
        // fused second operand pair (LoRA: [x | t] [W | B]^T), buffer-addressed: the steady-state loop stages from the first pair, a
        // short second loop from the adapter pair (one K-tile at r = 64), the last two steps stage nothing -- all branch-free
        int t = 0;
        for (; t + 2 < nt1; ++t) step(t, std::integral_constant<int, 1>{}, P0);
        for (; t + 2 < nt; ++t) step(t, std::integral_constant<int, 1>{}, P1);
        for (; t < nt; ++t) step(t, std::integral_constant<int, 0>{}, P0);
    } else {
        int t = 0;
        for (; t + 2 < nt; ++t) step(t, std::integral_constant<int, 1>{}, P0);
        for (; t < nt; ++t) step(t, std::integral_constant<int, 0>{}, P0);
    }
#ifdef RV_STAMPS
    if (MODE == 0 && P.ws && lane == 0) {   // per wave: [mid barrier, end vmcnt, end barrier] parked cycles of the whole K loop
        long long* o = (long long*)P.ws + (long)gridDim.x * 4 + ((long)blockIdx.x * 8 + wid) * 3;
        o[0] = dbg[0]; o[1] = dbg[1]; o[2] = dbg[2];
    }
#endif

    RV_STAMP(2);
    if (MODE == 3 && sliced) {
        // raw fp32 partial tile, tile-local [256][256] layout: slab = (tail tile index) * splits + kslice
        float* slab = P.ws + ((long)(vtile - P.n_full) * P.splits + kslice) * (BM2 * BN2);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                float* wp = slab + (wr * 128 + i * 16 + (lane & 15)) * BN2 + wc * 64 + 32 * a + 8 * (lane >> 4);
                *(f32x4*)wp = acc[i][2 * a];
                *(f32x4*)(wp + 4) = acc[i][2 * a + 1];
            }
        return;
    }
    // (every wave's LDS reads of this tile are behind the last step's barrier: the rings are free)
    const int m0e = m0, n0e = n0, tne = tn;
    bool more = false;
    if constexpr (PERSIST) {
        vt0 += MODE == 3 ? P.pgrid : (int)gridDim.x;
        more = !sliced && vt0 < (MODE == 3 ? P.n_full : nwg);
        if (more) {
            coords(vt0); make_plans();
            if constexpr (MODE == 1) make_plans2();
            stageA(0, 0, 0, P0); stageA(0, 0, 1, P0); stageB(0, 0, P0); stageB(0, 1, P0);
        }
    }
    if (EPI != EPI_NONE) epilogue_fused<EPI>(acc, P, m0e, tne, wr, wc, lane);
    else epilogue_256<MODE == 3 ? 0 : MODE>(acc, P, m0e, n0e, wr, wc, lane, kslice);
#ifdef RV_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    RV_STAMP(3);
    if (!more) break;
    } while (PERSIST);
}

// C = act(alpha * sum_s ws[s] + bias) + residual  (finishes a split-K GEMM)
__global__ void splitk_reduce_kernel(GemmParams P) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)P.M * P.N;
    if (i >= total) return;
    const int m = (int)(i / P.N), n = (int)(i % P.N);
    float s = 0.f;
    for (int k = 0; k < P.splits; ++k) s += P.ws[(long)k * total + i];
    s *= P.alpha;
    if (P.bias) s += bf2f(P.bias[n]);
    s = apply_act(s, P.act);
    if (P.R) s += P.res_f32 ? ((const float*)P.R)[(long)m * P.ldr + n] : bf2f(((const bf16*)P.R)[(long)m * P.ldr + n]);
    if (P.out_f32) ((float*)P.C)[(long)m * P.ldc + n] = s;
    else ((bf16*)P.C)[(long)m * P.ldc + n] = f2bf(s);
}

// ---------------------------------------------------------------------------------------------------------------
// LoRA A-gradient with the adapter's input dropout re-created in registers:
//     gA[R, K] (+)= 1 / (1 - p) * dT[M, R]^T * mask_p(X)[M, K]       (adjoint of lora_A(lora_dropout(x)), train/train.py:1515-1532)
// The unfused sequence wrote dropout(X) to HBM (read + write of X) and read it back in a split-K GEMM whose 256-row tile holds 64 useful rows:
// three passes over X per adapted module, 24 ms of the 13B LoRA step in the dropout kernel alone (profiles/r03c_lora_kernel_ms_per_step.json).
// Here X is read ONCE: a block owns 128 columns of X and a slice of the token range; per 64-token step every thread loads four 16-byte chunks
// of X and two of dT, masks X with the SAME counter-based mask the forward used (rv_keep8: element index m * K + k; the 1 / (1 - p) scale rides
// the reduce kernel's alpha, as it rides alpha in rv_lora_down_bf16), writes both to LDS contraction-major and the four waves read them back
// transposed (ds_read_b64_tr_b16) as MFMA operands.  The mask hashes (2 per chunk) and HBM bound it, not the MFMAs (16 per wave and step).
// Partial sums over the token slices go to the workspace in the layout splitk_reduce_kernel finishes (deterministic: no atomics, the replicas
// of a data-parallel job stay bit-identical).
constexpr int AG_BM = 64, AG_BN = 128;
constexpr int AG_XB = AG_BM * AG_BN * 2, AG_TB = AG_BM * 64 * 2, AG_STAGE = AG_XB + AG_TB;      // 16 KiB + 8 KiB per stage, two stages
struct LoraAGradParams {
    const bf16* X; const bf16* T; float* ws;
    long ldx, ldt;
    int M, R, K, splits;
    unsigned thr; unsigned long long seed;
};
// 128-byte rows ([64 tokens][64 adapter rows]): bank slot of a 32-byte granule = 4 (m & 1) + granule; the 8 rows a 32-lane half of a transposed
// read touches (m = 8 g + q, g < 2, q < 4) get 8 different slots when the granule is XORed with ((m >> 1) & 1) | ((m >> 3) & 1) << 1
DEVINL int sw128(int m) { return (((m >> 1) & 1) | (((m >> 3) & 1) << 1)) << 1; }
DEVINL bf16x8 tr_read16(const char* a, const char* b) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)b);
    const s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, r);
}
__global__ __launch_bounds__(256, 3) void lora_agrad_kernel(LoraAGradParams P) {
    __shared__ __attribute__((aligned(16))) char smem[2 * AG_STAGE];
    const int wid = wave_id(), lane = lane_id(), tid = (int)threadIdx.x;
    const int ncb = (P.K + AG_BN - 1) / AG_BN;
    const int cb = (int)blockIdx.x % ncb, split = (int)blockIdx.x / ncb;      // the column blocks of one token slice run side by side: they share dT in L2
    const int c0 = cb * AG_BN;
    const int nst = (P.M + AG_BM - 1) / AG_BM;
    const int s0 = (int)((long)split * nst / P.splits), s1 = (int)((long)(split + 1) * nst / P.splits);
    const int xr = tid >> 4, xc = tid & 15;     // X chunks of this thread: token rows xr + 16 i (i < 4), 16-byte chunk xc of the block's 128 columns
    const int tr = tid >> 3, tc = tid & 7;      // dT chunks: token rows tr + 32 i (i < 2), chunk tc of the (up to) 64 adapter rows
    const bool xok = c0 + xc * 8 < P.K, tok = tc * 8 < P.R;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    bf16x8 xv0, xv1, xv2, xv3, tv0, tv1;
    auto ldx8 = [&](int m) __attribute__((always_inline)) { return (xok && m < P.M) ? *(const bf16x8*)(P.X + (long)m * P.ldx + c0 + xc * 8) : zero8; };
    auto ldt8 = [&](int m) __attribute__((always_inline)) { return (tok && m < P.M) ? *(const bf16x8*)(P.T + (long)m * P.ldt + tc * 8) : zero8; };
    auto load = [&](int s) __attribute__((always_inline)) {
        const int m0 = s * AG_BM;
        xv0 = ldx8(m0 + xr); xv1 = ldx8(m0 + xr + 16); xv2 = ldx8(m0 + xr + 32); xv3 = ldx8(m0 + xr + 48);
        tv0 = ldt8(m0 + tr); tv1 = ldt8(m0 + tr + 32);
    };
    auto put_x = [&](bf16x8 v, int m0, int row, char* st) __attribute__((always_inline)) {
        if (P.thr) {
            const unsigned keep = rv_keep8(P.seed, (unsigned long long)(m0 + row) * (unsigned long long)P.K + (unsigned long long)(c0 + xc * 8), P.thr);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (keep >> j) & 1 ? v[j] : (bf16)0.f;
        }
        *(bf16x8*)(st + row * 256 + ((xc ^ tswz(row)) << 4)) = v;
    };
    auto store = [&](int s, char* st) __attribute__((always_inline)) {
        const int m0 = s * AG_BM;
        put_x(xv0, m0, xr, st); put_x(xv1, m0, xr + 16, st); put_x(xv2, m0, xr + 32, st); put_x(xv3, m0, xr + 48, st);
        *(bf16x8*)(st + AG_XB + tr * 128 + ((tc ^ sw128(tr)) << 4)) = tv0;
        *(bf16x8*)(st + AG_XB + (tr + 32) * 128 + ((tc ^ sw128(tr + 32)) << 4)) = tv1;
    };
    // transposed reads: lane (g, q, p) supplies the address of 4 consecutive features at token row 8 g + q (+ 4 for the second read, + 32 per k-step)
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int mrow = 8 * g + q;
    int xoff[2], toff[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) xoff[j] = mrow * 256 + (((((wid * 32 + 16 * j) >> 3) + (pp >> 1)) ^ tswz(mrow)) << 4) + (pp & 1) * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) toff[i] = AG_XB + mrow * 128 + ((((16 * i) >> 3) + (pp >> 1)) ^ sw128(mrow)) * 16 + (pp & 1) * 8;
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (s0 < s1) { load(s0); store(s0, smem); }
    __syncthreads();
    for (int s = s0; s < s1; ++s) {
        const char* cur = smem + ((s - s0) & 1) * AG_STAGE;
        char* nxt = smem + (((s - s0) & 1) ^ 1) * AG_STAGE;
        if (s + 1 < s1) load(s + 1);              // in flight behind this step's MFMAs
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 b[2], a[4];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = tr_read16(cur + xoff[j] + kk * 8192, cur + xoff[j] + kk * 8192 + 1024);
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = tr_read16(cur + toff[i] + kk * 4096, cur + toff[i] + kk * 4096 + 512);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
        }
        if (s + 1 < s1) store(s + 1, nxt);        // the other stage: its readers passed the barrier that ended step s - 1
        __syncthreads();
    }
    // lane holds D[adapter row 16 i + 4 g + r][column c0 + 32 wid + 16 j + (lane & 15)]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ar = 16 * i + 4 * g + r, col = c0 + wid * 32 + 16 * j + (lane & 15);
                if (ar < P.R && col < P.K) P.ws[((long)split * P.R + ar) * P.K + col] = acc[i][j][r];
            }
}

// Finishes MODE 3: for every tail tile, C = act(alpha * sum_slices partial + bias) + residual (8 columns per thread).
__global__ __launch_bounds__(256) void tail_reduce_kernel(GemmParams P) {
    const int nwg = P.tiles_m * P.tiles_n;
    const int tt = blockIdx.x >> 5;                       // tail tile index; 32 blocks x 256 threads x 8 columns per tile
    const int e = ((blockIdx.x & 31) * 256 + threadIdx.x) * 8;
    const int row = e >> 8, col = e & 255;
    const int pid = xcd_remap(P.n_full + tt, nwg);
    constexpr int GROUP_M = RV_GROUP_M;
    const int per_group = GROUP_M * P.tiles_n;
    const int group = pid / per_group;
    const int first_m = group * GROUP_M;
    const int gsz = min(P.tiles_m - first_m, GROUP_M);
    const int m = (first_m + (pid % per_group) % gsz) * BM2 + row;
    const int n = ((pid % per_group) / gsz) * BN2 + col;
    if (m >= P.M || n >= P.N) return;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < P.splits; ++k) {
        const float* sp = P.ws + ((long)tt * P.splits + k) * (BM2 * BN2) + row * BN2 + col;
        const f32x4 a = *(const f32x4*)sp, b = *(const f32x4*)(sp + 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] += a[r]; v[4 + r] += b[r]; }
    }
    const int nv = min(8, P.N - n);
    // whole 16-byte vectors when the row segment is complete and aligned (every decoder shape): one load per operand and one store per
    // thread instead of eight 2-byte accesses (a scalar short store costs ~12x a dwordx4 store per byte)
    const bool vec = nv == 8 && (P.N % 8 == 0) && (P.ldc % 8 == 0) && !P.out_f32 && (!P.R || (!P.res_f32 && P.ldr % 8 == 0)) &&
                     ((((uintptr_t)P.C) | ((uintptr_t)P.R) | ((uintptr_t)P.bias)) & 15) == 0;
    if (vec) {
        float x[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = v[r] * P.alpha;
        if (P.bias) {
            const bf16x8 bb = *(const bf16x8*)(P.bias + n);
#pragma unroll
            for (int r = 0; r < 8; ++r) x[r] += bf2f(bb[r]);
        }
        if (P.act != RV_ACT_NONE) {
#pragma unroll
            for (int r = 0; r < 8; ++r) x[r] = apply_act(x[r], P.act);
        }
        if (P.R) {
            const bf16x8 rr = *(const bf16x8*)((const bf16*)P.R + (long)m * P.ldr + n);
#pragma unroll
            for (int r = 0; r < 8; ++r) x[r] += bf2f(rr[r]);
        }
        *(bf16x8*)((bf16*)P.C + (long)m * P.ldc + n) = bf16x8{f2bf(x[0]), f2bf(x[1]), f2bf(x[2]), f2bf(x[3]), f2bf(x[4]), f2bf(x[5]), f2bf(x[6]), f2bf(x[7])};
        return;
    }
    for (int r = 0; r < nv; ++r) {
        float x = v[r] * P.alpha;
        if (P.bias) x += bf2f(P.bias[n + r]);
        x = apply_act(x, P.act);
        if (P.R) x += P.res_f32 ? ((const float*)P.R)[(long)m * P.ldr + n + r] : bf2f(((const bf16*)P.R)[(long)m * P.ldr + n + r]);
        if (P.out_f32) ((float*)P.C)[(long)m * P.ldc + n + r] = x;
        else ((bf16*)P.C)[(long)m * P.ldc + n + r] = f2bf(x);
    }
}

}  // namespace

// ---- process-wide launch configuration (the ONLY state this file keeps; one process drives one GPU) ----------
//   g_tail_split / g_force_kernel : measurement hooks (rv_gemm_select_kernel; RV_GEMM_KERNEL at first use)
//   g_persist                     : persistent blocks on (single GPU) / off (collectives share the CUs), rv_gemm_select_kernel(41 / 40)
//   g_cus                         : compute units the tile-round heuristics plan for = the device's multiProcessorCount minus the
//                                   units reserved for concurrently running collectives (rv_gemm_set_cu_budget / RV_GEMM_RESERVED_CUS);
//                                   queried once per process on first use
//   per-instantiation `attr` flags: hipFuncSetAttribute(MaxDynamicSharedMemorySize) is issued once per kernel instantiation
// None of it depends on the arguments of a call; results never depend on it (only which launch shape computes them).
static int g_tail_split = 1;   // rv_gemm_select_kernel(20) disables the tail split (A/B), (21) enables
static int g_force_kernel = 0;  // 0 auto, 1 = 128x128 kernel, 2 = 256x256 kernel (RV_GEMM_KERNEL or rv_gemm_select_kernel)
static int g_cus = 0;           // 0 = not yet queried
static int g_cus_device = -1;   // the device g_cus was derived for: a process that switches devices re-queries (one process normally drives one GPU)
static int g_reserved_cus = -1; // -1 = take RV_GEMM_RESERVED_CUS (default 0) at first use
static int g_no_buf = 0;         // rv_gemm_select_kernel(30 / 31): buffer-addressed staging off / on (A/B measurement)
static int g_persist = 1;        // rv_gemm_select_kernel(40 / 41): persistent tile-walking blocks off / on.  OFF when collectives share the GPU
                                 // (the engine does that for world size > 1): a persistent block that cannot start because an RCCL kernel holds
                                 // its CU delays its whole share of the tiles (up to 2x for the launch); one-tile blocks only lose part of a round
extern "C" int rv_gemm_select_kernel(int which) {
    if (which >= 40) { g_persist = which == 41; return RV_OK; }
    if (which >= 30) { g_no_buf = which == 30; return RV_OK; }
    if (which >= 20) { g_tail_split = which - 20; return RV_OK; }
    g_force_kernel = which;
    return RV_OK;
}
// Number of compute units the GEMM's round / tail-split / split-K heuristics plan for.  total_cus <= 0: ask the device
// (hipDeviceProp_t::multiProcessorCount of the current device); reserved_cus: units left to other streams (an RCCL all-reduce
// overlapped with backward occupies a few dozen), subtracted from the total.  Returns the resulting budget.
extern "C" int rv_gemm_set_cu_budget(int total_cus, int reserved_cus) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return RV_ERR_LAUNCH;
    if (total_cus <= 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return RV_ERR_LAUNCH;
        total_cus = n;
    }
    g_cus_device = dev;
    if (reserved_cus < 0) reserved_cus = 0;
    g_reserved_cus = reserved_cus;
    g_cus = total_cus - reserved_cus;
    if (g_cus < 8) g_cus = 8;
    return g_cus;
}
static int cu_budget() {
    int dev = 0;
    if (g_cus != 0 && hipGetDevice(&dev) == hipSuccess && dev != g_cus_device) g_cus = 0;     // another device became current: plan for ITS units
    if (g_cus == 0) {
        const char* e = getenv("RV_GEMM_RESERVED_CUS");
        rv_gemm_set_cu_budget(0, g_reserved_cus >= 0 ? g_reserved_cus : (e ? atoi(e) : 0));
    }
    return g_cus;
}

// Operand extents for the buffer-addressed kernels; false when a shape does not qualify (K tail inside a row, >= 2 GiB operand).
static bool buf_extents(GemmParams& P, int trans_a, int trans_b) {
    // a K tail is a resource boundary only for contraction-major operands (whole rows past the end read as zero); inside the rows of a
    // row-major operand it is not -- e.g. the weight gradients of a 14998-token batch (both operands contraction-major) qualify
    if ((P.K % BK) && !(trans_a && trans_b)) return false;
    const long ea = (trans_a ? (long)P.K : (long)P.M) * P.lda * 2, eb = (trans_b ? (long)P.K : (long)P.N) * P.ldb * 2;
    if (ea >= (1L << 31) || eb >= (1L << 31) || g_no_buf) return false;
    P.bytesA = (unsigned)ea; P.bytesB = (unsigned)eb;
    return true;
}
static bool buf_extents2(GemmParams& P, int trans_a, int trans_b) {
    if ((P.K2 % BK) && !(trans_a && trans_b)) return false;
    const long ea = (trans_a ? (long)P.K2 : (long)P.M) * P.lda2 * 2, eb = (trans_b ? (long)P.K2 : (long)P.N) * P.ldb2 * 2;
    if (ea >= (1L << 31) || eb >= (1L << 31)) return false;
    P.bytesA2 = (unsigned)ea; P.bytesB2 = (unsigned)eb;
    return true;
}
template <bool TA, bool TB, int MODE, bool BUF = false>
static void launch256m(const GemmParams& P, hipStream_t st) {
    static bool set = false;
    if (!set) { (void)hipFuncSetAttribute((const void*)gemm_kernel_256<TA, TB, MODE, EPI_NONE, BUF>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES2); set = true; }
    const int nwg = P.tiles_m * P.tiles_n;
    const int blocks = MODE == 2 ? nwg * P.splits : (MODE == 3 ? P.n_full + (nwg - P.n_full) * P.splits : nwg);
    int grid = blocks;
    GemmParams Q = P;
    Q.pgrid = MODE == 3 ? P.n_full : blocks;
    // persistent form: one block per CU walks the whole tiles (MODE 3: + the K-slice blocks of the tail tiles behind them)
    if (g_persist && BUF && (MODE == 0 || MODE == 1) && blocks > cu_budget()) { grid = cu_budget(); Q.pgrid = grid; }
    if (g_persist && BUF && MODE == 3 && P.n_full > cu_budget()) { Q.pgrid = cu_budget(); grid = Q.pgrid + (nwg - P.n_full) * P.splits; }
    hipLaunchKernelGGL((gemm_kernel_256<TA, TB, MODE, EPI_NONE, BUF>), dim3(grid), dim3(512), LDS_BYTES2, st, Q);
    if (MODE == 3) hipLaunchKernelGGL(tail_reduce_kernel, dim3((nwg - P.n_full) * 32), dim3(256), 0, st, P);
    if (MODE == 2) {
        const long total = (long)P.M * P.N;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, P);
    }
}
template <bool TA, bool TB>
static void launch256(GemmParams& P, int mode, hipStream_t st) {
    if (mode == 1) { if (buf_extents(P, TA, TB) && buf_extents2(P, TA, TB)) launch256m<TA, TB, 1, true>(P, st); else launch256m<TA, TB, 1>(P, st); }
    else if (mode == 2) { if (buf_extents(P, TA, TB)) launch256m<TA, TB, 2, true>(P, st); else launch256m<TA, TB, 2>(P, st); }
    else if (buf_extents(P, TA, TB)) { if (mode == 3) launch256m<TA, TB, 3, true>(P, st); else launch256m<TA, TB, 0, true>(P, st); }
    else if (mode == 3) launch256m<TA, TB, 3>(P, st);
    else launch256m<TA, TB, 0>(P, st);
}

static int gemm_ex_impl(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const void* bias,
                        const void* residual, int64_t ldr, int M, int N, int K, int trans_a, int trans_b, float alpha,
                        int act, int out_f32, int res_f32, const void* A2, int64_t lda2, const void* B2, int64_t ldb2,
                        int K2, void* workspace, int64_t workspace_bytes, const void* zeros16, void* stream, float drop_p, uint64_t drop_seed) {
    if (!A || !B || !C || !zeros16 || M <= 0 || N <= 0 || K <= 0) return RV_ERR_ARG;
    if ((lda & 7) || (ldb & 7)) return RV_ERR_ARG;
    if ((!trans_a && (K & 7)) || (trans_a && (M & 7)) || (!trans_b && (K & 7)) || (trans_b && (N & 7))) return RV_ERR_ARG;
    if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)zeros16) & 15) return RV_ERR_ARG;
    const bool ext = A2 && B2 && K2 > 0;
    if (ext && ((lda2 & 7) || (ldb2 & 7) || (!trans_a && (K2 & 7)) || (!trans_b && (K2 & 7)) || (((uintptr_t)A2 | (uintptr_t)B2) & 15))) return RV_ERR_ARG;
    GemmParams P;
    P.A = (const bf16*)A; P.B = (const bf16*)B; P.C = C; P.bias = (const bf16*)bias; P.R = residual;
    P.zeros = (const bf16*)zeros16;
    P.lda = lda; P.ldb = ldb; P.ldc = ldc; P.ldr = ldr;
    P.M = M; P.N = N; P.K = K; P.act = act; P.out_f32 = out_f32; P.res_f32 = res_f32; P.alpha = alpha;
    P.A2 = (const bf16*)A2; P.B2 = (const bf16*)B2; P.lda2 = lda2; P.ldb2 = ldb2; P.K2 = ext ? K2 : 0;
    P.ws = (float*)workspace; P.splits = 1;
    P.drop_thr = drop_p > 0.f ? rv_dropout_thr16(drop_p) : 0u; P.drop_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f; P.drop_seed = drop_seed;
    const bool dropping = P.drop_thr != 0;       // lives in the general epilogue of the 256x256 whole-tile kernel only
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_nt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE * STAGE_BYTES);
        const char* e = getenv("RV_GEMM_KERNEL");
        if (e) g_force_kernel = atoi(e);
        attr_set = true;
    }
    const long tiles256 = (long)((M + BM2 - 1) / BM2) * ((N + BN2 - 1) / BN2);
    const int nt = (K + BK - 1) / BK;
    const int force = g_force_kernel;
    const int cus = cu_budget();            // 256 on an idle MI355X; fewer when collectives are planned to run beside the GEMMs
    // split-K: few output tiles but a long contraction (LoRA / bias-like gradients): spread K over the idle CUs
    int mode = ext ? 1 : 0;
    if (dropping) workspace = nullptr;           // no K-split shapes: their reduce kernels do not carry the mask
    if (!ext && workspace && tiles256 <= cus / 4 && nt >= 16) {
        int sp = (int)(cus / tiles256);
        if (sp > nt / 4) sp = nt / 4;
        if (sp > 32) sp = 32;
        if (sp >= 2 && (int64_t)sp * M * N * 4 <= workspace_bytes) { mode = 2; P.splits = sp; }
    }
    // tail split: when the last round of `cus` blocks is at most half full, its tiles are cut into 2-4 K-slices so that the
    // round costs 1/2 - 1/4 of a full one (e.g. 1408 tiles on 256 CUs: 6 rounds -> 5.5)
    P.n_full = 0;
    if (mode == 0 && workspace && g_tail_split && tiles256 > cus) {
        const int rem = (int)(tiles256 % cus);
        if (rem > 0 && rem <= cus / 2 && nt >= 32) {
            int sp = cus / rem;
            if (sp > 4) sp = 4;
            if ((int64_t)rem * sp * BM2 * BN2 * 4 <= workspace_bytes) { mode = 3; P.splits = sp; P.n_full = (int)(tiles256 - rem); }
        }
    }
    // the 128x128 kernel only exists for the plain NT form
    // tile-shape choice for the plain NT form: whole rounds of `cus` blocks (256^2 tiles, 1 block/CU) against double-rounds of
    // 2 x cus blocks (128^2 tiles, 2 blocks/CU, ~15 % less efficient per flop but finer grained); measured crossover on
    // MI355X (tools/ab_kernel12.py): 292 / 352 tiles -> 128^2 wins by 8-27 %, >= 876 tiles -> 256^2 wins by 3-7 %.
    const long tiles128 = (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    const double cost256 = 4.0 * (double)((tiles256 + cus - 1) / cus);
    const double cost128 = 2.0 * 1.15 * (double)((tiles128 + 2 * cus - 1) / (2 * cus));
    const bool use256 = (trans_a || trans_b || mode || dropping) ? true : (force ? (force == 2) : (cost256 <= cost128));
    hipStream_t st = (hipStream_t)stream;
    if (use256) {
#ifdef RV_STAMPS
        if (mode == 0) P.ws = (float*)g_stamp_host;
#endif
        P.tiles_m = (M + BM2 - 1) / BM2; P.tiles_n = (N + BN2 - 1) / BN2;
        if (trans_a) { if (trans_b) launch256<true, true>(P, mode, st); else launch256<true, false>(P, mode, st); }
        else { if (trans_b) launch256<false, true>(P, mode, st); else launch256<false, false>(P, mode, st); }
    } else {
        P.tiles_m = (M + BM - 1) / BM; P.tiles_n = (N + BN - 1) / BN;
        hipLaunchKernelGGL(gemm_nt_kernel, dim3(P.tiles_m * P.tiles_n), dim3(256), NSTAGE * STAGE_BYTES, st, P);
    }
    return rv_check_launch();
}

extern "C" int rv_gemm_bf16_ex(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const void* bias,
                               const void* residual, int64_t ldr, int M, int N, int K, int trans_a, int trans_b, float alpha,
                               int act, int out_f32, int res_f32, const void* A2, int64_t lda2, const void* B2, int64_t ldb2,
                               int K2, void* workspace, int64_t workspace_bytes, const void* zeros16, void* stream) {
    return gemm_ex_impl(A, lda, B, ldb, C, ldc, bias, residual, ldr, M, N, K, trans_a, trans_b, alpha, act, out_f32, res_f32, A2, lda2, B2, ldb2, K2,
                        workspace, workspace_bytes, zeros16, stream, 0.f, 0);
}

// C[M, N] (+)= dropout_p(alpha * A[M, K] op(B)) with the mask of rv_dropout_bf16 over the M * N elements of the product: the adapter branch of
// a LoRA layer's input gradient, dx += dropout'(dt A) (peft LoraLayer: lora_dropout acts on the layer input, so its adjoint masks dt A with
// the forward's mask; reference wiring train/train.py:1515-1532).  The mask is applied to the accumulators in the epilogue: the product
// never reaches HBM unmasked (the unfused sequence wrote it, read it back and read + wrote dx).  accumulate != 0: C = C + ...; N % 8 == 0.
extern "C" int rv_gemm_dropout_add_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K,
                                        int trans_b, float alpha, float p, uint64_t seed, int accumulate, const void* zeros16, void* stream) {
    if (p < 0.f || p >= 1.f || (N & 7) || (ldc & 7) || (((uintptr_t)C) & 15)) return RV_ERR_ARG;
    return gemm_ex_impl(A, lda, B, ldb, C, ldc, nullptr, accumulate ? C : nullptr, ldc, M, N, K, 0, trans_b, alpha, RV_ACT_NONE, 0, 0, nullptr, 0,
                        nullptr, 0, 0, nullptr, 0, zeros16, stream, p, seed);
}

extern "C" int rv_gemm_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const void* bias,
                            const void* residual, int64_t ldr, int M, int N, int K, int trans_a, int trans_b, float alpha,
                            int act, int out_f32, int res_f32, const void* zeros16, void* stream) {
    return rv_gemm_bf16_ex(A, lda, B, ldb, C, ldc, bias, residual, ldr, M, N, K, trans_a, trans_b, alpha, act, out_f32, res_f32,
                           nullptr, 0, nullptr, 0, 0, nullptr, 0, zeros16, stream);
}

extern "C" int rv_gemm_nt_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                               const void* bias, const void* residual, int64_t ldr, int M, int N, int K, int act,
                               int out_f32, int res_f32, const void* zeros16, void* stream) {
    return rv_gemm_bf16(A, lda, B, ldb, C, ldc, bias, residual, ldr, M, N, K, 0, 0, 1.0f, act, out_f32, res_f32, zeros16, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// Fused-epilogue entry points (256x256 kernel, plain tiles).  Each falls back to the unfused sequence (GEMM, then the elementwise
// kernel of ops.hip) when the output has too few 256x256 tiles for that kernel to be the right choice, or when a shape constraint of
// the fused form does not hold -- results are bit-identical either way (the fused epilogues keep the unfused rounding points).
template <int EPI, bool TB, bool BUF>
static void launch_fused1(const GemmParams& P, hipStream_t st) {
    static bool attr = false;      // once per instantiation (see the launch-configuration note above)
    if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_kernel_256<false, TB, 0, EPI, BUF>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES2); attr = true; }
    int grid = P.tiles_m * P.tiles_n;
    if (g_persist && BUF && grid > cu_budget()) grid = cu_budget();
    hipLaunchKernelGGL((gemm_kernel_256<false, TB, 0, EPI, BUF>), dim3(grid), dim3(512), LDS_BYTES2, st, P);
}
template <int EPI, bool TB>
static int launch_fused(GemmParams& P, int tiles_m, int tiles_n, hipStream_t st) {
    P.tiles_m = tiles_m; P.tiles_n = tiles_n; P.n_full = 0; P.splits = 1; P.ws = nullptr;
    P.A2 = nullptr; P.B2 = nullptr; P.K2 = 0; P.lda2 = P.ldb2 = 0;
    if (buf_extents(P, 0, TB)) launch_fused1<EPI, TB, true>(P, st);
    else launch_fused1<EPI, TB, false>(P, st);
    return rv_check_launch();
}
static bool big_enough_for_256(int M, int N) {
    const int cus = cu_budget();
    const long tiles256 = (long)((M + BM2 - 1) / BM2) * ((N + BN2 - 1) / BN2);
    const long tiles128 = (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    const int force = g_force_kernel;
    if (force) return force == 2;
    return 4.0 * (double)((tiles256 + cus - 1) / cus) <= 2.0 * 1.15 * (double)((tiles128 + 2 * cus - 1) / (2 * cus));
}

extern "C" int rv_gemm_rope_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const void* bias,
                                 int M, int N, int K, const float* cos_sin, const int32_t* positions, int S, int rope_heads, int hd,
                                 void* workspace, int64_t workspace_bytes, const void* zeros16, void* stream) {
    if (!A || !B || !C || !cos_sin || !zeros16 || M <= 0 || N <= 0 || K <= 0 || rope_heads <= 0 || (long)rope_heads * hd > N) return RV_ERR_ARG;
    if ((hd & 15) || (!positions && S <= 0)) return RV_ERR_ARG;
    const bool fused = (hd == 128 || hd == 64) && (N % 8 == 0) && (ldc % 8 == 0) && (N % hd == 0) && big_enough_for_256(M, N) &&
                       ((((uintptr_t)C) | ((uintptr_t)bias) | ((uintptr_t)cos_sin)) & 15) == 0 && !(lda & 7) && !(ldb & 7) && !(K & 7);
    if (!fused) {
        int rc = rv_gemm_bf16_ex(A, lda, B, ldb, C, ldc, bias, nullptr, 0, M, N, K, 0, 0, 1.f, RV_ACT_NONE, 0, 0, nullptr, 0, nullptr, 0, 0,
                                 workspace, workspace_bytes, zeros16, stream);
        if (rc != RV_OK) return rc;
        return positions ? rv_rope_inplace_pos(C, ldc, cos_sin, positions, M, rope_heads, hd, 1, 1, stream)
                         : rv_rope_inplace(C, ldc, cos_sin, M, S, rope_heads, hd, 1, 1, stream);
    }
    if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)zeros16) & 15) return RV_ERR_ARG;
    GemmParams P = {};
    P.A = (const bf16*)A; P.B = (const bf16*)B; P.C = C; P.bias = (const bf16*)bias; P.zeros = (const bf16*)zeros16;
    P.lda = lda; P.ldb = ldb; P.ldc = ldc; P.M = M; P.N = N; P.K = K; P.alpha = 1.f;
    P.rope_cs = cos_sin; P.rope_pos = positions; P.rope_S = S > 0 ? S : 1; P.rope_cols = rope_heads * hd; P.rope_hd = hd;
    return launch_fused<EPI_ROPE, false>(P, (M + BM2 - 1) / BM2, (N + BN2 - 1) / BN2, (hipStream_t)stream);
}

extern "C" int rv_gemm_swiglu_fwd_bf16(const void* A, int64_t lda, const void* Wgu, int64_t ldb, void* GU, int64_t ldgu, void* ACT,
                                       int64_t ldact, int M, int F, int K, void* workspace, int64_t workspace_bytes, const void* zeros16,
                                       void* stream) {
    if (!A || !Wgu || !GU || !ACT || !zeros16 || M <= 0 || F <= 0 || K <= 0 || (F & 7) || (ldgu & 7) || (ldact & 7)) return RV_ERR_ARG;
    const bool fused = big_enough_for_256(M, 2 * F) && ((((uintptr_t)GU) | ((uintptr_t)ACT)) & 15) == 0 && !(lda & 7) && !(ldb & 7) && !(K & 7);
    if (!fused) {
        int rc = rv_gemm_bf16_ex(A, lda, Wgu, ldb, GU, ldgu, nullptr, nullptr, 0, M, 2 * F, K, 0, 0, 1.f, RV_ACT_NONE, 0, 0, nullptr, 0, nullptr,
                                 0, 0, workspace, workspace_bytes, zeros16, stream);
        if (rc != RV_OK) return rc;
        return rv_swiglu_fwd(GU, ldgu, ACT, ldact, M, F, stream);
    }
    if (((uintptr_t)A | (uintptr_t)Wgu | (uintptr_t)zeros16) & 15) return RV_ERR_ARG;
    GemmParams P = {};
    P.A = (const bf16*)A; P.B = (const bf16*)Wgu; P.C = GU; P.zeros = (const bf16*)zeros16;
    P.lda = lda; P.ldb = ldb; P.ldc = ldgu; P.M = M; P.N = 2 * F; P.K = K; P.alpha = 1.f;
    P.F = F; P.C2 = (bf16*)ACT; P.ldc2 = ldact;
    return launch_fused<EPI_SWIGLU_FWD, false>(P, (M + BM2 - 1) / BM2, (F + 127) / 128, (hipStream_t)stream);
}

extern "C" int rv_gemm_swiglu_bwd_bf16(const void* dY, int64_t ldy, const void* Wd, int64_t ldw, const void* GU, int64_t ldgu, void* dGU,
                                       int64_t lddgu, void* dact_scratch, int64_t ld_dact, int M, int F, int K, void* workspace,
                                       int64_t workspace_bytes, const void* zeros16, void* stream) {
    if (!dY || !Wd || !GU || !dGU || !zeros16 || M <= 0 || F <= 0 || K <= 0 || (F & 7) || (ldgu & 7) || (lddgu & 7)) return RV_ERR_ARG;
    const bool fused = big_enough_for_256(M, F) && ((((uintptr_t)GU) | ((uintptr_t)dGU)) & 15) == 0 && !(ldy & 7) && !(ldw & 7) && !(K & 7);
    if (!fused) {
        if (!dact_scratch || (ld_dact & 7)) return RV_ERR_ARG;      // the unfused sequence needs d(act) [M, F] as a real tensor
        int rc = rv_gemm_bf16_ex(dY, ldy, Wd, ldw, dact_scratch, ld_dact, nullptr, nullptr, 0, M, F, K, 0, 1, 1.f, RV_ACT_NONE, 0, 0, nullptr, 0,
                                 nullptr, 0, 0, workspace, workspace_bytes, zeros16, stream);
        if (rc != RV_OK) return rc;
        return rv_swiglu_bwd(dact_scratch, ld_dact, GU, ldgu, dGU, lddgu, M, F, stream);
    }
    if (((uintptr_t)dY | (uintptr_t)Wd | (uintptr_t)zeros16) & 15) return RV_ERR_ARG;
    GemmParams P = {};
    P.A = (const bf16*)dY; P.B = (const bf16*)Wd; P.C = dGU; P.zeros = (const bf16*)zeros16;
    P.lda = ldy; P.ldb = ldw; P.ldc = lddgu; P.M = M; P.N = F; P.K = K; P.alpha = 1.f;
    P.F = F; P.G = (const bf16*)GU; P.ldg = ldgu;
    return launch_fused<EPI_SWIGLU_BWD, true>(P, (M + BM2 - 1) / BM2, (F + BN2 - 1) / BN2, (hipStream_t)stream);
}

// gA[R, K] (+)= 1 / (1 - p) * dT[M, R]^T mask_p(X)[M, K] with the mask of rv_dropout_bf16(X viewed as M * K contiguous elements, p, seed): the
// gradient of a LoRA adapter's A matrix in ONE pass over X (lora_agrad_kernel above).  R <= 64, R % 8 == 0, K % 8 == 0; with p > 0 X must be
// contiguous (ldx == K).  workspace: fp32 scratch for the token-slice partial sums (>= R * K * 4 bytes; more slices fit in a larger one).
extern "C" int rv_lora_a_grad_bf16(const void* dT, int64_t ldt, const void* X, int64_t ldx, void* gA, int64_t ldg, int M, int R, int K, float p,
                                   uint64_t seed, int accumulate, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!dT || !X || !gA || !workspace || M <= 0 || R <= 0 || K <= 0 || p < 0.f || p >= 1.f) return RV_ERR_ARG;
    if (R > 64 || (R & 7) || (K & 7) || (ldx & 7) || (ldt & 7) || (p > 0.f && ldx != K)) return RV_ERR_ARG;
    if ((((uintptr_t)X) | ((uintptr_t)dT)) & 15) return RV_ERR_ARG;
    const int ncb = (K + AG_BN - 1) / AG_BN, nst = (M + AG_BM - 1) / AG_BM;
    int splits = (3 * cu_budget() + ncb - 1) / ncb;          // three blocks per CU are resident (48 KiB of LDS each)
    if (splits > nst) splits = nst;
    if (splits > 32) splits = 32;
    const int64_t per = (int64_t)R * K * 4;
    if (splits > workspace_bytes / per) splits = (int)(workspace_bytes / per);
    if (splits < 1) return RV_ERR_ARG;
    LoraAGradParams Q;
    Q.X = (const bf16*)X; Q.T = (const bf16*)dT; Q.ws = (float*)workspace; Q.ldx = ldx; Q.ldt = ldt; Q.M = M; Q.R = R; Q.K = K; Q.splits = splits;
    Q.thr = rv_dropout_thr16(p); Q.seed = seed;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(lora_agrad_kernel, dim3((unsigned)(ncb * splits)), dim3(256), 0, st, Q);
    GemmParams P{};
    P.M = R; P.N = K; P.splits = splits; P.ws = (float*)workspace; P.alpha = p > 0.f ? 1.f / (1.f - p) : 1.f; P.bias = nullptr; P.act = RV_ACT_NONE;
    P.R = accumulate ? gA : nullptr; P.ldr = ldg; P.res_f32 = 0; P.C = gA; P.ldc = ldg; P.out_f32 = 0;
    const long total = (long)R * K;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, P);
    return rv_check_launch();
}

// T[M, R] = alpha / (1 - p) * dropout_p(X)[M, K] A[R, K]^T with the mask of rv_dropout_bf16(X viewed as M * K contiguous elements, p, seed):
// the LoRA adapters' down-projection (peft LoraLayer: lora_A(lora_dropout(x)), reference wiring train/train.py:1515-1532) in ONE pass over X.
extern "C" int rv_lora_down_bf16(const void* X, int64_t ldx, const void* A, int64_t lda, void* T, int64_t ldt, int M, int R, int K, float alpha,
                                 float p, uint64_t seed, const void* zeros16, void* stream) {
    if (!X || !A || !T || !zeros16 || M <= 0 || R <= 0 || K <= 0 || p < 0.f || p >= 1.f) return RV_ERR_ARG;
    // the mask indexes X as a contiguous [M, K] tensor (ldx == K), like the dropout kernels backward re-creates it with
    if (R > LD_BN || (R & 3) || (K % BK) || (p > 0.f && ldx != K) || (ldx & 7) || (lda & 7) || (ldt & 3)) return RV_ERR_ARG;
    if ((((uintptr_t)X) | ((uintptr_t)A) | ((uintptr_t)zeros16)) & 15) return RV_ERR_ARG;
    if (((uintptr_t)T) & 7) return RV_ERR_ARG;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)lora_down_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LD_NSTAGE * LD_STAGE); attr = true; }
    LoraDownParams P;
    P.X = (const bf16*)X; P.A = (const bf16*)A; P.T = (bf16*)T; P.zeros = (const bf16*)zeros16;
    P.ldx = ldx; P.lda = lda; P.ldt = ldt; P.M = M; P.R = R; P.K = K;
    P.alpha = p > 0.f ? alpha / (1.f - p) : alpha;
    P.thr = rv_dropout_thr16(p); P.seed = seed;
    hipLaunchKernelGGL(lora_down_kernel, dim3((M + LD_BM - 1) / LD_BM), dim3(256), LD_NSTAGE * LD_STAGE, (hipStream_t)stream, P);
    return rv_check_launch();
}
