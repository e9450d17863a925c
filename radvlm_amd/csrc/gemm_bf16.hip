// bf16 MFMA GEMM for gfx950:  C[M,N] = act(A[M,K] * B[N,K]^T + bias[N]) + R[M,N]
// ("NT": both operands contraction-contiguous = torch.nn.Linear's  y = x W^T  with W stored [out,in]).
//
// Replaces (reference call sites): every nn.Linear / F.linear of the hot path -- HF Llama q/k/v/o/gate/up/down
// (finetuning/llava/model/language_model/modeling_llama.py:332-338,377,226), lm_head (:1323), CLIP q/k/v/out/fc1/fc2
// (HF:models/clip/modeling_clip.py:298-350), patch-embed conv as GEMM (:209), mm_projector
// (multimodal_projector/builder.py:41-48) and their autograd dgrad/wgrad (fed with transposed copies).
//
// Structure (v1): 128x128x64 block tile, 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32 tiles,
// LDS-DMA staging (global_load_lds 16 B) into a 2-stage ring, XOR-swizzled 128-B rows so that the
// ds_read_b128 fragment reads are bank-conflict free, XCD-aware + grouped block->tile map.
// Edge handling: out-of-range rows / k-chunks are sourced from a 16-byte zero page (per-lane source select),
// so any M, N and any K % 8 == 0 work; stores are masked.
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
};

DEVINL float apply_act(float x, int act) {
    if (act == RV_ACT_QUICK_GELU) return x / (1.f + __expf(-1.702f * x));
    if (act == RV_ACT_GELU) return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
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
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
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
// v2: 256x256x64 block tile, 8 waves (2 M x 4 N), 128x64 per wave, ~1 block/CU (128 KiB LDS, <=256 VGPR).
// Each K-tile is staged as four 16-KiB half-tiles (A rows 0-127 / 128-255, B rows 0-127 / 128-255), ONE half-tile
// per phase, into a 2-deep ring; a K-tile is computed in four phases of 16 MFMAs (one 64x32 quadrant each).
//   ph1: stage A0(t+1) | read A(mh0), B(nh0) | MFMA (mh0,nh0)
//   ph2: stage A1(t+1) | read B(nh1)         | MFMA (mh0,nh1)      -- barrier: all B reads of tile t retired
//   ph3: stage B0(t+2) | read A(mh1)         | MFMA (mh1,nh1)
//   ph4: stage B1(t+2) |                     | MFMA (mh1,nh0)      -- vmcnt(4): tile t+1 landed, B(t+2) still in flight
// Both B sub-tiles stay in registers, so the B half of a ring slot is free after ph2 and the A half after ph3;
// LDS-DMA loads therefore stay in flight across both barriers (counted vmcnt, raw s_barrier; never vmcnt(0) in the loop).
constexpr int BM2 = 256, BN2 = 256;
constexpr int HALF_BYTES = 128 * BK * 2;      // 16 KiB
constexpr int KT_BYTES2 = 4 * HALF_BYTES;     // 64 KiB per K-tile: A0 A1 B0 B1

DEVINL void stage_half(const bf16* __restrict__ src, long ld, int row0, int nrows_total, int k0, int K, const bf16* zeros,
                       char* lds, int wid, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int it = i * 8 + wid;           // wave-instruction index 0..15
        const int c = it * 64 + lane;         // 16-B chunk id 0..1023
        const int r = c >> 3, p = c & 7;
        const int lc = p ^ ((r >> 1) & 7);
        const int gr = row0 + r, gk = k0 + lc * 8;
        const bf16* g = (gr < nrows_total && gk < K) ? (src + (long)gr * ld + gk) : zeros;
        glds16(g, lds + it * 1024);
    }
}

#define BAR_LGKM() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

__global__ __launch_bounds__(512, 1) void gemm_nt_kernel_256(GemmParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wid = wave_id(), lane = lane_id();
    const int wr = wid >> 2, wc = wid & 3;

    const int nwg = P.tiles_m * P.tiles_n;
    int pid = xcd_remap(blockIdx.x, nwg);
    constexpr int GROUP_M = 4;
    const int per_group = GROUP_M * P.tiles_n;
    const int group = pid / per_group;
    const int first_m = group * GROUP_M;
    const int gsz = min(P.tiles_m - first_m, GROUP_M);
    const int tm = first_m + (pid % per_group) % gsz;
    const int tn = (pid % per_group) / gsz;
    const int m0 = tm * BM2, n0 = tn * BN2;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nt = (P.K + BK - 1) / BK;
    auto stageA = [&](int t, int h) { stage_half(P.A, P.lda, m0 + h * 128, P.M, t * BK, P.K, P.zeros, smem + (t & 1) * KT_BYTES2 + h * HALF_BYTES, wid, lane); };
    auto stageB = [&](int t, int h) { stage_half(P.B, P.ldb, n0 + h * 128, P.N, t * BK, P.K, P.zeros, smem + (t & 1) * KT_BYTES2 + (2 + h) * HALF_BYTES, wid, lane); };

    // prologue: all of tile 0, and the B halves of tile 1
    stageA(0, 0); stageA(0, 1); stageB(0, 0); stageB(0, 1);
    if (nt > 1) { stageB(1, 0); stageB(1, 1); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const int arow = lane & 15, kq = lane >> 4;
    for (int t = 0; t < nt; ++t) {
        const char* At = smem + (t & 1) * KT_BYTES2 + wr * HALF_BYTES;
        const char* Bt = smem + (t & 1) * KT_BYTES2 + (2 + (wc >> 1)) * HALF_BYTES;
        const int brow0 = (wc & 1) * 64;
        bf16x8 a[4][2], b[2][2][2];

        // ---- phase 1
        if (t + 1 < nt) stageA(t + 1, 0);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int j = 0; j < 2; ++j) b[0][j][kk] = read_frag(Bt, brow0 + j * 16 + arow, kk * 4 + kq);
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i][kk] = read_frag(At, i * 16 + arow, kk * 4 + kq);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(b[0][j][kk], a[i][kk], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);

        // ---- phase 2
        if (t + 1 < nt) stageA(t + 1, 1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 2; ++j) b[1][j][kk] = read_frag(Bt, brow0 + 32 + j * 16 + arow, kk * 4 + kq);
        BAR_LGKM();   // every wave's B reads of tile t are complete -> the B halves of this slot may be restaged
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][2 + j] = mfma16(b[1][j][kk], a[i][kk], acc[i][2 + j]);
        __builtin_amdgcn_s_setprio(0);

        // ---- phase 3
        if (t + 2 < nt) stageB(t + 2, 0);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i][kk] = read_frag(At, 64 + i * 16 + arow, kk * 4 + kq);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[4 + i][2 + j] = mfma16(b[1][j][kk], a[i][kk], acc[4 + i][2 + j]);
        __builtin_amdgcn_s_setprio(0);

        // ---- phase 4
        if (t + 2 < nt) stageB(t + 2, 1);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[4 + i][j] = mfma16(b[0][j][kk], a[i][kk], acc[4 + i][j]);
        __builtin_amdgcn_s_setprio(0);
        // tile t+1 (A issued in ph1/ph2 of this tile, B one tile earlier) must have landed; B(t+2) may stay in flight
        if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        BAR_LGKM();   // also: every wave's A reads of tile t are complete -> the A halves of this slot may be restaged
    }

    // epilogue: lane holds C[m = m0 + wr*128 + i*16 + (lane&15)][n = n0 + wc*64 + j*16 + 4*(lane>>4) + r]
    const bool n_vec_ok = (P.N % 4 == 0) && (P.ldc % 4 == 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = m0 + wr * 128 + i * 16 + (lane & 15);
        if (m >= P.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wc * 64 + j * 16 + 4 * (lane >> 4);
            if (n >= P.N) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
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

}  // namespace

static int g_force_kernel = 0;  // 0 auto, 1 = 128x128 kernel, 2 = 256x256 kernel (RV_GEMM_KERNEL or rv_gemm_select_kernel)
extern "C" int rv_gemm_select_kernel(int which) { g_force_kernel = which; return RV_OK; }

extern "C" int rv_gemm_nt_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                               const void* bias, const void* residual, int64_t ldr, int M, int N, int K, int act,
                               int out_f32, int res_f32, const void* zeros16, void* stream) {
    if (!A || !B || !C || !zeros16 || M <= 0 || N <= 0 || K <= 0) return RV_ERR_ARG;
    if ((K & 7) || (lda & 7) || (ldb & 7)) return RV_ERR_ARG;
    if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)zeros16) & 15) return RV_ERR_ARG;
    GemmParams P;
    P.A = (const bf16*)A; P.B = (const bf16*)B; P.C = C; P.bias = (const bf16*)bias; P.R = residual;
    P.zeros = (const bf16*)zeros16;
    P.lda = lda; P.ldb = ldb; P.ldc = ldc; P.ldr = ldr;
    P.M = M; P.N = N; P.K = K; P.act = act; P.out_f32 = out_f32; P.res_f32 = res_f32;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_nt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE * STAGE_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm_nt_kernel_256, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * KT_BYTES2);
        const char* e = getenv("RV_GEMM_KERNEL");
        if (e) g_force_kernel = atoi(e);
        attr_set = true;
    }
    const long tiles256 = (long)((M + BM2 - 1) / BM2) * ((N + BN2 - 1) / BN2);
    const int force = g_force_kernel;
    const bool use256 = force ? (force == 2) : (tiles256 >= 200);  // enough 256^2 tiles to fill the 256 CUs
    if (use256) {
        P.tiles_m = (M + BM2 - 1) / BM2; P.tiles_n = (N + BN2 - 1) / BN2;
        hipLaunchKernelGGL(gemm_nt_kernel_256, dim3(P.tiles_m * P.tiles_n), dim3(512), 2 * KT_BYTES2, (hipStream_t)stream, P);
    } else {
        P.tiles_m = (M + BM - 1) / BM; P.tiles_n = (N + BN - 1) / BN;
        hipLaunchKernelGGL(gemm_nt_kernel, dim3(P.tiles_m * P.tiles_n), dim3(256), NSTAGE * STAGE_BYTES, (hipStream_t)stream, P);
    }
    return rv_check_launch();
}
