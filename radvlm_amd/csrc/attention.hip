// Fused (flash-style) multi-head attention for gfx950, forward and backward, bf16 I/O, fp32 softmax.
//
// Replaces: LlamaAttention core  softmax_fp32(QK^T/sqrt(hd) + causal + key-padding) V
//   (finetuning/llava/model/language_model/modeling_llama.py:349-368, mask :1191-1225; fused contract
//    finetuning/llava/train/llama_flash_attn_monkey_patch.py:51-69: causal, key padding, dropout 0, scale 1/sqrt(hd))
//   and CLIPAttention's non-causal MHSA (HF:models/clip/modeling_clip.py:259-277,298-335), plus their autograd backward.
//
// MFMA: v_mfma_f32_16x16x32_bf16 only. Every operand is read contraction-contiguous from an XOR-swizzled LDS
// tile (ds_read_b128 / ds_read_b64); operands whose contraction index is the sequence axis come from
// pre-transposed global copies [b,h,hd,S_pad] (rv_transpose_bf16, perm32 order), so no LDS transposes are needed.
// The score tile produced by one MFMA chain is reused as the next chain's B operand straight from the
// accumulator registers (contraction order kappa(g,j) = 16*(2p + (j>>2)) + 4g + (j&3), matched by the A-side reads).
//
// Layouts: q,k,v,o,dq,dk,dv,dO are token-major rows [(b*S+s)*ld + h*HD + e]; lse/delta are fp32 [b,h,S_pad].
#include "common.h"
#include "radvlm_hip.h"

namespace {

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
};

template <int ROW_BYTES> DEVINL int swz(int r) { return ROW_BYTES == 128 ? ((r >> 1) & 7) : (r & 15); }

// Stage NROWS rows of ROW_BYTES each into an LDS tile; rows >= valid_rows come from the zero page.
template <int ROW_BYTES, int NROWS, int NW = 4>
DEVINL void stage_rows(const bf16* src, long row_stride, int valid_rows, const bf16* zeros, char* lds, int wid, int lane) {
    constexpr int CPR = ROW_BYTES / 16;
    constexpr int NI = NROWS * CPR / 64;  // wave-instructions for the tile
    static_assert(NI % NW == 0, "tile must split evenly over the block's waves");
#pragma unroll
    for (int i = 0; i < NI / NW; ++i) {
        const int it = i * NW + wid;
        const int c = it * 64 + lane;
        const int r = c / CPR, p = c % CPR;
        const int lc = p ^ swz<ROW_BYTES>(r);
        const bf16* g = (r < valid_rows) ? (src + (long)r * row_stride + lc * 8) : zeros;
        glds16(g, lds + it * 1024);
    }
}

template <int ROW_BYTES> DEVINL bf16x8 rd128(const char* tile, int row, int chunk) {
    return *(const bf16x8*)(tile + row * ROW_BYTES + ((chunk ^ swz<ROW_BYTES>(row)) << 4));
}
// A-operand for a contraction over the tile's 64 (sequence) columns: pair p in {0,1} = 32 columns, lane group g.
// The transposed global copies are written in MFMA contraction order (rv_transpose_bf16 perm32), so the 8 operands
// kappa(g, 0..7) of a lane are the 16 contiguous bytes of chunk 4p + g: one conflict-free ds_read_b128.
DEVINL bf16x8 rdT(const char* tile, int row, int p, int g) { return rd128<128>(tile, row, 4 * p + g); }
// asm-issued fragment reads + counted waits (LDS operations return in order; LEFT = younger operations that may stay in
// flight).  hipcc's own bookkeeping put an `s_waitcnt lgkmcnt(0)` right behind every compiler-visible ds_read in these loops
// (one LDS round trip per pair of MFMAs); with the reads in asm a batch of four is fetched two batches ahead of its MFMAs.
template <int ROW_BYTES> DEVINL void rd128_asm(bf16x8& dst, const char* tile, int row, int chunk) {
    const char* a = tile + row * ROW_BYTES + ((chunk ^ swz<ROW_BYTES>(row)) << 4);
    const unsigned o = (unsigned)(unsigned long)(__attribute__((address_space(3))) const char*)a;
    asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(o) : "memory");
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

// ------------------------------------------------------------------------------------------------ forward
template <int HD, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnParams P) {
    constexpr int KS = HD / 32, DB = HD / 16, KROW = HD * 2;
    constexpr int KT_BYTES = 64 * KROW, VT_BYTES = HD * 128, STAGE = KT_BYTES + VT_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wid = wave_id(), lane = lane_id(), g = lane >> 4, c = lane & 15;
    const int qblk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const long rb = P.cu ? (long)P.cu[b] : (long)b * P.S;              // first token row of this sample
    const int S = P.cu ? P.cu[b + 1] - P.cu[b] : P.S, q0 = qblk * 128 + wid * 32;
    if (qblk * 128 >= S) return;                                        // packed batches: shorter samples need fewer blocks
    const int len = (P.lens && !P.cu) ? P.lens[b] : S;
    const float sl2 = P.scale * LOG2E;

    bf16x8 qf[2][KS];
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
        const int row = min(q0 + qs * 16 + c, S - 1);
        const bf16* p = P.q + (rb + row) * P.ld_q + h * HD + g * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[qs][ks] = *(const bf16x8*)(p + ks * 32);
    }
    const int kv_end = CAUSAL ? min(len, qblk * 128 + 128) : len;
    const int ntiles = (kv_end + 63) >> 6;
    const int hk = h / P.nrep;
    const bf16* kbase = P.k + rb * P.ld_k + hk * HD;
    const bf16* vtbase = P.vT + (long)(b * P.Hkv + hk) * HD * P.S_pad;

    float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};
    f32x4 o[2][DB];
#pragma unroll
    for (int qs = 0; qs < 2; ++qs)
#pragma unroll
        for (int db = 0; db < DB; ++db) o[qs][db] = f32x4{0.f, 0.f, 0.f, 0.f};

    stage_rows<KROW, 64>(kbase, P.ld_k, S, P.zeros, smem, wid, lane);
    stage_rows<128, HD>(vtbase, P.S_pad, HD, P.zeros, smem + KT_BYTES, wid, lane);

    for (int t = 0; t < ntiles; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const char* Kt = smem + (t & 1) * STAGE;
        const char* Vt = Kt + KT_BYTES;
        if (t + 1 < ntiles) {
            char* nx = smem + ((t + 1) & 1) * STAGE;
            const int kv1 = (t + 1) * 64;
            stage_rows<KROW, 64>(kbase + (long)kv1 * P.ld_k, P.ld_k, S - kv1, P.zeros, nx, wid, lane);
            stage_rows<128, HD>(vtbase + kv1, P.S_pad, HD, P.zeros, nx + KT_BYTES, wid, lane);
        }
        const int kv0 = t * 64;
        if (CAUSAL && kv0 > q0 + 31) continue;  // every key of this tile is in the future of this wave's rows

        f32x4 s[2][4];
#pragma unroll
        for (int qs = 0; qs < 2; ++qs)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) s[qs][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
        {   // S^T = K Q^T: the KS fragments of key block kb are one batch; batches kb+1, kb+2 are in flight while kb's MFMAs run
            bf16x8 kq[3][KS];
            auto issue_k = [&](int kb, bf16x8 (&dst)[KS]) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) rd128_asm<KROW>(dst[ks], Kt, kb * 16 + c, ks * 4 + g);
            };
            issue_k(0, kq[0]);
            issue_k(1, kq[1]);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                if (kb + 2 < 4) issue_k(kb + 2, kq[(kb + 2) % 3]);
                if (kb < 2) lds_wait<KS, 2 * KS>(kq[kb % 3]);
                else if (kb == 2) lds_wait<KS, KS>(kq[kb % 3]);
                else lds_wait<KS, 0>(kq[kb % 3]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int qs = 0; qs < 2; ++qs) s[qs][kb] = mfma16(kq[kb % 3][ks], qf[qs][ks], s[qs][kb]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // lane holds S^T[key = kv0 + 16kb + 4g + r][q = q0 + 16qs + c]
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
            const int qidx = q0 + qs * 16 + c;
            float mx = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kidx = kv0 + kb * 16 + 4 * g + r;
                    const bool ok = (kidx < len) && (!CAUSAL || kidx <= qidx);
                    const float tv = ok ? s[qs][kb][r] * sl2 : -INFINITY;
                    s[qs][kb][r] = tv;
                    mx = fmaxf(mx, tv);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mnew = fmaxf(m[qs], mx);
            const float alpha = fexp2(m[qs] - mnew);
            float rs = 0.f;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = fexp2(s[qs][kb][r] - mnew);
                    s[qs][kb][r] = p;
                    rs += p;
                }
            l[qs] = l[qs] * alpha + rs;
            m[qs] = mnew;
#pragma unroll
            for (int db = 0; db < DB; ++db) o[qs][db] *= alpha;
        }
        // O^T[d][q] += V^T[d][key] P^T[key][q]: V^T fragments in batches of four (db), two batches ahead of their MFMAs
        {
            bf16x8 vq[3][4];
            constexpr int NB = 2 * DB / 4;     // batches over (kp, db)
            auto issue_v = [&](int bi, bf16x8 (&dst)[4]) {
                const int kp = bi / (DB / 4), db0 = (bi % (DB / 4)) * 4;
#pragma unroll
                for (int u = 0; u < 4; ++u) rd128_asm<128>(dst[u], Vt, (db0 + u) * 16 + c, 4 * kp + g);
            };
            issue_v(0, vq[0]);
            issue_v(1, vq[1]);
            bf16x8 pf[2][2];
#pragma unroll
            for (int kp = 0; kp < 2; ++kp)
#pragma unroll
                for (int qs = 0; qs < 2; ++qs) pf[kp][qs] = pack8(s[qs][2 * kp], s[qs][2 * kp + 1]);
#pragma unroll
            for (int bi = 0; bi < NB; ++bi) {
                if (bi + 2 < NB) issue_v(bi + 2, vq[(bi + 2) % 3]);
                if (bi + 2 < NB) lds_wait<4, 8>(vq[bi % 3]);
                else if (bi + 1 < NB) lds_wait<4, 4>(vq[bi % 3]);
                else lds_wait<4, 0>(vq[bi % 3]);
                __builtin_amdgcn_sched_barrier(0);
                const int kp = bi / (DB / 4), db0 = (bi % (DB / 4)) * 4;
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int qs = 0; qs < 2; ++qs) o[qs][db0 + u] = mfma16(vq[bi % 3][u], pf[kp][qs], o[qs][db0 + u]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
        float lt = l[qs];
        lt += __shfl_xor(lt, 16, 64);
        lt += __shfl_xor(lt, 32, 64);
        const int qidx = q0 + qs * 16 + c;
        if (qidx >= S) continue;
        const float inv = 1.f / lt;
        bf16* op = P.out + (rb + qidx) * P.ld_o + h * HD + 4 * g;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            const f32x4 v = o[qs][db] * inv;
            *(bf16x4*)(op + db * 16) = bf16x4{f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        }
        if (g == 0 && P.lse) P.lse[(long)(b * P.H + h) * P.S_pad + qidx] = (m[qs] + __builtin_amdgcn_logf(lt)) * LN2;
    }
}

// ------------------------------------------------------------------------------------------------ backward: dQ
template <int HD, bool CAUSAL, int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void attn_bwd_dq_kernel(AttnParams P) {
    constexpr int KS = HD / 32, DB = HD / 16, KROW = HD * 2;
    constexpr int NAT_BYTES = 64 * KROW, T_BYTES = HD * 128, STAGE = 2 * NAT_BYTES + T_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wid = wave_id(), lane = lane_id(), g = lane >> 4, c = lane & 15;
    const int qblk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const long rb = P.cu ? (long)P.cu[b] : (long)b * P.S;
    const int S = P.cu ? P.cu[b + 1] - P.cu[b] : P.S, q0 = qblk * (32 * NW) + wid * 32;
    if (qblk * (32 * NW) >= S) return;
    const int len = (P.lens && !P.cu) ? P.lens[b] : S;
    const float sl2 = P.scale * LOG2E;

    bf16x8 qf[2][KS], dof[2][KS];
    float lse2[2], dl[2];
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
        const int row = min(q0 + qs * 16 + c, S - 1);
        const bf16* p = P.q + (rb + row) * P.ld_q + h * HD + g * 8;
        const bf16* d = P.dout + (rb + row) * P.ld_do + h * HD + g * 8;
        const bf16* op = P.o + (rb + row) * P.ld_o + h * HD + g * 8;
        float dsum = 0.f;   // delta = rowsum(dO * O), fused here (this lane owns 8 * KS of the row's HD products)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[qs][ks] = *(const bf16x8*)(p + ks * 32);
            dof[qs][ks] = *(const bf16x8*)(d + ks * 32);
            const bf16x8 ov = *(const bf16x8*)(op + ks * 32);
#pragma unroll
            for (int j = 0; j < 8; ++j) dsum += bf2f(ov[j]) * bf2f(dof[qs][ks][j]);
        }
        dsum += __shfl_xor(dsum, 16, 64);
        dsum += __shfl_xor(dsum, 32, 64);
        lse2[qs] = P.lse[(long)(b * P.H + h) * P.S_pad + row] * LOG2E;
        dl[qs] = dsum;
        if (g == 0 && q0 + qs * 16 + c < S) P.delta[(long)(b * P.H + h) * P.S_pad + row] = dsum;   // for the dK/dV pass
    }
    const int kv_end = CAUSAL ? min(len, (qblk + 1) * (32 * NW)) : len;
    const int ntiles = (kv_end + 63) >> 6;
    const int hk = h / P.nrep;
    const bf16* kbase = P.k + rb * P.ld_k + hk * HD;
    const bf16* vbase = P.v + rb * P.ld_v + hk * HD;
    const bf16* ktbase = P.kT + (long)(b * P.Hkv + hk) * HD * P.S_pad;

    f32x4 dq[2][DB];
#pragma unroll
    for (int qs = 0; qs < 2; ++qs)
#pragma unroll
        for (int db = 0; db < DB; ++db) dq[qs][db] = f32x4{0.f, 0.f, 0.f, 0.f};

    stage_rows<KROW, 64, NW>(kbase, P.ld_k, S, P.zeros, smem, wid, lane);
    stage_rows<KROW, 64, NW>(vbase, P.ld_v, S, P.zeros, smem + NAT_BYTES, wid, lane);
    stage_rows<128, HD, NW>(ktbase, P.S_pad, HD, P.zeros, smem + 2 * NAT_BYTES, wid, lane);

    for (int t = 0; t < ntiles; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const char* Kt = smem + (t & 1) * STAGE;
        const char* Vt = Kt + NAT_BYTES;
        const char* KTt = Kt + 2 * NAT_BYTES;
        if (t + 1 < ntiles) {
            char* nx = smem + ((t + 1) & 1) * STAGE;
            const int kv1 = (t + 1) * 64;
            stage_rows<KROW, 64, NW>(kbase + (long)kv1 * P.ld_k, P.ld_k, S - kv1, P.zeros, nx, wid, lane);
            stage_rows<KROW, 64, NW>(vbase + (long)kv1 * P.ld_v, P.ld_v, S - kv1, P.zeros, nx + NAT_BYTES, wid, lane);
            stage_rows<128, HD, NW>(ktbase + kv1, P.S_pad, HD, P.zeros, nx + 2 * NAT_BYTES, wid, lane);
        }
        const int kv0 = t * 64;
        if (CAUSAL && kv0 > q0 + 31) continue;

        f32x4 s[2][4], dp[2][4];
#pragma unroll
        for (int qs = 0; qs < 2; ++qs)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) { s[qs][kb] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[qs][kb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 kf = rd128<KROW>(Kt, kb * 16 + c, ks * 4 + g);
                const bf16x8 vf = rd128<KROW>(Vt, kb * 16 + c, ks * 4 + g);
#pragma unroll
                for (int qs = 0; qs < 2; ++qs) {
                    s[qs][kb] = mfma16(kf, qf[qs][ks], s[qs][kb]);
                    dp[qs][kb] = mfma16(vf, dof[qs][ks], dp[qs][kb]);
                }
            }
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
            const int qidx = q0 + qs * 16 + c;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kidx = kv0 + kb * 16 + 4 * g + r;
                    const bool ok = (kidx < len) && (!CAUSAL || kidx <= qidx);
                    const float p = fexp2(s[qs][kb][r] * sl2 - lse2[qs]);
                    s[qs][kb][r] = ok ? p * (dp[qs][kb][r] - dl[qs]) : 0.f;
                }
        }
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) {
            bf16x8 dsf[2];
#pragma unroll
            for (int qs = 0; qs < 2; ++qs) dsf[qs] = pack8(s[qs][2 * kp], s[qs][2 * kp + 1]);
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const bf16x8 ktf = rdT(KTt, db * 16 + c, kp, g);
#pragma unroll
                for (int qs = 0; qs < 2; ++qs) dq[qs][db] = mfma16(ktf, dsf[qs], dq[qs][db]);
            }
        }
    }
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
        const int qidx = q0 + qs * 16 + c;
        if (qidx >= S) continue;
        bf16* op = P.dq + (rb + qidx) * P.ld_dq + h * HD + 4 * g;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            const f32x4 v = dq[qs][db] * P.scale;
            *(bf16x4*)(op + db * 16) = bf16x4{f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
template <int HD, bool CAUSAL, int NW, int NKB>
__global__ __launch_bounds__(NW * 64, NW / 4) void attn_bwd_dkv_kernel(AttnParams P) {
    constexpr int KS = HD / 32, DB = HD / 16, KROW = HD * 2;
    constexpr int NAT_BYTES = 64 * KROW, T_BYTES = HD * 128, STAGE = 2 * NAT_BYTES + 2 * T_BYTES + 1024;  // + lse|delta
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wid = wave_id(), lane = lane_id(), g = lane >> 4, c = lane & 15;
    const int kblk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    constexpr int KPB = NW * NKB * 16;   // keys per block
    const long rb = P.cu ? (long)P.cu[b] : (long)b * P.S;
    const int S = P.cu ? P.cu[b + 1] - P.cu[b] : P.S, k0 = kblk * KPB + wid * (16 * NKB);
    if (kblk * KPB >= S) return;
    const int len = (P.lens && !P.cu) ? P.lens[b] : S;
    const float sl2 = P.scale * LOG2E;

    bf16x8 kf[NKB][KS], vf[NKB][KS];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        const int row = min(k0 + kb * 16 + c, S - 1);
        const bf16* kp = P.k + (rb + row) * P.ld_k + (h / P.kdiv) * HD + g * 8;
        const bf16* vp = P.v + (rb + row) * P.ld_v + (h / P.kdiv) * HD + g * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { kf[kb][ks] = *(const bf16x8*)(kp + ks * 32); vf[kb][ks] = *(const bf16x8*)(vp + ks * 32); }
    }
    const int q_end = len;  // query rows >= len carry zero dO
    const int t0 = CAUSAL ? (kblk * KPB) >> 6 : 0;
    const int t1 = (q_end + 63) >> 6;
    // grouped-query attention, two launch shapes:
    //  (kdiv 1, qrep n): blockIdx.y is the KEY/VALUE head; its n query heads are walked by one flattened (head, query tile)
    //    loop, so dK/dV accumulate in registers across the group and the prefetch runs across head boundaries;
    //  (kdiv n, qrep 1): blockIdx.y is a QUERY head writing per-query-head partials that group_sum_heads_kernel folds --
    //    n times more blocks, for grids too small to balance the causal triangle over 256 CUs.
    const int nt = max(t1 - t0, 0);
    const int n_it = nt * P.qrep;

    f32x4 dv[DB][NKB], dk[DB][NKB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) { dv[db][kb] = f32x4{0.f, 0.f, 0.f, 0.f}; dk[db][kb] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    auto stage = [&](int it, char* dst) {
        const int hq = h * P.qrep + it / nt;
        const int qt0 = (t0 + it % nt) * 64;
        const bf16* qbase = P.q + rb * P.ld_q + hq * HD;
        const bf16* dobase = P.dout + rb * P.ld_do + hq * HD;
        const bf16* qtbase = P.qT + (long)(b * P.H + hq) * HD * P.S_pad;
        const bf16* dotbase = P.doT + (long)(b * P.H + hq) * HD * P.S_pad;
        const float* lsebase = P.lse + (long)(b * P.H + hq) * P.S_pad;
        const float* dlbase = P.delta + (long)(b * P.H + hq) * P.S_pad;
        stage_rows<KROW, 64, NW>(qbase + (long)qt0 * P.ld_q, P.ld_q, S - qt0, P.zeros, dst, wid, lane);
        stage_rows<KROW, 64, NW>(dobase + (long)qt0 * P.ld_do, P.ld_do, S - qt0, P.zeros, dst + NAT_BYTES, wid, lane);
        stage_rows<128, HD, NW>(qtbase + qt0, P.S_pad, HD, P.zeros, dst + 2 * NAT_BYTES, wid, lane);
        stage_rows<128, HD, NW>(dotbase + qt0, P.S_pad, HD, P.zeros, dst + 2 * NAT_BYTES + T_BYTES, wid, lane);
        // lse / delta of the tile's 64 query rows ride the same LDS-DMA stream (a plain global load here would make
        // hipcc drain vmcnt(0) -- i.e. the whole prefetched tile -- before every use)
        if (wid == 0) {
            const float* g = lane < 16 ? lsebase + qt0 + lane * 4 : (lane < 32 ? dlbase + qt0 + (lane - 16) * 4 : (const float*)P.zeros);
            glds16(g, dst + 2 * NAT_BYTES + 2 * T_BYTES);
        }
    };
    if (n_it > 0) stage(0, smem);

    for (int it = 0; it < n_it; ++it) {
        const int t = t0 + it % nt;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const char* Qt = smem + (it & 1) * STAGE;
        const char* dOt = Qt + NAT_BYTES;
        const char* QTt = Qt + 2 * NAT_BYTES;
        const char* dOTt = QTt + T_BYTES;
        const char* LSt = dOTt + T_BYTES;   // [64 lse | 64 delta] fp32
        if (it + 1 < n_it) stage(it + 1, smem + ((it + 1) & 1) * STAGE);
        const int qt0 = t * 64;
        if (CAUSAL && qt0 + 63 < k0) continue;  // every query of this tile precedes this wave's keys

#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
            f32x4 s[2][NKB], dp[2][NKB];
#pragma unroll
            for (int qq = 0; qq < 2; ++qq)
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) { s[qq][kb] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[qq][kb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const int qb = 2 * qp + qq;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 qa = rd128<KROW>(Qt, qb * 16 + c, ks * 4 + g);
                    const bf16x8 da = rd128<KROW>(dOt, qb * 16 + c, ks * 4 + g);
#pragma unroll
                    for (int kb = 0; kb < NKB; ++kb) {
                        s[qq][kb] = mfma16(qa, kf[kb][ks], s[qq][kb]);
                        dp[qq][kb] = mfma16(da, vf[kb][ks], dp[qq][kb]);
                    }
                }
            }
            // lane holds S[q = qt0 + 16qb + 4g + r][key = k0 + 16kb + c]
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const int qrow = qt0 + (2 * qp + qq) * 16 + 4 * g;
                const f32x4 ls = *(const f32x4*)(LSt + (qrow - qt0) * 4);
                const f32x4 dl = *(const f32x4*)(LSt + 256 + (qrow - qt0) * 4);
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) {
                    const int kidx = k0 + kb * 16 + c;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int qidx = qrow + r;
                        const bool ok = (kidx < len) && (qidx < q_end) && (!CAUSAL || kidx <= qidx);
                        const float p = fexp2(s[qq][kb][r] * sl2 - ls[r] * LOG2E);
                        s[qq][kb][r] = ok ? p : 0.f;
                        dp[qq][kb][r] = ok ? p * (dp[qq][kb][r] - dl[r]) : 0.f;
                    }
                }
            }
            bf16x8 pf[NKB], dsf[NKB];
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) { pf[kb] = pack8(s[0][kb], s[1][kb]); dsf[kb] = pack8(dp[0][kb], dp[1][kb]); }
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const bf16x8 dota = rdT(dOTt, db * 16 + c, qp, g);
                const bf16x8 qta = rdT(QTt, db * 16 + c, qp, g);
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) {
                    dv[db][kb] = mfma16(dota, pf[kb], dv[db][kb]);
                    dk[db][kb] = mfma16(qta, dsf[kb], dk[db][kb]);
                }
            }
        }
    }
    // lane holds dV^T[d = 16db + 4g + r][key = k0 + 16kb + c]
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        const int kidx = k0 + kb * 16 + c;
        if (kidx >= S) continue;
        bf16* vp = P.dv + (rb + kidx) * P.ld_dv + h * HD + 4 * g;
        bf16* kp = P.dk + (rb + kidx) * P.ld_dk + h * HD + 4 * g;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            const f32x4 a = dv[db][kb], bb = dk[db][kb] * P.scale;
            *(bf16x4*)(vp + db * 16) = bf16x4{f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3])};
            *(bf16x4*)(kp + db * 16) = bf16x4{f2bf(bb[0]), f2bf(bb[1]), f2bf(bb[2]), f2bf(bb[3])};
        }
    }
}

// out[m, hk*HD + e] = sum_r tmp[m, (hk*nrep + r)*HD + e]  (fp32 sum of the group's per-query-head dK or dV partials)
__global__ void group_sum_heads_kernel(const bf16* tmp, long ld_tmp, bf16* out, long ld_out, long rows, int Hkv, int nrep, int hd) {
    const int per_row = Hkv * hd / 8;
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= rows * per_row) return;
    const long m = tid / per_row;
    const int col = (tid % per_row) * 8, hk = col / hd, e = col % hd;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = 0; r < nrep; ++r) {
        const bf16x8 v = *(const bf16x8*)(tmp + m * ld_tmp + (long)(hk * nrep + r) * hd + e);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += (float)v[i];
    }
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (__bf16)acc[i];
    *(bf16x8*)(out + m * ld_out + col) = o;
}

template <typename K>
int set_smem(K kern, int bytes) {
    return hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess ? 0 : -1;
}

bool aligned_ok(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

extern "C" int rv_attn_fwd_gqa(const void* q, int64_t ld_q, const void* k, int64_t ld_k, const void* vT, void* out,
                               int64_t ld_o, float* lse, const int32_t* lens, const int32_t* cu_rows, int B, int H, int H_kv,
                               int S, int S_pad, int HD, int causal, float scale, const void* zeros16, void* stream) {
    if (!q || !k || !vT || !out || !zeros16 || B <= 0 || H <= 0 || S <= 0 || H_kv <= 0 || H % H_kv) return RV_ERR_ARG;
    if ((HD != 64 && HD != 128) || (S_pad & 63) || S_pad < S) return RV_ERR_ARG;
    if ((ld_q & 7) || (ld_k & 7) || (ld_o & 3) || !aligned_ok(q) || !aligned_ok(k) || !aligned_ok(vT)) return RV_ERR_ARG;
    AttnParams P = {};
    P.q = (const bf16*)q; P.k = (const bf16*)k; P.vT = (const bf16*)vT; P.out = (bf16*)out; P.lse = lse; P.lens = lens; P.cu = cu_rows;
    P.zeros = (const bf16*)zeros16; P.ld_q = ld_q; P.ld_k = ld_k; P.ld_o = ld_o;
    P.B = B; P.H = H; P.S = S; P.S_pad = S_pad; P.scale = scale; P.Hkv = H_kv; P.nrep = H / H_kv;
    dim3 grid((S + 127) / 128, H, B);
    const int smem = 2 * (64 * HD * 2 + HD * 128);
#define LAUNCH_FWD(HD_, C_)                                                                             \
    do {                                                                                                \
        set_smem(attn_fwd_kernel<HD_, C_>, smem);                                                       \
        hipLaunchKernelGGL((attn_fwd_kernel<HD_, C_>), grid, dim3(256), smem, (hipStream_t)stream, P); \
    } while (0)
    if (HD == 128) { if (causal) LAUNCH_FWD(128, true); else LAUNCH_FWD(128, false); }
    else { if (causal) LAUNCH_FWD(64, true); else LAUNCH_FWD(64, false); }
#undef LAUNCH_FWD
    return rv_check_launch();
}

extern "C" int rv_attn_fwd(const void* q, int64_t ld_q, const void* k, int64_t ld_k, const void* vT, void* out,
                           int64_t ld_o, float* lse, const int32_t* lens, int B, int H, int S, int S_pad, int HD,
                           int causal, float scale, const void* zeros16, void* stream) {
    return rv_attn_fwd_gqa(q, ld_q, k, ld_k, vT, out, ld_o, lse, lens, nullptr, B, H, H, S, S_pad, HD, causal, scale, zeros16, stream);
}

extern "C" int rv_attn_bwd_gqa(const void* q, int64_t ld_q, const void* k, int64_t ld_k, const void* v, int64_t ld_v,
                               const void* o, int64_t ld_o, const void* dout, int64_t ld_do, const void* qT, const void* kT,
                               const void* doT, const float* lse, float* delta, void* dq, int64_t ld_dq, void* dk,
                               int64_t ld_dk, void* dv, int64_t ld_dv, const int32_t* lens, const int32_t* cu_rows, int total_rows,
                               int B, int H, int H_kv, int S, int S_pad, int HD, int causal, float scale, void* workspace, int64_t workspace_bytes,
                               const void* zeros16, void* stream) {
    if (!q || !k || !v || !o || !dout || !qT || !kT || !doT || !lse || !delta || !dq || !dk || !dv || !zeros16) return RV_ERR_ARG;
    if (H_kv <= 0 || H <= 0 || H % H_kv) return RV_ERR_ARG;
    if ((HD != 64 && HD != 128) || (S_pad & 63) || S_pad < S || B <= 0 || H <= 0 || S <= 0) return RV_ERR_ARG;
    if ((ld_q & 7) || (ld_k & 7) || (ld_v & 7) || (ld_do & 7) || (ld_o & 7) || (ld_dq & 3) || (ld_dk & 3) || (ld_dv & 3)) return RV_ERR_ARG;
    if (!aligned_ok(q) || !aligned_ok(k) || !aligned_ok(v) || !aligned_ok(dout) || !aligned_ok(qT) || !aligned_ok(kT) || !aligned_ok(doT)) return RV_ERR_ARG;
    AttnParams P = {};
    P.q = (const bf16*)q; P.k = (const bf16*)k; P.v = (const bf16*)v; P.o = (const bf16*)o; P.dout = (const bf16*)dout;
    P.qT = (const bf16*)qT; P.kT = (const bf16*)kT; P.doT = (const bf16*)doT;
    P.dq = (bf16*)dq; P.dk = (bf16*)dk; P.dv = (bf16*)dv; P.lse = (float*)lse; P.delta = delta; P.lens = lens; P.cu = cu_rows;
    P.zeros = (const bf16*)zeros16;
    P.ld_q = ld_q; P.ld_k = ld_k; P.ld_v = ld_v; P.ld_o = ld_o; P.ld_do = ld_do; P.ld_dq = ld_dq; P.ld_dk = ld_dk; P.ld_dv = ld_dv;
    P.B = B; P.H = H; P.S = S; P.S_pad = S_pad; P.scale = scale; P.Hkv = H_kv; P.nrep = H / H_kv;
    hipStream_t st = (hipStream_t)stream;
    // 8 waves per block (2 per SIMD): dQ pass = 256 query rows per block, dK/dV pass = 128 keys per block (16 per wave)
    // and one block per KEY/VALUE head (it walks the head's H / H_kv query heads)
    dim3 grid_dq((S + 255) / 256, H, B), grid_dkv((S + 127) / 128, H_kv, B);
    AttnParams PK = P;        // parameters of the dK/dV pass
    PK.kdiv = 1; PK.qrep = P.nrep;
    // few key/value heads and a long causal sequence: per-query-head blocks + a group sum balance better than per-group blocks
    const long rows_all = cu_rows ? (long)total_rows : (long)B * S;      // token rows of q/k/v (packed: cu_rows[B])
    if (cu_rows && total_rows <= 0) return RV_ERR_ARG;
    const int64_t need = 2 * (int64_t)rows_all * H * HD * 2;
    const bool expand = P.nrep > 1 && workspace && workspace_bytes >= need && (((uintptr_t)workspace) & 15) == 0 &&
                        (long)grid_dkv.x * H_kv * B < 2048;
    if (expand) {
        PK.kdiv = P.nrep; PK.qrep = 1;
        PK.dk = (bf16*)workspace; PK.dv = (bf16*)workspace + (int64_t)rows_all * H * HD;
        PK.ld_dk = PK.ld_dv = (long)H * HD;
        grid_dkv.y = H;
    }
    const int smem_dq = 2 * (2 * 64 * HD * 2 + HD * 128);
    const int smem_dkv = 2 * (2 * 64 * HD * 2 + 2 * HD * 128 + 1024);
#define LAUNCH_BWD(HD_, C_)                                                                                    \
    do {                                                                                                       \
        set_smem(attn_bwd_dq_kernel<HD_, C_, 8>, smem_dq);                                                     \
        set_smem(attn_bwd_dkv_kernel<HD_, C_, 8, 1>, smem_dkv);                                                \
        hipLaunchKernelGGL((attn_bwd_dq_kernel<HD_, C_, 8>), grid_dq, dim3(512), smem_dq, st, P);              \
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<HD_, C_, 8, 1>), grid_dkv, dim3(512), smem_dkv, st, PK);       \
    } while (0)
    if (HD == 128) { if (causal) LAUNCH_BWD(128, true); else LAUNCH_BWD(128, false); }
    else { if (causal) LAUNCH_BWD(64, true); else LAUNCH_BWD(64, false); }
#undef LAUNCH_BWD
    if (expand) {
        const long rows = rows_all, total = rows * (H_kv * HD / 8);
        hipLaunchKernelGGL(group_sum_heads_kernel, dim3((total + 255) / 256), dim3(256), 0, st, PK.dk, PK.ld_dk, P.dk, P.ld_dk, rows, H_kv, P.nrep, HD);
        hipLaunchKernelGGL(group_sum_heads_kernel, dim3((total + 255) / 256), dim3(256), 0, st, PK.dv, PK.ld_dv, P.dv, P.ld_dv, rows, H_kv, P.nrep, HD);
    }
    return rv_check_launch();
}

extern "C" int rv_attn_bwd(const void* q, int64_t ld_q, const void* k, int64_t ld_k, const void* v, int64_t ld_v,
                           const void* o, int64_t ld_o, const void* dout, int64_t ld_do, const void* qT, const void* kT,
                           const void* doT, const float* lse, float* delta, void* dq, int64_t ld_dq, void* dk,
                           int64_t ld_dk, void* dv, int64_t ld_dv, const int32_t* lens, int B, int H, int S, int S_pad,
                           int HD, int causal, float scale, const void* zeros16, void* stream) {
    return rv_attn_bwd_gqa(q, ld_q, k, ld_k, v, ld_v, o, ld_o, dout, ld_do, qT, kT, doT, lse, delta, dq, ld_dq, dk, ld_dk, dv, ld_dv,
                           lens, nullptr, 0, B, H, H, S, S_pad, HD, causal, scale, nullptr, 0, zeros16, stream);
}
