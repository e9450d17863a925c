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
// Compile-time switches that remain (measurement only): RV_ATTN_STAMPS (tools/attn_stamps.py), RV_DQ_NW=<4|8> waves of the dQ block,
// RV_ATTN_ROW_MAX_EVERY_TILE (round 3's cross-lane row maximum on every tile: A side of profiles/r04_ab_attn_fwd_lane_local_max.txt).
#include "attn_common.h"

namespace {

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
#ifdef RV_ATTN_STAMPS
// Diagnostic build only (tools/attn_stamps.py): per-wave cycle accumulators of the forward kernel's tile phases -> P.delta (unused by the
// forward), viewed as long long [blocks][4 waves][6]: wait + barrier, K reads + score MFMAs, softmax, P V, tiles, whole kernel.
#define ATT_T0() long long att_t = __builtin_readcyclecounter()
#define ATT_ACC(k) do { asm volatile("" ::: "memory"); const long long n_ = __builtin_readcyclecounter(); att_dbg[k] += n_ - att_t; att_t = n_; } while (0)
#else
#define ATT_T0() do { } while (0)
#define ATT_ACC(k) do { } while (0)
#endif


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



// ------------------------------------------------------------------------------------------------ forward (natural V)
// Softmax arithmetic per score: one v_max, one v_fma (s * scale*log2e - m) and one v_exp; the causal / key-padding mask is only
// evaluated on the tiles that touch the diagonal or the end of the sample.
template <bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_fwd_nat_kernel(AttnParams P) {
    constexpr int HD = 128, KS = 4, DB = 8, TILE = 64 * 256, STAGE = 2 * TILE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wid = wave_id(), lane = lane_id(), g = lane >> 4, c = lane & 15;
    // CAUSAL: a block owns the query-block PAIR (nq - 1 - xb, xb) of its (sample, head): every pair costs the same 2 nq + 2 key tiles (no long /
    // short blocks, no tail round: S = 704, b = 32, 32 heads is exactly 6 rounds of 512 resident blocks), and the second block's Q rows and first
    // K / V tile are fetched behind the first block's last tile, so half of the exposed prologues disappear (S = 704: a prologue was a third of a
    // block's life).  Non-causal: one query block per block, as before.
    int xb, h, b;
    block_coords<false>(xb, h, b);
    const long rb = P.cu ? (long)P.cu[b] : (long)b * P.S;
    const int S = P.cu ? P.cu[b + 1] - P.cu[b] : P.S;
    const int nq = (S + 127) >> 7;
    const bool paired = CAUSAL && P.paired;
    if ((paired ? 2 * xb : xb) >= nq) return;
    int qblk = CAUSAL ? nq - 1 - xb : xb;          // unpaired causal: the long blocks first
    const int npass = (paired && qblk != xb) ? 2 : 1;
    int q0 = qblk * 128 + wid * 32;
    const int len = (P.lens && !P.cu) ? P.lens[b] : S;
    const float sl2 = P.scale * LOG2E;

    bf16x8 qf[2][KS], qn[2][KS];
    auto load_q = [&](bf16x8 (&dst)[2][KS], int q0_) __attribute__((always_inline)) {
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
            const int row = min(q0_ + qs * 16 + c, S - 1);
            const bf16* p = P.q + (rb + row) * P.ld_q + h * HD + g * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) dst[qs][ks] = *(const bf16x8*)(p + ks * 32);
        }
    };
    load_q(qf, q0);
    int kv_end = CAUSAL ? min(len, qblk * 128 + 128) : len;
    int ntiles = (kv_end + 63) >> 6;
    int toff = 0;                 // tiles run before this pass: LDS stage parity and the read-address toggle continue across the seam
    bool more = npass > 1;        // another query block follows this one
    bool landed = false;          // this block's first K / V tile and Q rows are known to have landed (wave-uniform)
    const int hk = h / P.nrep;
    // staging: buffer resources over this sample's rows (rows >= S read as zero), loop-invariant lane offsets, scalar tile offset
    const __amdgpu_buffer_rsrc_t rsK = rows_rsrc(P.k + rb * P.ld_k, S, P.ld_k), rsV = rows_rsrc(P.v + rb * P.ld_v, S, P.ld_v);
    NatPlan<64, 4> plK, plV;
    plK.init(P.ld_k, wid, lane);
    plV.init(P.ld_v, wid, lane);
    const int kstep = (int)(64 * P.ld_k * 2), vstep = (int)(64 * P.ld_v * 2), hoff = hk * HD * 2;

    float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};
    float thr[2] = {-INFINITY, -INFINITY};        // (m + RESCALE_LAG) / sl2: the first tile always takes the rescale branch
    const float inv_sl2 = 1.f / sl2;
    f32x4 o[2][DB];
#pragma unroll
    for (int qs = 0; qs < 2; ++qs)
#pragma unroll
        for (int db = 0; db < DB; ++db) o[qs][db] = f32x4{0.f, 0.f, 0.f, 0.f};

    plK.stage(rsK, hoff, smem, wid);
    plV.stage(rsV, hoff, smem + TILE, wid);
    NatAddr ad;
    ad.init(smem + STAGE, lane);          // the first tile's toggle brings it to stage 0
#ifdef RV_ATTN_STAMPS
    long long att_dbg[6] = {0, 0, 0, 0, 0, 0}, att_dma = 0;
    const long long att_start = __builtin_readcyclecounter();
#endif

    // One key tile.  INTERIOR tiles lie wholly below the diagonal of every wave of the block and inside the sample: no mask is
    // evaluated, no wave skips, the next tile always exists -- straight-line code.  The (at most two + one partial) tiles at the
    // diagonal / end of the sample take the general form.
    auto tile = [&](int t, auto interior_tag) {
        constexpr bool INTERIOR = decltype(interior_tag)::value;
        ATT_T0();
        if (!landed) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the pair's second block waited before the first block's epilogue stores)
        landed = false;
        __syncthreads();
        ATT_ACC(0);
        ad.shift((t & 1) ? STAGE : -STAGE);
        // interior tiles issue the next tile's staging pieces (8 per wave, 60-100 issue cycles each) behind this tile's score MFMAs (K) and
        // softmax (V) instead of in front of its first fragment read: +2 % (same-box A/B)
        char* nx = smem + ((t + 1) & 1) * STAGE;
        const int u = t - toff;           // key tile of this query block
        const int kv0 = u * 64;
        constexpr bool LATE = INTERIOR;
        if (!LATE) {
            if (u + 1 < ntiles) {
                plK.stage(rsK, (u + 1) * kstep + hoff, nx, wid);
                plV.stage(rsV, (u + 1) * vstep + hoff, nx + TILE, wid);
            } else if (more) {            // last tile of the pair's first block: the second block's first K / V tile and Q rows
                plK.stage(rsK, hoff, nx, wid);
                plV.stage(rsV, hoff, nx + TILE, wid);
                load_q(qn, xb * 128 + wid * 32);
            }
        }
        if (!INTERIOR && CAUSAL && kv0 > q0 + 31) return;  // every key of this tile is in the future of this wave's rows

        f32x4 s[2][4];
#pragma unroll
        for (int qs = 0; qs < 2; ++qs)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) s[qs][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
        {   // S^T = K Q^T
            bf16x8 kq[3][KS];
            auto issue_k = [&](auto kbt, bf16x8 (&dst)[KS]) {
                sfor<KS>([&](auto ks) { rdrow_imm<decltype(kbt)::value * 4096>(dst[decltype(ks)::value], ad.row[decltype(ks)::value]); });
            };
            issue_k(std::integral_constant<int, 0>{}, kq[0]);
            issue_k(std::integral_constant<int, 1>{}, kq[1]);
            sfor<4>([&](auto kbt) {
                constexpr int kb = decltype(kbt)::value;
                if constexpr (kb + 2 < 4) issue_k(std::integral_constant<int, (kb + 2) % 4>{}, kq[(kb + 2) % 3]);
                if (kb < 2) lds_wait<KS, 2 * KS>(kq[kb % 3]);
                else if (kb == 2) lds_wait<KS, KS>(kq[kb % 3]);
                else lds_wait<KS, 0>(kq[kb % 3]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int qs = 0; qs < 2; ++qs) s[qs][kb] = mfma16(kq[kb % 3][ks], qf[qs][ks], s[qs][kb]);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        ATT_ACC(1);
        // V column fragments of the first two batches are fetched behind the softmax arithmetic
        TFrag vq[3][4];
        constexpr int NB = 2 * DB / 4;
        auto issue_v = [&](auto bit, TFrag (&dst)[4]) {
            constexpr int kp = decltype(bit)::value / (DB / 4), db0 = (decltype(bit)::value % (DB / 4)) * 4;
            sfor<4>([&](auto u) { rdcol_imm<TILE + kp * 8192>(dst[decltype(u)::value], ad.col[db0 + decltype(u)::value]); });
        };
        issue_v(std::integral_constant<int, 0>{}, vq[0]);
        issue_v(std::integral_constant<int, 1>{}, vq[1]);
#ifdef RV_ATTN_STAMPS
        { asm volatile("" ::: "memory"); const long long a_ = __builtin_readcyclecounter();
          if (LATE) plK.stage(rsK, (u + 1) * kstep + hoff, nx, wid);
          asm volatile("" ::: "memory"); att_dma += __builtin_readcyclecounter() - a_; }
#else
        if (LATE) plK.stage(rsK, (u + 1) * kstep + hoff, nx, wid);
#endif
        // lane holds S^T[key = kv0 + 16kb + 4g + r][q = q0 + 16qs + c]
        const bool edge = !INTERIOR && ((kv0 + 64 > len) || (CAUSAL && kv0 + 63 > q0));     // wave-uniform
        if (edge) {
#pragma unroll
            for (int qs = 0; qs < 2; ++qs) {
                const int qidx = q0 + qs * 16 + c;
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int kidx = kv0 + kb * 16 + 4 * g + r;
                        const bool ok = (kidx < len) && (!CAUSAL || kidx <= qidx);
                        s[qs][kb][r] = ok ? s[qs][kb][r] : -INFINITY;
                    }
            }
        }
        // row maxima: two v_max3 chains per 16-query block, the four chains interleaved (the asm statements keep their order)
        float mxa[2], mxb[2];
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) { mxa[qs] = max3_asm(s[qs][0][0], s[qs][0][1], s[qs][0][2]); mxb[qs] = max3_asm(s[qs][0][3], s[qs][1][0], s[qs][1][1]); }
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) { mxa[qs] = max3_asm(mxa[qs], s[qs][1][2], s[qs][1][3]); mxb[qs] = max3_asm(mxb[qs], s[qs][2][0], s[qs][2][1]); }
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) { mxa[qs] = max3_asm(mxa[qs], s[qs][2][2], s[qs][2][3]); mxb[qs] = max3_asm(mxb[qs], s[qs][3][0], s[qs][3][1]); }
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) mxa[qs] = max3_asm(mxa[qs], s[qs][3][2], s[qs][3][3]);
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) mxa[qs] = max3_asm(mxa[qs], mxb[qs], mxb[qs]);
        // Running maximum kept as an INTEGER of the exp2 domain (ceil of the scaled row maximum): every rescale factor is then an exact
        // power of two, so WHEN a row is rescaled does not change a bit of the result -- and it is deferred until some row of the wave has
        // outgrown its stored maximum by more than 2^RESCALE_LAG (p then stays <= 2^RESCALE_LAG: nothing for fp32 sums or bf16 P).  After
        // the first tile that practically never happens, and the 64 accumulator multiplies + 2 exp per tile (a third of the tile's VALU
        // issue cycles; the loop was VALU-bound: 182 VALU for 64 MFMA per tile) leave the loop.
        constexpr float RESCALE_LAG = 8.f;
#ifdef RV_ATTN_ROW_MAX_EVERY_TILE     // A/B switch (measurement only): round 3's cross-lane row maximum + candidate on every tile
        float cand[2];
        bool need = false;
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
            float mx = mxa[qs];
            float sh = __shfl_xor(mx, 16, 64);
            mx = max3_asm(mx, sh, sh);
            sh = __shfl_xor(mx, 32, 64);
            mx = max3_asm(mx, sh, sh);
            cand[qs] = ceilf(mx * sl2);                       // scale > 0: the maximum commutes with the scaling
            need |= cand[qs] > m[qs] + RESCALE_LAG;           // m = -inf before the first tile
        }
        if (__builtin_amdgcn_ballot_w64(need) != 0) {         // wave-uniform
#pragma unroll
            for (int qs = 0; qs < 2; ++qs) {
                const float mnew = max3_asm(m[qs], cand[qs], cand[qs]);
                const float alpha = fexp2(m[qs] - mnew);      // 2^(integer) or 0
                l[qs] *= alpha;
                m[qs] = mnew;
#pragma unroll
                for (int db = 0; db < DB; ++db) o[qs][db] *= alpha;
            }
        }
#else
        // A row needs a new maximum only when one of its scores outgrows the stored one by more than 2^RESCALE_LAG -- a test every lane
        // makes on its OWN 16 scores against thr = (m + RESCALE_LAG) / sl2 (raw-score units).  The cross-lane row maximum (two LDS
        // permutes whose wait also drained the V fragment reads in flight), the candidate and the rescale only run in the rare branch.
        const bool need = (mxa[0] > thr[0]) | (mxa[1] > thr[1]);
        if (__builtin_amdgcn_ballot_w64(need) != 0) {         // wave-uniform
#pragma unroll
            for (int qs = 0; qs < 2; ++qs) {
                float mx = mxa[qs];
                float sh = __shfl_xor(mx, 16, 64);
                mx = max3_asm(mx, sh, sh);
                sh = __shfl_xor(mx, 32, 64);
                mx = max3_asm(mx, sh, sh);
                const float cand = ceilf(mx * sl2);               // scale > 0: the maximum commutes with the scaling
                const float mnew = max3_asm(m[qs], cand, cand);
                const float alpha = fexp2(m[qs] - mnew);          // 2^(integer) or 0 (m = -inf before the first tile)
                l[qs] *= alpha;
                m[qs] = mnew;
                thr[qs] = (mnew + RESCALE_LAG) * inv_sl2;
#pragma unroll
                for (int db = 0; db < DB; ++db) o[qs][db] *= alpha;
            }
        }
#endif
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
            float rs = 0.f;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = fexp2(__builtin_fmaf(s[qs][kb][r], sl2, -m[qs]));
                    s[qs][kb][r] = p;
                    rs += p;
                }
            l[qs] += rs;
        }
#ifdef RV_ATTN_STAMPS
        { asm volatile("" ::: "memory"); const long long a_ = __builtin_readcyclecounter();
          if (LATE) plV.stage(rsV, (u + 1) * vstep + hoff, nx + TILE, wid);
          asm volatile("" ::: "memory"); att_dma += __builtin_readcyclecounter() - a_; }
#else
        if (LATE) plV.stage(rsV, (u + 1) * vstep + hoff, nx + TILE, wid);
#endif
        ATT_ACC(2);
        // O^T[d][q] += V^T[d][key] P^T[key][q]
        {
            bf16x8 pf[2][2];
#pragma unroll
            for (int kp = 0; kp < 2; ++kp)
#pragma unroll
                for (int qs = 0; qs < 2; ++qs) pf[kp][qs] = pack8(s[qs][2 * kp], s[qs][2 * kp + 1]);
            sfor<NB>([&](auto bit) {
                constexpr int bi = decltype(bit)::value;
                if constexpr (bi + 2 < NB) issue_v(std::integral_constant<int, (bi + 2) % NB>{}, vq[(bi + 2) % 3]);
                if (bi + 2 < NB) tf_wait4<15>(vq[bi % 3]);           // 16 younger reads in flight (the 4-bit counter saturates at 15)
                else if (bi + 1 < NB) tf_wait4<8>(vq[bi % 3]);
                else tf_wait4<0>(vq[bi % 3]);
                __builtin_amdgcn_sched_barrier(0);
                constexpr int kp = bi / (DB / 4), db0 = (bi % (DB / 4)) * 4;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bf16x8 vf = tf_get(vq[bi % 3][u]);
#pragma unroll
                    for (int qs = 0; qs < 2; ++qs) o[qs][db0 + u] = mfma16(vf, pf[kp][qs], o[qs][db0 + u]);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        }
#ifdef RV_ATTN_STAMPS
        ATT_ACC(3);
        att_dbg[4] += 1;
#endif
    };
    for (int pass = 0;; ++pass) {
        // interior tiles: below the block's first query row (causal) and wholly inside the sample; never the last tile
        const int n_int = min(CAUSAL ? min(qblk * 2, len >> 6) : (len >> 6), ntiles - 1);
        int t = toff;
        for (; t < toff + n_int; ++t) tile(t, std::true_type{});
        for (; t < toff + ntiles; ++t) tile(t, std::false_type{});
        // the second block's first tile and Q rows were issued a whole key tile ago: retire them HERE, so that its first tile does not wait for
        // the output stores below (vmcnt counts them too)
        if (more) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); landed = true; }
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
        if (!more) break;
        // the pair's second (short) query block: its Q rows and first K / V tile are already in flight
        more = false;
        toff += ntiles;
        qblk = xb;
        q0 = qblk * 128 + wid * 32;
        kv_end = min(len, qblk * 128 + 128);
        ntiles = (kv_end + 63) >> 6;
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
            m[qs] = -INFINITY; l[qs] = 0.f; thr[qs] = -INFINITY;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) qf[qs][ks] = qn[qs][ks];
#pragma unroll
            for (int db = 0; db < DB; ++db) o[qs][db] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
#ifdef RV_ATTN_STAMPS
    if (P.delta && lane == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        att_dbg[5] = __builtin_readcyclecounter() - att_start;
        att_dbg[4] |= att_dma << 20;          // tiles (low 20 bits) | cycles spent issuing the 8 LDS-DMA pieces of the next tile
        const long blk = blockIdx.x + (long)gridDim.x * (blockIdx.y + (long)gridDim.y * blockIdx.z);
        long long* o_ = (long long*)P.delta + (blk * 4 + wid) * 6;
        for (int i = 0; i < 6; ++i) o_[i] = att_dbg[i];
        if (wid == 0) {   // where and when the block ran: [start, end, HW_ID | XCC_ID << 32] behind the per-wave records
            const long nblk = (long)gridDim.x * gridDim.y * gridDim.z;
            long long* w_ = (long long*)P.delta + nblk * 4 * 6 + blk * 3;
            const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
            w_[0] = att_start; w_[1] = att_start + att_dbg[5]; w_[2] = (long long)hw | ((long long)xcc << 32);
        }
    }
#endif
}


// ------------------------------------------------------------------------------------------------ backward dQ (natural K, V)
// 8 waves x 32 query rows per block; per 64-key tile: S^T = K Q^T and dP^T = V dO^T from row reads, dQ^T += K^T dS^T from column
// reads of the SAME K image (no K^T copy).  p = exp2(s * scale*log2e - lse*log2e) is one fma + one exp per score.
template <bool CAUSAL, int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn_bwd_dq_nat_kernel(AttnParams P) {
    constexpr int HD = 128, KS = 4, DB = 8, TILE = 64 * 256, STAGE = 2 * TILE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wid = wave_id(), lane = lane_id(), g = lane >> 4, c = lane & 15;
    int qblk, h, b;
    block_coords<CAUSAL>(qblk, h, b);
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
    const __amdgpu_buffer_rsrc_t rsK = rows_rsrc(P.k + rb * P.ld_k, S, P.ld_k), rsV = rows_rsrc(P.v + rb * P.ld_v, S, P.ld_v);
    NatPlan<64, NW> plK, plV;
    plK.init(P.ld_k, wid, lane);
    plV.init(P.ld_v, wid, lane);
    const int kstep = (int)(64 * P.ld_k * 2), vstep = (int)(64 * P.ld_v * 2), hoff = hk * HD * 2;

    f32x4 dq[2][DB];
#pragma unroll
    for (int qs = 0; qs < 2; ++qs)
#pragma unroll
        for (int db = 0; db < DB; ++db) dq[qs][db] = f32x4{0.f, 0.f, 0.f, 0.f};

    plK.stage(rsK, hoff, smem, wid);
    plV.stage(rsV, hoff, smem + TILE, wid);
    NatAddr ad;
    ad.init(smem + STAGE, lane);          // the first tile's toggle brings it to stage 0

    // INTERIOR tiles (wholly below the block's first query row and inside the sample): no mask, no skipping wave, the next tile
    // always exists -- straight-line code; the tiles at the diagonal / the end of the sample take the general form.
    auto tile = [&](int t, auto interior_tag) {
        constexpr bool INTERIOR = decltype(interior_tag)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        ad.shift((t & 1) ? STAGE : -STAGE);
        if (INTERIOR || t + 1 < ntiles) {   // (issuing these behind the tile's first MFMA phase, as the forward does, gains < 0.5 % here)
            char* nx = smem + ((t + 1) & 1) * STAGE;
            plK.stage(rsK, (t + 1) * kstep + hoff, nx, wid);
            plV.stage(rsV, (t + 1) * vstep + hoff, nx + TILE, wid);
        }
        const int kv0 = t * 64;
        if (!INTERIOR && CAUSAL && kv0 > q0 + 31) return;

        f32x4 s[2][4], dp[2][4];
#pragma unroll
        for (int qs = 0; qs < 2; ++qs)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) { s[qs][kb] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[qs][kb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        {   // row fragments of K and V in batches of two k-steps: batch i + 1 is in flight while batch i's MFMAs run
            bf16x8 kq[2][2], vq[2][2];
            auto issue = [&](auto bit, bf16x8 (&kd)[2], bf16x8 (&vd)[2]) {
                constexpr int kb = decltype(bit)::value >> 1, ks0 = (decltype(bit)::value & 1) * 2;
                sfor<2>([&](auto u) {
                    rdrow_imm<kb * 4096>(kd[decltype(u)::value], ad.row[ks0 + decltype(u)::value]);
                    rdrow_imm<TILE + kb * 4096>(vd[decltype(u)::value], ad.row[ks0 + decltype(u)::value]);
                });
            };
            issue(std::integral_constant<int, 0>{}, kq[0], vq[0]);
            sfor<8>([&](auto bit) {
                constexpr int bi = decltype(bit)::value;
                if constexpr (bi + 1 < 8) issue(std::integral_constant<int, (bi + 1) % 8>{}, kq[(bi + 1) & 1], vq[(bi + 1) & 1]);
                if (bi + 1 < 8) lds_wait4<4>(kq[bi & 1][0], kq[bi & 1][1], vq[bi & 1][0], vq[bi & 1][1]);
                else lds_wait4<0>(kq[bi & 1][0], kq[bi & 1][1], vq[bi & 1][0], vq[bi & 1][1]);
                __builtin_amdgcn_sched_barrier(0);
                constexpr int kb = bi >> 1, ks0 = (bi & 1) * 2;
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int qs = 0; qs < 2; ++qs) {
                        s[qs][kb] = mfma16(kq[bi & 1][u], qf[qs][ks0 + u], s[qs][kb]);
                        dp[qs][kb] = mfma16(vq[bi & 1][u], dof[qs][ks0 + u], dp[qs][kb]);
                    }
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        // the K column fragments of the first dQ batch are fetched behind the elementwise part
        TFrag kc[4];
        auto issue_c = [&](auto bit) {
            constexpr int kp = decltype(bit)::value / (DB / 4), db0 = (decltype(bit)::value % (DB / 4)) * 4;
            sfor<4>([&](auto u) { rdcol_imm<kp * 8192>(kc[decltype(u)::value], ad.col[db0 + decltype(u)::value]); });
        };
        issue_c(std::integral_constant<int, 0>{});
        const bool edge = !INTERIOR && ((kv0 + 64 > len) || (CAUSAL && kv0 + 63 > q0));
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
            const int qidx = q0 + qs * 16 + c;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = fexp2(__builtin_fmaf(s[qs][kb][r], sl2, -lse2[qs]));
                    float ds = p * (dp[qs][kb][r] - dl[qs]);
                    if (edge) {
                        const int kidx = kv0 + kb * 16 + 4 * g + r;
                        ds = ((kidx < len) && (!CAUSAL || kidx <= qidx)) ? ds : 0.f;
                    }
                    s[qs][kb][r] = ds;
                }
        }
        constexpr int NB = 2 * DB / 4;
        bf16x8 dsf[2][2];
#pragma unroll
        for (int kp = 0; kp < 2; ++kp)
#pragma unroll
            for (int qs = 0; qs < 2; ++qs) dsf[kp][qs] = pack8(s[qs][2 * kp], s[qs][2 * kp + 1]);
        sfor<NB>([&](auto bit) {
            constexpr int bi = decltype(bit)::value;
            tf_wait4<0>(kc);
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 ktf[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) ktf[u] = tf_get(kc[u]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (bi + 1 < NB) issue_c(std::integral_constant<int, (bi + 1) % NB>{});          // the next batch lands behind this batch's MFMAs
            constexpr int kp = bi / (DB / 4), db0 = (bi % (DB / 4)) * 4;
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int qs = 0; qs < 2; ++qs) dq[qs][db0 + u] = mfma16(ktf[u], dsf[kp][qs], dq[qs][db0 + u]);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    const int n_int = min(CAUSAL ? min(qblk * (NW / 2), len >> 6) : (len >> 6), ntiles - 1);
    int t = 0;
    for (; t < n_int; ++t) tile(t, std::true_type{});
    for (; t < ntiles; ++t) tile(t, std::false_type{});
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
        const int qidx = q0 + qs * 16 + c;
        if (qidx >= S) continue;
        bf16* op = P.dq + (rb + qidx) * P.ld_dq + h * HD + 4 * g;
        f32x4 v[DB];
#pragma unroll
        for (int db = 0; db < DB; ++db) v[db] = dq[qs][db] * P.scale;
        if (P.rope_cs) unrope<DB>(v, P.rope_cs + (long)(P.rope_pos ? P.rope_pos[rb + qidx] : qidx) * HD, g);
#pragma unroll
        for (int db = 0; db < DB; ++db) *(bf16x4*)(op + db * 16) = bf16x4{f2bf(v[db][0]), f2bf(v[db][1]), f2bf(v[db][2]), f2bf(v[db][3])};
    }
}

// ------------------------------------------------------------------------------------------------ backward dK, dV (natural Q, dO)
// 8 waves x 16 keys per block; per 64-query tile: S = Q K^T and dP = dO V^T from row reads of the Q / dO images, dV^T += dO^T P and
// dK^T += Q^T dS from column reads of the SAME images (no Q^T / dO^T copies: half the staging traffic of the transposed-copy form).
template <bool CAUSAL, int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn_bwd_dkv_nat_kernel(AttnParams P) {
    constexpr int HD = 128, KS = 4, DB = 8, TILE = 64 * 256, STAGE = 2 * TILE + 1024;   // Q | dO | lse, delta
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wid = wave_id(), lane = lane_id(), g = lane >> 4, c = lane & 15;
    int kblk, h, b;
    block_coords<false>(kblk, h, b);          // low key blocks (the longest under a causal mask) are already first
    constexpr int KPB = NW * 16;   // keys per block
    const long rb = P.cu ? (long)P.cu[b] : (long)b * P.S;
    const int S = P.cu ? P.cu[b + 1] - P.cu[b] : P.S, k0 = kblk * KPB + wid * 16;
    if (kblk * KPB >= S) return;
    const int len = (P.lens && !P.cu) ? P.lens[b] : S;
    const float sl2 = P.scale * LOG2E;

    bf16x8 kf[KS], vf[KS];
    {
        const int row = min(k0 + c, S - 1);
        const bf16* kp = P.k + (rb + row) * P.ld_k + (h / P.kdiv) * HD + g * 8;
        const bf16* vp = P.v + (rb + row) * P.ld_v + (h / P.kdiv) * HD + g * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { kf[ks] = *(const bf16x8*)(kp + ks * 32); vf[ks] = *(const bf16x8*)(vp + ks * 32); }
    }
    const int q_end = len;  // query rows >= len carry zero dO
    const int t0 = CAUSAL ? (kblk * KPB) >> 6 : 0;
    const int t1 = (q_end + 63) >> 6;
    const int nt = max(t1 - t0, 0);
    const int n_it = nt * P.qrep;

    f32x4 dv[DB], dk[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) { dv[db] = f32x4{0.f, 0.f, 0.f, 0.f}; dk[db] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    const __amdgpu_buffer_rsrc_t rsQ = rows_rsrc(P.q + rb * P.ld_q, S, P.ld_q), rsO = rows_rsrc(P.dout + rb * P.ld_do, S, P.ld_do);
    NatPlan<64, NW> plQ, plO;
    plQ.init(P.ld_q, wid, lane);
    plO.init(P.ld_do, wid, lane);
    auto stage = [&](int it, char* dst) {
        const int hq = h * P.qrep + it / nt;
        const int qt0 = (t0 + it % nt) * 64;
        const float* lsebase = P.lse + (long)(b * P.H + hq) * P.S_pad;
        const float* dlbase = P.delta + (long)(b * P.H + hq) * P.S_pad;
        plQ.stage(rsQ, (int)((long)qt0 * P.ld_q * 2) + hq * HD * 2, dst, wid);
        plO.stage(rsO, (int)((long)qt0 * P.ld_do * 2) + hq * HD * 2, dst + TILE, wid);
        if (wid == 0) {   // lse / delta of the tile's 64 query rows ride the same LDS-DMA stream
            const float* gp = lane < 16 ? lsebase + qt0 + lane * 4 : (lane < 32 ? dlbase + qt0 + (lane - 16) * 4 : (const float*)P.zeros);
            glds16(gp, dst + 2 * TILE);
        }
    };
    if (n_it > 0) stage(0, smem);
    NatAddr ad;
    ad.init(smem + STAGE, lane);          // the first iteration's toggle brings it to stage 0

    // One 64-query tile.  EDGE tiles (the diagonal, the end of the sample, a key block that crosses the end of the sample) evaluate
    // the mask and may be skipped by a wave; all others run mask-free straight-line code (three sequential loops per query head
    // below: two copies of the body under one branch made hipcc keep both register sets live -> scratch).
    auto iter = [&](int it, int j, auto edge_tag) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        ad.shift((it & 1) ? STAGE : -STAGE);
        const char* LSt = smem + (it & 1) * STAGE + 2 * TILE;   // [64 lse | 64 delta] fp32 behind the Q | dO images
        if (it + 1 < n_it) stage(it + 1, smem + ((it + 1) & 1) * STAGE);
        const int qt0 = (t0 + j) * 64;
        if (EDGE && CAUSAL && qt0 + 63 < k0) return;  // every query of this tile precedes this wave's keys
        {
            sfor<2>([&](auto qpt) {
                constexpr int qp = decltype(qpt)::value;
                if (EDGE && CAUSAL && qt0 + qp * 32 + 31 < k0) return;   // diagonal tile: this 32-query step precedes all of the wave's keys
                // column fragments of this 32-query step (for dV, dK) are fetched first: they are consumed last
                TFrag dc[2][4], qc[2][4];
                auto issue_c = [&](auto bit, TFrag (&dd)[4], TFrag (&qd)[4]) {
                    constexpr int bi = decltype(bit)::value;
                    sfor<4>([&](auto u) {
                        rdcol_imm<TILE + qp * 8192>(dd[decltype(u)::value], ad.col[bi * 4 + decltype(u)::value]);
                        rdcol_imm<qp * 8192>(qd[decltype(u)::value], ad.col[bi * 4 + decltype(u)::value]);
                    });
                };
                f32x4 s[2], dp[2];
                s[0] = s[1] = dp[0] = dp[1] = f32x4{0.f, 0.f, 0.f, 0.f};
                bf16x8 qa[2][KS], da[2][KS];
                sfor<2>([&](auto qq) {
                    sfor<KS>([&](auto ks) {
                        rdrow_imm<(2 * qp + decltype(qq)::value) * 4096>(qa[decltype(qq)::value][decltype(ks)::value], ad.row[decltype(ks)::value]);
                        rdrow_imm<TILE + (2 * qp + decltype(qq)::value) * 4096>(da[decltype(qq)::value][decltype(ks)::value], ad.row[decltype(ks)::value]);
                    });
                });
                lds_wait8<2 * KS>(qa[0], da[0]);       // the 8 reads of the second 16-query block stay in flight
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) { s[0] = mfma16(qa[0][ks], kf[ks], s[0]); dp[0] = mfma16(da[0][ks], vf[ks], dp[0]); }
                __builtin_amdgcn_sched_barrier(0);
                issue_c(std::integral_constant<int, 0>{}, dc[0], qc[0]);        // 16 column reads, consumed after the elementwise part
                lds_wait8<15>(qa[1], da[1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) { s[1] = mfma16(qa[1][ks], kf[ks], s[1]); dp[1] = mfma16(da[1][ks], vf[ks], dp[1]); }
                __builtin_amdgcn_sched_barrier(0);
                // lane holds S[q = qt0 + 16qb + 4g + r][key = k0 + c]
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    const int qrow = qt0 + (2 * qp + qq) * 16 + 4 * g;
                    const f32x4 ls = *(const f32x4*)(LSt + (qrow - qt0) * 4);
                    const f32x4 dl = *(const f32x4*)(LSt + 256 + (qrow - qt0) * 4);
                    const int kidx = k0 + c;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float p = fexp2(__builtin_fmaf(s[qq][r], sl2, -ls[r] * LOG2E));
                        if (EDGE) {
                            const int qidx = qrow + r;
                            p = ((kidx < len) && (qidx < q_end) && (!CAUSAL || kidx <= qidx)) ? p : 0.f;
                        }
                        s[qq][r] = p;
                        dp[qq][r] = p * (dp[qq][r] - dl[r]);
                    }
                }
                const bf16x8 pf = pack8(s[0], s[1]), dsf = pack8(dp[0], dp[1]);
                sfor<2>([&](auto bit) {
                    constexpr int bi = decltype(bit)::value;
                    if constexpr (bi == 0) { issue_c(std::integral_constant<int, 1>{}, dc[1], qc[1]); tf_wait8<15>(dc[0], qc[0]); }
                    else tf_wait8<0>(dc[1], qc[1]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        dv[bi * 4 + u] = mfma16(tf_get(dc[bi][u]), pf, dv[bi * 4 + u]);
                        dk[bi * 4 + u] = mfma16(tf_get(qc[bi][u]), dsf, dk[bi * 4 + u]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
        }
    };
    {
        const bool key_partial = kblk * KPB + KPB > len;
        const int ne0 = key_partial ? nt : (CAUSAL ? min(nt, (KPB + 63) / 64) : 0);          // leading edge tiles (the diagonal)
        const int ni = max(ne0, min(nt, (q_end >> 6) - t0));                                   // [ne0, ni): interior; [ni, nt): trailing edge
        int it = 0;
        for (int rep = 0; rep < P.qrep; ++rep) {
            for (int j = 0; j < ne0; ++j, ++it) iter(it, j, std::true_type{});
            for (int j = ne0; j < ni; ++j, ++it) iter(it, j, std::false_type{});
            for (int j = ni; j < nt; ++j, ++it) iter(it, j, std::true_type{});
        }
    }
    // lane holds dV^T[d = 16db + 4g + r][key = k0 + c]
    {
        const int kidx = k0 + c;
        if (kidx < S) {
            bf16* vp = P.dv + (rb + kidx) * P.ld_dv + h * HD + 4 * g;
            bf16* kp = P.dk + (rb + kidx) * P.ld_dk + h * HD + 4 * g;
            f32x4 kk[DB];
#pragma unroll
            for (int db = 0; db < DB; ++db) kk[db] = dk[db] * P.scale;
            if (P.rope_cs && P.rope_dk) unrope<DB>(kk, P.rope_cs + (long)(P.rope_pos ? P.rope_pos[rb + kidx] : kidx) * HD, g);
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const f32x4 a = dv[db];
                *(bf16x4*)(vp + db * 16) = bf16x4{f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3])};
                *(bf16x4*)(kp + db * 16) = bf16x4{f2bf(kk[db][0]), f2bf(kk[db][1]), f2bf(kk[db][2]), f2bf(kk[db][3])};
            }
        }
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
        f32x4 v[DB];
#pragma unroll
        for (int db = 0; db < DB; ++db) v[db] = dq[qs][db] * P.scale;
        if (P.rope_cs) unrope<DB>(v, P.rope_cs + (long)(P.rope_pos ? P.rope_pos[rb + qidx] : qidx) * HD, g);
#pragma unroll
        for (int db = 0; db < DB; ++db) *(bf16x4*)(op + db * 16) = bf16x4{f2bf(v[db][0]), f2bf(v[db][1]), f2bf(v[db][2]), f2bf(v[db][3])};
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
        f32x4 kk[DB];
#pragma unroll
        for (int db = 0; db < DB; ++db) kk[db] = dk[db][kb] * P.scale;
        if (P.rope_cs && P.rope_dk) unrope<DB>(kk, P.rope_cs + (long)(P.rope_pos ? P.rope_pos[rb + kidx] : kidx) * HD, g);
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            const f32x4 a = dv[db][kb];
            *(bf16x4*)(vp + db * 16) = bf16x4{f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3])};
            *(bf16x4*)(kp + db * 16) = bf16x4{f2bf(kk[db][0]), f2bf(kk[db][1]), f2bf(kk[db][2]), f2bf(kk[db][3])};
        }
    }
}

// out[m, hk*HD + e] = sum_r tmp[m, (hk*nrep + r)*HD + e]  (fp32 sum of the group's per-query-head dK or dV partials); with `cs` the
// summed dK row is un-rotated too (a thread owns dimensions e .. e+7 and their partners e + hd/2 ..).
__global__ void group_sum_heads_kernel(const bf16* tmp, long ld_tmp, bf16* out, long ld_out, long rows, int Hkv, int nrep, int hd,
                                       const float* cs, const int* rope_pos, const int* cu, int S) {
    const int per_row = Hkv * hd / 16;
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= rows * per_row) return;
    const long m = tid / per_row;
    const int t = tid % per_row, half = hd / 2, hk = t / (half / 8), e = (t % (half / 8)) * 8;
    float lo[8] = {0, 0, 0, 0, 0, 0, 0, 0}, hi[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = 0; r < nrep; ++r) {
        const bf16* p = tmp + m * ld_tmp + (long)(hk * nrep + r) * hd + e;
        const bf16x8 a = *(const bf16x8*)p, b = *(const bf16x8*)(p + half);
#pragma unroll
        for (int i = 0; i < 8; ++i) { lo[i] += (float)a[i]; hi[i] += (float)b[i]; }
    }
    if (cs) {
        const int pos = rope_pos ? rope_pos[m] : (int)(m % S);
        const float* c = cs + ((long)pos * half + e) * 2;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float a = bf2f(f2bf(lo[i])), b = bf2f(f2bf(hi[i]));
            lo[i] = a * c[2 * i] + b * c[2 * i + 1];
            hi[i] = b * c[2 * i] - a * c[2 * i + 1];
        }
    }
    bf16x8 o0, o1;
#pragma unroll
    for (int i = 0; i < 8; ++i) { o0[i] = (__bf16)lo[i]; o1[i] = (__bf16)hi[i]; }
    *(bf16x8*)(out + m * ld_out + (long)hk * hd + e) = o0;
    *(bf16x8*)(out + m * ld_out + (long)hk * hd + e + half) = o1;
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

// explicit instantiations (hipcc left the host stub of one internal-linkage instantiation undefined when it was only named inside a
// launch macro next to a generic lambda)
namespace {
template __global__ void attn_fwd_nat_kernel<true>(AttnParams);
template __global__ void attn_fwd_nat_kernel<false>(AttnParams);
#ifndef RV_DQ_NW
#define RV_DQ_NW 4
#endif
template __global__ void attn_bwd_dq_nat_kernel<true, RV_DQ_NW>(AttnParams);
template __global__ void attn_bwd_dq_nat_kernel<false, RV_DQ_NW>(AttnParams);
template __global__ void attn_bwd_dkv_nat_kernel<true, 4>(AttnParams);
template __global__ void attn_bwd_dkv_nat_kernel<false, 4>(AttnParams);
}  // namespace

// one-wave-per-SIMD kernels (attention_w64.hip)
int rv_attn_fwd_w64_launch(const AttnParams& P, int causal, hipStream_t st);
// Kernel-family selection of the head_dim 128 natural-layout entry points, process-wide (measurement hook: same-process A/B and tests):
// 0 = default (the two-waves-per-SIMD kernels of this file: faster on every shape measured, profiles/r04_ab_attn_w64_*), 1 = the same,
// explicitly, 2 = the one-wave-per-SIMD forward of attention_w64.hip.
static int g_attn_family = 0;
// compute units of the current device (one process drives one GPU: read once)
static int g_attn_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
        else cus = 256;
    }
    return cus;
}
extern "C" int rv_attn_select_kernel(int which) {
    if (which < 0 || which > 2) return RV_ERR_ARG;
    g_attn_family = which;
    return RV_OK;
}

// Causal forward: one block per PAIR of query blocks (attn_fwd_nat_kernel) -- all blocks then cost the same, so the pairing is only taken when the
// last round of resident blocks (2 per CU) is nearly full; otherwise single query blocks, the long ones first, pack the CUs better.
static int attn_fwd_pairs(int B, int H, int S, int causal) {
    const int nq = (S + 127) / 128;
    const long pairs = (long)((nq + 1) / 2) * H * B;
    const long slots = 2L * g_attn_cus();
    const long rounds = (pairs + slots - 1) / slots;
    return causal && nq >= 2 && (double)(rounds * slots - pairs) <= 0.06 * (double)(rounds * slots);
}
// which of the two forms rv_attn_fwd_nat takes for a shape on this device (1 = pairs): lets a test assert that it covers both
extern "C" int rv_attn_fwd_nat_pairs(int B, int H, int S, int causal) { return attn_fwd_pairs(B, H, S, causal); }

#if defined(RV_ATTN_STAMPS) || defined(RV_W64_STAMPS)
static float* g_attn_stamp = nullptr;
extern "C" int rv_debug_set_attn_stamp_buffer(void* p) { g_attn_stamp = (float*)p; return 0; }
#endif
extern "C" int rv_attn_fwd_nat(const void* q, int64_t ld_q, const void* k, int64_t ld_k, const void* v, int64_t ld_v, void* out, int64_t ld_o,
                               float* lse, const int32_t* lens, const int32_t* cu_rows, int B, int H, int H_kv, int S, int S_pad, int HD,
                               int causal, float scale, const void* zeros16, void* stream) {
    if (!q || !k || !v || !out || !zeros16 || B <= 0 || H <= 0 || S <= 0 || H_kv <= 0 || H % H_kv) return RV_ERR_ARG;
    if (HD != 128 || (S_pad & 63) || S_pad < S) return RV_ERR_ARG;
    if ((ld_q & 7) || (ld_k & 7) || (ld_v & 7) || (ld_o & 3) || !aligned_ok(q) || !aligned_ok(k) || !aligned_ok(v)) return RV_ERR_ARG;
    AttnParams P = {};
    P.q = (const bf16*)q; P.k = (const bf16*)k; P.v = (const bf16*)v; P.out = (bf16*)out; P.lse = lse; P.lens = lens; P.cu = cu_rows;
    P.zeros = (const bf16*)zeros16; P.ld_q = ld_q; P.ld_k = ld_k; P.ld_v = ld_v; P.ld_o = ld_o;
    P.B = B; P.H = H; P.S = S; P.S_pad = S_pad; P.scale = scale; P.Hkv = H_kv; P.nrep = H / H_kv;
#if defined(RV_ATTN_STAMPS) || defined(RV_W64_STAMPS)
    P.delta = g_attn_stamp;
#endif
    if (g_attn_family == 2) return rv_attn_fwd_w64_launch(P, causal, (hipStream_t)stream);
    const int nq = (S + 127) / 128;
    P.paired = attn_fwd_pairs(B, H, S, causal);
    dim3 grid(P.paired ? (nq + 1) / 2 : nq, H, B);
    const int smem = 2 * 2 * 64 * 256;
    if (causal) { set_smem(attn_fwd_nat_kernel<true>, smem); hipLaunchKernelGGL(attn_fwd_nat_kernel<true>, grid, dim3(256), smem, (hipStream_t)stream, P); }
    else { set_smem(attn_fwd_nat_kernel<false>, smem); hipLaunchKernelGGL(attn_fwd_nat_kernel<false>, grid, dim3(256), smem, (hipStream_t)stream, P); }
    return rv_check_launch();
}

extern "C" int rv_attn_fwd(const void* q, int64_t ld_q, const void* k, int64_t ld_k, const void* vT, void* out,
                           int64_t ld_o, float* lse, const int32_t* lens, int B, int H, int S, int S_pad, int HD,
                           int causal, float scale, const void* zeros16, void* stream) {
    return rv_attn_fwd_gqa(q, ld_q, k, ld_k, vT, out, ld_o, lse, lens, nullptr, B, H, H, S, S_pad, HD, causal, scale, zeros16, stream);
}

extern "C" int rv_attn_bwd_gqa_rope(const void* q, int64_t ld_q, const void* k, int64_t ld_k, const void* v, int64_t ld_v,
                               const void* o, int64_t ld_o, const void* dout, int64_t ld_do, const void* qT, const void* kT,
                               const void* doT, const float* lse, float* delta, void* dq, int64_t ld_dq, void* dk,
                               int64_t ld_dk, void* dv, int64_t ld_dv, const int32_t* lens, const int32_t* cu_rows, int total_rows,
                               int B, int H, int H_kv, int S, int S_pad, int HD, int causal, float scale, void* workspace, int64_t workspace_bytes,
                               const float* rope_cos_sin, const int32_t* rope_positions, const void* zeros16, void* stream) {
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
    if (rope_cos_sin && ((((uintptr_t)rope_cos_sin) & 15) || (cu_rows && !rope_positions))) return RV_ERR_ARG;
    P.rope_cs = rope_cos_sin; P.rope_pos = rope_positions; P.rope_dk = 1;
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
        PK.kdiv = P.nrep; PK.qrep = 1; PK.rope_dk = 0;       // partial dK rows are summed first, then un-rotated once
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
        const long rows = rows_all, total = rows * (H_kv * HD / 16);
        hipLaunchKernelGGL(group_sum_heads_kernel, dim3((total + 255) / 256), dim3(256), 0, st, PK.dk, PK.ld_dk, P.dk, P.ld_dk, rows, H_kv, P.nrep, HD,
                           rope_cos_sin, rope_positions, cu_rows, S);
        hipLaunchKernelGGL(group_sum_heads_kernel, dim3((total + 255) / 256), dim3(256), 0, st, PK.dv, PK.ld_dv, P.dv, P.ld_dv, rows, H_kv, P.nrep, HD,
                           (const float*)nullptr, (const int32_t*)nullptr, cu_rows, S);
    }
    return rv_check_launch();
}


extern "C" int rv_attn_bwd_nat(const void* q, int64_t ld_q, const void* k, int64_t ld_k, const void* v, int64_t ld_v, const void* o, int64_t ld_o,
                               const void* dout, int64_t ld_do, const float* lse, float* delta, void* dq, int64_t ld_dq, void* dk, int64_t ld_dk,
                               void* dv, int64_t ld_dv, const int32_t* lens, const int32_t* cu_rows, int total_rows, int B, int H, int H_kv, int S,
                               int S_pad, int HD, int causal, float scale, void* workspace, int64_t workspace_bytes, const float* rope_cos_sin,
                               const int32_t* rope_positions, const void* zeros16, void* stream) {
    if (!q || !k || !v || !o || !dout || !lse || !delta || !dq || !dk || !dv || !zeros16) return RV_ERR_ARG;
    if (H_kv <= 0 || H <= 0 || H % H_kv || HD != 128 || (S_pad & 63) || S_pad < S || B <= 0 || S <= 0) return RV_ERR_ARG;
    if ((ld_q & 7) || (ld_k & 7) || (ld_v & 7) || (ld_do & 7) || (ld_o & 7) || (ld_dq & 3) || (ld_dk & 3) || (ld_dv & 3)) return RV_ERR_ARG;
    if (!aligned_ok(q) || !aligned_ok(k) || !aligned_ok(v) || !aligned_ok(dout) || !aligned_ok(o)) return RV_ERR_ARG;
    if (rope_cos_sin && ((((uintptr_t)rope_cos_sin) & 15) || (cu_rows && !rope_positions))) return RV_ERR_ARG;
    AttnParams P = {};
    P.q = (const bf16*)q; P.k = (const bf16*)k; P.v = (const bf16*)v; P.o = (const bf16*)o; P.dout = (const bf16*)dout;
    P.dq = (bf16*)dq; P.dk = (bf16*)dk; P.dv = (bf16*)dv; P.lse = (float*)lse; P.delta = delta; P.lens = lens; P.cu = cu_rows;
    P.zeros = (const bf16*)zeros16;
    P.ld_q = ld_q; P.ld_k = ld_k; P.ld_v = ld_v; P.ld_o = ld_o; P.ld_do = ld_do; P.ld_dq = ld_dq; P.ld_dk = ld_dk; P.ld_dv = ld_dv;
    P.B = B; P.H = H; P.S = S; P.S_pad = S_pad; P.scale = scale; P.Hkv = H_kv; P.nrep = H / H_kv;
    P.rope_cs = rope_cos_sin; P.rope_pos = rope_positions; P.rope_dk = 1;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid_dq((S + 32 * RV_DQ_NW - 1) / (32 * RV_DQ_NW), H, B), grid_dkv((S + 63) / 64, H_kv, B);
    AttnParams PK = P;
    PK.kdiv = 1; PK.qrep = P.nrep;
    const long rows_all = cu_rows ? (long)total_rows : (long)B * S;
    if (cu_rows && total_rows <= 0) return RV_ERR_ARG;
    const int64_t need = 2 * (int64_t)rows_all * H * HD * 2;
    const bool expand = P.nrep > 1 && workspace && workspace_bytes >= need && (((uintptr_t)workspace) & 15) == 0 &&
                        (long)grid_dkv.x * H_kv * B < 4096;
    if (expand) {     // few key/value heads and a long causal sequence: per-query-head blocks + a group sum balance better
        PK.kdiv = P.nrep; PK.qrep = 1; PK.rope_dk = 0;
        PK.dk = (bf16*)workspace; PK.dv = (bf16*)workspace + (int64_t)rows_all * H * HD;
        PK.ld_dk = PK.ld_dv = (long)H * HD;
        grid_dkv.y = H;
    }
    const int smem_dq = 2 * 2 * 64 * 256, smem_dkv = 2 * (2 * 64 * 256 + 1024);
    // dK/dV pass: 4-wave blocks of 64 keys, two per CU (independent blocks hide each other's barriers and staging latency; measured
    // against 8-wave blocks of 128 keys: -10 % at S = 704, -3 % at S = 3056, -1 % at S = 7499)
#define LAUNCH_BWD_NAT(C_)                                                                                 \
    do {                                                                                                   \
        set_smem(attn_bwd_dq_nat_kernel<C_, RV_DQ_NW>, smem_dq);                                                  \
        set_smem(attn_bwd_dkv_nat_kernel<C_, 4>, smem_dkv);                                                \
        hipLaunchKernelGGL((attn_bwd_dq_nat_kernel<C_, RV_DQ_NW>), grid_dq, dim3(64 * RV_DQ_NW), smem_dq, st, P);           \
        hipLaunchKernelGGL((attn_bwd_dkv_nat_kernel<C_, 4>), grid_dkv, dim3(256), smem_dkv, st, PK);       \
    } while (0)
    if (causal) LAUNCH_BWD_NAT(true); else LAUNCH_BWD_NAT(false);
#undef LAUNCH_BWD_NAT
    if (expand) {
        const long total = rows_all * (H_kv * HD / 16);
        hipLaunchKernelGGL(group_sum_heads_kernel, dim3((total + 255) / 256), dim3(256), 0, st, PK.dk, PK.ld_dk, P.dk, P.ld_dk, rows_all, H_kv, P.nrep, HD,
                           rope_cos_sin, rope_positions, cu_rows, S);
        hipLaunchKernelGGL(group_sum_heads_kernel, dim3((total + 255) / 256), dim3(256), 0, st, PK.dv, PK.ld_dv, P.dv, P.ld_dv, rows_all, H_kv, P.nrep, HD,
                           (const float*)nullptr, (const int32_t*)nullptr, cu_rows, S);
    }
    return rv_check_launch();
}

extern "C" int rv_attn_bwd_gqa(const void* q, int64_t ld_q, const void* k, int64_t ld_k, const void* v, int64_t ld_v,
                               const void* o, int64_t ld_o, const void* dout, int64_t ld_do, const void* qT, const void* kT,
                               const void* doT, const float* lse, float* delta, void* dq, int64_t ld_dq, void* dk,
                               int64_t ld_dk, void* dv, int64_t ld_dv, const int32_t* lens, const int32_t* cu_rows, int total_rows,
                               int B, int H, int H_kv, int S, int S_pad, int HD, int causal, float scale, void* workspace, int64_t workspace_bytes,
                               const void* zeros16, void* stream) {
    return rv_attn_bwd_gqa_rope(q, ld_q, k, ld_k, v, ld_v, o, ld_o, dout, ld_do, qT, kT, doT, lse, delta, dq, ld_dq, dk, ld_dk, dv, ld_dv,
                                lens, cu_rows, total_rows, B, H, H_kv, S, S_pad, HD, causal, scale, workspace, workspace_bytes, nullptr, nullptr,
                                zeros16, stream);
}

extern "C" int rv_attn_bwd(const void* q, int64_t ld_q, const void* k, int64_t ld_k, const void* v, int64_t ld_v,
                           const void* o, int64_t ld_o, const void* dout, int64_t ld_do, const void* qT, const void* kT,
                           const void* doT, const float* lse, float* delta, void* dq, int64_t ld_dq, void* dk,
                           int64_t ld_dk, void* dv, int64_t ld_dv, const int32_t* lens, int B, int H, int S, int S_pad,
                           int HD, int causal, float scale, const void* zeros16, void* stream) {
    return rv_attn_bwd_gqa(q, ld_q, k, ld_k, v, ld_v, o, ld_o, dout, ld_do, qT, kT, doT, lse, delta, dq, ld_dq, dk, ld_dk, dv, ld_dv,
                           lens, nullptr, 0, B, H, H, S, S_pad, HD, causal, scale, nullptr, 0, zeros16, stream);
}
