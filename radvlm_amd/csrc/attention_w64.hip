// head_dim 128 attention, one wave per SIMD: 4-wave workgroups, one per CU (the whole 512-register file per wave), 64 query rows
// (forward, dQ) or 64 keys (dK/dV) per wave, K/V (Q/dO) tiles in a 4-stage LDS-DMA ring behind counted vmcnt + raw s_barrier.
//
// Replaces the same reference code as attention.hip (LlamaAttention core, modeling_llama.py:349-368; fused contract
// llama_flash_attn_monkey_patch.py:16-72): causal, key padding / packed samples, grouped-query heads, dropout 0, scale 1/sqrt(hd).
//
// Why this shape (DESIGN section 5, round 3 stamps): with 32 rows per wave and two waves per SIMD every wave paid 8 staging pieces,
// 48 fragment reads and one softmax per 64 MFMAs and the two waves shared one issue port.  Here a fragment read feeds FOUR MFMAs (the
// wave's four 16-row query blocks), a tile's 8 staging pieces per wave are paid per 128 MFMAs, and the softmax arithmetic of one
// 32-key half tile is hand-placed into the MFMA gaps of the NEXT half tile's score product / the PREVIOUS half tile's P V product:
//
//   slot 1  S(t, keys 0-31)  = K Q^T   32 MFMA  |  exp + row sums + bf16 pack of tile t-1's second half
//   slot 2  O += V P (t-1, keys 32-63) 32 MFMA  |  row maxima of S(t, 0-31); rescale decision at the end of the slot
//   slot 3  S(t, keys 32-63)           32 MFMA  |  exp + row sums + pack of S(t, 0-31)
//   slot 4  O += V P (t, keys 0-31)    32 MFMA  |  row maxima of S(t, 32-63); decision
//
// The O accumulator is only rescaled at slot boundaries, after every P V product that was exponentiated against the old maximum has
// been issued (guide T13: never between the MFMAs of a pending tile).  The running maximum is the integer of the exp2 domain of
// attention.hip (every rescale an exact power of two, deferred until a row outgrows its stored maximum by 2^8).
#include "attn_common.h"

namespace {

constexpr int W_TILE = 64 * 256;            // one natural image: 64 token rows x 128 dims
constexpr int W_NST = 4;                    // ring stages, K and V each
constexpr int W_VBASE = W_NST * W_TILE;     // V ring behind the K ring
constexpr int W_SMEM = 2 * W_NST * W_TILE;  // 128 KiB of the CU's 160

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

DEVINL unsigned fbits(float x) { return __builtin_bit_cast(unsigned, x); }
DEVINL float bitsf(unsigned x) { return __builtin_bit_cast(float, x); }
// maximum / sum over the four 16-lane rows of the wave (lanes c, c+16, c+32, c+48), VALU only: no LDS round trip, nothing on lgkmcnt
DEVINL float xrow_max(float x) {
    auto r = __builtin_amdgcn_permlane16_swap(fbits(x), fbits(x), false, false);
    const float y = fmaxf(bitsf(r[0]), bitsf(r[1]));
    auto q = __builtin_amdgcn_permlane32_swap(fbits(y), fbits(y), false, false);
    return fmaxf(bitsf(q[0]), bitsf(q[1]));
}
DEVINL float xrow_sum(float x) {
    auto r = __builtin_amdgcn_permlane16_swap(fbits(x), fbits(x), false, false);
    const float y = bitsf(r[0]) + bitsf(r[1]);
    auto q = __builtin_amdgcn_permlane32_swap(fbits(y), fbits(y), false, false);
    return bitsf(q[0]) + bitsf(q[1]);
}
DEVINL unsigned pk2(float a, float b) { return __builtin_bit_cast(unsigned, bf16x2{f2bf(a), f2bf(b)}); }
// counted LDS wait: every consumer of the asm-issued reads is itself a volatile asm statement behind this one (no operand ties: a tie made
// hipcc pad one wait state between the wait and the MFMA, eight times per slot)
template <int LEFT> DEVINL void lwait() { asm volatile("s_waitcnt lgkmcnt(%0)" : : "i"(LEFT) : "memory"); }
// single VALU instructions as volatile asm statements: they stay where they are written (between two MFMAs), one instruction each
DEVINL void a_fma_neg(float& d, float x, float k, float mneg) { asm volatile("v_fma_f32 %0, %1, %2, -%3" : "=v"(d) : "v"(x), "s"(k), "v"(mneg)); }   // d = x * k - m
DEVINL void a_exp(float& x) { asm volatile("v_exp_f32 %0, %0" : "+v"(x)); }
DEVINL void a_add(float& acc, float x) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(x)); }
DEVINL void a_max(float& a, float b) { asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(b)); }
DEVINL void a_mov(float& d, float x) { asm volatile("v_mov_b32 %0, %1" : "=v"(d) : "v"(x)); }
DEVINL void a_mulk(float& x, float k) { asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x) : "s"(k)); }
DEVINL void a_ceil(float& d, float x) { asm volatile("v_ceil_f32 %0, %1" : "=v"(d) : "v"(x)); }
DEVINL void a_cvt_pk(unsigned& d, float lo, float hi) { asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(lo), "v"(hi)); }
// operands written >= 2 instructions earlier (VALU write -> permlane read: 2 wait states; the callers interleave two rows' chains)
DEVINL void a_swap16(float& a, float& b) { asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
DEVINL void a_swap32(float& a, float& b) { asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }

// One 1-KiB LDS-DMA piece of a natural tile (see NatPlan): piece i of wave wid = rows (4 i + wid) 4 .. + 3 of the tile.
DEVINL void w_piece(__amdgpu_buffer_rsrc_t rsrc, int v0, int soff, char* lds) {
#if defined(RV_W64_EXP) && RV_W64_EXP == 4
    return;
#endif
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, v0, soff, 0, 0);
}

// MFMAs with hand-allocated accumulator registers.  hipcc gives a 512-register kernel the AGPR form of every MFMA builtin (scores would
// land in AGPRs and cost a v_accvgpr_read each, and its allocator spilled this kernel to scratch); here the two operands that live for
// the whole kernel sit in AGPRs named literally -- O (or dQ) in a[0:127], the Q (Q / dO) fragments behind them -- and everything the VALU
// touches stays in the 256 architectural VGPRs.  Rules (guide 5.7): the compiler must never emit a v_accvgpr_* of its own in these
// kernels (build.sh audits the ISA), every wait state between these statements and their neighbours is placed by hand.
template <int QA> DEVINL void mfma_s0(f32x4& s, const bf16x8& a) {          // s = a x acc[QA..]   (score product, first k-step)
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, a[%c2:%c3], 0" : "=&v"(s) : "v"(a), "i"(QA), "i"(QA + 3));
}
template <int QA> DEVINL void mfma_s(f32x4& s, const bf16x8& a) {           // s += a x acc[QA..]
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(s) : "v"(a), "i"(QA), "i"(QA + 3));
}
template <int OA> DEVINL void mfma_acc(const bf16x8& a, const bf16x8& b) {  // acc[OA..] += a x b
    asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" : : "v"(a), "v"(b), "i"(OA), "i"(OA + 3));
}
template <int A> DEVINL void acc_write(unsigned v) { asm volatile("v_accvgpr_write_b32 a%c1, %0" : : "v"(v), "i"(A)); }
template <int A> DEVINL void acc_zero() { asm volatile("v_accvgpr_write_b32 a%c0, 0" : : "i"(A)); }
template <int A> DEVINL float acc_read() { float v; asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(v) : "i"(A)); return v; }
// wait states between an MFMA of one asm statement and a non-MFMA reader / writer of its result in another (4-pass MFMA: < 16)
DEVINL void mfma_settle() { asm volatile("s_nop 7\n\ts_nop 7" ::: "memory"); }
#define RV_ACC_CLOBBER_192 \
    "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31", \
    "a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63", \
    "a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79","a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95", \
    "a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111","a112","a113","a114","a115","a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127", \
    "a128","a129","a130","a131","a132","a133","a134","a135","a136","a137","a138","a139","a140","a141","a142","a143","a144","a145","a146","a147","a148","a149","a150","a151","a152","a153","a154","a155","a156","a157","a158","a159", \
    "a160","a161","a162","a163","a164","a165","a166","a167","a168","a169","a170","a171","a172","a173","a174","a175","a176","a177","a178","a179","a180","a181","a182","a183","a184","a185","a186","a187","a188","a189","a190","a191"

#ifdef RV_W64_STAMPS
// Diagnostic build only (tools/attn_w64_stamps.py; build with -DRV_W64_STAMPS): per-wave cycle sums of the tile loop's sections -> P.delta
// (unused by the forward), as long long [blocks][4 waves][8].  The stamp's own lgkmcnt(0) drains the fragment reads in flight: read SHARES.
#define W_T0() long long w_t = __builtin_readcyclecounter()
#define W_ACC(k) do { const long long n_ = __builtin_readcyclecounter(); w_dbg[k] += n_ - w_t; w_t = n_; } while (0)
#else
#define W_T0() do { } while (0)
#define W_ACC(k) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------ forward
// accumulator file: O^T[qs][db] = a[(qs * 8 + db) * 4 ..], Q fragment [qs][ks] = a[128 + (qs * 4 + ks) * 4 ..]
constexpr int fw_oa(int qs, int db) { return (qs * 8 + db) * 4; }
constexpr int fw_qa(int qs, int ks) { return 128 + (qs * 4 + ks) * 4; }
// Stream of the 32 MFMA groups (4 MFMAs each: one fragment x the wave's four query blocks) of one tile; FIRST tiles have no slot 2.
//   group 0-7 K rows of keys 0-31 | 8-15 V columns of the PREVIOUS tile's keys 32-63 | 16-23 K rows of keys 32-63 | 24-31 V columns, keys 0-31
constexpr int fw_n(bool first) { return first ? 24 : 32; }
constexpr int fw_group(bool first, int i) { return (first && i >= 8) ? i + 8 : i; }
constexpr bool fw_is_k(int G) { return G < 8 || (G >= 16 && G < 24); }
constexpr int fw_ops(bool first, int i) { return fw_is_k(fw_group(first, i)) ? 1 : 2; }     // LDS operations of the group's fragment
constexpr int FW_AHEAD = 3;
constexpr int fw_left(bool first, int i) {      // LDS operations younger than group i's fragment at its wait
    int n = 0;
    for (int j = i + 1; j <= i + FW_AHEAD && j < fw_n(first); ++j) n += fw_ops(first, j);
    return n;
}

template <bool CAUSAL>
__global__ __launch_bounds__(256, 1) void attn_fwd_w64_kernel(AttnParams P) {
    constexpr int HD = 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wid = wave_id(), lane = lane_id(), g = lane >> 4, c = lane & 15;
    int qblk, h, b;
    block_coords<CAUSAL>(qblk, h, b);
    const long rb = P.cu ? (long)P.cu[b] : (long)b * P.S;
    const int S = P.cu ? P.cu[b + 1] - P.cu[b] : P.S;
    if (qblk * 256 >= S) return;
    const int q0 = qblk * 256 + wid * 64;
    const int len = (P.lens && !P.cu) ? P.lens[b] : S;
    const float sl2 = P.scale * LOG2E;

    asm volatile("" ::: RV_ACC_CLOBBER_192);          // the kernel descriptor allocates the accumulator registers named below
    u32x4 qf[4][4];
#pragma unroll
    for (int qs = 0; qs < 4; ++qs) {
        const int row = min(q0 + qs * 16 + c, S - 1);
        const bf16* p = P.q + (rb + row) * P.ld_q + h * HD + g * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[qs][ks] = *(const u32x4*)(p + ks * 32);
    }
    const int kv_end = CAUSAL ? min(len, qblk * 256 + 256) : len;
    const int ntiles = (kv_end + 63) >> 6;                                       // block-uniform: every wave stages and meets every barrier
    const int t_last = q0 < S ? (CAUSAL ? min(ntiles - 1, q0 >> 6) : ntiles - 1) : -1;   // this wave's last tile with an unmasked key
    const int hk = h / P.nrep;
#if defined(RV_W64_EXP) && RV_W64_EXP == 3
    const __amdgpu_buffer_rsrc_t rsK = rows_rsrc(P.k + rb * P.ld_k, 0, P.ld_k), rsV = rows_rsrc(P.v + rb * P.ld_v, 0, P.ld_v);
#else
    const __amdgpu_buffer_rsrc_t rsK = rows_rsrc(P.k + rb * P.ld_k, S, P.ld_k), rsV = rows_rsrc(P.v + rb * P.ld_v, S, P.ld_v);
#endif
    int kv0_, vv0_, kpstep, vpstep;
    {
        const int cc = wid * 64 + lane, r = cc >> 4, pch = cc & 15;
        kv0_ = (int)(((long)r * P.ld_k + (pch ^ nswz(r)) * 8) * 2);
        vv0_ = (int)(((long)r * P.ld_v + (pch ^ nswz(r)) * 8) * 2);
        kpstep = (int)(16 * P.ld_k * 2);
        vpstep = (int)(16 * P.ld_v * 2);
    }
    const int ktstep = (int)(64 * P.ld_k * 2), vtstep = (int)(64 * P.ld_v * 2), hoff = hk * HD * 2;
    // pieces [i0, i0 + n) of tile tt's K / V image (4 per wave and image)
    auto stage_k = [&](int tt, int i0 = 0, int n = 4) __attribute__((always_inline)) {
        char* dst = smem + (tt & 3) * W_TILE + wid * 1024;
        for (int i = i0; i < i0 + n; ++i) w_piece(rsK, kv0_, tt * ktstep + hoff + i * kpstep, dst + i * 4096);
    };
    auto stage_v = [&](int tt, int i0 = 0, int n = 4) __attribute__((always_inline)) {
        char* dst = smem + W_VBASE + (tt & 3) * W_TILE + wid * 1024;
        for (int i = i0; i < i0 + n; ++i) w_piece(rsV, vv0_, tt * vtstep + hoff + i * vpstep, dst + i * 4096);
    };

    float m[4], l[4];
    sfor<4>([&](auto qt) __attribute__((always_inline)) { m[decltype(qt)::value] = -INFINITY; l[decltype(qt)::value] = 0.f; });
    const float inv_sl2 = 1.f / sl2;
    sfor<128>([&](auto it) __attribute__((always_inline)) { acc_zero<decltype(it)::value>(); });
    if (ntiles > 0) { stage_k(0); stage_v(0); }
    if (ntiles > 1) { stage_k(1); stage_v(1); }
    // Q fragments -> accumulator registers (the compiler's wait for the Q loads lands here: they are older than the staging pieces)
    sfor<16>([&](auto it) __attribute__((always_inline)) {
        constexpr int i = decltype(it)::value, qs = i >> 2, ks = i & 3;
        sfor<4>([&](auto wt) __attribute__((always_inline)) { acc_write<fw_qa(qs, ks) + decltype(wt)::value>(qf[qs][ks][decltype(wt)::value]); });
    });

    // fragment read addresses (see NatAddr): K rows of stage 0; V columns one stage BEFORE stage 0 (slot 2 reads the previous tile)
    unsigned arow[4], acol[8];
    {
        const int r1 = (lane >> 2) & 15, pp = lane & 3;
        const unsigned base = lds_off(smem);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) arow[ks] = base + c * 256 + (((ks * 4 + g) ^ nswz(c)) << 4);
#pragma unroll
        for (int db = 0; db < 8; ++db) acol[db] = base + (W_VBASE - W_TILE) + r1 * 256 + (((2 * db + (pp >> 1)) ^ nswz(r1)) << 4) + ((pp & 1) << 3);
    }

    f32x4 s0[4][2], s1[4][2];       // scores of the two 32-key halves: s[qs][kbl][r] = S^T[key = 32 kp + 16 kbl + 4 g + r][q = 16 qs + c]
    u32x4 pf[4];                    // bf16 probabilities of one half, packed in contraction order (the P V product's B operand)
    float mx[4] = {0.f, 0.f, 0.f, 0.f}, pv[32];            // pv: t = s * sl2 - m, then p = exp2(t), of the half that is being exponentiated
    constexpr float RESCALE_LAG = 8.f;

    // ---- VALU pieces, each placed behind one MFMA: volatile asm statements, one instruction each (compiler-visible arithmetic did not
    // stay put: IR-level sinking moved every exp of a slot behind the slot's closing branch, instruction selection hoisted the rest to
    // the slot's head; operand-pinning empty asm statements cost a pad wait state each).  Hazards by construction: no statement reads a
    // register written by the statement right before it unless the pair is a plain VALU dependency (interlocked in hardware).
    // exp / row sum / bf16 pack of one 32-key half, software-rotated over its 32 scores j = qs * 8 + kbl * 4 + r:
    //   step J:  t[J+1] = s[J+1] * sl2 - m      p[J] = exp2(t[J])      l += p[J-1]      (J even) pack(p[J-2], p[J-1])
    auto exp_step = [&](f32x4 (&s)[4][2], auto jt) __attribute__((always_inline)) {
        constexpr int J = decltype(jt)::value - 1;          // -1 .. 32
        if constexpr (J + 1 < 32) {
            constexpr int j = J + 1, qs = j >> 3, kbl = (j >> 2) & 1, r = j & 3;
            a_fma_neg(pv[j], s[qs][kbl][r], sl2, m[qs]);
        }
        if constexpr (J >= 0 && J < 32) {
            a_exp(pv[J]);
        }
        if constexpr (J >= 1) {
            a_add(l[(J - 1) >> 3], pv[J - 1]);
        }
        if constexpr (J >= 2 && (J & 1) == 0) {
            constexpr int j = J - 2, qs = j >> 3, kbl = (j >> 2) & 1, r = j & 3;     // r = 0 or 2
            unsigned w; a_cvt_pk(w, pv[j], pv[j + 1]); pf[qs][kbl * 2 + (r >> 1)] = w;
        }
    };
    // Row maxima of one half, 16 pieces (one v_max3 each, query block k >> 2): only the LANE's eight scores are reduced.  A row needs a new
    // maximum only when some score outgrows the stored one by more than 2^RESCALE_LAG, and that is a per-lane test against
    // thr = (m + RESCALE_LAG) / sl2 in raw-score units -- the cross-row reduction, the candidate and the rescale run in the rare branch.
    bool need;
    float thr[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};        // m = -inf: the first half always takes the branch
    auto max_piece = [&](f32x4 (&s)[4][2], auto kt) __attribute__((always_inline)) {
        constexpr int k = decltype(kt)::value, qs = k >> 2, u = k & 3;
        if constexpr (u == 0) mx[qs] = max3_asm(s[qs][0][0], s[qs][0][1], s[qs][0][2]);
        if constexpr (u == 1) mx[qs] = max3_asm(mx[qs], s[qs][0][3], s[qs][1][0]);
        if constexpr (u == 2) mx[qs] = max3_asm(mx[qs], s[qs][1][1], s[qs][1][2]);
        if constexpr (u == 3) { mx[qs] = max3_asm(mx[qs], s[qs][1][3], s[qs][1][3]); need |= mx[qs] > thr[qs]; }
    };
    auto decide = [&]() __attribute__((always_inline)) {
        if (__builtin_amdgcn_ballot_w64(need) != 0) {           // wave-uniform, rare after the first tile
            float alpha[4];
            sfor<4>([&](auto qt) __attribute__((always_inline)) {
                constexpr int qs = decltype(qt)::value;
                const float cand = ceilf(xrow_max(mx[qs]) * sl2);   // scale > 0: the maximum commutes with the scaling
                const float mnew = fmaxf(m[qs], cand);
                alpha[qs] = fexp2(m[qs] - mnew);                    // 2^(integer) or 0 (m = -inf before the first tile)
                l[qs] *= alpha[qs];
                m[qs] = mnew;
                thr[qs] = (mnew + RESCALE_LAG) * inv_sl2;
            });
            mfma_settle();                                      // the slot's last P V MFMAs have written O
            sfor<128>([&](auto it) __attribute__((always_inline)) {
                constexpr int a = decltype(it)::value;
                acc_write<a>(fbits(acc_read<a>() * alpha[a >> 5]));
            });
            asm volatile("s_nop 3" ::: "memory");              // v_accvgpr_write -> MFMA reading it as C
        }
    };
    auto mask_half = [&](f32x4 (&s)[4][2], int kbase) __attribute__((always_inline)) {         // keys kbase + 16 kbl + 4 g + r
        sfor<32>([&](auto it) __attribute__((always_inline)) {
            constexpr int i = decltype(it)::value, qs = i >> 3, kbl = (i >> 2) & 1, r = i & 3;
            const int qidx = q0 + qs * 16 + c, kidx = kbase + kbl * 16 + 4 * g + r;
            const bool ok = (kidx < len) && (!CAUSAL || kidx <= qidx);
            s[qs][kbl][r] = ok ? s[qs][kbl][r] : -INFINITY;
        });
    };

#ifdef RV_W64_STAMPS
    long long w_dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long w_start = __builtin_readcyclecounter();
    W_T0();
#endif
    // ---- one tile: the stream of MFMA groups with the fragment reads FW_AHEAD groups ahead
    auto body = [&](int t, auto first_tag) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr int N = fw_n(FIRST);
        const int kv0 = t * 64;
        const bool edge = (kv0 + 64 > len) || (CAUSAL && kv0 + 63 > q0);      // wave-uniform
        bf16x8 kq[4];
        TFrag vq[4];
        auto issue = [&](auto it) __attribute__((always_inline)) {
            constexpr int i = decltype(it)::value, G = fw_group(FIRST, i);
            if constexpr (G < 8) rdrow_imm<(G >> 2) * 4096>(kq[i & 3], arow[G & 3]);
            else if constexpr (G < 16) rdcol_imm<8192>(vq[i & 3], acol[G - 8]);
            else if constexpr (G < 24) rdrow_imm<(2 + ((G - 16) >> 2)) * 4096>(kq[i & 3], arow[G & 3]);
            else rdcol_imm<0>(vq[i & 3], acol[G - 24]);
        };
        sfor<FW_AHEAD>([&](auto it) __attribute__((always_inline)) { issue(it); });
        if constexpr (FIRST) need = false;
        sfor<N>([&](auto it) __attribute__((always_inline)) {
            constexpr int i = decltype(it)::value, G = fw_group(FIRST, i);
            if constexpr (G == 16) {
                // between slots 2 and 3: the V column addresses move on to this tile's stage; the V pieces of tile t + 2 are issued
                const int dcol = ((t & 3) == 0 && t > 0) ? -3 * W_TILE : W_TILE;
                sfor<8>([&](auto dt) __attribute__((always_inline)) { acol[decltype(dt)::value] += dcol; });
            }
            // the 8 staging pieces of tile t + 2 go out two at a time behind the P V slots' MFMA groups (light VALU there); a first tile
            // has no slot 2: its K pieces ride in slot 3
            if constexpr ((!FIRST && (G == 9 || G == 13)) || (FIRST && (G == 17 || G == 21)))
                if (t + 2 < ntiles) stage_k(t + 2, ((G >> 2) & 1) ? 2 : 0, 2);
            if constexpr (G == 25 || G == 29)
                if (t + 2 < ntiles) stage_v(t + 2, G == 29 ? 2 : 0, 2);
            if constexpr (G == 8 || G == 24) {
                if (edge) { mfma_settle(); mask_half(G == 8 ? s0 : s1, kv0 + (G == 8 ? 0 : 32)); }
                if constexpr (G == 24 || !FIRST) need = false;
            }
            if constexpr (i + FW_AHEAD < N) {
                // a V read of THIS tile's stage (groups 24-31) may only be issued once the column addresses have moved (group 16)
                issue(std::integral_constant<int, i + FW_AHEAD>{});
            }
            lwait<fw_left(FIRST, i)>();
            if constexpr (G == 0 && !FIRST) exp_step(s1, std::integral_constant<int, 0>{});       // t[0] of the previous tile's second half
            if constexpr (G == 16) exp_step(s0, std::integral_constant<int, 0>{});
            sfor<4>([&](auto ut) __attribute__((always_inline)) {
                constexpr int u = decltype(ut)::value;      // query block of this MFMA; index of the VALU piece behind it
                if constexpr (G < 8) {
                    constexpr int kbl = G >> 2, ks = G & 3;
                    if constexpr (ks == 0) mfma_s0<fw_qa(u, ks)>(s0[u][kbl], kq[i & 3]); else mfma_s<fw_qa(u, ks)>(s0[u][kbl], kq[i & 3]);
                    if constexpr (!FIRST) exp_step(s1, std::integral_constant<int, G * 4 + u + 1>{});
                } else if constexpr (G < 16) {
                    mfma_acc<fw_oa(u, G - 8)>(tf_get(vq[i & 3]), __builtin_bit_cast(bf16x8, pf[u]));
                    if constexpr (G < 12) max_piece(s0, std::integral_constant<int, (G - 8) * 4 + u>{});
                } else if constexpr (G < 24) {
                    constexpr int kbl = (G - 16) >> 2, ks = G & 3;
                    if constexpr (ks == 0) mfma_s0<fw_qa(u, ks)>(s1[u][kbl], kq[i & 3]); else mfma_s<fw_qa(u, ks)>(s1[u][kbl], kq[i & 3]);
                    exp_step(s0, std::integral_constant<int, (G - 16) * 4 + u + 1>{});
                } else {
                    mfma_acc<fw_oa(u, G - 24)>(tf_get(vq[i & 3]), __builtin_bit_cast(bf16x8, pf[u]));
                    if constexpr (G < 28) max_piece(s1, std::integral_constant<int, (G - 24) * 4 + u>{});
                }
            });
            if constexpr (G == 7 && !FIRST) exp_step(s1, std::integral_constant<int, 33>{});
            if constexpr (G == 23) exp_step(s0, std::integral_constant<int, 33>{});
            if constexpr (FIRST && G == 7) {
                // first tile: slot 2 has no P V product -- the row maxima of S(0, keys 0-31) run alone (wait states between the last
                // score MFMA and the asm maximum that reads its result: the hazard recogniser does not look into inline asm)
                mfma_settle();
                if (edge) mask_half(s0, kv0);
                sfor<16>([&](auto kt) __attribute__((always_inline)) { max_piece(s0, kt); });
                decide();
            }
            if constexpr (G == 15 || G == 31) decide();
            if constexpr (G == 7) W_ACC(1);
            if constexpr (G == 15) W_ACC(2);
            if constexpr (G == 23) W_ACC(3);
            if constexpr (G == 31) W_ACC(4);
        });
        // K row addresses -> next tile's stage
        const int drow = ((t & 3) == 3) ? -3 * W_TILE : W_TILE;
        sfor<4>([&](auto kt) __attribute__((always_inline)) { arow[decltype(kt)::value] += drow; });
    };
    // the last half tile of the wave: exp / pack, then its P V product
    auto drain = [&]() __attribute__((always_inline)) {
        TFrag vq[8];
        sfor<8>([&](auto dt) __attribute__((always_inline)) { rdcol_imm<8192>(vq[decltype(dt)::value], acol[decltype(dt)::value]); });
        sfor<34>([&](auto jt) __attribute__((always_inline)) { exp_step(s1, jt); });
        sfor<8>([&](auto dt) __attribute__((always_inline)) {
            constexpr int db = decltype(dt)::value;
            lwait<2 * (7 - db)>();
            sfor<4>([&](auto qt) __attribute__((always_inline)) { mfma_acc<fw_oa(decltype(qt)::value, db)>(tf_get(vq[db]), __builtin_bit_cast(bf16x8, pf[decltype(qt)::value])); });
        });
    };

    for (int t = 0; t < ntiles; ++t) {
#ifndef RV_W64_EXP
#define RV_W64_EXP 0     // timing experiments only (wrong results): 1 no barrier, 2 no barrier + no staging waits, 3 zero-record staging descriptors, 4 no staging
#endif
#if RV_W64_EXP != 2 && RV_W64_EXP != 4
        if (t + 1 < ntiles) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // tile t landed; tile t + 1 (8 pieces) stays in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#if RV_W64_EXP == 0 || RV_W64_EXP == 3
        __builtin_amdgcn_s_barrier();
#endif
        W_ACC(0);
        // every wave has left tile t - 1: stage (t + 2) & 3 (tile t - 2's images) is free from here on
        W_ACC(5);
#ifdef RV_W64_STAMPS
        if (t <= t_last) w_dbg[7] += 1;
#endif
        if (t <= t_last) {
            if (t == 0) body(t, std::true_type{});
            else body(t, std::false_type{});
        } else {
            if (t + 2 < ntiles) { stage_k(t + 2); stage_v(t + 2); }
            if (t == t_last + 1 && t_last >= 0) drain();
            W_ACC(6);
        }
    }
    if (t_last >= 0 && t_last == ntiles - 1) drain();

#ifdef RV_W64_STAMPS
    if (P.delta && lane == 0) {
        const long long tot = __builtin_readcyclecounter() - w_start;
        const long blk = blockIdx.x + (long)gridDim.x * (blockIdx.y + (long)gridDim.y * blockIdx.z);
        long long* o_ = (long long*)P.delta + (blk * 4 + wid) * 8;
        for (int i = 0; i < 8; ++i) o_[i] = i == 6 ? tot : w_dbg[i];
    }
#endif
    // ---- epilogue: O / l, 16-byte stores (two 16-dim blocks exchanged between the wave's 16-lane rows: 8 consecutive dims per lane)
    mfma_settle();
    sfor<4>([&](auto qt) __attribute__((always_inline)) {
        constexpr int qs = decltype(qt)::value;
        const float lt = xrow_sum(l[qs]);
        const int qidx = q0 + qs * 16 + c;
        const float inv = 1.f / lt;
        u32x2 pk[8];
        sfor<8>([&](auto dt) __attribute__((always_inline)) {
            constexpr int db = decltype(dt)::value, a = fw_oa(qs, db);
            pk[db] = u32x2{pk2(acc_read<a>() * inv, acc_read<a + 1>() * inv), pk2(acc_read<a + 2>() * inv, acc_read<a + 3>() * inv)};
        });
        bf16* op = P.out + (rb + qidx) * P.ld_o + h * HD + 16 * (g & 1) + 8 * (g >> 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            auto x = __builtin_amdgcn_permlane16_swap(pk[2 * j][0], pk[2 * j + 1][0], false, false);
            auto y = __builtin_amdgcn_permlane16_swap(pk[2 * j][1], pk[2 * j + 1][1], false, false);
            if (qidx < S) *(u32x4*)(op + 32 * j) = u32x4{x[0], y[0], x[1], y[1]};
        }
        if (g == 0 && qidx < S && P.lse) P.lse[(long)(b * P.H + h) * P.S_pad + qidx] = (m[qs] + __builtin_amdgcn_logf(lt)) * LN2;
    });
}

template __global__ void attn_fwd_w64_kernel<true>(AttnParams);
template __global__ void attn_fwd_w64_kernel<false>(AttnParams);

}  // namespace

// launchers called from attention.hip's C entry points (arguments already validated there)
int rv_attn_fwd_w64_launch(const AttnParams& P, int causal, hipStream_t st) {
    dim3 grid((P.S + 255) / 256, P.H, P.B);
    if (causal) {
        hipFuncSetAttribute((const void*)attn_fwd_w64_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, W_SMEM);
        hipLaunchKernelGGL(attn_fwd_w64_kernel<true>, grid, dim3(256), W_SMEM, st, P);
    } else {
        hipFuncSetAttribute((const void*)attn_fwd_w64_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, W_SMEM);
        hipLaunchKernelGGL(attn_fwd_w64_kernel<false>, grid, dim3(256), W_SMEM, st, P);
    }
    return rv_check_launch();
}
