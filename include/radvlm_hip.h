/* radvlm_hip.h -- C ABI of libradvlm_hip.so: the MI355X (gfx950) kernels of the LLaVA training hot path.
 *
 * Drop-in boundary: the reference (rfahrn/RadVLM) has no C/FFI surface of its own -- its
 * hot path bottoms out in torch/HF Python calls.  Each entry point below therefore names the reference call site
 * (file:line under /root/reference/finetuning/llava, or HF: = transformers as pinned by the reference) whose
 * arithmetic it replaces.  Conventions for every function:
 *   - plain device pointers + sizes, no torch types; bf16 = raw uint16 storage; strides (ld*) in ELEMENTS;
 *   - asynchronous on `stream` (a hipStream_t), no allocation, no host sync;
 *   - no state that depends on a call's arguments.  What the library does keep, per process (one process drives one GPU): the
 *     GEMM launch configuration (CU budget, rv_gemm_set_cu_budget; tile-selection hook, rv_gemm_select_kernel) and "dynamic LDS
 *     size attribute already set" flags per kernel instantiation.  Results never depend on it, only the launch shape does;
 *   - `zeros16` is any 16-byte-aligned device buffer of >= 16 zero bytes (source for out-of-range tile chunks);
 *   - returns 0 (RV_OK) or a negative error code (RV_ERR_*), which the Python binding raises as an exception.
 */
#ifndef RADVLM_HIP_H
#define RADVLM_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RV_ACT_NONE 0
#define RV_ACT_QUICK_GELU 1 /* x*sigmoid(1.702x): HF:activations.py QuickGELUActivation (CLIP MLP) */
#define RV_ACT_GELU 2       /* erf GELU: torch.nn.GELU in multimodal_projector/builder.py:44 */
#define RV_ACT_GELU_TANH 3  /* gelu_pytorch_tanh: SigLipMLP, multimodal_encoder/siglip_encoder.py:83,:243-256 */

/* Library version / build arch string ("gfx950"). */
const char* rv_version(void);

/* ---- GEMM -------------------------------------------------------------------------------------------------
 * C[M,N] = act(A[M,K] * B[N,K]^T + bias[N]) + residual[M,N]          (bf16 in, fp32 accumulate)
 * = torch.nn.functional.linear at: modeling_llama.py:332-338,377 (q/k/v/o), :226 (gate/up/down), :1323 (lm_head);
 *   HF:models/clip/modeling_clip.py:298-350 (CLIP q/k/v/out/fc1/fc2), :209 (patch conv as GEMM over im2col rows);
 *   multimodal_projector/builder.py:41-48.  Backward (dgrad / wgrad) goes through rv_gemm_bf16 below, which reads its operands
 *   contraction-major in place (no transposed copies).
 * K % 8 == 0, lda % 8 == 0, ldb % 8 == 0.  out_f32: C is fp32 (else bf16).  residual may alias C (accumulate).
 */
int rv_gemm_nt_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const void* bias,
                    const void* residual, int64_t ldr, int M, int N, int K, int act, int out_f32, int res_f32,
                    const void* zeros16, void* stream);
/* General form: op(A)[M,K] * op(B)[N,K]^T with either operand stored contraction-major instead:
 *   trans_a != 0: A is stored [K, M] (row stride lda);  trans_b != 0: B is stored [K, N] (row stride ldb).
 * Lets the autograd GEMMs read their operands in place (hardware-transposed LDS reads, ds_read_b64_tr_b16):
 *   dgrad dX[M,Kin] = dY[M,Nout] * W[Nout,Kin]        -> trans_b (W is [contraction, Kin]);
 *   wgrad dW[Nout,Kin] = dY[M,Nout]^T * X[M,Kin]      -> trans_a and trans_b (both are [contraction=tokens, features]).
 * Feature dimensions of transposed operands must be multiples of 8.  C = act(alpha * op(A) op(B)^T + bias) + residual;
 * alpha carries the LoRA scaling lora_alpha / r (peft LoraLayer, train/train.py:1515-1532). */
int rv_gemm_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const void* bias,
                 const void* residual, int64_t ldr, int M, int N, int K, int trans_a, int trans_b, float alpha, int act,
                 int out_f32, int res_f32, const void* zeros16, void* stream);
/* Extended form:  C = act(alpha * (op(A) op(B)^T + op(A2) op(B2)^T) + bias) + residual.
 *  - (A2, B2, K2): optional second operand pair sharing the transposition flags -- the fused LoRA GEMM
 *    y = [x | t] [W | B]^T (forward) and dx = [dy | dt] [W ; A] (dgrad) of peft's LoraLayer (train/train.py:1515-1532)
 *    in one launch, without a second pass over y.  NULL / 0 disables it.
 *  - workspace (optional fp32 scratch): lets outputs with few tiles and a long contraction (LoRA dA/dB) run split-K over
 *    the idle CUs; partial sums are combined by a deterministic reduce kernel (no atomics). */
int rv_gemm_bf16_ex(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const void* bias,
                    const void* residual, int64_t ldr, int M, int N, int K, int trans_a, int trans_b, float alpha, int act,
                    int out_f32, int res_f32, const void* A2, int64_t lda2, const void* B2, int64_t ldb2, int K2,
                    void* workspace, int64_t workspace_bytes, const void* zeros16, void* stream);
/* ---- GEMMs with a fused elementwise epilogue (HBM passes removed from the decoder layer) ----------------------------------
 * Each runs on the 256x256-tile kernel when the output is large enough for it and otherwise performs the unfused sequence itself
 * (GEMM, then the elementwise kernel) -- bit-identical results either way: the fused epilogues round where the unfused path stored.
 *
 * rv_gemm_rope_bf16: C[M,N] = A[M,K] B[N,K]^T + bias, then rotary embedding on the first rope_heads heads of hd columns each
 *   (q heads followed by k heads of the fused q|k|v projection; the v columns pass through):
 *   LlamaAttention q_proj/k_proj/v_proj + apply_rotary_pos_emb, modeling_llama.py:332-338 and :167-198 (Qwen2: with q/k/v bias).
 *   cos_sin: fp32 [positions, hd/2, 2] (LlamaRotaryEmbedding.forward :123-139, rounded through bf16 by the caller);
 *   position of token row m: positions[m], or m % S when positions is NULL (training: position_ids = arange(S), llava_arch.py:534-545).
 *   The B rows of a tile are staged in an order that puts the rotation partners (e, e + hd/2) into one lane.  hd 64 or 128 fused. */
int rv_gemm_rope_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const void* bias, int M, int N, int K,
                      const float* cos_sin, const int32_t* positions, int S, int rope_heads, int hd, void* workspace,
                      int64_t workspace_bytes, const void* zeros16, void* stream);
/* rv_gemm_swiglu_fwd_bf16: LlamaMLP's gate/up projections and activation in one launch (modeling_llama.py:226):
 *   GU[M, 2F] = A[M,K] [Wgate; Wup][2F,K]^T (kept for backward), ACT[M,F] = silu(GU[:, :F]) * GU[:, F:].  Wgu = the stacked [2F, K]
 *   weight (gate rows then up rows); a tile holds gate and up of the same 128 features, so the product is lane-local. */
int rv_gemm_swiglu_fwd_bf16(const void* A, int64_t lda, const void* Wgu, int64_t ldb, void* GU, int64_t ldgu, void* ACT, int64_t ldact,
                            int M, int F, int K, void* workspace, int64_t workspace_bytes, const void* zeros16, void* stream);
/* rv_gemm_swiglu_bwd_bf16: backward of the same through down_proj's input gradient: dGU[M, 2F] = swiglu'(GU) * (dY[M,K] Wd[K,F])
 *   with Wd = down_proj.weight stored [K = hidden, F] (read contraction-major, in place); d(act) never reaches memory.
 *   dact_scratch [M, F] is only used by the unfused fallback (may be NULL when the fused form is certain to run). */
int rv_gemm_swiglu_bwd_bf16(const void* dY, int64_t ldy, const void* Wd, int64_t ldw, const void* GU, int64_t ldgu, void* dGU, int64_t lddgu,
                            void* dact_scratch, int64_t ld_dact, int M, int F, int K, void* workspace, int64_t workspace_bytes,
                            const void* zeros16, void* stream);
/* Launch-shape selection, process-wide.  Measurement hooks (A/B tools and tests only): 0 = automatic tile selection (default),
 * 1 = 128x128 tile kernel, 2 = 256x256 tile kernel, 20 / 21 = tail split off / on, 30 / 31 = buffer-addressed staging off / on.
 * 40 / 41 = persistent tile-walking blocks off / on (default on): the engine turns them OFF for world size > 1, where RCCL kernels
 * share the CUs -- a persistent block that cannot start delays its whole share of the tiles.  Results never depend on any of it. */
int rv_gemm_select_kernel(int which);
/* Compute units the GEMM plans its tile rounds for (whole rounds of one 256x256 tile per CU, the K-split of a half-empty last
 * round, split-K of small outputs).  total_cus <= 0: the current device's multiProcessorCount (256 on MI355X); reserved_cus:
 * units left to concurrently running work -- the bucketed RCCL all-reduce that overlaps backward in data-parallel runs
 * (what DDP's NCCL kernels take on the reference's GPUs).  Default without a call: all units, or
 * RV_GEMM_RESERVED_CUS from the environment.  Returns the resulting budget (>= 8) or a negative error code.  Process-wide. */
int rv_gemm_set_cu_budget(int total_cus, int reserved_cus);

/* Batched strided transpose of bf16 matrices: out[bz][c][r] = in[bz][r][c], r < R, c < C; columns r in [R, R_pad)
 * of every output row are written as zero.  bz = b0 * nb1 + b1; offsets in elements.
 * Used for the [b,h,hd,S_pad] head-transposed attention operands (perm32 = 1: within every aligned group of 32
 * output columns, column 8g + 4h + j holds input row 16h + 4g + j -- the order in which an MFMA accumulator tile
 * presents the sequence axis as the next MFMA's contraction index, so each lane's 8 operands are one 16-byte read;
 * requires R_pad % 64 == 0). */
int rv_transpose_bf16(const void* in, int64_t in_ld, int64_t in_bs0, int64_t in_bs1, void* out, int64_t out_ld,
                      int64_t out_bs0, int64_t out_bs1, int R, int C, int R_pad, int nb0, int nb1, int perm32,
                      void* stream);

/* ---- normalisation ------------------------------------------------------------------------------------------
 * LlamaRMSNorm.forward (modeling_llama.py:82-87): y = w * bf16(x * rsqrt(mean(x^2) + eps)), fp32 internal.
 * rstd (fp32 [rows], optional) is saved for backward. */
int rv_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int rows, int d, float eps, void* stream);
/* dx = rstd * (w*dy - xhat * mean(w*dy*xhat)); dw_partial[blk, d] (fp32, nblk rows) holds per-block sums of dy*xhat
 * (finish with rv_colsum_f32).  If dx_add != 0, dx += (residual-stream gradient accumulation). */
int rv_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, void* dx, int dx_add,
                   float* dw_partial, int nblk, int rows, int d, void* stream);
/* torch.nn.LayerNorm as used by CLIP (HF:modeling_clip.py:362-384, pre_layrnorm :744).  stats (optional, fp32
 * [rows,2] = mean, rstd) is saved for backward.  Backward: dx (+)= ..., partial[blk] = [sum dy*xhat (d) | sum dy (d)]
 * (fp32 [nblk, 2d]; finish both halves with rv_colsum_f32) -- used when the vision tower is tunable
 * (mm_tunable_parts contains mm_vision_tower, train/train.py:1658-1661). */
int rv_layernorm_fwd(const void* x, const void* w, const void* b, void* y, float* stats, int rows, int d, float eps,
                     void* stream);
int rv_layernorm_bwd(const void* dy, const void* x, const void* w, const float* stats, void* dx, int dx_add,
                     float* partial, int nblk, int rows, int d, void* stream);
/* CLIP MLP activation x*sigmoid(1.702x) (HF:activations.py QuickGELUActivation) and its derivative. */
int rv_quick_gelu_fwd(const void* x, void* y, int64_t n, void* stream);
int rv_quick_gelu_bwd(const void* dy, const void* x, void* dx, int64_t n, void* stream);

/* SigLIP MLP activation 0.5x(1+tanh(sqrt(2/pi)(x+0.044715x^3))) (siglip_encoder.py:83 hidden_act) and its derivative. */
int rv_gelu_tanh_fwd(const void* x, void* y, int64_t n, void* stream);
int rv_gelu_tanh_bwd(const void* dy, const void* x, void* dx, int64_t n, void* stream);

/* out[c] (+)= sum_r in[r, c]  (fp32 partial rows -> bf16 vector). */
int rv_colsum_f32(const float* in, int rows, int cols, void* out_bf16, int accumulate, void* stream);
/* partial[blk, c] = sum over the block's rows of x[r, c]  (bf16 [rows, cols] with stride ld -> fp32 [nblk, cols]);
 * bias gradients of nn.Linear. */
int rv_colsum_partial_bf16(const void* x, int64_t ld, int rows, int cols, float* partial, int nblk, void* stream);

/* ---- rotary embedding ------------------------------------------------------------------------------------------
 * apply_rotary_pos_emb (modeling_llama.py:167-198), half-split convention, in place on `nsec` consecutive
 * [heads*hd] sections of each token row (q and k of a fused qkv row).  cos_sin: fp32 [S, hd/2, 2];
 * position = row % S (position_ids = arange(S) for every sample, llava_arch.py:534-545).  dir = +1 fwd, -1 bwd. */
int rv_rope_inplace(void* x, int64_t ld, const float* cos_sin, int rows, int S, int heads, int hd, int nsec, int dir,
                    void* stream);

/* ---- attention ----------------------------------------------------------------------------------------------------
 * softmax_fp32(scale * Q K^T + causal + key-padding) V: modeling_llama.py:349-368 + :1191-1225,
 * llama_flash_attn_monkey_patch.py:51-69; non-causal for HF:modeling_clip.py:259-277.
 * q/k/out: token-major [(b*S+s)*ld + h*hd + e]; vT: [b,h,hd,S_pad] (zero padded); lse: fp32 [b,h,S_pad];
 * lens (optional int32 [B]): valid keys per sample (right padding).  hd in {64,128}; S_pad % 64 == 0. */
int rv_attn_fwd(const void* q, int64_t ld_q, const void* k, int64_t ld_k, const void* vT, void* out, int64_t ld_o,
                float* lse, const int32_t* lens, int B, int H, int S, int S_pad, int hd, int causal, float scale,
                const void* zeros16, void* stream);
/* Backward of the above: dQ (query-block pass, which also produces delta = rowsum(dO*O)), then dK/dV (key-block pass).
 * qT/kT/doT are [b,h,hd,S_pad] transposed copies of q, k, dout (rv_transpose_bf16). */
int rv_attn_bwd(const void* q, int64_t ld_q, const void* k, int64_t ld_k, const void* v, int64_t ld_v, const void* o,
                int64_t ld_o, const void* dout, int64_t ld_do, const void* qT, const void* kT, const void* doT,
                const float* lse, float* delta, void* dq, int64_t ld_dq, void* dk, int64_t ld_dk, void* dv,
                int64_t ld_dv, const int32_t* lens, int B, int H, int S, int S_pad, int hd, int causal, float scale,
                const void* zeros16, void* stream);

/* Grouped-query form (Qwen2: language_model/llava_qwen.py:46-58 -> HF Qwen2Attention with num_key_value_heads < heads;
 * repeat_kv modeling_llama.py:201-210): k / v / kT / vT / dk / dv hold H_kv heads, query head h uses key/value head
 * h / (H / H_kv); the dK/dV pass sums the group's query heads in registers (no expanded copies).  H % H_kv == 0.
 * Packed (variable-length) batches -- SURVEY 8f.2, no padding rows: cu_rows (int32 [B+1], device; may be NULL) gives the
 * first token row of every sample in q/k/v/out/dq/dk/dv (total_rows = cu_rows[B], host copy); S is then the LONGEST sample (grid extent, S_pad >= S), `lens` is
 * ignored, and the [b,h,hd,S_pad] transposed operands and lse/delta stay per-sample padded (rv_transpose_bf16_varlen).
 * Optional `workspace` (>= 2*B*S*H*hd*2 bytes, 16-byte aligned; may be NULL): when the dK/dV grid (key blocks x H_kv x B)
 * is too small to balance a long causal sequence over the chip, the pass runs one block per QUERY head into the
 * workspace and a deterministic group sum folds the partials. */
int rv_attn_fwd_gqa(const void* q, int64_t ld_q, const void* k, int64_t ld_k, const void* vT, void* out, int64_t ld_o,
                    float* lse, const int32_t* lens, const int32_t* cu_rows, int B, int H, int H_kv, int S, int S_pad, int hd,
                    int causal, float scale, const void* zeros16, void* stream);
/* Forward for head_dim 128 on the operands as they lie in memory: v is the token-major [(b*S+s), H_kv*hd] view like k (no V^T copy).
 * Every tile is staged once into an LDS image that serves row reads (contraction over head_dim) and hardware-transposed column
 * reads (contraction over the tile's keys, ds_read_b64_tr_b16).  Same semantics, masks and packed-batch conventions as
 * rv_attn_fwd_gqa. */
int rv_attn_fwd_nat(const void* q, int64_t ld_q, const void* k, int64_t ld_k, const void* v, int64_t ld_v, void* out, int64_t ld_o,
                    float* lse, const int32_t* lens, const int32_t* cu_rows, int B, int H, int H_kv, int S, int S_pad, int hd, int causal,
                    float scale, const void* zeros16, void* stream);
/* Kernel family behind the head_dim 128 natural-layout forward (rv_attn_fwd_nat), process-wide.  Measurement hook (A/B tools and tests
 * only): 0 = default, 1 = two waves per SIMD, 32 query rows per wave (attention.hip; what the default selects), 2 = one wave per SIMD,
 * 64 query rows per wave, hand-placed softmax (attention_w64.hip; measured slower, kept for the A/B record).  Same results up to the
 * rounding of the running maximum's granularity (64- vs 32-key steps). */
int rv_attn_select_kernel(int which);
/* 1 when rv_attn_fwd_nat runs a causal [B, H, S] batch as query-block pairs on the current device, 0 for single query blocks (test hook). */
int rv_attn_fwd_nat_pairs(int B, int H, int S, int causal);
int rv_attn_bwd_gqa(const void* q, int64_t ld_q, const void* k, int64_t ld_k, const void* v, int64_t ld_v, const void* o,
                    int64_t ld_o, const void* dout, int64_t ld_do, const void* qT, const void* kT, const void* doT,
                    const float* lse, float* delta, void* dq, int64_t ld_dq, void* dk, int64_t ld_dk, void* dv,
                    int64_t ld_dv, const int32_t* lens, const int32_t* cu_rows, int total_rows, int B, int H, int H_kv, int S,
                    int S_pad, int hd, int causal, float scale, void* workspace, int64_t workspace_bytes, const void* zeros16,
                    void* stream);
/* The same with the adjoint of the rotary embedding folded into the dQ / dK epilogues (the backward of apply_rotary_pos_emb,
 * modeling_llama.py:167-198, on q and k that are stored rotated): dq and dk come out as gradients of the UN-rotated projections,
 * no separate pass over d(q|k|v).  rope_cos_sin: fp32 [positions, hd/2, 2] (NULL = plain rv_attn_bwd_gqa); position of a token row =
 * rope_positions[row] (required for packed batches) or its index inside the sample. */
int rv_attn_bwd_gqa_rope(const void* q, int64_t ld_q, const void* k, int64_t ld_k, const void* v, int64_t ld_v, const void* o,
                         int64_t ld_o, const void* dout, int64_t ld_do, const void* qT, const void* kT, const void* doT,
                         const float* lse, float* delta, void* dq, int64_t ld_dq, void* dk, int64_t ld_dk, void* dv,
                         int64_t ld_dv, const int32_t* lens, const int32_t* cu_rows, int total_rows, int B, int H, int H_kv, int S,
                         int S_pad, int hd, int causal, float scale, void* workspace, int64_t workspace_bytes,
                         const float* rope_cos_sin, const int32_t* rope_positions, const void* zeros16, void* stream);

/* Backward for head_dim 128 on the operands as they lie in memory (no q^T / k^T / dO^T copies: every staged tile serves row reads and
 * hardware-transposed column reads, see rv_attn_fwd_nat), with the optional rotary-embedding adjoint of rv_attn_bwd_gqa_rope.
 * delta [B,H,S_pad] fp32 is scratch written by the dQ pass (rowsum(dO * O)) and read by the dK/dV pass. */
int rv_attn_bwd_nat(const void* q, int64_t ld_q, const void* k, int64_t ld_k, const void* v, int64_t ld_v, const void* o, int64_t ld_o,
                    const void* dout, int64_t ld_do, const float* lse, float* delta, void* dq, int64_t ld_dq, void* dk, int64_t ld_dk,
                    void* dv, int64_t ld_dv, const int32_t* lens, const int32_t* cu_rows, int total_rows, int B, int H, int H_kv, int S,
                    int S_pad, int hd, int causal, float scale, void* workspace, int64_t workspace_bytes, const float* rope_cos_sin,
                    const int32_t* rope_positions, const void* zeros16, void* stream);

/* rv_transpose_bf16 for packed batches: batch b0 reads rows [cu_rows[b0], cu_rows[b0+1]) of `in` (R_max = longest). */
int rv_transpose_bf16_varlen(const void* in, int64_t in_ld, const int32_t* cu_rows, int64_t in_bs1, void* out, int64_t out_ld,
                             int64_t out_bs0, int64_t out_bs1, int R_max, int C, int R_pad, int nb0, int nb1, int perm32,
                             void* stream);
/* rv_rope_inplace with an explicit position per token row (packed batches: position = row - first row of its sample). */
int rv_rope_inplace_pos(void* x, int64_t ld, const float* cos_sin, const int32_t* positions, int rows, int heads, int hd,
                        int nsec, int dir, void* stream);

/* ---- MLP activations ------------------------------------------------------------------------------------------------
 * LlamaMLP (modeling_llama.py:226): act[r, f] = silu(gu[r, f]) * gu[r, F + f]   (gu = fused gate|up output). */
int rv_swiglu_fwd(const void* gu, int64_t ld_gu, void* act, int64_t ld_act, int rows, int F, void* stream);
int rv_swiglu_bwd(const void* dact, int64_t ld_dact, const void* gu, int64_t ld_gu, void* dgu, int64_t ld_dgu, int rows,
                  int F, void* stream);
/* Inverted dropout with a counter-based mask: y[i] = keep(seed, i) ? x[i] / (1 - p) : 0; keep: 16 bits of a splitmix64 value shared by the
 * four elements i >> 2, compared with round(p * 2^16) (p is honoured to 1.5e-5).
 * The same (seed, p) regenerates the mask, so backward applies the same call to the gradient (lora_dropout). */
int rv_dropout_bf16(const void* x, void* y, int64_t n, float p, uint64_t seed, void* stream);
/* y += dropout(x) with the same mask as rv_dropout_bf16(p, seed): the adapter branch of a LoRA layer's input gradient,
 * dx += dropout'(dt A), in one pass (peft LoraLayer: lora_dropout is applied to the layer input, so its adjoint masks dt A). */
int rv_dropout_add_bf16(const void* x, void* y, int64_t n, float p, uint64_t seed, void* stream);
/* LoRA down-projection with the adapter's input dropout inside: T[M, R] = alpha / (1 - p) * mask(seed) o X[M, K] * A[R, K]^T
 * = lora_A(lora_dropout(x)) * (alpha / r) of a peft LoraLayer (reference wiring train/train.py:1515-1532, defaults :152-157).
 * The mask is that of rv_dropout_bf16(X as M * K contiguous elements, p, seed) -- backward re-creates dropout(X) with that call --
 * and is applied to the operand fragments in registers: one pass over X instead of three.  K % 64 == 0, R <= 64, R % 4 == 0; with p > 0 X
 * must be contiguous (ldx == K: the mask indexes it as M * K elements); p = 0: the plain skinny product at HBM speed, any ldx % 8 == 0
 * (also used for dT = alpha * dY B, the adapters' backward, with B^T as the row-major operand). */
/* dx[M, N] (+)= dropout'(alpha * dT[M, K] op(A)): the adapter branch of a LoRA layer's input gradient with the forward's dropout mask
 * (that of rv_dropout_bf16 over M * N elements) applied to the accumulators in the GEMM epilogue.  trans_b: A stored [K, N] (lora_A [r, in]).
 * accumulate != 0: added to dx.  N % 8 == 0. */
int rv_gemm_dropout_add_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K, int trans_b,
                             float alpha, float p, uint64_t seed, int accumulate, const void* zeros16, void* stream);
int rv_lora_down_bf16(const void* X, int64_t ldx, const void* A, int64_t lda, void* T, int64_t ldt, int M, int R, int K, float alpha,
                      float p, uint64_t seed, const void* zeros16, void* stream);
/* gA[R, K] (+)= 1 / (1 - p) * dT[M, R]^T mask_p(X)[M, K]: the gradient of a LoRA adapter's A matrix (the adjoint of lora_A(lora_dropout(x)),
 * reference wiring train/train.py:1515-1532) with the forward's dropout mask -- that of rv_dropout_bf16(X as M * K contiguous elements, p, seed) --
 * re-created in registers: X is read once and dropout(X) never reaches HBM.  dT is the (alpha / r)-scaled gradient of the adapter's inner
 * activation.  R <= 64, R % 8 == 0, K % 8 == 0, 16-byte aligned X and dT; p > 0 needs ldx == K.  accumulate != 0: added to gA.  workspace: fp32
 * scratch for the token-slice partial sums, at least R * K * 4 bytes (deterministic reduction: no atomics). */
int rv_lora_a_grad_bf16(const void* dT, int64_t ldt, const void* X, int64_t ldx, void* gA, int64_t ldg, int M, int R, int K, float p, uint64_t seed,
                        int accumulate, void* workspace, int64_t workspace_bytes, void* stream);
/* torch.nn.GELU (erf) of the mm_projector (multimodal_projector/builder.py:44) and its derivative. */
int rv_gelu_fwd(const void* x, void* y, int64_t n, void* stream);
int rv_gelu_bwd(const void* dy, const void* x, void* dx, int64_t n, void* stream);

/* ---- loss ----------------------------------------------------------------------------------------------------------
 * LlamaForCausalLM loss (modeling_llama.py:1323-1337): logits.float(), CE with ignore_index -100.
 * labels[r] is the ALREADY SHIFTED target of row r.  loss_rows[r] = -log softmax(logits[r])[label] (0 if ignored).
 * If dlogits != NULL: dlogits[r] = (softmax - onehot) * inv_count (0 rows if ignored).  In-place use (dlogits == logits, same
 * ld) is supported: the target logit is read before any store of the row.  A label >= V is treated like ignore_index (it
 * never indexes the row); callers reject such labels on the host (radvlm_amd.engine does).
 * V may be any size; rows are ld >= ceil8(V) wide, the pad columns are ignored on read and get zero gradient. */
int rv_cross_entropy(const void* logits, int64_t ld, const int64_t* labels, float* loss_rows, void* dlogits,
                     int64_t ld_d, int rows, int V, float inv_count, void* stream);
/* out[0] = scale * sum(in[0..n))  (deterministic, single block). */
int rv_sum_f32(const float* in, int64_t n, float scale, float* out, void* stream);

/* ---- embedding splice ---------------------------------------------------------------------------------------------
 * prepare_inputs_labels_for_multimodal steps (vi)-(viii), llava_arch.py:449-531, as one gather:
 * dst[r] = idx[r] >= 0 ? table_a[idx[r]] : (idx[r] == -1 ? 0 : table_b[-idx[r] - 2]).  (exact-zero pad rows) */
int rv_gather_rows(void* dst, int64_t ld_dst, const void* table_a, int64_t ld_a, const void* table_b, int64_t ld_b,
                   const int32_t* idx, int rows, int d, void* stream);
/* Embedding gradient without atomics: for segment s, out[out_row[s]] = sum_{j in [off[s], off[s+1])} src[pos[j]]. */
int rv_segment_sum_rows(const void* src, int64_t ld_src, const int32_t* seg_off, const int32_t* pos,
                        const int32_t* out_row, int nseg, void* out, int64_t ld_out, int d, void* stream);

/* Weighted form: out[out_row[s]] = sum_j w[j] * src[pos[j]].  Forward and adjoint of the anyres_max bilinear
 * down-sampling nn.functional.interpolate(mode="bilinear") at llava_arch.py:381-392 (4 taps per output row; the
 * adjoint's CSR is the transposed tap list, built on the host), deterministic (no atomics). */
int rv_weighted_segment_sum_rows(const void* src, int64_t ld_src, const int32_t* seg_off, const int32_t* pos,
                                 const float* w, const int32_t* out_row, int nseg, void* out, int64_t ld_out, int d,
                                 void* stream);

/* 'maxpool2x2' patch merge (nn.functional.max_pool2d(grid, 2), llava_arch.py:375-379): out[out_row[s]] = elementwise max of the
 * four rows src[idx4[4s..4s+3]] (window scan order; `which` keeps the winning slot per element); backward routes the gradient
 * of every pooled row to the winning source element and zero to the other three (windows do not overlap: no atomics). */
int rv_max4_rows_fwd(const void* src, int64_t ld_src, const int32_t* idx4, const int32_t* out_row, int n, void* out, int64_t ld_out,
                     uint8_t* which, int d, void* stream);
int rv_max4_rows_bwd(const void* dout, int64_t ld_dout, const int32_t* idx4, const int32_t* dout_row, int n, const uint8_t* which,
                     void* dsrc, int64_t ld_dsrc, int d, void* stream);

/* Device-side image normalisation and tiling (the host image path of train/train.py:1060-1099 + mm_utils.py:243-293 ends in
 * processor.preprocess: rescale 1/255, (x - mean) / std, HWC -> CHW, per tile): n uint8 canvases [gh*tile, gw*tile, 3] (HOST memory is
 * not accepted: device pointers only) -> bf16 [n*gh*gw, 3, tile, tile], tiles in row-major grid order (divide_to_patches, mm_utils.py:191-210).
 * mode 0: CLIPImageProcessor arithmetic (float32(u8) / 255), mode 1: SigLipImageProcessor (float64(u8) * factor -> float32); mean3 / std3
 * are HOST arrays of 3 floats (passed by value).  Bit-identical to normalising on the host in fp32 and casting on the device; the upload
 * is a quarter of the fp32 bytes. */
int rv_normalize_tiles_u8(const uint8_t* img, void* out, int n, int gh, int gw, int tile, int mode, double factor, const float* mean3,
                          const float* std3, void* stream);
/* ---- CLIP embeddings -----------------------------------------------------------------------------------------------
 * HF:modeling_clip.py:202-218: patches of pix [n,3,H,W] (bf16) -> rows [n*gh*gw, Kp], k = c*p*p + i*p + j (zero
 * padded to Kp); then out[n, 0] = cls + pos[0], out[n, 1+i] = patch_out[n, i] + pos[1+i]. */
int rv_im2col_patches(const void* pix, void* out, int n, int H, int W, int p, int Kp, void* stream);
int rv_clip_embed(const void* patch_out, const void* cls, const void* pos, void* out, int n, int P, int d, void* stream);
/* SigLipVisionEmbeddings (siglip_encoder.py:169-174): x[n, i] += pos[i] in place (no class token; the conv bias rides the
 * patch GEMM).  rv_im2col_patches accepts H, W that are not multiples of p ('valid' conv: trailing pixels unused). */
int rv_add_pos_rows(void* x, const void* pos, int n, int P, int d, void* stream);

/* ---- optimizer / misc ----------------------------------------------------------------------------------------------
 * torch.optim.AdamW step (optim="adamw_torch", train/train.py:140) on a flat slice: fp32 master/m/v, bf16 params and
 * grads; grad is multiplied by *gscale (device scalar, e.g. the clip coefficient) if gscale != NULL. */
int rv_adamw(void* p_bf16, float* master, const void* g_bf16, float* m, float* v, int64_t n, float lr, float b1,
             float b2, float eps, float wd, float bc1, float bc2, const float* gscale, void* stream);
/* partial[blk] = sum of squares of the block's slice (fp32); finish with rv_clip_coef. */
int rv_sumsq_partial_bf16(const void* g, int64_t n, float* partial, int nblk, void* stream);
/* norm = sqrt(sum partial); out[0] = norm, out[1] = min(1, max_norm / (norm + 1e-6))  (torch clip_grad_norm_). */
int rv_clip_coef(const float* partial, int nblk, float max_norm, float* out2, void* stream);
int rv_cast_f32_to_bf16(const float* in, void* out, int64_t n, void* stream);
int rv_cast_bf16_to_f32(const void* in, float* out, int64_t n, void* stream);
/* y[i] = a[i] + b[i] (bf16). */
int rv_add_bf16(const void* a, const void* b, void* y, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif
