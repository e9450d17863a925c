"""The bf16-exact one-ulp gate (tests/test_backward_parity_gpu.py, tests/test_e2e_gpu.py::test_bf16_emulated_parity) extended from decoder
layer 0 to every other op of the trainable path (-m gpu):

  * one CLIP-L/14-336 layer (d 1024, 16 heads x 64, quick_gelu) and one SigLIP-so400m layer (d 1152, 16 heads x 72 padded to 128,
    gelu_tanh), forward AND backward: LayerNorm fwd / bwd, the q|k|v / out / fc1 / fc2 GEMMs with bias, residual and activation epilogues,
    head_dim-64 and padded-head attention fwd / bwd, activation backward, bias column sums
    (HF modeling_clip.py:202-384, siglip_encoder.py:148-306);
  * the mlp2x_gelu projector fwd / bwd (multimodal_projector/builder.py:41-48);
  * lm_head + rv_cross_entropy fwd / bwd (modeling_llama.py:1323-1337);
  * the splice backward: rv_segment_sum_rows, rv_weighted_segment_sum_rows, rv_max4_rows_bwd (llava_arch.py:381-392,442-531 adjoints);
  * the LoRA kernels against an emulation that applies the SAME counter-based dropout mask: rv_lora_down_bf16, the fused second
    operand pair (MODE 1 GEMM), rv_gemm_dropout_add_bf16, rv_dropout_bf16 (peft LoraLayer; train/train.py:1515-1532 -- peft is absent and
    the reference holds no LoRA fixture: parity unpinned upstream, the emulation restates the published formula).

Every op is run on the inputs the emulated chain (oracle/bf16_emulation.py, pinned to torch autograd of the reference-pinned oracle by
tests/test_oracle_golden.py::test_bf16_emulation_tower_head_backward_reduces_to_autograd) feeds it and must reproduce that op's emulated
output.  Gates (the same classes as the decoder-layer gate):
  single-rounding bf16 outputs   <= 1e-3 of the elements differ, each by one ulp (elements below 2^-12 of the tensor's largest are measured
                                 in the ulp of that floor);
  column sums (bias / norm grads) <= 1 ulp, <= 2e-2 of the elements (two-stage fp32 reduction of another order);
  attention forward              <= 1e-3 of the elements differ, every error below 2^-7 of the tensor's largest element (P is rounded inside);
  attention backward             <= 2e-3 of the elements off by more than half an ulp (ulp floored at 2^-4 of the tensor's largest), every
                                 error below 2^-8 of the tensor's largest element (P and dS are rounded inside);
  two store points in one op     (LoRA input gradient: base product stored, adapter branch added in a second pass) <= 1e-3 of the elements,
                                 every error below 2^-8 of the tensor's largest element;
  fp32 weight-gradient sums      ||d||_inf / ||ref||_inf <= 1e-5 against a float64 product.
"""
import json
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
RESULTS = {}
DEV = "cuda:0"
F32 = torch.float32


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


class Gate:
    def __init__(self, rnd):
        self.res, self.rnd = {}, rnd

    def bf(self, name, hip, emu, kind="one_ulp", floor_exp=-12):
        hip = hip.detach().float().cpu().reshape(-1, emu.shape[-1])
        ref = self.rnd(emu.reshape(-1, emu.shape[-1]).float())
        mag = torch.maximum(ref.abs(), ref.abs().max() * 2.0 ** floor_exp).clamp_min(1e-37)
        ulp = torch.exp2(torch.floor(torch.log2(mag)) - 7)
        err = (hip - ref).abs() / ulp
        self.res[name] = dict(kind=kind, mismatch_frac=float((err > 0.5).float().mean()), max_ulp=float(err.max()),
                              max_abs_err_over_tensor_max=float((hip - ref).abs().max() / ref.abs().max().clamp_min(1e-37)))

    def f32(self, name, hip, ref):
        self.res[name] = dict(kind="f32", relinf=float((hip.detach().float().cpu() - ref).abs().max() / ref.abs().max()))

    def check(self, tag):
        RESULTS[tag] = self.res
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/bf16_tower_head_parity.json", "w") as f:
            json.dump(RESULTS, f, indent=1)
        print(tag, json.dumps(self.res))
        for k, v in self.res.items():
            if v["kind"] == "f32":
                assert v["relinf"] <= 1e-5, (k, v)
            elif v["kind"] == "colsum":
                assert v["max_ulp"] <= 1.0 and v["mismatch_frac"] <= 2e-2, (k, v)
            elif v["kind"] == "attn_bwd":
                assert v["mismatch_frac"] <= 2e-3 and v["max_abs_err_over_tensor_max"] <= 2.0 ** -8, (k, v)
            elif v["kind"] == "attn_fwd":
                # one internal store point (P is rounded to bf16 before the P V product; v_exp_f32 vs the CPU's exp2 flips a rounding of P
                # now and then): the forward gate of tests/test_e2e_gpu.py::test_bf16_emulated_parity -- <= 1e-3 of the elements differ,
                # every error below one ulp of the tensor's largest element
                assert v["mismatch_frac"] <= 1e-3 and v["max_abs_err_over_tensor_max"] <= 2.0 ** -7, (k, v)
            elif v["kind"] == "two_stores":
                # an op with an internal bf16 store point (the base input gradient is stored, then read-modified-written by the adapter
                # branch): a one-ulp flip of the stored intermediate is many ulps of a final value that mostly cancelled -- same fraction
                # gate, every error below one ulp of the tensor's LARGEST element (as for the attention backward)
                assert v["mismatch_frac"] <= 1e-3 and v["max_abs_err_over_tensor_max"] <= 2.0 ** -8, (k, v)
            elif v["kind"] == "exact":
                assert v["max_ulp"] == 0.0, (k, v)
            else:
                assert v["max_ulp"] <= 1.0 and v["mismatch_frac"] <= 1e-3, (k, v)


def up(t):
    return t.reshape(-1, t.shape[-1]).to(torch.bfloat16).to(DEV).contiguous()


@pytest.mark.parametrize("case", ["clip_l14_336_layer", "siglip_so400m_layer"])
def test_tower_layer_ops_on_bf16_exact_inputs(case):
    _need_gpu()
    from oracle import bf16_emulation as E
    from radvlm_amd import ops
    from radvlm_amd.engine import LlavaEngine
    rnd = E.bf16_round
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    gen = torch.Generator().manual_seed(41)
    siglip = case.startswith("siglip")
    vis = dict(d=1152, heads=16, ffn=4304, layers=2, image=378, patch=14, kind="siglip") if siglip else dict(d=1024, heads=16, ffn=4096, layers=2, image=336, patch=14)
    geo = {"vision": vis, "lm": dict(d=256, heads=2, ffn=512, layers=1, vocab=2048)}
    eng = LlavaEngine(geo, device=DEV, init="fast", seed=3, train_vision_tower=True)
    p = E.VP + "encoder.layers.0."
    for nm in eng.vis.names():       # fast init leaves norms at one and biases at zero: give them values
        if nm.startswith(p) and len(eng.vis.shapes[nm]) == 1:
            base = 1.0 if nm.endswith("norm1.weight") or nm.endswith("norm2.weight") else 0.0
            eng.vis.view(nm).copy_((base + torch.randn(eng.vis.shapes[nm], generator=gen) * 0.1).to(DEV))
    eng.weights_changed()
    Pr = {k: v.float().cpu() for k, v in eng.state_dict().items() if k.startswith(p)}
    dv, H = vis["d"], vis["heads"]
    hd, hp = dv // H, eng.vhd_pad
    n, N = 2, eng.N_vis
    x = rnd(torch.randn(n, N, dv, generator=gen))
    with torch.no_grad():
        y, T = E.vision_layer(x, Pr, p, vis, rnd, fused_act=False)
        _, Tf = E.vision_layer(x, Pr, p, vis, rnd, fused_act=True)
        dx_out = rnd(torch.randn(n, N, dv, generator=gen) * 2e-3)
        R = E.vision_layer_backward(T, Pr, p, vis, dx_out, rnd)
    G = Gate(rnd)
    W = lambda nm: eng.vis.view(p + nm)
    eps = eng.ln_eps
    n_pad = (N + 63) // 64 * 64
    dvp = H * hp

    def pad_heads(t, parts):          # [rows, parts * H * hd] -> [rows, parts * H * hp] (zero lanes), as the padded projection weights produce it
        t = t.reshape(-1, parts, H, hd)
        o = torch.zeros(t.shape[0], parts, H, hp)
        o[..., :hd] = t
        return o.reshape(t.shape[0], parts * H * hp)

    def unpad(t, parts):
        return t.detach().float().cpu().reshape(-1, parts, H, hp)[..., :hd].reshape(-1, parts * H * hd)

    def stats(t):      # (mean, rstd) rows as the forward kernel leaves them
        t = t.reshape(-1, t.shape[-1]).float()
        mean = t.mean(-1)
        return torch.stack((mean, torch.rsqrt((t - mean[:, None]).pow(2).mean(-1) + eps)), 1).to(DEV).contiguous()

    act_code = ops.ACT_GELU_TANH if siglip else ops.ACT_QUICK_GELU
    act_fwd, act_bwd = (ops.gelu_tanh_fwd, ops.gelu_tanh_bwd) if siglip else (ops.quick_gelu_fwd, ops.quick_gelu_bwd)
    wqkv, bqkv, wo = eng._vis_attn_weights(0)
    # ---------------------------------------------------------------- forward
    x_e = up(x)
    G.bf("layernorm 1", ops.layernorm_fwd(x_e, W("layer_norm1.weight"), W("layer_norm1.bias"), eps=eps), T["h"])
    qkv_hip = ops.gemm_nt(up(T["h"]), wqkv, bias=bqkv)
    G.bf("q|k|v gemm + bias", unpad(qkv_hip, 3), T["qkv"])
    qkv_e = up(pad_heads(T["qkv"], 3))
    if hp == 128:
        a_hip, lse_hip = ops.attn_fwd(qkv_e[:, :dvp], qkv_e[:, dvp:2 * dvp], None, n, N, H, hp, n_pad, causal=False, scale=hd ** -0.5, v=qkv_e[:, 2 * dvp:])
    else:
        vT = ops.transpose_heads(qkv_e[:, 2 * dvp:], n, N, H, hp, n_pad)
        a_hip, lse_hip = ops.attn_fwd(qkv_e[:, :dvp], qkv_e[:, dvp:2 * dvp], vT, n, N, H, hp, n_pad, causal=False, scale=hd ** -0.5)
    G.bf("attention fwd (non-causal)", unpad(a_hip, 1), T["a"], kind="attn_fwd")
    a_e = up(pad_heads(T["a"], 1))
    G.bf("out_proj + bias + residual", ops.gemm_nt(a_e, wo, bias=W("self_attn.out_proj.bias"), residual=x_e), T["x1"])
    x1_e = up(T["x1"])
    G.bf("layernorm 2", ops.layernorm_fwd(x1_e, W("layer_norm2.weight"), W("layer_norm2.bias"), eps=eps), T["h2"])
    h2_e = up(T["h2"])
    G.bf("fc1 gemm + bias", ops.gemm_nt(h2_e, W("mlp.fc1.weight"), bias=W("mlp.fc1.bias")), T["z"])
    G.bf("activation fwd", act_fwd(up(T["z"])), T["g"])
    G.bf("fc1 gemm + bias + activation epilogue (frozen tower)", ops.gemm_nt(h2_e, W("mlp.fc1.weight"), bias=W("mlp.fc1.bias"), act=act_code), Tf["g"])
    g_e = up(T["g"])
    G.bf("fc2 gemm + bias + residual", ops.gemm_nt(g_e, W("mlp.fc2.weight"), bias=W("mlp.fc2.bias"), residual=x1_e), y)
    # ---------------------------------------------------------------- backward (the order of LlavaEngine.vision_backward)
    dxo = up(dx_out)
    row = lambda t: t.reshape(1, -1)
    G.bf("bias grad fc2", row(ops.bias_grad(dxo)), row(R["g_fc2_b"]), kind="colsum")
    G.f32("wgrad fc2", ops.gemm(dxo, g_e, ta=True, tb=True, out_dtype=F32), R["gW_fc2"])
    G.bf("dgrad fc2", ops.gemm(dxo, W("mlp.fc2.weight"), tb=True), R["dg"])
    G.bf("activation bwd", act_bwd(up(R["dg"]), up(T["z"])), R["dz"])
    dz_e = up(R["dz"])
    G.bf("bias grad fc1", row(ops.bias_grad(dz_e)), row(R["g_fc1_b"]), kind="colsum")
    G.f32("wgrad fc1", ops.gemm(dz_e, h2_e, ta=True, tb=True, out_dtype=F32), R["gW_fc1"])
    G.bf("dgrad fc1", ops.gemm(dz_e, W("mlp.fc1.weight"), tb=True), R["dh2"])
    gw, gb = torch.empty(dv, dtype=torch.bfloat16, device=DEV), torch.empty(dv, dtype=torch.bfloat16, device=DEV)
    dx1 = ops.layernorm_bwd(up(R["dh2"]), x1_e, W("layer_norm2.weight"), stats(T["x1"]), gw, gb, dx=dxo.clone(), dx_add=True)
    G.bf("layernorm bwd 2 (dx accumulate)", dx1, R["dx1"])
    G.bf("layernorm bwd 2 (dw)", row(gw), row(R["g_ln2_w"]), kind="colsum")
    G.bf("layernorm bwd 2 (db)", row(gb), row(R["g_ln2_b"]), kind="colsum")
    dx1_e = up(R["dx1"])
    G.bf("bias grad out_proj", row(ops.bias_grad(dx1_e)), row(R["g_out_b"]), kind="colsum")
    gwo = ops.gemm(dx1_e, a_e, ta=True, tb=True, out_dtype=F32)
    G.f32("wgrad out_proj", gwo.view(dv, H, hp)[:, :, :hd].reshape(dv, dv), R["gW_out"])
    G.bf("dgrad out_proj", unpad(ops.gemm(dx1_e, wo, tb=True), 1), R["da"])
    da_e = up(pad_heads(R["da"], 1))
    lse = torch.zeros(n, H, n_pad, dtype=F32)
    lse[:, :, :N] = R["lse"]
    dqkv = torch.empty_like(qkv_e)
    ops.attn_bwd(qkv_e[:, :dvp], qkv_e[:, dvp:2 * dvp], qkv_e[:, 2 * dvp:], a_e, da_e, lse.to(DEV), n, N, H, hp, n_pad, False, scale=hd ** -0.5,
                 dq=dqkv[:, :dvp], dk=dqkv[:, dvp:2 * dvp], dv=dqkv[:, 2 * dvp:])
    dq_hip = unpad(dqkv, 3)
    for j, nm in enumerate(("dQ", "dK", "dV")):
        G.bf(f"attention bwd {nm} (non-causal)", dq_hip[:, j * dv:(j + 1) * dv], R["dqkv"][..., j * dv:(j + 1) * dv], kind="attn_bwd", floor_exp=-4)
    G.f32("attention fwd lse", lse_hip[:, :, :N], R["lse"])
    dqkv_e = up(pad_heads(R["dqkv"], 3))
    G.bf("bias grad q|k|v", row(unpad(ops.bias_grad(dqkv_e).view(1, -1), 3)), row(R["g_bqkv"]), kind="colsum")
    gwq = ops.gemm(dqkv_e, up(T["h"]), ta=True, tb=True, out_dtype=F32)
    G.f32("wgrad q|k|v", gwq.view(3, H, hp, dv)[:, :, :hd].reshape(3 * dv, dv), R["gW_qkv"])
    G.bf("dgrad q|k|v", ops.gemm(dqkv_e, wqkv, tb=True), R["dh"])
    dx0 = ops.layernorm_bwd(up(R["dh"]), x_e, W("layer_norm1.weight"), stats(x), gw, gb, dx=dx1_e.clone(), dx_add=True)
    G.bf("layernorm bwd 1 (dx accumulate)", dx0, R["dx_in"])
    G.bf("layernorm bwd 1 (dw)", row(gw), row(R["g_ln1_w"]), kind="colsum")
    G.bf("layernorm bwd 1 (db)", row(gb), row(R["g_ln1_b"]), kind="colsum")
    torch.cuda.synchronize()
    G.check(case)


def test_projector_head_and_splice_backward_on_bf16_exact_inputs():
    _need_gpu()
    from oracle import bf16_emulation as E
    from radvlm_amd import lib, ops
    rnd = E.bf16_round
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    gen = torch.Generator().manual_seed(43)
    G = Gate(rnd)
    row = lambda t: t.reshape(1, -1)
    # ---- mlp2x_gelu projector at LLaVA-1.5 widths (1024 -> 4096 -> 4096), two images of 576 patch rows
    dv, d, rows = 1024, 4096, 1152
    P = {"model.mm_projector.0.weight": rnd(torch.randn(d, dv, generator=gen) * 0.02), "model.mm_projector.0.bias": rnd(torch.randn(d, generator=gen) * 0.05),
         "model.mm_projector.2.weight": rnd(torch.randn(d, d, generator=gen) * 0.02), "model.mm_projector.2.bias": rnd(torch.randn(d, generator=gen) * 0.05)}
    Pd = {k: v.to(torch.bfloat16).to(DEV) for k, v in P.items()}
    f0 = rnd(torch.randn(rows, dv, generator=gen))
    E.TRACE = {}
    try:
        with torch.no_grad():
            proj = E.mm_projector(P, f0, rnd)
        T = E.TRACE
    finally:
        E.TRACE = None
    dproj = rnd(torch.randn(rows, d, generator=gen) * 2e-3)
    with torch.no_grad():
        R = E.mm_projector_backward(T, P, dproj, rnd)
    w0, b0, w2, b2 = (Pd["model.mm_projector." + k] for k in ("0.weight", "0.bias", "2.weight", "2.bias"))
    G.bf("projector fc0 + bias", ops.gemm_nt(up(f0), w0, bias=b0), T["z1"])
    G.bf("projector gelu fwd", ops.gelu_fwd(up(T["z1"])), T["a1"])
    G.bf("projector fc2 + bias", ops.gemm_nt(up(T["a1"]), w2, bias=b2), proj)
    dp = up(dproj)
    G.bf("projector bias grad 2", row(ops.bias_grad(dp)), row(R["g_b2"]), kind="colsum")
    G.f32("projector wgrad 2", ops.gemm(dp, up(T["a1"]), ta=True, tb=True, out_dtype=F32), R["gW2"])
    G.bf("projector dgrad 2", ops.gemm(dp, w2, tb=True), R["da1"])
    G.bf("projector gelu bwd", ops.gelu_bwd(up(R["da1"]), up(T["z1"])), R["dz1"])
    dz1 = up(R["dz1"])
    G.bf("projector bias grad 0", row(ops.bias_grad(dz1)), row(R["g_b0"]), kind="colsum")
    G.f32("projector wgrad 0", ops.gemm(dz1, up(f0), ta=True, tb=True, out_dtype=F32), R["gW0"])
    G.bf("projector dgrad 0", ops.gemm(dz1, w0, tb=True), R["df0"])
    # ---- lm_head + cross entropy: hidden 4096, an 8192-row vocabulary slice, 512 token rows, some ignored
    V, M = 8192, 512
    hN = rnd(torch.randn(M, d, generator=gen))
    wh = rnd(torch.randn(V, d, generator=gen) * 0.02)
    tgt = torch.randint(0, V, (M,), generator=gen)
    tgt[:37] = -100
    gscale = 0.5
    with torch.no_grad():
        Hd = E.lm_head_cross_entropy(hN, wh, tgt, gscale, rnd)
    wh_d, hN_d = wh.to(torch.bfloat16).to(DEV), up(hN)
    logits = ops.gemm_nt(hN_d, wh_d)
    G.bf("lm_head gemm", logits, Hd["logits"])
    lg = up(Hd["logits"])
    count = int((tgt != -100).sum())
    loss_rows = torch.empty(M, dtype=F32, device=DEV)
    lib.call("rv_cross_entropy", lg, lg.stride(0), tgt.to(DEV), loss_rows, lg, lg.stride(0), M, V, gscale / count)
    G.f32("cross entropy loss rows", loss_rows, Hd["loss_rows"])
    G.bf("cross entropy dlogits (in place)", lg, Hd["dlogits"])
    dl = up(Hd["dlogits"])
    G.bf("lm_head dgrad", ops.gemm(dl, wh_d, tb=True), Hd["dhN"])
    G.f32("lm_head wgrad", ops.gemm(dl, hN_d, ta=True, tb=True, out_dtype=F32), Hd["gW_head"])
    # ---- splice backward: plain and weighted segment sums (embedding rows / bilinear taps), 2x2 max-pool adjoint
    src = rnd(torch.randn(900, 512, generator=gen) * 1e-2)
    src_d = up(src)
    nseg = 150
    cuts = np.sort(np.random.default_rng(5).choice(np.arange(1, 900), nseg - 1, replace=False))
    seg_off = np.concatenate([[0], cuts, [900]]).astype(np.int32)
    pos = np.random.default_rng(6).permutation(900).astype(np.int32)
    out_row = np.random.default_rng(7).permutation(nseg).astype(np.int32)
    ref = torch.zeros(nseg, 512, dtype=torch.float64)
    wts = torch.rand(900, generator=gen)
    refw = torch.zeros(nseg, 512, dtype=torch.float64)
    for s_ in range(nseg):
        j = torch.from_numpy(pos[seg_off[s_]:seg_off[s_ + 1]].astype(np.int64))
        ref[out_row[s_]] = src[j].double().sum(0)
        refw[out_row[s_]] = (src[j].double() * wts[seg_off[s_]:seg_off[s_ + 1], None].double()).sum(0)
    t = lambda a: torch.from_numpy(a).to(DEV)
    G.bf("rv_segment_sum_rows", ops.segment_sum_rows(src_d, t(seg_off), t(pos), t(out_row), torch.zeros(nseg, 512, dtype=torch.bfloat16, device=DEV)), ref.float())
    G.bf("rv_weighted_segment_sum_rows", ops.weighted_segment_sum_rows(src_d, t(seg_off), t(pos), wts.to(DEV), t(out_row),
                                                                      torch.zeros(nseg, 512, dtype=torch.bfloat16, device=DEV)), refw.float())
    npool = 225
    idx4 = np.random.default_rng(8).permutation(900).astype(np.int32).reshape(npool, 4)
    prow = np.arange(npool, dtype=np.int32)
    pooled = torch.zeros(npool, 512, dtype=torch.bfloat16, device=DEV)
    which = ops.max4_rows_fwd(src_d, t(idx4.reshape(-1)), t(prow), pooled)
    win = src[torch.from_numpy(idx4.astype(np.int64))]                     # [npool, 4, 512]
    G.bf("rv_max4_rows_fwd", pooled, win.max(1).values, kind="exact")
    dpool = rnd(torch.randn(npool, 512, generator=gen))
    dsrc = torch.full((900, 512), 7.0, dtype=torch.bfloat16, device=DEV)
    ops.max4_rows_bwd(up(dpool), t(idx4.reshape(-1)), t(prow), which, dsrc)
    arg = win.argmax(1)                                                      # first maximum, as the kernel's scan order
    refd = torch.zeros(900, 512)
    for j in range(4):
        refd[torch.from_numpy(idx4[:, j].astype(np.int64))] = torch.where(arg == j, dpool, torch.zeros_like(dpool))
    G.bf("rv_max4_rows_bwd", dsrc, refd, kind="exact")
    torch.cuda.synchronize()
    G.check("projector_head_splice")


@pytest.mark.parametrize("pdrop", [0.0, 0.05])
def test_lora_kernels_against_the_emulation_with_the_same_mask(pdrop):
    """One adapted linear at Vicuna-13B widths (d 5120, r 64, alpha 16: BASELINE config 5), forward and backward."""
    _need_gpu()
    from oracle import bf16_emulation as E
    from radvlm_amd import ops
    rnd = E.bf16_round
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    gen = torch.Generator().manual_seed(47)
    M, K, N, r, scale, seed = 704, 5120, 5120, 64, 16.0 / 64, 918273645
    x = rnd(torch.randn(M, K, generator=gen))
    Wb = rnd(torch.randn(N, K, generator=gen) * 0.02)
    A = rnd((torch.rand(r, K, generator=gen) * 2 - 1) / math.sqrt(K))
    Bm = rnd(torch.randn(N, r, generator=gen) * 0.02)
    res = rnd(torch.randn(M, N, generator=gen))
    dy = rnd(torch.randn(M, N, generator=gen) * 2e-3)
    with torch.no_grad():
        y, t = E.lora_linear(x, Wb, A, Bm, scale, pdrop, seed, rnd, residual=res)
        Rb = E.lora_linear_backward(dy, x, Wb, A, Bm, t, scale, pdrop, seed, rnd)
    G = Gate(rnd)
    dev = lambda u: u.to(torch.bfloat16).to(DEV).contiguous()
    x_d, W_d, A_d, B_d = dev(x), dev(Wb), dev(A), dev(Bm)
    if pdrop > 0:
        keep = E.dropout_keep_mask(x.shape, pdrop, seed)
        G.bf("rv_dropout_bf16 (mask + scale)", ops.dropout(x_d, pdrop, seed), rnd(x * keep / (1.0 - pdrop)))
    G.bf("rv_lora_down_bf16 (mask on the operand fragments)", ops.lora_down(x_d, A_d, scale, pdrop, seed), t)
    t_d = dev(t)
    G.bf("fused second operand pair: x W^T + t B^T + residual", ops.gemm(x_d, W_d, residual=dev(res), a2=t_d, b2=B_d), y)
    dy_d = dev(dy)
    G.bf("d(t) = scale dy B (skinny one-pass kernel)", ops.lora_down(dy_d, ops.transpose(B_d), scale, 0.0, 0), Rb["dts"])
    dts_d = dev(Rb["dts"])
    G.f32("wgrad lora_B", ops.gemm(dy_d, t_d, ta=True, tb=True, out_dtype=F32), Rb["gB"])
    if pdrop > 0:      # the product path: one pass over x, the forward's mask re-created in registers, fp32 token-slice sums, one bf16 store
        ws = torch.empty(8 << 20, dtype=F32, device=DEV)
        g_a = ops.lora_a_grad(dts_d, x_d, torch.empty(r, K, dtype=torch.bfloat16, device=DEV), pdrop, seed, False, ws)
        G.bf("rv_lora_a_grad_bf16: wgrad lora_A, mask re-created in registers", g_a, rnd(Rb["gA"]))
    else:
        G.f32("wgrad lora_A", ops.gemm(dts_d, x_d, ta=True, tb=True, out_dtype=F32), Rb["gA"])
    if pdrop > 0:
        dx = ops.gemm(dy_d, W_d, tb=True)
        ops.gemm_dropout_add(dts_d, A_d, dx, pdrop, seed)
        G.bf("dgrad: dy W, then += dropout'(dts A) in the GEMM epilogue", dx, Rb["dx"], kind="two_stores")
    else:
        G.bf("dgrad: dy W + dts A as one fused-pair GEMM", ops.gemm(dy_d, W_d, tb=True, a2=dts_d, b2=A_d), Rb["dx"])
    torch.cuda.synchronize()
    G.check(f"lora_linear_p{pdrop}")
