"""Thin typed wrappers over the C ABI (radvlm_amd.lib): shape checks on the host, then one kernel launch.

Names follow the reference's operators.  No op here has a torch fallback.
"""
import math

import torch

from . import lib
from .lib import ACT_GELU, ACT_GELU_TANH, ACT_NONE, ACT_QUICK_GELU  # noqa: F401

BF16 = torch.bfloat16


def _chk(t, dtype=BF16):
    assert t.is_cuda and t.dtype == dtype, (t.device, t.dtype)
    return t


def round_up(x, m):
    return (x + m - 1) // m * m


_WS = {}


def default_workspace(device):
    """128 MiB fp32 scratch per device for the GEMM's split-K / tail-split modes (stream-ordered reuse)."""
    key = (device.type, device.index)
    if key not in _WS:
        _WS[key] = torch.empty(32 << 20, dtype=torch.float32, device=device)
    return _WS[key]


def gemm_nt(a, b, out=None, bias=None, residual=None, act=ACT_NONE, out_dtype=BF16):
    """out[M,N] = act(a[M,K] @ b[N,K]^T + bias) + residual.  a, b: 2-D bf16 views with unit inner stride."""
    return gemm(a, b, out=out, bias=bias, residual=residual, act=act, out_dtype=out_dtype)


def _gemm_nt_direct(a, b, out=None, bias=None, residual=None, act=ACT_NONE, out_dtype=BF16):
    """rv_gemm_nt_bf16 entry point (no scratch: plain tiles only); kept for ABI coverage in the tests."""
    _chk(a), _chk(b)
    M, K = a.shape
    N, K2 = b.shape
    assert K == K2 and a.stride(1) == 1 and b.stride(1) == 1
    if out is None:
        out = torch.empty(M, N, dtype=out_dtype, device=a.device)
    assert out.shape == (M, N) and out.stride(1) == 1 and out.dtype in (BF16, torch.float32)
    if bias is not None:
        _chk(bias)
        assert bias.numel() == N and bias.is_contiguous()
    res_f32 = 0
    ldr = 0
    if residual is not None:
        assert residual.shape == (M, N) and residual.stride(1) == 1
        res_f32 = int(residual.dtype == torch.float32)
        ldr = residual.stride(0)
    lib.call("rv_gemm_nt_bf16", a, a.stride(0), b, b.stride(0), out, out.stride(0), bias, residual, ldr, M, N, K, act,
             int(out.dtype == torch.float32), res_f32, lib.zeros16(a.device))
    return out


def gemm(a, b, ta=False, tb=False, out=None, bias=None, residual=None, act=ACT_NONE, out_dtype=BF16, alpha=1.0, a2=None, b2=None,
         workspace=None):
    """out[M,N] = act(op(a) @ op(b)^T + bias) + residual with a stored [K,M] if ta else [M,K], b stored [K,N] if tb else [N,K]."""
    _chk(a), _chk(b)
    assert a.stride(1) == 1 and b.stride(1) == 1
    (K, M) = a.shape if ta else a.shape[::-1]
    (K2, N) = b.shape if tb else b.shape[::-1]
    assert K == K2, (a.shape, b.shape, ta, tb)
    if out is None:
        out = torch.empty(M, N, dtype=out_dtype, device=a.device)
    assert out.shape == (M, N) and out.stride(1) == 1 and out.dtype in (BF16, torch.float32)
    res_f32, ldr = 0, 0
    if residual is not None:
        assert residual.shape == (M, N) and residual.stride(1) == 1
        res_f32, ldr = int(residual.dtype == torch.float32), residual.stride(0)
    if bias is not None:
        assert bias.numel() == N and bias.is_contiguous()
    if workspace is None:
        workspace = default_workspace(a.device)
    # fused second operand pair (a2 like a, b2 like b, contraction K2) and/or split-K scratch
    K2 = 0
    if a2 is not None:
        _chk(a2), _chk(b2)
        assert a2.stride(1) == 1 and b2.stride(1) == 1
        (K2, M2) = a2.shape if ta else a2.shape[::-1]
        (K2b, N2) = b2.shape if tb else b2.shape[::-1]
        assert M2 == M and N2 == N and K2 == K2b
    ws_bytes = workspace.numel() * workspace.element_size() if workspace is not None else 0
    lib.call("rv_gemm_bf16_ex", a, a.stride(0), b, b.stride(0), out, out.stride(0), bias, residual, ldr, M, N, K, int(ta), int(tb),
             float(alpha), act, int(out.dtype == torch.float32), res_f32, a2, a2.stride(0) if a2 is not None else 0, b2,
             b2.stride(0) if b2 is not None else 0, K2, workspace, ws_bytes, lib.zeros16(a.device))
    return out


def gemm_rope(a, b, cos_sin, S, rope_heads, hd, bias=None, positions=None, out=None):
    """out[M,N] = rope(a[M,K] @ b[N,K]^T + bias): rotary embedding applied to the first rope_heads heads (q then k) in the epilogue."""
    _chk(a), _chk(b)
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K and a.stride(1) == 1 and b.stride(1) == 1
    if out is None:
        out = torch.empty(M, N, dtype=BF16, device=a.device)
    assert out.shape == (M, N) and out.stride(1) == 1 and out.dtype == BF16
    ws = default_workspace(a.device)
    lib.call("rv_gemm_rope_bf16", a, a.stride(0), b, b.stride(0), out, out.stride(0), bias, M, N, K, cos_sin, positions, S, rope_heads, hd,
             ws, ws.numel() * ws.element_size(), lib.zeros16(a.device))
    return out


def gemm_swiglu_fwd(a, wgu, F, gu=None, act=None):
    """(gu [M,2F], act [M,F]) = (a @ wgu^T, silu(gate) * up) in one launch; wgu = stacked [gate; up] rows [2F, K]."""
    _chk(a), _chk(wgu)
    M, K = a.shape
    assert wgu.shape == (2 * F, K) and a.stride(1) == 1 and wgu.stride(1) == 1
    gu = torch.empty(M, 2 * F, dtype=BF16, device=a.device) if gu is None else gu
    act = torch.empty(M, F, dtype=BF16, device=a.device) if act is None else act
    ws = default_workspace(a.device)
    lib.call("rv_gemm_swiglu_fwd_bf16", a, a.stride(0), wgu, wgu.stride(0), gu, gu.stride(0), act, act.stride(0), M, F, K,
             ws, ws.numel() * ws.element_size(), lib.zeros16(a.device))
    return gu, act


def gemm_swiglu_bwd(dy, wd, gu, F, dgu=None):
    """dgu [M,2F] = swiglu'(gu) * (dy [M,d] @ wd [d,F]): down_proj's input gradient with the activation backward in the epilogue."""
    _chk(dy), _chk(wd), _chk(gu)
    M, K = dy.shape
    assert wd.shape == (K, F) and gu.shape == (M, 2 * F) and dy.stride(1) == 1 and wd.stride(1) == 1 and gu.stride(1) == 1
    dgu = torch.empty_like(gu) if dgu is None else dgu
    ws = default_workspace(dy.device)
    # d(act) as a tensor exists only in the unfused fallback (small shapes): allocate it there, not for the 7B step
    small = M * F <= (1 << 24)
    scratch = torch.empty(M, F, dtype=BF16, device=dy.device) if small else None
    try:
        lib.call("rv_gemm_swiglu_bwd_bf16", dy, dy.stride(0), wd, wd.stride(0), gu, gu.stride(0), dgu, dgu.stride(0), scratch,
                 F if small else 0, M, F, K, ws, ws.numel() * ws.element_size(), lib.zeros16(dy.device))
    except lib.RadvlmHipError:
        if small:
            raise
        scratch = torch.empty(M, F, dtype=BF16, device=dy.device)     # a large shape that still took the fallback (forced kernel choice)
        lib.call("rv_gemm_swiglu_bwd_bf16", dy, dy.stride(0), wd, wd.stride(0), gu, gu.stride(0), dgu, dgu.stride(0), scratch, F, M, F, K,
                 ws, ws.numel() * ws.element_size(), lib.zeros16(dy.device))
    return dgu


def transpose(x, r_pad=None, out=None):
    """x [R,C] (unit inner stride) -> [C, r_pad] with zero padding."""
    _chk(x)
    R, C = x.shape
    r_pad = r_pad or round_up(R, 8)
    if out is None:
        out = torch.empty(C, r_pad, dtype=BF16, device=x.device)
    assert out.shape == (C, r_pad) and out.stride(1) == 1 and x.stride(1) == 1
    lib.call("rv_transpose_bf16", x, x.stride(0), 0, 0, out, out.stride(0), 0, 0, R, C, r_pad, 1, 1, 0)
    return out


def transpose_heads(x, B, S, H, hd, s_pad, out=None, perm32=True, cu=None):
    """x: token-major view [(b*S+s), H*hd] (row stride ld) -> [B,H,hd,s_pad] (zero padded along s).
    perm32 (what the attention kernels expect): the sequence axis is stored in MFMA contraction order per group of 32.
    cu (int32 [B+1], packed batches): sample b owns rows [cu[b], cu[b+1]) of x; S = the longest sample."""
    _chk(x)
    assert x.shape[1] == H * hd and x.stride(1) == 1 and (cu is not None or x.shape[0] == B * S)
    if out is None:
        out = torch.empty(B, H, hd, s_pad, dtype=BF16, device=x.device)
    ld = x.stride(0)
    if cu is not None:
        lib.call("rv_transpose_bf16_varlen", x, ld, cu, hd, out, s_pad, H * hd * s_pad, hd * s_pad, S, hd, s_pad, B, H, int(perm32))
    else:
        lib.call("rv_transpose_bf16", x, ld, S * ld, hd, out, s_pad, H * hd * s_pad, hd * s_pad, S, hd, s_pad, B, H, int(perm32))
    return out


def rmsnorm_fwd(x, w, eps=1e-5, y=None, rstd=None):
    _chk(x), _chk(w)
    rows, d = x.shape
    assert x.is_contiguous()
    y = torch.empty_like(x) if y is None else y
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if rstd is None else rstd
    lib.call("rv_rmsnorm_fwd", x, w, y, rstd, rows, d, eps)
    return y, rstd


def rmsnorm_bwd(dy, x, w, rstd, dx=None, dx_add=False, dw=None, dw_accumulate=False):
    rows, d = x.shape
    assert dy.is_contiguous() and x.is_contiguous()
    nblk = min(rows, 1024)
    part = torch.empty(nblk, d, dtype=torch.float32, device=x.device)
    if dx is None:
        dx = torch.empty_like(x)
        dx_add = False
    lib.call("rv_rmsnorm_bwd", dy, x, w, rstd, dx, int(dx_add), part, nblk, rows, d)
    if dw is None:
        dw = torch.empty(d, dtype=BF16, device=x.device)
        dw_accumulate = False
    lib.call("rv_colsum_f32", part, nblk, d, dw, int(dw_accumulate))
    return dx, dw


def layernorm_fwd(x, w, b, eps=1e-5, y=None, save_stats=False):
    rows, d = x.shape
    assert x.is_contiguous()
    y = torch.empty_like(x) if y is None else y
    stats = torch.empty(rows, 2, dtype=torch.float32, device=x.device) if save_stats else None
    lib.call("rv_layernorm_fwd", x, w, b, y, stats, rows, d, eps)
    return (y, stats) if save_stats else y


def layernorm_bwd(dy, x, w, stats, dw, db, dx=None, dx_add=False, accumulate=False):
    """dx (+)= LN'(dy); dw/db (bf16 [d] views) (+)= parameter gradients."""
    rows, d = x.shape
    assert dy.is_contiguous() and x.is_contiguous()
    nblk = min(rows, 512)
    part = torch.empty(nblk, 2 * d, dtype=torch.float32, device=x.device)
    if dx is None:
        dx = torch.empty_like(x)
        dx_add = False
    lib.call("rv_layernorm_bwd", dy, x, w, stats, dx, int(dx_add), part, nblk, rows, d)
    wb = torch.empty(2 * d, dtype=BF16, device=x.device)
    lib.call("rv_colsum_f32", part, nblk, 2 * d, wb, 0)
    if accumulate:
        lib.call("rv_add_bf16", dw, wb[:d], dw, d)
        lib.call("rv_add_bf16", db, wb[d:], db, d)
    else:
        dw.copy_(wb[:d])
        db.copy_(wb[d:])
    return dx


def quick_gelu_fwd(x, y=None):
    assert x.is_contiguous()
    y = torch.empty_like(x) if y is None else y
    lib.call("rv_quick_gelu_fwd", x, y, x.numel())
    return y


def quick_gelu_bwd(dy, x, dx=None):
    assert x.is_contiguous() and dy.is_contiguous()
    dx = torch.empty_like(x) if dx is None else dx
    lib.call("rv_quick_gelu_bwd", dy, x, dx, x.numel())
    return dx


def gelu_tanh_fwd(x, y=None):
    assert x.is_contiguous()
    y = torch.empty_like(x) if y is None else y
    lib.call("rv_gelu_tanh_fwd", x, y, x.numel())
    return y


def gelu_tanh_bwd(dy, x, dx=None):
    assert x.is_contiguous() and dy.is_contiguous()
    dx = torch.empty_like(x) if dx is None else dx
    lib.call("rv_gelu_tanh_bwd", dy, x, dx, x.numel())
    return dx


def bias_grad(dy, out=None, accumulate=False):
    """db[c] = sum_r dy[r, c]."""
    rows, cols = dy.shape
    nblk = min(rows, 256)
    part = torch.empty(nblk, cols, dtype=torch.float32, device=dy.device)
    lib.call("rv_colsum_partial_bf16", dy, dy.stride(0), rows, cols, part, nblk)
    if out is None:
        out = torch.empty(cols, dtype=BF16, device=dy.device)
        accumulate = False
    lib.call("rv_colsum_f32", part, nblk, cols, out, int(accumulate))
    return out


def rope_table(S, hd, theta=10000.0, device="cuda", round_bf16=True):
    """fp32 [S, hd/2, 2] (cos, sin); computed like LlamaRotaryEmbedding.forward (modeling_llama.py:123-139):
    fp32 trig, then rounded through bf16 like the reference's ``.to(dtype=x.dtype)``."""
    inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.int64).float() / hd))
    fr = torch.outer(torch.arange(S, dtype=torch.float32), inv)
    cs = torch.stack((fr.cos(), fr.sin()), dim=-1)
    if round_bf16:
        cs = cs.to(BF16).float()
    return cs.contiguous().to(device)


def rope_inplace(x, cos_sin, S, heads, hd, nsec, direction=1, positions=None):
    """positions (int32 [rows], packed batches): explicit position of every token row; default row % S."""
    rows = x.shape[0]
    if positions is not None:
        lib.call("rv_rope_inplace_pos", x, x.stride(0), cos_sin, positions, rows, heads, hd, nsec, direction)
    else:
        lib.call("rv_rope_inplace", x, x.stride(0), cos_sin, rows, S, heads, hd, nsec, direction)
    return x


def attn_fwd(q, k, vT, B, S, H, hd, s_pad, causal, lens=None, scale=None, out=None, lse=None, kv_heads=None, cu=None, v=None):
    """q, k: token-major [(B*S), H*hd] / [(B*S), kv_heads*hd] views; vT [B,kv_heads,hd,s_pad], or (head_dim 128) v = the
    token-major value view like k with vT=None: no transposed copy.  Returns (out [(B*S), H*hd], lse fp32 [B,H,s_pad])."""
    scale = scale if scale is not None else 1.0 / math.sqrt(hd)
    Hkv = kv_heads or H
    if out is None:
        out = torch.empty(q.shape[0], H * hd, dtype=BF16, device=q.device)
    if lse is None:
        lse = torch.zeros(B, H, s_pad, dtype=torch.float32, device=q.device)
    if v is not None:
        assert vT is None and hd == 128
        lib.call("rv_attn_fwd_nat", q, q.stride(0), k, k.stride(0), v, v.stride(0), out, out.stride(0), lse, lens, cu, B, H, Hkv, S, s_pad, hd,
                 int(causal), scale, lib.zeros16(q.device))
        return out, lse
    lib.call("rv_attn_fwd_gqa", q, q.stride(0), k, k.stride(0), vT, out, out.stride(0), lse, lens, cu, B, H, Hkv, S, s_pad, hd,
             int(causal), scale, lib.zeros16(q.device))
    return out, lse


def attn_bwd(q, k, v, o, dout, lse, B, S, H, hd, s_pad, causal, lens=None, scale=None, dq=None, dk=None, dv=None,
             kv_heads=None, use_workspace=True, cu=None, rope=None, natural=True, delta=None):
    """rope = (cos_sin table, positions or None): dq / dk are returned as gradients of the un-rotated q / k (the rotary embedding's
    adjoint runs in the epilogues)."""
    scale = scale if scale is not None else 1.0 / math.sqrt(hd)
    dev = q.device
    Hkv = kv_heads or H
    rows = q.shape[0]                       # B*S, or the packed row count cu[B]
    if hd == 128 and natural:
        delta = torch.zeros(B, H, s_pad, dtype=torch.float32, device=dev) if delta is None else delta
        dq = torch.empty(rows, H * hd, dtype=BF16, device=dev) if dq is None else dq
        dk = torch.empty(rows, Hkv * hd, dtype=BF16, device=dev) if dk is None else dk
        dv = torch.empty(rows, Hkv * hd, dtype=BF16, device=dev) if dv is None else dv
        ws = torch.empty(2 * rows * H * hd, dtype=BF16, device=dev) if (Hkv != H and use_workspace) else None
        lib.call("rv_attn_bwd_nat", q, q.stride(0), k, k.stride(0), v, v.stride(0), o, o.stride(0), dout, dout.stride(0), lse, delta,
                 dq, dq.stride(0), dk, dk.stride(0), dv, dv.stride(0), lens, cu, rows if cu is not None else 0, B, H, Hkv, S, s_pad, hd,
                 int(causal), scale, ws, ws.numel() * ws.element_size() if ws is not None else 0,
                 rope[0] if rope else None, rope[1] if rope else None, lib.zeros16(dev))
        return dq, dk, dv
    qT = transpose_heads(q, B, S, H, hd, s_pad, cu=cu)
    kT = transpose_heads(k, B, S, Hkv, hd, s_pad, cu=cu)
    doT = transpose_heads(dout, B, S, H, hd, s_pad, cu=cu)
    delta = torch.zeros(B, H, s_pad, dtype=torch.float32, device=dev) if delta is None else delta
    dq = torch.empty(rows, H * hd, dtype=BF16, device=dev) if dq is None else dq
    dk = torch.empty(rows, Hkv * hd, dtype=BF16, device=dev) if dk is None else dk
    dv = torch.empty(rows, Hkv * hd, dtype=BF16, device=dev) if dv is None else dv
    # per-query-head dK/dV partials for the small-grid grouped-query case (the C side decides whether to use it)
    ws = torch.empty(2 * rows * H * hd, dtype=BF16, device=dev) if (Hkv != H and use_workspace) else None
    lib.call("rv_attn_bwd_gqa_rope", q, q.stride(0), k, k.stride(0), v, v.stride(0), o, o.stride(0), dout, dout.stride(0), qT, kT, doT,
             lse, delta, dq, dq.stride(0), dk, dk.stride(0), dv, dv.stride(0), lens, cu, rows if cu is not None else 0, B, H, Hkv, S,
             s_pad, hd, int(causal), scale,
             ws, ws.numel() * ws.element_size() if ws is not None else 0, rope[0] if rope else None, rope[1] if rope else None, lib.zeros16(dev))
    return dq, dk, dv


def swiglu_fwd(gu, F, act=None):
    rows = gu.shape[0]
    if act is None:
        act = torch.empty(rows, F, dtype=BF16, device=gu.device)
    lib.call("rv_swiglu_fwd", gu, gu.stride(0), act, act.stride(0), rows, F)
    return act


def swiglu_bwd(dact, gu, F, dgu=None):
    rows = gu.shape[0]
    dgu = torch.empty_like(gu) if dgu is None else dgu
    lib.call("rv_swiglu_bwd", dact, dact.stride(0), gu, gu.stride(0), dgu, dgu.stride(0), rows, F)
    return dgu


def dropout(x, p, seed, y=None):
    """Inverted dropout with a regenerable counter-based mask (same (p, seed) in backward)."""
    assert x.is_contiguous()
    y = torch.empty_like(x) if y is None else y
    lib.call("rv_dropout_bf16", x, y, x.numel(), float(p), int(seed))
    return y


def lora_down(x, a, alpha, p, seed, out=None):
    """t[M, r] = alpha * dropout_p(x) @ a^T with the mask of dropout(x, p, seed), in one pass over x (x contiguous [M, K], K % 64 == 0,
    r <= 64); other shapes take the two-launch sequence."""
    M, K = x.shape
    r = a.shape[0]
    # rv_lora_down_bf16 takes 16-byte vector accesses: base pointers 16-byte aligned, adapter rows a multiple of 8 elements apart
    if not ((x.is_contiguous() or (p <= 0 and x.stride(1) == 1 and x.stride(0) % 8 == 0)) and K % 64 == 0 and r <= 64 and r % 4 == 0 and a.stride(1) == 1
            and x.data_ptr() % 16 == 0 and a.data_ptr() % 16 == 0 and a.stride(0) % 8 == 0):
        return gemm(dropout(x.contiguous(), p, seed) if p > 0 else x, a, alpha=alpha, out=out)
    if out is None:
        out = torch.empty(M, r, dtype=BF16, device=x.device)
    lib.call("rv_lora_down_bf16", x, x.stride(0), a, a.stride(0), out, out.stride(0), M, r, K, float(alpha), float(p), int(seed), lib.zeros16(x.device))
    return out


def lora_a_grad(dt, x, ga, p, seed, accumulate, workspace):
    """ga[r, K] (+)= dt[M, r]^T @ dropout(x, p, seed)[M, K] in one pass over x, the mask re-created in registers (rv_lora_a_grad_bf16: x contiguous,
    r <= 64, r % 8 == 0, K % 8 == 0); other shapes take the two-launch sequence (dropout kernel, split-K weight-gradient GEMM)."""
    M, K = x.shape
    r = dt.shape[1]
    ok = (x.is_contiguous() and dt.stride(1) == 1 and dt.stride(0) % 8 == 0 and ga.stride(1) == 1 and r <= 64 and r % 8 == 0 and K % 8 == 0
          and x.data_ptr() % 16 == 0 and dt.data_ptr() % 16 == 0 and workspace is not None and workspace.numel() * workspace.element_size() >= r * K * 4)
    if not ok:
        return gemm(dt, dropout(x.contiguous(), p, seed) if p > 0 else x, ta=True, tb=True, out=ga, residual=ga if accumulate else None, workspace=workspace)
    lib.call("rv_lora_a_grad_bf16", dt, dt.stride(0), x, x.stride(0), ga, ga.stride(0), M, r, K, float(p), int(seed), int(bool(accumulate)),
             workspace, workspace.numel() * workspace.element_size())
    return ga


def gemm_dropout_add(a, b, y, p, seed, tb=True, alpha=1.0, accumulate=True):
    """y (+)= dropout(alpha * a @ op(b)) with the mask of dropout(., p, seed) over y's elements, applied in the GEMM epilogue (the product is
    never stored unmasked).  a [M, K], b [K, N] (tb) or [N, K]; y [M, N] contiguous rows."""
    _chk(a), _chk(b), _chk(y)
    M, K = a.shape
    N = b.shape[1] if tb else b.shape[0]
    assert y.shape == (M, N) and (b.shape[0] if tb else b.shape[1]) == K and a.stride(1) == 1 and b.stride(1) == 1 and y.stride(1) == 1
    if p <= 0 or N % 8 or y.stride(0) != N or y.data_ptr() % 16:       # the mask indexes y as M * N contiguous elements, 16-byte accesses
        t = gemm(a, b, tb=tb, alpha=alpha)
        return dropout_add(t, y, p, seed) if p > 0 else y.add_(t) if accumulate else y.copy_(t)
    lib.call("rv_gemm_dropout_add_bf16", a, a.stride(0), b, b.stride(0), y, y.stride(0), M, N, K, int(tb), float(alpha), float(p), int(seed),
             int(accumulate), lib.zeros16(a.device))
    return y


def dropout_add(x, y, p, seed):
    """y += dropout(x) (same regenerable mask as dropout(x, p, seed)), one pass."""
    assert x.is_contiguous() and y.is_contiguous() and x.shape == y.shape
    lib.call("rv_dropout_add_bf16", x, y, x.numel(), float(p), int(seed))
    return y


def gelu_fwd(x, y=None):
    assert x.is_contiguous()
    y = torch.empty_like(x) if y is None else y
    lib.call("rv_gelu_fwd", x, y, x.numel())
    return y


def gelu_bwd(dy, x, dx=None):
    assert x.is_contiguous() and dy.is_contiguous()
    dx = torch.empty_like(x) if dx is None else dx
    lib.call("rv_gelu_bwd", dy, x, dx, x.numel())
    return dx


def cross_entropy(logits, labels_shifted, V, inv_count, grad_inplace=True):
    """Returns (loss scalar fp32 tensor [1], loss_rows). If grad_inplace, logits are overwritten by dlogits."""
    rows = logits.shape[0]
    loss_rows = torch.empty(rows, dtype=torch.float32, device=logits.device)
    dl = logits if grad_inplace else None
    lib.call("rv_cross_entropy", logits, logits.stride(0), labels_shifted, loss_rows, dl, logits.stride(0) if grad_inplace else 0,
             rows, V, inv_count)
    loss = torch.empty(1, dtype=torch.float32, device=logits.device)
    lib.call("rv_sum_f32", loss_rows, rows, inv_count, loss)
    return loss, loss_rows


def gather_rows(idx, d, table_a, table_b=None, out=None):
    rows = idx.numel()
    assert idx.dtype == torch.int32
    if out is None:
        out = torch.empty(rows, d, dtype=BF16, device=idx.device)
    lib.call("rv_gather_rows", out, out.stride(0), table_a, table_a.stride(0) if table_a is not None else 0, table_b,
             table_b.stride(0) if table_b is not None else 0, idx, rows, d)
    return out


def segment_sum_rows(src, seg_off, pos, out_row, out):
    nseg = out_row.numel()
    if nseg == 0:
        return out
    lib.call("rv_segment_sum_rows", src, src.stride(0), seg_off, pos, out_row, nseg, out, out.stride(0), src.shape[1])
    return out


def weighted_segment_sum_rows(src, seg_off, pos, w, out_row, out):
    """out[out_row[s]] = sum_{j in segment s} w[j] * src[pos[j]]  (bilinear taps / their adjoint)."""
    nseg = out_row.numel()
    if nseg == 0:
        return out
    lib.call("rv_weighted_segment_sum_rows", src, src.stride(0), seg_off, pos, w, out_row, nseg, out, out.stride(0), src.shape[1])
    return out


def max4_rows_fwd(src, idx4, out_row, out):
    """out[out_row[s]] = max over the four rows src[idx4[s]]; returns the per-element winning slot (uint8 [n, d])."""
    n, d = out_row.numel(), src.shape[1]
    which = torch.empty(n, d, dtype=torch.uint8, device=src.device)
    lib.call("rv_max4_rows_fwd", src, src.stride(0), idx4, out_row, n, out, out.stride(0), which, d)
    return which


def max4_rows_bwd(dout, idx4, dout_row, which, dsrc):
    lib.call("rv_max4_rows_bwd", dout, dout.stride(0), idx4, dout_row, dout_row.numel(), which, dsrc, dsrc.stride(0), dout.shape[1])
    return dsrc


def add_pos_rows(x, pos, n, P):
    """x [n*P, d] += pos [P, d] broadcast over images, in place."""
    assert x.is_contiguous() and pos.is_contiguous() and x.shape[0] == n * P and pos.shape[0] == P
    lib.call("rv_add_pos_rows", x, pos, n, P, x.shape[1])
    return x


def normalize_tiles_u8(img, tile, mean, std, mode=0, factor=1.0 / 255, gh=1, gw=1):
    """uint8 device canvases [n, gh*tile, gw*tile, 3] -> normalised bf16 tiles [n*gh*gw, 3, tile, tile] (host processors' fp32 arithmetic:
    mode 0 = CLIPImageProcessor, 1 = SigLipImageProcessor)."""
    import ctypes
    assert img.is_cuda and img.dtype == torch.uint8 and img.is_contiguous() and img.shape[1:] == (gh * tile, gw * tile, 3), img.shape
    n = img.shape[0]
    out = torch.empty(n * gh * gw, 3, tile, tile, dtype=BF16, device=img.device)
    m3, s3 = (ctypes.c_float * 3)(*[float(x) for x in mean]), (ctypes.c_float * 3)(*[float(x) for x in std])
    lib.call("rv_normalize_tiles_u8", img, out, n, gh, gw, tile, int(mode), float(factor), m3, s3)
    return out


def im2col_patches(pix, p, kp):
    n, c, H, W = pix.shape
    assert c == 3 and pix.is_contiguous()
    _chk(pix)
    out = torch.empty(n * (H // p) * (W // p), kp, dtype=BF16, device=pix.device)
    lib.call("rv_im2col_patches", pix, out, n, H, W, p, kp)
    return out


def clip_embed(patch_out, cls, pos, n, P, d):
    out = torch.empty(n * (P + 1), d, dtype=BF16, device=patch_out.device)
    lib.call("rv_clip_embed", patch_out, cls, pos, out, n, P, d)
    return out


def adamw(p, master, g, m, v, lr, b1, b2, eps, wd, step, gscale=None):
    n = p.numel()
    lib.call("rv_adamw", p, master, g, m, v, n, lr, b1, b2, eps, wd, 1.0 - b1 ** step, 1.0 - b2 ** step, gscale)


def grad_norm_clip_coef(g, max_norm):
    """Returns fp32[2] device tensor: (global grad norm, clip coefficient) with no host sync."""
    nblk = 1024
    part = torch.empty(nblk, dtype=torch.float32, device=g.device)
    lib.call("rv_sumsq_partial_bf16", g, g.numel(), part, nblk)
    out = torch.empty(2, dtype=torch.float32, device=g.device)
    lib.call("rv_clip_coef", part, nblk, max_norm, out)
    return out


def to_bf16(x):
    out = torch.empty(x.shape, dtype=BF16, device=x.device)
    lib.call("rv_cast_f32_to_bf16", x.contiguous(), out, x.numel())
    return out


def to_f32(x):
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    lib.call("rv_cast_bf16_to_f32", x.contiguous(), out, x.numel())
    return out
