"""bf16-emulating mode of the CPU oracle: the forward pass of ``llava_oracle`` with every tensor rounded to bf16 at the
points where the HIP path STORES bf16 (and nowhere else), arithmetic in fp32 in between.

TEST INFRASTRUCTURE ONLY (same rule as llava_oracle.py: imported by ``tests/``, ``smoke()`` and ``bench.py``'s cpu_baseline
leg alone).

Why it exists: the contract gate is ``||logits - ref||_inf / ||ref||_inf <= 1e-3`` and ``|loss - ref| <= 1e-3``.  Against the
fp32 reference that gate mixes two things -- the quantisation every bf16 implementation has (the reference's own GPU runs
are bf16: finetune_radio_7b.sh ``--bf16 True``) and errors of these kernels.  This module separates them:
  * emulated-vs-fp32 oracle  = the quantisation floor of the storage format (reported, not gated);
  * HIP-vs-emulated          = kernel error proper, gated at the contract's 1e-3.
Pinning: with ``rnd`` = identity this module must reproduce ``llava_oracle.llava_forward`` (itself pinned to the reference's
golden vectors) to <= 1e-5 -- ``tests/test_oracle_golden.py::test_bf16_emulation_reduces_to_oracle``.

Store points followed (reference text each one shadows):
  weights, pixels                      bf16 storage of ``--bf16 True`` runs
  every GEMM output                    one rounding AFTER bias / activation / residual (fused epilogue; the reference rounds after
                                       the matmul and again after each elementwise op)
  LayerNorm / RMSNorm output           modeling_llama.py:84-87 (normalised value rounded, then * weight, rounded)
  RoPE                                 cos/sin rounded (modeling_llama.py:134-139 ``.to(dtype=x.dtype)``), rotation in fp32, one rounding
  attention                            scores / softmax statistics fp32 (modeling_llama.py:361 softmax(dtype=float32)), probabilities
                                       rounded to bf16 before P V, 64-key tiles with a running maximum (head_dim 128 kernels: the
                                       maximum rounded up to an integer of the exp2 domain), output rounded once
  SwiGLU                               silu(gate) rounded, * up, rounded (modeling_llama.py:226)
  logits                               bf16 (lm_head output), read back as fp32 for the loss (modeling_llama.py:1324 logits.float())
"""
import math

import torch
import torch.nn.functional as F

from . import llava_oracle as O

BF16 = torch.bfloat16
LOG2E = 1.4426950408889634


TRACE = None     # tools/bf16_flip_trace.py sets a dict here to collect every stored tensor by name (diagnostics only)


def _t(name, x):
    if TRACE is not None:
        TRACE[name] = x.detach().clone()
    return x


def bf16_round(x):
    return x.to(BF16).to(torch.float32)


def identity(x):
    return x


def rmsnorm(x, w, eps, rnd):
    rstd = torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps)
    return rnd(w * rnd(x * rstd))


def layernorm(x, w, b, eps, rnd):
    mean = x.mean(-1, keepdim=True)
    var = (x - mean).pow(2).mean(-1, keepdim=True)
    return rnd((x - mean) * torch.rsqrt(var + eps) * w + b)


def rope(x, cos, sin, rnd):
    """x [b,h,S,hd]; cos/sin [S,hd/2] already rounded; half-split rotation, one rounding."""
    h = x.shape[-1] // 2
    a, b = x[..., :h], x[..., h:]
    return rnd(torch.cat((a * cos - b * sin, b * cos + a * sin), dim=-1))


def nat_kernel_head_dim(hd):
    """True when the engine runs this head_dim on the natural-layout attention kernels (head_dim 128, or padded to 128): those keep
    the running maximum as an integer of the exp2 domain (radvlm_amd/csrc/attention.hip, attn_fwd_nat_kernel)."""
    return (hd if hd in (64, 128) else (64 if hd < 64 else 128)) == 128


def attention(q, k, v, lens, causal, scale, rnd, pos0=None, tile=64, int_max=False):
    """Flash-style attention as the kernel evaluates it: per 64-key tile, running maximum in the exp2 domain (int_max: rounded up to
    an integer, as the natural-layout kernels keep it -- every rescale is then an exact power of two), probabilities rounded before
    the P V product, fp32 accumulators, one rounding of O / l.  With rnd = identity it is plain softmax attention."""
    b, h, S, hd = q.shape
    sl2 = scale * LOG2E
    o = torch.zeros(b, h, S, hd)
    m = torch.full((b, h, S, 1), -math.inf)
    l = torch.zeros(b, h, S, 1)
    qi = torch.arange(S)[:, None]
    for k0 in range(0, S, tile):
        k1 = min(S, k0 + tile)
        s = torch.matmul(q, k[:, :, k0:k1].transpose(2, 3)) * sl2
        ki = torch.arange(k0, k1)[None, :]
        ok = torch.ones(S, k1 - k0, dtype=torch.bool)
        if causal:
            ok = ok & (ki <= qi)
        ok = ok[None, None].expand(b, 1, S, k1 - k0)
        if lens is not None:
            ok = ok & (ki[None, None] < torch.as_tensor(lens)[:, None, None, None])
        s = torch.where(ok, s, torch.full_like(s, -math.inf))
        tmax = s.amax(-1, keepdim=True)
        mnew = torch.maximum(m, torch.ceil(tmax) if int_max else tmax)
        safe = torch.where(torch.isinf(mnew), torch.zeros_like(mnew), mnew)      # rows with no valid key so far
        alpha = torch.exp2(m - safe)
        p = torch.exp2(s - safe)
        l = l * alpha + p.sum(-1, keepdim=True)
        o = o * alpha + torch.matmul(rnd(p), v[:, :, k0:k1])
        m = mnew
    return rnd(o / l.clamp_min(1e-30))


def decoder_layer(x, P, pre, l, lens, cos, sin, eps, rnd):
    b, S, d = x.shape
    heads, kvh = l["heads"], l.get("kv_heads", l["heads"])
    hd = d // heads
    bias = lambda n: P.get(pre + f"self_attn.{n}.bias")
    _t(pre + "x", x)
    h1 = _t(pre + "h1", rmsnorm(x, P[pre + "input_layernorm.weight"], eps, rnd))
    q = rnd(F.linear(h1, P[pre + "self_attn.q_proj.weight"], bias("q_proj"))).view(b, S, heads, hd).transpose(1, 2)
    k = rnd(F.linear(h1, P[pre + "self_attn.k_proj.weight"], bias("k_proj"))).view(b, S, kvh, hd).transpose(1, 2)
    v = rnd(F.linear(h1, P[pre + "self_attn.v_proj.weight"], bias("v_proj"))).view(b, S, kvh, hd).transpose(1, 2)
    _t(pre + "v", v.transpose(1, 2).reshape(b, S, -1))
    q, k = rope(q, cos, sin, rnd), rope(k, cos, sin, rnd)
    _t(pre + "q_roped", q.transpose(1, 2).reshape(b, S, -1)), _t(pre + "k_roped", k.transpose(1, 2).reshape(b, S, -1))
    if kvh != heads:
        k = k.repeat_interleave(heads // kvh, dim=1)
        v = v.repeat_interleave(heads // kvh, dim=1)
    a = _t(pre + "attn", attention(q, k, v, lens, True, 1.0 / math.sqrt(hd), rnd, int_max=nat_kernel_head_dim(hd)).transpose(1, 2).reshape(b, S, d))
    x_mid = _t(pre + "x_mid", rnd(F.linear(a, P[pre + "self_attn.o_proj.weight"]) + x))
    h2 = _t(pre + "h2", rmsnorm(x_mid, P[pre + "post_attention_layernorm.weight"], eps, rnd))
    g = rnd(F.linear(h2, P[pre + "mlp.gate_proj.weight"]))
    u = rnd(F.linear(h2, P[pre + "mlp.up_proj.weight"]))
    _t(pre + "gu", torch.cat((g, u), -1))
    act = _t(pre + "act", rnd(rnd(g * torch.sigmoid(g)) * u))
    return rnd(F.linear(act, P[pre + "mlp.down_proj.weight"]) + x_mid)


def llama_forward(P, geo, embeds, lens, rnd):
    l = geo["lm"]
    b, S, d = embeds.shape
    eps = l.get("rms_eps", 1e-5)
    hd = d // l["heads"]
    inv = 1.0 / (l.get("rope_theta", 10000.0) ** (torch.arange(0, hd, 2, dtype=torch.int64).float() / hd))
    fr = torch.outer(torch.arange(S, dtype=torch.float32), inv)
    cos, sin = rnd(fr.cos()), rnd(fr.sin())
    x = embeds
    for i in range(l["layers"]):
        x = decoder_layer(x, P, f"model.layers.{i}.", l, lens, cos, sin, eps, rnd)
    _t("x_last", x)
    x = _t("hN", rmsnorm(x, P["model.norm.weight"], eps, rnd))
    raw = F.linear(x, P["lm_head.weight"])       # fp32 accumulators of the lm_head GEMM (what the engine hands out as .logits)
    return rnd(raw), raw


def vision_tower(P, geo, pixels, rnd, fused_act=True):
    """CLIP (hidden_states[-2], class token dropped) or SigLIP (last layer dropped) patch features; fused_act: the frozen tower
    applies fc1's activation inside the GEMM epilogue (one rounding), a tunable tower stores the pre-activation too (two)."""
    v = geo["vision"]
    vp = "model.vision_tower.vision_tower.vision_model."
    siglip = v.get("kind") == "siglip"
    n = pixels.shape[0]
    eps = 1e-6 if siglip else 1e-5
    x = F.conv2d(pixels, P[vp + "embeddings.patch_embedding.weight"], P.get(vp + "embeddings.patch_embedding.bias") if siglip else None,
                 stride=v["patch"]).flatten(2).transpose(1, 2)
    x = rnd(x)
    if siglip:
        x = rnd(x + P[vp + "embeddings.position_embedding.weight"][None])
    else:
        cls = P[vp + "embeddings.class_embedding"].expand(n, 1, -1)
        x = rnd(torch.cat([cls, x], dim=1) + P[vp + "embeddings.position_embedding.weight"][None])
        x = layernorm(x, P[vp + "pre_layrnorm.weight"], P[vp + "pre_layrnorm.bias"], eps, rnd)
    for i in range(v["layers"] - 1):
        x, _ = vision_layer(x, P, vp + f"encoder.layers.{i}.", v, rnd, fused_act=fused_act)
    return x if siglip else x[:, 1:]


def mm_projector(P, x, rnd):
    _t("f0", x)
    z = _t("z1", rnd(F.linear(x, P["model.mm_projector.0.weight"], P["model.mm_projector.0.bias"])))
    a1 = _t("a1", rnd(F.gelu(z)))
    return _t("proj", rnd(F.linear(a1, P["model.mm_projector.2.weight"], P["model.mm_projector.2.bias"])))


def llava_forward(P, geo, input_ids, attention_mask, labels, images, image_sizes=None, cfg=None, emulate=True, tower_tunable=False):
    """llava_oracle.llava_forward with the HIP path's bf16 store points (emulate=True) or without any rounding (emulate=False:
    must equal the fp32 oracle).  Weights and pixels are rounded once up front.  Returns (loss, logits, aux): the loss is computed
    from the bf16-stored logits like the training path does, `logits` are the unrounded fp32 accumulators of the lm_head GEMM."""
    rnd = bf16_round if emulate else identity
    cfg = dict(cfg or {})
    cfg.setdefault("tower_image_size", geo["vision"]["image"])
    with torch.no_grad():
        P = {k: rnd(v.detach().float()) for k, v in P.items()}
        imgs = [rnd((x[None] if x.ndim == 3 else x).float()) for x in images]
        split = [x.shape[0] for x in imgs]
        feats = mm_projector(P, vision_tower(P, geo, torch.cat(imgs, 0), rnd, fused_act=not tower_tunable), rnd)
        per = list(torch.split(feats, split))
        side = geo["vision"]["image"] // geo["vision"]["patch"]
        mt = cfg.get("mm_patch_merge_type", "flat")
        if "anyres_max" in cfg.get("image_aspect_ratio", "") or "maxpool" in mt:
            # the resampled / pooled rows are stored once more by the device (weighted gather, max4): one extra rounding
            merged = [rnd(m) for m in O.merge_image_features(per, cfg, P, image_sizes, side)]
        else:
            merged = O.merge_image_features(per, cfg, P, image_sizes, side)
        E, L, M, lens = O.splice(P, input_ids, attention_mask, labels, merged, cfg.get("tokenizer_model_max_length"))
        stored, logits = llama_forward(P, geo, E, lens, rnd)
        loss = O.causal_lm_loss(stored, L)       # the loss reads the bf16-stored logits (modeling_llama.py:1324 logits.float())
    return loss, logits, dict(inputs_embeds=E, labels=L, attention_mask=M, lens=lens, image_features=feats)


# ----------------------------------------------------------------------------- backward of one decoder layer, HIP store points
# The forward above shadows the reference's forward; its backward is what torch autograd derives from modeling_llama.py:852-911
# (decoder layer), :82-87 (RMSNorm), :174-198 (rotary embedding), :349-368 (attention) and :226 (SwiGLU).  The functions below are
# those derivatives written out op by op, in fp32, with a bf16 rounding exactly where the HIP backward STORES bf16:
#   every dgrad GEMM output (one rounding; "+ dx" of the norm backward is inside the same store)
#   SwiGLU backward            d(act) rounded (it was a tensor of the unfused sequence and the fused epilogue keeps that point), then
#                              d gate / d up rounded once
#   attention backward         P and dS rounded before the P^T dO / dS K / dS^T Q products (MFMA operands), dQ / dK rounded before the
#                              rotary adjoint (the unfused sequence stored them first), one more rounding after it; dV rounded once;
#                              grouped-query heads: the group's query heads are summed in fp32 before the rounding
#   weight gradients           returned as fp32 sums (the GEMM's fp32 accumulators; the bf16 store into the flat gradient buffer is one
#                              rounding of these)
# Pinning: with rnd = identity `decoder_layer_backward` must equal torch autograd of llava_oracle.decoder_layer to 1e-5
# (tests/test_oracle_golden.py::test_bf16_emulation_backward_reduces_to_autograd).
def swiglu_bwd(dact, gu, F_, rnd):
    g, u = gu[..., :F_], gu[..., F_:]
    d = rnd(dact)
    sg = torch.sigmoid(g)
    return torch.cat((rnd(d * u * (sg * (1.0 + g * (1.0 - sg)))), rnd(d * (g * sg))), -1)


def rmsnorm_bwd(dy, x, w, eps, dx_in, rnd):
    """d/dx of w * (x * rstd): returns (rnd(dx_in + dx), fp32 dw).  dw sums dy * (x * rstd) over rows."""
    rstd = torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps)
    xh = x * rstd
    gg = dy * w
    dot = (gg * xh).mean(-1, keepdim=True)
    dx = rstd * (gg - xh * dot)
    return rnd((dx_in if dx_in is not None else 0.0) + dx), (dy * xh).reshape(-1, x.shape[-1]).sum(0)


def unrope(x, cos, sin, rnd):
    """Adjoint of `rope` (a rotation: its transpose); x [b,h,S,hd] rounded first, rounded once more after."""
    h = x.shape[-1] // 2
    a, b = rnd(x[..., :h]), rnd(x[..., h:])
    return rnd(torch.cat((a * cos + b * sin, b * cos - a * sin), dim=-1))


def attention_lse(q, k, lens, scale, causal=True):
    """Natural-log softmax normaliser of the causal + key-padding masked scores, [b,h,S] (what the forward kernel leaves for backward)."""
    b, h, S, hd = q.shape
    nrep = h // k.shape[1]
    s = torch.matmul(q, k.repeat_interleave(nrep, dim=1).transpose(2, 3)) * scale
    ok = (torch.tril(torch.ones(S, S, dtype=torch.bool)) if causal else torch.ones(S, S, dtype=torch.bool))[None, None]
    if lens is not None:
        ok = ok & (torch.arange(S)[None, None, None, :] < torch.as_tensor(lens)[:, None, None, None])
    return torch.logsumexp(s.masked_fill(~ok, -math.inf), dim=-1)


def attention_bwd(q, k, v, o, do, lse, lens, scale, rnd, cos=None, sin=None, group_partials_bf16=False, causal=True):
    """q, o, do [b,h,S,hd]; k, v [b,kvh,S,hd] (q, k rotated); lse [b,h,S].  Returns (dq, dk, dv) as the kernels store them: gradients of
    the UN-rotated q / k when cos / sin are given.  group_partials_bf16: the grouped-query launch shape that writes one bf16 dK / dV
    partial per QUERY head and sums the group afterwards (rv_attn_bwd_nat with a workspace and a small grid) -- one more store point."""
    b, h, S, hd = q.shape
    kvh = k.shape[1]
    nrep = h // kvh
    kk, vv = k.repeat_interleave(nrep, dim=1), v.repeat_interleave(nrep, dim=1)
    ok = (torch.tril(torch.ones(S, S, dtype=torch.bool)) if causal else torch.ones(S, S, dtype=torch.bool))[None, None]
    if lens is not None:
        ln = torch.as_tensor(lens)[:, None, None, None]
        ok = ok & (torch.arange(S)[None, None, None, :] < ln) & (torch.arange(S)[None, None, :, None] < ln)
    sl2 = scale * LOG2E
    p = torch.exp2(torch.matmul(q, kk.transpose(2, 3)) * sl2 - (lse * LOG2E)[..., None])
    p = torch.where(ok, p, torch.zeros_like(p))
    delta = (do * o).sum(-1, keepdim=True)
    ds = p * (torch.matmul(do, vv.transpose(2, 3)) - delta)
    pr, dsr = rnd(p), rnd(ds)
    dq = torch.matmul(dsr, kk) * scale
    part = rnd if (group_partials_bf16 and nrep > 1) else identity
    dk = part(torch.matmul(dsr.transpose(2, 3), q) * scale).view(b, kvh, nrep, S, hd).sum(2)
    dv = part(torch.matmul(pr.transpose(2, 3), do)).view(b, kvh, nrep, S, hd).sum(2)
    if cos is not None:
        return unrope(dq, cos, sin, rnd), unrope(dk, cos, sin, rnd), rnd(dv)
    return rnd(dq), rnd(dk), rnd(dv)


def decoder_layer_backward(T, P, pre, l, lens, dx_out, rnd, rope_adjoint=True, group_partials_bf16=False):
    """Backward of `decoder_layer` through the stored activations T (a TRACE of the forward: bf16-exact when rnd = bf16_round) for the
    upstream gradient dx_out [b,S,d].  Returns an ordered dict: every tensor the HIP backward stores for this layer, in its order --
      gW_down, dgu, gW_gu, dh2, dx_mid, g_ln2, gW_o, dattn, dqkv (un-rotated q|k|v gradient), g_bqkv, gW_qkv, dh1, dx_in, g_ln1
    (gW_* / g_* in fp32).  Each op reads the previous op's STORED output, so every entry is also the exact input of the next op."""
    from collections import OrderedDict
    d, F_ = l["d"], l["ffn"]
    heads, kvh = l["heads"], l.get("kv_heads", l["heads"])
    hd = d // heads
    eps = l.get("rms_eps", 1e-5)
    b, S, _ = dx_out.shape
    flat = lambda t: t.reshape(-1, t.shape[-1])
    W = lambda n: P[pre + n]
    out = OrderedDict()
    x, h1, attn, x_mid, h2, gu, act = (T[pre + n] for n in ("x", "h1", "attn", "x_mid", "h2", "gu", "act"))
    out["gW_down"] = flat(dx_out).t().double().matmul(flat(act).double()).float()
    out["dgu"] = dgu = swiglu_bwd(F.linear(dx_out, W("mlp.down_proj.weight").t()), gu, F_, rnd)
    wgu = torch.cat((W("mlp.gate_proj.weight"), W("mlp.up_proj.weight")), 0)
    out["gW_gu"] = flat(dgu).t().double().matmul(flat(h2).double()).float()
    out["dh2"] = dh2 = rnd(F.linear(dgu, wgu.t()))
    out["dx_mid"], out["g_ln2"] = rmsnorm_bwd(dh2, x_mid, W("post_attention_layernorm.weight"), eps, dx_out, rnd)
    dx_mid = out["dx_mid"]
    out["gW_o"] = flat(dx_mid).t().double().matmul(flat(attn).double()).float()
    out["dattn"] = dattn = rnd(F.linear(dx_mid, W("self_attn.o_proj.weight").t()))
    heads_of = lambda t, n: t.view(b, S, n, hd).transpose(1, 2)
    q, k, v = heads_of(T[pre + "q_roped"], heads), heads_of(T[pre + "k_roped"], kvh), heads_of(T[pre + "v"], kvh)
    inv = 1.0 / (l.get("rope_theta", 10000.0) ** (torch.arange(0, hd, 2, dtype=torch.int64).float() / hd))
    fr = torch.outer(torch.arange(S, dtype=torch.float32), inv)
    cos, sin = rnd(fr.cos()), rnd(fr.sin())
    scale = 1.0 / math.sqrt(hd)
    out["lse"] = lse = attention_lse(q, k, lens, scale)
    dq, dk, dv = attention_bwd(q, k, v, heads_of(attn, heads), heads_of(dattn, heads), lse, lens, scale, rnd,
                               cos if rope_adjoint else None, sin if rope_adjoint else None, group_partials_bf16)
    back = lambda t: t.transpose(1, 2).reshape(b, S, -1)
    out["dqkv"] = dqkv = torch.cat((back(dq), back(dk), back(dv)), -1)
    out["g_bqkv"] = flat(dqkv).sum(0)
    wqkv = torch.cat((W("self_attn.q_proj.weight"), W("self_attn.k_proj.weight"), W("self_attn.v_proj.weight")), 0)
    out["gW_qkv"] = flat(dqkv).t().double().matmul(flat(h1).double()).float()
    out["dh1"] = dh1 = rnd(F.linear(dqkv, wqkv.t()))
    out["dx_in"], out["g_ln1"] = rmsnorm_bwd(dh1, x, W("input_layernorm.weight"), eps, dx_mid, rnd)
    return out


# ----------------------------------------------------------------------------- vision tower layer, projector, head: traces + backward
# Same construction as the decoder layer above, for the rest of the trainable path (the RadVLM recipe tunes the tower too,
# finetune_radio_7b.sh:52): forward functions that record every stored tensor, and their derivatives written out op by op with a bf16
# rounding exactly where the HIP backward stores bf16.  Reference text: HF modeling_clip.py:202-384 (CLIPEncoderLayer: pre-LN attention +
# quick_gelu MLP), siglip_encoder.py:148-306 (same block with gelu_tanh, head_dim 72), multimodal_projector/builder.py:41-48 (mlp2x_gelu),
# modeling_llama.py:1323-1337 (lm_head + shifted cross entropy).  Pinning: with rnd = identity every backward below equals torch autograd
# of its own forward to 1e-5 (tests/test_oracle_golden.py::test_bf16_emulation_tower_head_backward_reduces_to_autograd), and the forwards
# are the code `vision_tower` / `mm_projector` / `llama_forward` run, which are pinned to the reference-pinned oracle.
VP = "model.vision_tower.vision_tower.vision_model."


def _act_pair(siglip):
    if siglip:
        k = 0.7978845608028654

        def fwd(z):
            return F.gelu(z, approximate="tanh")

        def grad(z):
            t = torch.tanh(k * (z + 0.044715 * z ** 3))
            return 0.5 * (1.0 + t) + 0.5 * z * (1.0 - t * t) * k * (1.0 + 0.134145 * z * z)
        return fwd, grad

    def grad(z):
        sg = torch.sigmoid(1.702 * z)
        return sg * (1.0 + 1.702 * z * (1.0 - sg))
    return O.quick_gelu, grad


def layernorm_bwd(dy, x, w, eps, dx_in, rnd):
    """Backward of `layernorm`: (rnd(dx_in + dx), fp32 dw, fp32 db); dw / db are sums over all rows."""
    mean = x.mean(-1, keepdim=True)
    rstd = torch.rsqrt((x - mean).pow(2).mean(-1, keepdim=True) + eps)
    xh = (x - mean) * rstd
    g = dy * w
    dx = rstd * (g - g.mean(-1, keepdim=True) - xh * (g * xh).mean(-1, keepdim=True))
    flat = lambda t: t.reshape(-1, t.shape[-1])
    return rnd((dx_in if dx_in is not None else 0.0) + dx), flat(dy * xh).sum(0), flat(dy).sum(0)


def vision_layer(x, P, p, v, rnd, fused_act=False):
    """One pre-LN encoder layer of the CLIP / SigLIP tower as the engine runs it (LlavaEngine.vision_forward): returns (x_out, T) with T
    the stored tensors {x, h, qkv, a, x1, h2, z, g}.  fused_act: fc1's activation in the GEMM epilogue (frozen tower: z is not stored)."""
    siglip = v.get("kind") == "siglip"
    eps = 1e-6 if siglip else 1e-5
    n, N, d = x.shape
    H = v["heads"]
    hd = d // H
    act, _ = _act_pair(siglip)
    T = {"x": x}
    T["h"] = h = layernorm(x, P[p + "layer_norm1.weight"], P[p + "layer_norm1.bias"], eps, rnd)
    T["qkv"] = qkv = torch.cat([rnd(F.linear(h, P[p + f"self_attn.{t}_proj.weight"], P[p + f"self_attn.{t}_proj.bias"])) for t in "qkv"], -1)
    heads = lambda t: t.view(n, N, H, hd).transpose(1, 2)
    q, k, vv = heads(qkv[..., :d]), heads(qkv[..., d:2 * d]), heads(qkv[..., 2 * d:])
    T["a"] = a = attention(q, k, vv, None, False, hd ** -0.5, rnd, int_max=nat_kernel_head_dim(hd)).transpose(1, 2).reshape(n, N, d)
    T["x1"] = x1 = rnd(F.linear(a, P[p + "self_attn.out_proj.weight"], P[p + "self_attn.out_proj.bias"]) + x)
    T["h2"] = h2 = layernorm(x1, P[p + "layer_norm2.weight"], P[p + "layer_norm2.bias"], eps, rnd)
    z = F.linear(h2, P[p + "mlp.fc1.weight"], P[p + "mlp.fc1.bias"])
    if fused_act:
        T["g"] = g = rnd(act(z))
    else:
        T["z"] = z = rnd(z)
        T["g"] = g = rnd(act(z))
    return rnd(F.linear(g, P[p + "mlp.fc2.weight"], P[p + "mlp.fc2.bias"]) + x1), T


def vision_layer_backward(T, P, p, v, dx_out, rnd):
    """Backward of `vision_layer` (tower tunable: z stored) for the upstream gradient dx_out [n,N,d], in the order of
    LlavaEngine.vision_backward.  Every entry is a tensor the HIP backward stores (weight / bias / norm gradients as fp32 sums)."""
    from collections import OrderedDict
    siglip = v.get("kind") == "siglip"
    eps = 1e-6 if siglip else 1e-5
    n, N, d = dx_out.shape
    H = v["heads"]
    hd = d // H
    _, dact = _act_pair(siglip)
    flat = lambda t: t.reshape(-1, t.shape[-1])
    wsum = lambda dy, xin: flat(dy).t().double().matmul(flat(xin).double()).float()
    W = lambda nm: P[p + nm]
    out = OrderedDict()
    out["g_fc2_b"] = flat(dx_out).sum(0)
    out["gW_fc2"] = wsum(dx_out, T["g"])
    out["dg"] = dg = rnd(F.linear(dx_out, W("mlp.fc2.weight").t()))
    out["dz"] = dz = rnd(dg * dact(T["z"]))
    out["g_fc1_b"] = flat(dz).sum(0)
    out["gW_fc1"] = wsum(dz, T["h2"])
    out["dh2"] = dh2 = rnd(F.linear(dz, W("mlp.fc1.weight").t()))
    out["dx1"], out["g_ln2_w"], out["g_ln2_b"] = layernorm_bwd(dh2, T["x1"], W("layer_norm2.weight"), eps, dx_out, rnd)
    dx1 = out["dx1"]
    out["g_out_b"] = flat(dx1).sum(0)
    out["gW_out"] = wsum(dx1, T["a"])
    out["da"] = da = rnd(F.linear(dx1, W("self_attn.out_proj.weight").t()))
    heads = lambda t: t.view(n, N, H, hd).transpose(1, 2)
    qkv = T["qkv"]
    q, k, vv = heads(qkv[..., :d]), heads(qkv[..., d:2 * d]), heads(qkv[..., 2 * d:])
    scale = hd ** -0.5
    out["lse"] = lse = attention_lse(q, k, None, scale, causal=False)
    dq, dk, dv = attention_bwd(q, k, vv, heads(T["a"]), heads(da), lse, None, scale, rnd, causal=False)
    back = lambda t: t.transpose(1, 2).reshape(n, N, d)
    out["dqkv"] = dqkv = torch.cat((back(dq), back(dk), back(dv)), -1)
    out["g_bqkv"] = flat(dqkv).sum(0)
    out["gW_qkv"] = wsum(dqkv, T["h"])
    wqkv = torch.cat([W(f"self_attn.{t}_proj.weight") for t in "qkv"], 0)
    out["dh"] = dh = rnd(F.linear(dqkv, wqkv.t()))
    out["dx_in"], out["g_ln1_w"], out["g_ln1_b"] = layernorm_bwd(dh, T["x"], W("layer_norm1.weight"), eps, dx1, rnd)
    return out


def gelu_grad(z):
    """d/dz of the exact (erf) GELU of mlp2x_gelu."""
    return 0.5 * (1.0 + torch.erf(z * 0.7071067811865476)) + z * 0.3989422804014327 * torch.exp(-0.5 * z * z)


def mm_projector_backward(T, P, dproj, rnd):
    """Backward of `mm_projector` through its trace {f0, z1, a1} for the gradient of its output rows dproj [rows, d_lm]."""
    from collections import OrderedDict
    w0, w2 = P["model.mm_projector.0.weight"], P["model.mm_projector.2.weight"]
    flat = lambda t: t.reshape(-1, t.shape[-1])
    wsum = lambda dy, xin: flat(dy).t().double().matmul(flat(xin).double()).float()
    out = OrderedDict()
    out["g_b2"] = flat(dproj).sum(0)
    out["gW2"] = wsum(dproj, T["a1"])
    out["da1"] = da1 = rnd(F.linear(dproj, w2.t()))
    out["dz1"] = dz1 = rnd(da1 * gelu_grad(T["z1"]))
    out["g_b0"] = flat(dz1).sum(0)
    out["gW0"] = wsum(dz1, T["f0"])
    out["df0"] = rnd(F.linear(dz1, w0.t()))
    return out


def lm_head_cross_entropy(hN, w_head, targets, gscale, rnd):
    """lm_head + shifted cross entropy as the engine evaluates them (modeling_llama.py:1323-1337): logits stored bf16, per-row loss and
    softmax in fp32 from the STORED logits, dlogits = (softmax - onehot) * gscale / count stored bf16 over the logits, then the head's
    input gradient (one rounding) and weight gradient (fp32 sum).  targets: already shifted, -100 = ignored row.  Returns a dict."""
    logits = rnd(F.linear(hN, w_head))
    rows = targets != -100
    count = int(rows.sum())
    lse = torch.logsumexp(logits, -1)
    tgt = logits.gather(-1, targets.clamp_min(0)[..., None])[..., 0]
    loss_rows = torch.where(rows, lse - tgt, torch.zeros_like(lse))
    p = torch.exp(logits - lse[..., None])
    onehot = F.one_hot(targets.clamp_min(0), logits.shape[-1]).to(p.dtype)
    dlogits = rnd(torch.where(rows[..., None], (p - onehot) * (gscale / max(count, 1)), torch.zeros_like(p)))
    flat = lambda t: t.reshape(-1, t.shape[-1])
    return dict(logits=logits, loss_rows=loss_rows, loss=loss_rows.sum() / max(count, 1), dlogits=dlogits,
                dhN=rnd(F.linear(dlogits, w_head.t())), gW_head=flat(dlogits).t().double().matmul(flat(hN).double()).float())


# ----------------------------------------------------------------------------- LoRA: the counter-based dropout mask and the adapted linear
# peft LoraLayer (train/train.py:1515-1532): y = x W^T + (alpha / r) * dropout_p(x) A^T B^T, an independent mask per adapted module, on
# the adapter's input only.  The HIP path regenerates its masks from (seed, element index) -- common.h rv_hash64 / rv_keep8, restated here
# bit for bit -- so an emulation can apply THE SAME mask (peft is absent and the reference holds no LoRA fixture: parity unpinned upstream).
_M64 = (1 << 64) - 1


def _hash64(seed, b):
    import numpy as np
    with np.errstate(over="ignore"):
        z = (b + np.uint64((seed * 0x9E3779B97F4A7C15) & _M64)) + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def dropout_keep_mask(shape, p, seed):
    """Boolean keep mask of rv_dropout_bf16(x viewed as contiguous elements, p, seed): one splitmix64 value per four consecutive
    elements, element i kept when bits [16 (i & 3), +16) of hash(seed, i >> 2) >= round(p * 2^16)."""
    import numpy as np
    n = 1
    for s in shape:
        n *= int(s)
    thr = int(p * 65536.0 + 0.5)
    i = np.arange(n, dtype=np.uint64)
    h = _hash64(int(seed), i >> np.uint64(2))
    field = (h >> (np.uint64(16) * (i & np.uint64(3)))) & np.uint64(0xFFFF)
    return torch.from_numpy((field >= np.uint64(thr)).reshape(tuple(int(s) for s in shape)))


def lora_linear(x, W, A, B, scale, p, seed, rnd, bias=None, residual=None):
    """(y, t): t = rnd(scale / (1 - p) * (keep * x) A^T) -- the 1 / (1 - p) rides the scale, the masked x is not rounded
    (rv_lora_down_bf16) -- and y = rnd(x W^T + t B^T + bias + residual) in ONE store (the adapter rides the base GEMM)."""
    keep = dropout_keep_mask(x.shape, p, seed).to(x.dtype) if p > 0 else torch.ones_like(x)
    t = rnd(F.linear(x * keep, A) * (scale / (1.0 - p) if p > 0 else scale))
    y = F.linear(x, W, bias) + F.linear(t, B)
    return rnd(y + (residual if residual is not None else 0.0)), t


def lora_linear_backward(dy, x, W, A, B, t, scale, p, seed, rnd):
    """Backward of `lora_linear` as LlavaEngine._lora_linear_bwd stores it: dts = rnd(scale * dy B) (r-wide), gB = dy^T t and
    gA = dts^T (keep * x) / (1 - p) (fp32 sums; since round 4 the masked x is an operand re-created in registers, not a stored
    tensor -- rv_lora_a_grad_bf16 -- so it is not rounded and the 1 / (1 - p) scales the sum, as in the forward; peft materialises
    dropout(x) in bf16, a rounding of <= 2^-9 per term), the base input gradient rnd(dy W) and the adapter branch added through
    the masked epilogue: dx = rnd(dx_base + keep * (dts A) / (1 - p))."""
    flat = lambda u: u.reshape(-1, u.shape[-1])
    keep = dropout_keep_mask(x.shape, p, seed).to(x.dtype) if p > 0 else torch.ones_like(x)
    dts = rnd(F.linear(dy, B.t()) * scale)
    gB = flat(dy).t().double().matmul(flat(t).double()).float()
    gA = (flat(dts).t().double().matmul(flat(x * keep).double()) / ((1.0 - p) if p > 0 else 1.0)).float()
    if p > 0:
        dx = rnd(rnd(F.linear(dy, W.t())) + keep * F.linear(dts, A.t()) / (1.0 - p))
    else:
        dx = rnd(F.linear(dy, W.t()) + F.linear(dts, A.t()))      # second operand pair of the same GEMM: one store
    return dict(dts=dts, gA=gA, gB=gB, dx=dx)
