"""CPU oracle: a plain torch-fp32 restatement of the reference's LLaVA training hot path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module, and only as the checker / reported baseline; the product path
(``radvlm_amd``) never imports it and fails loudly when its HIP library is missing.

Pinning: the reference ships no tests or golden vectors for this path, so this
restatement is pinned against outputs of the reference itself run in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``; checked by ``tests/test_oracle_golden.py``).

Every function cites the reference text it follows (paths relative to /root/reference/finetuning/llava;
``HF:`` = transformers 5.15 as installed, whose CLIP/Llama arithmetic the reference executes).
"""
import math
import re

import numpy as np
import torch
import torch.nn.functional as F

IGNORE_INDEX = -100       # constants.py:7-12
IMAGE_TOKEN_INDEX = -200  # constants.py:7-12


# ----------------------------------------------------------------------------- weights
def make_params(geo, seed=0, dtype=torch.float32, with_newline=False):
    """All parameters of the LLaVA model for ``geo`` from the portable generator, by canonical (4.x) name."""
    from radvlm_amd import portable_rng as prng
    from radvlm_amd.config import init_std_for
    out = {}
    for name, shape in param_shapes(geo, with_newline).items():
        kind, std = init_std_for(name, geo["lm"]["d"])
        w = prng.normal(seed, prng.name_tag(name), shape, std)
        if kind == "norm_weight":
            w = 1.0 + w
        out[name] = torch.from_numpy(w).to(dtype)
    return out


def param_shapes(geo, with_newline=False):
    """Canonical state-dict layout (SURVEY.md section 8b 'Parameter naming')."""
    v, l = geo["vision"], geo["lm"]
    s = {}
    s["model.embed_tokens.weight"] = (l["vocab"], l["d"])
    kvd = l["d"] // l["heads"] * l.get("kv_heads", l["heads"])   # Qwen2 GQA: k/v project to kv_heads * head_dim
    for i in range(l["layers"]):
        p = f"model.layers.{i}."
        for n, rows in (("q_proj", l["d"]), ("k_proj", kvd), ("v_proj", kvd), ("o_proj", l["d"])):
            s[p + f"self_attn.{n}.weight"] = (rows, l["d"])
            if l.get("qkv_bias") and n != "o_proj":              # HF:models/qwen2/modeling_qwen2.py Qwen2Attention
                s[p + f"self_attn.{n}.bias"] = (rows,)
        s[p + "mlp.gate_proj.weight"] = (l["ffn"], l["d"])
        s[p + "mlp.up_proj.weight"] = (l["ffn"], l["d"])
        s[p + "mlp.down_proj.weight"] = (l["d"], l["ffn"])
        s[p + "input_layernorm.weight"] = (l["d"],)
        s[p + "post_attention_layernorm.weight"] = (l["d"],)
    s["model.norm.weight"] = (l["d"],)
    vp = "model.vision_tower.vision_tower.vision_model."
    siglip = v.get("kind") == "siglip"
    npos = (v["image"] // v["patch"]) ** 2 + (0 if siglip else 1)
    if not siglip:
        s[vp + "embeddings.class_embedding"] = (v["d"],)
    s[vp + "embeddings.patch_embedding.weight"] = (v["d"], 3, v["patch"], v["patch"])
    if siglip:   # multimodal_encoder/siglip_encoder.py:148-174: conv with bias, no CLS, no pre-LN
        s[vp + "embeddings.patch_embedding.bias"] = (v["d"],)
    s[vp + "embeddings.position_embedding.weight"] = (npos, v["d"])
    if not siglip:
        s[vp + "pre_layrnorm.weight"] = (v["d"],)
        s[vp + "pre_layrnorm.bias"] = (v["d"],)
    for i in range(v["layers"]):
        p = vp + f"encoder.layers.{i}."
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            s[p + f"self_attn.{n}.weight"] = (v["d"], v["d"])
            s[p + f"self_attn.{n}.bias"] = (v["d"],)
        s[p + "layer_norm1.weight"] = (v["d"],)
        s[p + "layer_norm1.bias"] = (v["d"],)
        s[p + "mlp.fc1.weight"] = (v["ffn"], v["d"])
        s[p + "mlp.fc1.bias"] = (v["ffn"],)
        s[p + "mlp.fc2.weight"] = (v["d"], v["ffn"])
        s[p + "mlp.fc2.bias"] = (v["d"],)
        s[p + "layer_norm2.weight"] = (v["d"],)
        s[p + "layer_norm2.bias"] = (v["d"],)
    s[vp + "post_layernorm.weight"] = (v["d"],)
    s[vp + "post_layernorm.bias"] = (v["d"],)
    s["model.mm_projector.0.weight"] = (l["d"], v["d"])
    s["model.mm_projector.0.bias"] = (l["d"],)
    s["model.mm_projector.2.weight"] = (l["d"], l["d"])
    s["model.mm_projector.2.bias"] = (l["d"],)
    if with_newline:
        s["model.image_newline"] = (l["d"],)
    s["lm_head.weight"] = (l["vocab"], l["d"])
    return s


# ----------------------------------------------------------------------------- decoder ops
def rmsnorm(x, w, eps=1e-5):
    """LlamaRMSNorm.forward, language_model/modeling_llama.py:82-87 (fp32 internal, cast, then * weight)."""
    dt = x.dtype
    xf = x.to(torch.float32)
    var = xf.pow(2).mean(-1, keepdim=True)
    xf = xf * torch.rsqrt(var + eps)
    return w * xf.to(dt)


def rope_cos_sin(S, hd, theta=10000.0, dtype=torch.float32):
    """LlamaRotaryEmbedding.forward, modeling_llama.py:123-139: inv_freq = theta^(-2i/hd), fp32 trig, then cast."""
    inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.int64).float() / hd))
    pos = torch.arange(S, dtype=torch.float32)
    fr = torch.outer(pos, inv)
    emb = torch.cat((fr, fr), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rotate_half(x):
    """modeling_llama.py:167-171 (half-split convention)."""
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def apply_rope(q, k, cos, sin):
    """apply_rotary_pos_emb, modeling_llama.py:174-198; q,k [b,h,S,hd], cos/sin [S,hd]."""
    return q * cos + rotate_half(q) * sin, k * cos + rotate_half(k) * sin


def attention(q, k, v, lens=None, causal=True, scale=None):
    """LlamaAttention.forward eager core, modeling_llama.py:349-368: softmax_fp32(QK^T/sqrt(hd) + mask) V.

    Mask = causal AND key-padding (``_update_causal_mask`` :1191-1225); ``lens[b]`` = valid keys (right padding).
    """
    b, h, S, hd = q.shape
    scale = scale if scale is not None else 1.0 / math.sqrt(hd)
    s = torch.matmul(q, k.transpose(2, 3)) * scale
    neg = torch.finfo(s.dtype).min
    if causal:
        m = torch.triu(torch.ones(S, S, dtype=torch.bool), 1)
        s = s.masked_fill(m, neg)
    if lens is not None:
        kp = torch.arange(S)[None, :] >= torch.as_tensor(lens)[:, None]
        s = s.masked_fill(kp[:, None, None, :], neg)
    p = F.softmax(s, dim=-1, dtype=torch.float32).to(q.dtype)
    return torch.matmul(p, v)


def swiglu_mlp(x, wg, wu, wd):
    """LlamaMLP.forward, modeling_llama.py:226: down(silu(gate(x)) * up(x))."""
    return F.linear(F.silu(F.linear(x, wg)) * F.linear(x, wu), wd)


def decoder_layer(x, P, pre, heads, lens, cos, sin, eps=1e-5, kv_heads=None):
    """LlamaDecoderLayer.forward, modeling_llama.py:852-911 (pre-norm residual block).  With kv_heads < heads and
    q/k/v biases it is the Qwen2 layer LlavaQwenForCausalLM runs (language_model/llava_qwen.py:46-58 ->
    HF:models/qwen2/modeling_qwen2.py Qwen2DecoderLayer): same block, k/v heads repeated (repeat_kv, modeling_llama.py:201-210)."""
    b, S, d = x.shape
    hd = d // heads
    kvh = kv_heads or heads
    bias = lambda n: P.get(pre + f"self_attn.{n}.bias")
    h = rmsnorm(x, P[pre + "input_layernorm.weight"], eps)
    q = F.linear(h, P[pre + "self_attn.q_proj.weight"], bias("q_proj")).view(b, S, heads, hd).transpose(1, 2)
    k = F.linear(h, P[pre + "self_attn.k_proj.weight"], bias("k_proj")).view(b, S, kvh, hd).transpose(1, 2)
    v = F.linear(h, P[pre + "self_attn.v_proj.weight"], bias("v_proj")).view(b, S, kvh, hd).transpose(1, 2)
    q, k = apply_rope(q, k, cos, sin)
    if kvh != heads:
        k = k.repeat_interleave(heads // kvh, dim=1)
        v = v.repeat_interleave(heads // kvh, dim=1)
    a = attention(q, k, v, lens=lens, causal=True).transpose(1, 2).reshape(b, S, d)
    x = x + F.linear(a, P[pre + "self_attn.o_proj.weight"])
    h = rmsnorm(x, P[pre + "post_attention_layernorm.weight"], eps)
    return x + swiglu_mlp(h, P[pre + "mlp.gate_proj.weight"], P[pre + "mlp.up_proj.weight"], P[pre + "mlp.down_proj.weight"])


def llama_forward(P, geo, embeds, lens, eps=None):
    """LlamaModel.forward + lm_head, modeling_llama.py:1083-1185, 1323; position_ids = arange(S) for every row
    (the reference discards the spliced position_ids in training, llava_arch.py:534-545)."""
    l = geo["lm"]
    b, S, d = embeds.shape
    eps = eps if eps is not None else l.get("rms_eps", 1e-5)
    cos, sin = rope_cos_sin(S, d // l["heads"], theta=l.get("rope_theta", 10000.0), dtype=embeds.dtype)
    x = embeds
    for i in range(l["layers"]):
        x = decoder_layer(x, P, f"model.layers.{i}.", l["heads"], lens, cos, sin, eps, l.get("kv_heads"))
    x = rmsnorm(x, P["model.norm.weight"], eps)
    return F.linear(x, P["lm_head.weight"])


def causal_lm_loss(logits, labels):
    """LlamaForCausalLM.forward loss, modeling_llama.py:1323-1337 (logits.float, shift, mean CE, ignore -100)."""
    lg = logits.float()[:, :-1, :].reshape(-1, logits.shape[-1])
    lb = labels[:, 1:].reshape(-1)
    return F.cross_entropy(lg, lb, ignore_index=IGNORE_INDEX)


# ----------------------------------------------------------------------------- vision tower + projector
def quick_gelu(x):
    """HF:activations QuickGELU (x * sigmoid(1.702 x)), the act of OpenAI CLIP checkpoints."""
    return x * torch.sigmoid(1.702 * x)


def clip_vision_hidden(P, geo, pixels, select_layer=-2):
    """HF:models/clip/modeling_clip.py CLIPVisionTransformer up to hidden_states[select_layer].

    Embeddings :202-218 (conv patch embed no bias, CLS first, learned positions), pre_layrnorm, pre-LN encoder
    blocks :362-384 with non-causal MHSA (:298-335, scale hd^-0.5) and quick_gelu MLP (:346-350).
    hidden_states[0] is the embedding output after pre_layrnorm; [-2] is the input of the last layer.
    """
    v = geo["vision"]
    vp = "model.vision_tower.vision_tower.vision_model."
    n = pixels.shape[0]
    x = F.conv2d(pixels, P[vp + "embeddings.patch_embedding.weight"], stride=v["patch"])
    x = x.flatten(2).transpose(1, 2)
    cls = P[vp + "embeddings.class_embedding"].expand(n, 1, -1)
    x = torch.cat([cls, x], dim=1) + P[vp + "embeddings.position_embedding.weight"][None]
    x = F.layer_norm(x, (v["d"],), P[vp + "pre_layrnorm.weight"], P[vp + "pre_layrnorm.bias"], 1e-5)
    n_run = v["layers"] + 1 + select_layer if select_layer < 0 else select_layer
    N, d, H = x.shape[1], v["d"], v["heads"]
    hd = d // H
    for i in range(n_run):
        p = vp + f"encoder.layers.{i}."
        h = F.layer_norm(x, (d,), P[p + "layer_norm1.weight"], P[p + "layer_norm1.bias"], 1e-5)
        q = F.linear(h, P[p + "self_attn.q_proj.weight"], P[p + "self_attn.q_proj.bias"]).view(n, N, H, hd).transpose(1, 2)
        k = F.linear(h, P[p + "self_attn.k_proj.weight"], P[p + "self_attn.k_proj.bias"]).view(n, N, H, hd).transpose(1, 2)
        vv = F.linear(h, P[p + "self_attn.v_proj.weight"], P[p + "self_attn.v_proj.bias"]).view(n, N, H, hd).transpose(1, 2)
        a = attention(q, k, vv, lens=None, causal=False).transpose(1, 2).reshape(n, N, d)
        x = x + F.linear(a, P[p + "self_attn.out_proj.weight"], P[p + "self_attn.out_proj.bias"])
        h = F.layer_norm(x, (d,), P[p + "layer_norm2.weight"], P[p + "layer_norm2.bias"], 1e-5)
        h = quick_gelu(F.linear(h, P[p + "mlp.fc1.weight"], P[p + "mlp.fc1.bias"]))
        x = x + F.linear(h, P[p + "mlp.fc2.weight"], P[p + "mlp.fc2.bias"])
    return x


def siglip_vision_hidden(P, geo, pixels):
    """SigLipVisionTower.forward, multimodal_encoder/siglip_encoder.py:576-590: hidden_states[-1] of the encoder whose
    LAST layer was deleted at load (:570), head = Identity (:571), post_layernorm not applied to hidden_states.

    Embeddings :148-174 (conv WITH bias, 'valid' padding, learned positions, no CLS / pre-LN); encoder layer :259-306
    (pre-LN, MHSA :177-240 with scale head_dim^-0.5 and biases, MLP :243-256 with gelu_pytorch_tanh); eps 1e-6 (:83)."""
    v = geo["vision"]
    vp = "model.vision_tower.vision_tower.vision_model."
    n = pixels.shape[0]
    x = F.conv2d(pixels, P[vp + "embeddings.patch_embedding.weight"], P[vp + "embeddings.patch_embedding.bias"], stride=v["patch"])
    x = x.flatten(2).transpose(1, 2) + P[vp + "embeddings.position_embedding.weight"][None]
    N, d, H = x.shape[1], v["d"], v["heads"]
    hd = d // H
    for i in range(v["layers"] - 1):
        p = vp + f"encoder.layers.{i}."
        h = F.layer_norm(x, (d,), P[p + "layer_norm1.weight"], P[p + "layer_norm1.bias"], 1e-6)
        q = F.linear(h, P[p + "self_attn.q_proj.weight"], P[p + "self_attn.q_proj.bias"]).view(n, N, H, hd).transpose(1, 2)
        k = F.linear(h, P[p + "self_attn.k_proj.weight"], P[p + "self_attn.k_proj.bias"]).view(n, N, H, hd).transpose(1, 2)
        vv = F.linear(h, P[p + "self_attn.v_proj.weight"], P[p + "self_attn.v_proj.bias"]).view(n, N, H, hd).transpose(1, 2)
        a = attention(q, k, vv, lens=None, causal=False).transpose(1, 2).reshape(n, N, d)
        x = x + F.linear(a, P[p + "self_attn.out_proj.weight"], P[p + "self_attn.out_proj.bias"])
        h = F.layer_norm(x, (d,), P[p + "layer_norm2.weight"], P[p + "layer_norm2.bias"], 1e-6)
        h = F.gelu(F.linear(h, P[p + "mlp.fc1.weight"], P[p + "mlp.fc1.bias"]), approximate="tanh")
        x = x + F.linear(h, P[p + "mlp.fc2.weight"], P[p + "mlp.fc2.bias"])
    return x


def vision_tower(P, geo, pixels):
    """CLIPVisionTower.forward + feature_select, multimodal_encoder/clip_encoder.py:46-79 (layer -2, drop CLS); or the
    SigLIP tower (all 729 patch tokens)."""
    if geo["vision"].get("kind") == "siglip":
        return siglip_vision_hidden(P, geo, pixels)
    return clip_vision_hidden(P, geo, pixels, -2)[:, 1:]


def mm_projector(P, x):
    """build_vision_projector 'mlp2x_gelu', multimodal_projector/builder.py:41-48: Linear, GELU(erf), Linear."""
    h = F.gelu(F.linear(x, P["model.mm_projector.0.weight"], P["model.mm_projector.0.bias"]))
    return F.linear(h, P["model.mm_projector.2.weight"], P["model.mm_projector.2.bias"])


def encode_images(P, geo, pixels):
    """LlavaMetaForCausalLM.encode_images, llava_arch.py:192-196."""
    return mm_projector(P, vision_tower(P, geo, pixels))


# ----------------------------------------------------------------------------- host geometry (mm_utils.py)
def select_best_resolution(original_size, possible_resolutions):
    """mm_utils.py:119-149: maximise effective resolution, then minimise wasted area."""
    ow, oh = original_size
    best, max_eff, min_waste = None, 0, float("inf")
    for w, h in possible_resolutions:
        scale = min(w / ow, h / oh)
        dw, dh = int(ow * scale), int(oh * scale)
        eff = min(dw * dh, ow * oh)
        waste = w * h - eff
        if eff > max_eff or (eff == max_eff and waste < min_waste):
            max_eff, min_waste, best = eff, waste, (w, h)
    return best


def get_anyres_image_grid_shape(image_size, grid_pinpoints, patch_size):
    """mm_utils.py:213-240 -> (grid_w, grid_h) for list-form pinpoints."""
    w, h = select_best_resolution(image_size, [tuple(p) for p in grid_pinpoints])
    return w // patch_size, h // patch_size


def unpad_image(t, original_size):
    """llava_arch.py:127-159; t [C,H,W], original_size (W,H)."""
    ow, oh = original_size
    ch, cw = t.shape[1:]
    if ow / oh > cw / ch:
        nh = int(oh * (cw / ow))
        pad = (ch - nh) // 2
        return t[:, pad:ch - pad, :]
    nw = int(ow * (ch / oh))
    pad = (cw - nw) // 2
    return t[:, :, pad:cw - pad]


def tokenizer_image_token(prompt, tokenizer, image_token_index=IMAGE_TOKEN_INDEX):
    """mm_utils.py:341-360: split at '<image>', tokenise chunks, interleave -200, keep one BOS."""
    chunks = [tokenizer(c).input_ids for c in prompt.split("<image>")]
    ids, offset = [], 0
    if chunks and chunks[0] and chunks[0][0] == tokenizer.bos_token_id:
        offset = 1
        ids.append(chunks[0][0])
    sep = [image_token_index] * (offset + 1)
    inter = []
    for i, c in enumerate(chunks):
        inter.append(c)
        if i < len(chunks) - 1:
            inter.append(sep)
    for x in inter:
        ids.extend(x[offset:])
    return ids


# ----------------------------------------------------------------------------- splice (llava_arch.py:251-555)
def merge_image_features(feats_per_sample, cfg, P, image_sizes, side):
    """Patch-merge step, llava_arch.py:293-413 ('flat' :298-299; 'spatial_unpad' + anyres :350-412)."""
    mt = cfg.get("mm_patch_merge_type", "flat")
    if mt == "flat":
        return [f.flatten(0, 1) for f in feats_per_sample]
    assert mt.startswith("spatial")
    out = []
    for idx, f in enumerate(feats_per_sample):
        if f.shape[0] > 1:
            base, rest = f[0], f[1:]
            gw, gh = get_anyres_image_grid_shape(image_sizes[idx], cfg["image_grid_pinpoints"], cfg["tower_image_size"])
            rest = rest.view(gh, gw, side, side, -1)
            if "maxpool2x2" in mt:   # llava_arch.py:375-379
                rest = rest.permute(4, 0, 2, 1, 3).contiguous().flatten(1, 2).flatten(2, 3)
                rest = F.max_pool2d(rest, 2).flatten(1, 2).transpose(0, 1)
            elif "unpad" in mt:
                rest = rest.permute(4, 0, 2, 1, 3).contiguous().flatten(1, 2).flatten(2, 3)
                rest = unpad_image(rest, image_sizes[idx])
                mx = re.match(r"anyres_max_(\d+)", cfg.get("image_aspect_ratio", ""))
                if mx:   # llava_arch.py:381-392: bilinear down-sampling once the grid exceeds max_num_patches tiles
                    c, hh, ww = rest.shape
                    times = math.sqrt(hh * ww / (int(mx.group(1)) * side ** 2))
                    if times > 1.1:
                        rest = F.interpolate(rest[None], [int(hh // times), int(ww // times)], mode="bilinear")[0]
                nl = P["model.image_newline"][:, None, None].expand(*rest.shape[:-1], 1)
                rest = torch.cat((rest, nl), dim=-1).flatten(1, 2).transpose(0, 1)
            else:
                rest = rest.permute(0, 2, 1, 3, 4).contiguous().flatten(0, 3)
            out.append(torch.cat((base, rest), dim=0))
        else:
            f0 = f[0]
            if "unpad" in mt:
                f0 = torch.cat((f0, P["model.image_newline"][None]), dim=0)
            out.append(f0)
    return out


def splice(P, input_ids, attention_mask, labels, image_features, max_len=None):
    """prepare_inputs_labels_for_multimodal steps (v)-(ix), llava_arch.py:442-545 (right padding).

    Returns (inputs_embeds [b,S,d] with exact-zero pad rows, labels [b,S], attention_mask [b,S] bool, lens).
    """
    emb_w = P["model.embed_tokens.weight"]
    new_e, new_l = [], []
    img_i = 0
    for b in range(input_ids.shape[0]):
        ids = input_ids[b][attention_mask[b]]
        lab = labels[b][attention_mask[b]]
        pos = (ids == IMAGE_TOKEN_INDEX).nonzero().flatten().tolist()
        if not pos:
            new_e.append(torch.cat([emb_w[ids], image_features[img_i][0:0]], 0))
            new_l.append(lab)
            img_i += 1
            continue
        bounds = [-1] + pos + [ids.shape[0]]
        pe, pl = [], []
        for i in range(len(bounds) - 1):
            seg = slice(bounds[i] + 1, bounds[i + 1])
            pe.append(emb_w[ids[seg]])
            pl.append(lab[seg])
            if i < len(pos):
                f = image_features[min(img_i, len(image_features) - 1)]
                img_i += 1
                pe.append(f)
                pl.append(torch.full((f.shape[0],), IGNORE_INDEX, dtype=lab.dtype))
        new_e.append(torch.cat(pe, 0)[:max_len])
        new_l.append(torch.cat(pl, 0)[:max_len])
    S = max(e.shape[0] for e in new_e)
    B = len(new_e)
    E = torch.zeros(B, S, emb_w.shape[1], dtype=emb_w.dtype)
    L = torch.full((B, S), IGNORE_INDEX, dtype=labels.dtype)
    M = torch.zeros(B, S, dtype=torch.bool)
    lens = []
    for b, (e, l) in enumerate(zip(new_e, new_l)):
        n = e.shape[0]
        E[b, :n], L[b, :n], M[b, :n] = e, l, True
        lens.append(n)
    return E, L, M, lens


def llava_forward(P, geo, input_ids, attention_mask, labels, images, image_sizes=None, cfg=None):
    """LlavaLlamaForCausalLM.forward, language_model/llava_llama.py:69-120 -> (loss, logits, aux dict)."""
    cfg = dict(cfg or {})
    cfg.setdefault("tower_image_size", geo["vision"]["image"])
    imgs = [x[None] if x.ndim == 3 else x for x in images]
    split = [x.shape[0] for x in imgs]
    feats = encode_images(P, geo, torch.cat(imgs, 0))
    per = list(torch.split(feats, split))
    side = geo["vision"]["image"] // geo["vision"]["patch"]
    merged = merge_image_features(per, cfg, P, image_sizes, side)
    E, L, M, lens = splice(P, input_ids, attention_mask, labels, merged, cfg.get("tokenizer_model_max_length"))
    logits = llama_forward(P, geo, E, lens)
    loss = causal_lm_loss(logits, L)
    return loss, logits, dict(inputs_embeds=E, labels=L, attention_mask=M, lens=lens, image_features=feats)


# ----------------------------------------------------------------------------- LoRA
LORA_TARGETS = ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj",
                "mlp.down_proj")


def apply_lora(P, L, geo, scale):
    """peft 0.4.0 LoraLayer at dropout 0 (train/train.py:1515-1532 wires r=64, alpha=16 on every LM linear except
    lm_head, :242-255): y = x W^T + (alpha/r) x A^T B^T  ==  x (W + (alpha/r) B A)^T.  peft is not installed here and the
    reference holds no LoRA fixtures: PARITY UNPINNED for this function (restated from the published formula).
    Returns a parameter dict with the effective weights; autograd reaches A and B through it."""
    out = dict(P)
    for i in range(geo["lm"]["layers"]):
        for t in LORA_TARGETS:
            n = f"model.layers.{i}.{t}."
            out[n + "weight"] = P[n + "weight"] + scale * (L[n + "lora_B.weight"] @ L[n + "lora_A.weight"])
    return out


# ----------------------------------------------------------------------------- optimizer (torch.optim.AdamW semantics)
def adamw_step(p, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8, wd=0.0):
    """torch.optim.AdamW single-tensor update (optim='adamw_torch', train/train.py:140), fp32, in place."""
    p.mul_(1 - lr * wd)
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)
