"""Flat parameter storage for the LLaVA hot path.

MI355X-first layout: every trainable tensor lives in ONE contiguous bf16 buffer (and its gradient, fp32 master
copy and AdamW moments in buffers of the same layout), in forward execution order.  Consequences:
  * q/k/v and gate/up weights are adjacent, so the fused [3d,d] / [2F,d] GEMM operands are plain views;
  * data-parallel gradient buckets are contiguous slices that become ready back-to-front during backward
    and are all-reduced in place (no bucket copies);
  * the optimizer is a handful of launches over large slices.
State-dict names are the reference's: model.embed_tokens, model.layers.{i}.self_attn.{q,k,v,o}_proj,
mlp.{gate,up,down}_proj, input_layernorm, post_attention_layernorm, model.norm, lm_head, model.mm_projector.{0,2},
model.image_newline, model.vision_tower.vision_tower.vision_model.* (transformers-4 layout; the 5.x layout without the
``vision_model.`` level is accepted on load).
"""
from collections import OrderedDict

import numpy as np
import torch

from .config import canonical_name, init_std_for

ALIGN = 64  # elements (128 B)
VP = "model.vision_tower.vision_tower.vision_model."


def _ru(x, m):
    return (x + m - 1) // m * m


def lm_param_shapes(geo, with_newline=False):
    """Trainable tensors in forward order."""
    v, l = geo["vision"], geo["lm"]
    s = OrderedDict()
    s["model.embed_tokens.weight"] = (l["vocab"], l["d"])
    s["model.mm_projector.0.weight"] = (l["d"], v["d"])
    s["model.mm_projector.0.bias"] = (l["d"],)
    s["model.mm_projector.2.weight"] = (l["d"], l["d"])
    s["model.mm_projector.2.bias"] = (l["d"],)
    if with_newline:
        s["model.image_newline"] = (l["d"],)
    kvd = l["d"] // l["heads"] * l.get("kv_heads", l["heads"])   # grouped-query k/v width (Qwen2)
    for i in range(l["layers"]):
        p = f"model.layers.{i}."
        s[p + "input_layernorm.weight"] = (l["d"],)
        for n, rows in (("q_proj", l["d"]), ("k_proj", kvd), ("v_proj", kvd)):
            s[p + f"self_attn.{n}.weight"] = (rows, l["d"])
        if l.get("qkv_bias"):
            for n, rows in (("q_proj", l["d"]), ("k_proj", kvd), ("v_proj", kvd)):
                s[p + f"self_attn.{n}.bias"] = (rows,)
        s[p + "self_attn.o_proj.weight"] = (l["d"], l["d"])
        s[p + "post_attention_layernorm.weight"] = (l["d"],)
        s[p + "mlp.gate_proj.weight"] = (l["ffn"], l["d"])
        s[p + "mlp.up_proj.weight"] = (l["ffn"], l["d"])
        s[p + "mlp.down_proj.weight"] = (l["d"], l["ffn"])
    s["model.norm.weight"] = (l["d"],)
    s["lm_head.weight"] = (l["vocab"], l["d"])
    return s


LORA_TARGETS = (("self_attn.q_proj", "d", "d"), ("self_attn.k_proj", "d", "d"), ("self_attn.v_proj", "d", "d"),
                ("self_attn.o_proj", "d", "d"), ("mlp.gate_proj", "ffn", "d"), ("mlp.up_proj", "ffn", "d"),
                ("mlp.down_proj", "d", "ffn"))


def lora_trainable_shapes(geo, r, with_newline=False):
    """Trainables of a LoRA run (train/train.py:1515-1532 + :1613-1665): the projector (non-LoRA trainable) and, for every
    LM linear except lm_head (find_all_linear_names :242-255), lora_A [r, in] and lora_B [out, r] -- forward order."""
    v, l = geo["vision"], geo["lm"]
    s = OrderedDict()
    s["model.mm_projector.0.weight"] = (l["d"], v["d"])
    s["model.mm_projector.0.bias"] = (l["d"],)
    s["model.mm_projector.2.weight"] = (l["d"], l["d"])
    s["model.mm_projector.2.bias"] = (l["d"],)
    if with_newline:
        s["model.image_newline"] = (l["d"],)
    kvd = l["d"] // l["heads"] * l.get("kv_heads", l["heads"])      # grouped-query k/v projections are narrower (Qwen2)
    for i in range(l["layers"]):
        for t, o, k in LORA_TARGETS:
            out = kvd if t in ("self_attn.k_proj", "self_attn.v_proj") else l[o]
            s[f"model.layers.{i}.{t}.lora_A.weight"] = (r, l[k])
            s[f"model.layers.{i}.{t}.lora_B.weight"] = (out, r)
    return s


def vision_param_shapes(geo):
    v = geo["vision"]
    siglip = v.get("kind") == "siglip"   # siglip_encoder.py:148-174: conv bias, no class token, no pre-LN
    npos = (v["image"] // v["patch"]) ** 2 + (0 if siglip else 1)
    s = OrderedDict()
    if not siglip:
        s[VP + "embeddings.class_embedding"] = (v["d"],)
    s[VP + "embeddings.patch_embedding.weight"] = (v["d"], 3, v["patch"], v["patch"])
    if siglip:
        s[VP + "embeddings.patch_embedding.bias"] = (v["d"],)
    s[VP + "embeddings.position_embedding.weight"] = (npos, v["d"])
    if not siglip:
        s[VP + "pre_layrnorm.weight"] = (v["d"],)
        s[VP + "pre_layrnorm.bias"] = (v["d"],)
    for i in range(v["layers"]):
        p = VP + f"encoder.layers.{i}."
        s[p + "layer_norm1.weight"] = (v["d"],)
        s[p + "layer_norm1.bias"] = (v["d"],)
        for n in ("q_proj", "k_proj", "v_proj"):
            s[p + f"self_attn.{n}.weight"] = (v["d"], v["d"])
        for n in ("q_proj", "k_proj", "v_proj"):
            s[p + f"self_attn.{n}.bias"] = (v["d"],)
        s[p + "self_attn.out_proj.weight"] = (v["d"], v["d"])
        s[p + "self_attn.out_proj.bias"] = (v["d"],)
        s[p + "layer_norm2.weight"] = (v["d"],)
        s[p + "layer_norm2.bias"] = (v["d"],)
        s[p + "mlp.fc1.weight"] = (v["ffn"], v["d"])
        s[p + "mlp.fc1.bias"] = (v["ffn"],)
        s[p + "mlp.fc2.weight"] = (v["d"], v["ffn"])
        s[p + "mlp.fc2.bias"] = (v["d"],)
    s[VP + "post_layernorm.weight"] = (v["d"],)
    s[VP + "post_layernorm.bias"] = (v["d"],)
    return s


class FlatParams:
    """name -> view table over one flat buffer. Tensors that must stay adjacent (q|k|v, gate|up, and their biases)
    are packed without padding between them; every group starts ALIGN-aligned."""

    FUSE_AFTER = ("q_proj.weight", "k_proj.weight", "gate_proj.weight", "q_proj.bias", "k_proj.bias")

    def __init__(self, shapes, device, dtype=torch.bfloat16):
        self.shapes = shapes
        self.offsets = OrderedDict()
        off = 0
        prev = None
        for name, shp in shapes.items():
            n = int(np.prod(shp))
            if not (prev is not None and prev.endswith(self.FUSE_AFTER)):
                off = _ru(off, ALIGN)
            self.offsets[name] = (off, n)
            off += n
            prev = name
        self.numel = _ru(off, ALIGN)
        self.device = device
        self.flat = torch.zeros(self.numel, dtype=dtype, device=device)

    def like(self, dtype):
        return torch.zeros(self.numel, dtype=dtype, device=self.device)

    def view(self, name, flat=None):
        off, n = self.offsets[name]
        return (self.flat if flat is None else flat)[off:off + n].view(self.shapes[name])

    def span(self, first, last):
        """[start, end) element range covering tensors first..last (inclusive, must be adjacent in order)."""
        o0, _ = self.offsets[first]
        o1, n1 = self.offsets[last]
        return o0, o1 + n1

    def fused(self, first, last, rows, cols, flat=None):
        s, e = self.span(first, last)
        assert e - s == rows * cols, (first, last, e - s, rows, cols)
        return (self.flat if flat is None else flat)[s:e].view(rows, cols)

    def names(self):
        return list(self.shapes.keys())


def portable_init_(fp: FlatParams, lm_hidden: int, seed=0):
    """Fill from the portable generator (same convention as tests/golden/make_golden.py)."""
    from . import portable_rng as prng
    for name, shp in fp.shapes.items():
        kind, std = init_std_for(name, lm_hidden)
        w = prng.normal(seed, prng.name_tag(name), tuple(shp), std)
        if kind == "norm_weight":
            w = 1.0 + w
        fp.view(name).copy_(torch.from_numpy(w).to(fp.flat.dtype))


def fast_random_init_(fp: FlatParams, lm_hidden: int, seed=0):
    """Device-side random init of the same distribution family (bench only: 7B of portable_rng is too slow)."""
    g = torch.Generator(device=fp.device)
    g.manual_seed(seed)
    for name, shp in fp.shapes.items():
        kind, std = init_std_for(name, lm_hidden)
        v = fp.view(name)
        t = torch.empty(v.shape, dtype=torch.float32, device=fp.device).normal_(0.0, std, generator=g)
        if kind == "norm_weight":
            t += 1.0
        v.copy_(t.to(v.dtype))


def load_named(fp: FlatParams, state_dict, strict=False):
    """Copy tensors of a (reference-format) state dict into the flat views. Returns (missing, unexpected)."""
    seen = set()
    unexpected = []
    for k, t in state_dict.items():
        ck = canonical_name(k)
        if ck in fp.offsets:
            dst = fp.view(ck)
            if t.ndim == 2 and t.shape[0] < dst.shape[0] and t.shape[1:] == dst.shape[1:]:
                dst[:t.shape[0]].copy_(t.to(fp.flat.dtype))     # vocabulary tables padded to a multiple of 8 rows
            else:
                dst.copy_(t.to(fp.flat.dtype).reshape(fp.shapes[ck]))
            seen.add(ck)
        else:
            unexpected.append(k)
    missing = [n for n in fp.offsets if n not in seen]
    if strict and (missing or unexpected):
        raise KeyError(f"missing={missing[:5]} unexpected={unexpected[:5]}")
    return missing, unexpected
