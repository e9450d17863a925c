"""Reading pretrained checkpoints from LOCAL directories into the engine's flat stores.

What the reference gets from ``model_class.from_pretrained(model_name_or_path)`` (finetuning/llava/train/train.py:1358-1427)
and ``CLIPVisionModel.from_pretrained(vision_tower)`` (model/multimodal_encoder/clip_encoder.py:35-44): the language model
and the vision tower start from saved weights, never from random values.  Only loaders that execute nothing from the file
are used: safetensors, or ``torch.load(..., weights_only=True)`` for ``*.bin`` shards.  Nothing is ever fetched: a name
that is not a local directory raises.
"""
import json
import os

import torch

from .params import VP

TOWER_PREFIX = "model.vision_tower.vision_tower."


def _weight_files(path):
    """Weight files of an HF-format checkpoint directory, shard order: the safetensors / bin index if present, else the
    single-file names."""
    for index, single in (("model.safetensors.index.json", "model.safetensors"), ("pytorch_model.bin.index.json", "pytorch_model.bin")):
        ip = os.path.join(path, index)
        if os.path.exists(ip):
            with open(ip) as f:
                files = sorted(set(json.load(f)["weight_map"].values()))
            return [os.path.join(path, x) for x in files]
        sp = os.path.join(path, single)
        if os.path.exists(sp):
            return [sp]
    return []


def iter_tensors(path):
    """(name, cpu tensor) for every tensor of the checkpoint directory ``path``, one shard in memory at a time."""
    files = _weight_files(path)
    if not files:
        raise FileNotFoundError(f"{path}: no model.safetensors / pytorch_model.bin (or their sharded index) found")
    for f in files:
        if f.endswith(".safetensors"):
            from safetensors import safe_open
            with safe_open(f, framework="pt", device="cpu") as sf:
                for k in sf.keys():
                    yield k, sf.get_tensor(k)
        else:
            sd = torch.load(f, map_location="cpu", weights_only=True)
            for k, v in sd.items():
                yield k, v
            del sd


def read_config(path):
    cp = os.path.join(path, "config.json")
    if not os.path.exists(cp):
        raise FileNotFoundError(f"{path}: config.json not found (a local HF-format checkpoint directory is required; nothing is downloaded)")
    with open(cp) as f:
        return json.load(f)


def lm_geometry_from_config(cfg):
    """The decoder half of a geometry dict from an HF Llama / Qwen2 (or LLaVA) config.json."""
    heads = cfg["num_attention_heads"]
    g = dict(d=cfg["hidden_size"], heads=heads, ffn=cfg["intermediate_size"], layers=cfg["num_hidden_layers"], vocab=cfg["vocab_size"])
    kv = cfg.get("num_key_value_heads", heads) or heads
    if kv != heads:
        g["kv_heads"] = kv
    if "qwen" in (cfg.get("model_type") or "").lower():
        g["qkv_bias"] = True
    if cfg.get("rope_theta") is not None:
        g["rope_theta"] = float(cfg["rope_theta"])
    if cfg.get("rms_norm_eps") is not None:
        g["rms_eps"] = float(cfg["rms_norm_eps"])
    return g


def vision_geometry_from_config(cfg):
    """The tower half of a geometry dict from a CLIP / SigLIP config.json (either the vision config itself or a full
    CLIPConfig / SiglipConfig holding it under 'vision_config')."""
    v = cfg.get("vision_config", cfg)
    g = dict(d=v["hidden_size"], heads=v["num_attention_heads"], ffn=v["intermediate_size"], layers=v["num_hidden_layers"],
             image=v["image_size"], patch=v["patch_size"])
    if "siglip" in (v.get("model_type") or cfg.get("model_type") or "").lower():
        g["kind"] = "siglip"
    return g


def tower_key(k):
    """A CLIP / SigLIP checkpoint key -> the LLaVA state-dict name (model.vision_tower.vision_tower.vision_model.*), or None for
    tensors outside the vision model (text tower, projections, logit scale, the pooling head)."""
    if k.startswith(TOWER_PREFIX):
        k = k[len(TOWER_PREFIX):]
    if k.startswith("vision_model."):
        k = k[len("vision_model."):]
    elif k.startswith(("text_model.", "visual_projection", "text_projection", "logit_")):
        return None
    if k.startswith("head."):
        return None
    if not k.startswith(("embeddings.", "encoder.", "pre_layrnorm.", "post_layernorm.")):
        return None
    if k.endswith("embeddings.position_ids"):
        return None
    return VP + k


def load_pretrained(engine, lm_path=None, tower_path=None, allow_missing=("model.mm_projector.", "model.image_newline")):
    """Fill the engine's stores from checkpoint directories.  ``lm_path``: an HF Llama / Qwen2 (or full LLaVA) checkpoint;
    ``tower_path``: a CLIP / SigLIP checkpoint (skipped when the LM checkpoint already holds the tower).  Raises KeyError if
    any language-model or tower tensor stays unset -- there is no fall-back to random values (the projector and
    image_newline may be absent: the reference initialises them fresh too, llava_arch.py:99-113)."""
    from .config import canonical_name
    stores = [engine.lm] + ([] if engine.vis is engine.lm else [engine.vis]) + ([engine.base] if engine.base is not None else [])
    seen = set()

    def put(name, t):
        name = canonical_name(name)
        hit = False
        for st in stores:
            if name in st.offsets:
                dst = st.view(name)
                if t.ndim == 2 and t.shape[0] < dst.shape[0] and tuple(t.shape[1:]) == tuple(dst.shape[1:]):
                    dst[:t.shape[0]].copy_(t.to(dst.dtype))      # vocabulary tables padded to a multiple of 8 rows
                    dst[t.shape[0]:].zero_()
                elif tuple(t.shape) == tuple(dst.shape):
                    dst.copy_(t.to(dst.dtype))
                elif t.numel() == dst.numel() and (t.ndim == 1 or dst.ndim == 1 or (t.ndim == 4 and dst.ndim == 2 and t.shape[0] == dst.shape[0])):
                    # the only reshapes a checkpoint may need: the conv patch embedding [d,3,p,p] -> [d,3*p*p] and 1-D tensors stored
                    # with a unit axis; any other same-size shape (a transposed or fused projection) would load as wrong weights
                    dst.copy_(t.to(dst.dtype).reshape(dst.shape))
                else:
                    raise ValueError(f"{name}: checkpoint shape {tuple(t.shape)} does not match the model's {tuple(dst.shape)}")
                hit = True
        if hit:
            seen.add(name)
        return hit

    if lm_path:
        for k, t in iter_tensors(lm_path):
            if k.endswith("rotary_emb.inv_freq"):
                continue
            put(k, t)
    if tower_path and not any(n.startswith(VP) for n in seen):
        for k, t in iter_tensors(tower_path):
            tk = tower_key(k)
            if tk is not None:
                put(tk, t)
    missing = []
    for st in stores:
        for n in st.offsets:
            if n not in seen and not n.startswith(tuple(allow_missing)) and ".lora_" not in n:
                if st is engine.base and n in engine.lm.offsets:
                    continue
                missing.append(n)
    if missing:
        raise KeyError(f"pretrained weights missing for {len(missing)} tensors (first: {sorted(set(missing))[:4]}); "
                       f"lm_path={lm_path!r} tower_path={tower_path!r}")
    engine.weights_changed()
    if engine.master is not None:
        from . import ops
        engine.master.copy_(ops.to_f32(engine.lm.flat))
    return sorted(seen)
