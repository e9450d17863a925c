"""LLaVA-OneVision -> HF converter contract.

The reference converts a trained checkpoint with radvlm/evaluation/convert_llava_onevision_weights_to_hf.py: it reads every
``*.safetensors`` of the checkpoint directory, ties ``lm_head.weight`` to ``model.embed_tokens.weight`` when the head is absent
(:65-77), drops ``*.inv_freq``, renames keys through an ORDERED substring table (:49-59, :79-90) and casts to fp16; ``config.json`` must
name the tower under ``mm_vision_tower`` (:102-117).  This module states the contract from the producing side:

* ``hf_key(name)`` / ``convert_state_dict_to_hf(sd)`` -- the same renaming, so that a run can emit the HF layout directly;
* ``check_checkpoint_dir(path)`` -- what the reference converter needs from a directory written by ``save_pretrained``: safetensors under
  LLaVA names whose converted keys are exactly the HF LlavaOnevision parameter names, and ``mm_vision_tower`` in config.json.

Pinned by tests/golden/host_convert_keys.json (the reference's own table and function executed on this package's key set).
"""
import json
import os

# applied in this order to every key; each replacement sees the result of the previous ones (convert_llava_onevision_weights_to_hf.py:49-59)
RENAMES = (
    ("model.vision_tower.", ""),
    ("model.mm_projector", "multi_modal_projector"),
    ("model", "model.model"),
    ("vision_model.model", "vision_model"),
    ("lm_head", "language_model.lm_head"),
    ("model.model", "language_model.model"),
    ("multi_modal_projector.0", "multi_modal_projector.linear_1"),
    ("multi_modal_projector.2", "multi_modal_projector.linear_2"),
    ("language_model.model.image_newline", "image_newline"),
)


def hf_key(name):
    """HF LlavaOnevisionForConditionalGeneration parameter name of a LLaVA state-dict key; None for keys the converter drops."""
    if name.endswith(".inv_freq"):
        return None
    for old, new in RENAMES:
        if old in name:
            name = name.replace(old, new)
    return name


def convert_state_dict_to_hf(state_dict, dtype=None):
    """Renamed (and, with dtype, cast: the reference writes fp16) copy of a LLaVA state dict; lm_head tied to the embeddings if absent."""
    sd = dict(state_dict)
    if "lm_head.weight" not in sd and "model.embed_tokens.weight" in sd:
        sd["lm_head.weight"] = sd["model.embed_tokens.weight"].clone()
    out = {}
    for k, v in sd.items():
        nk = hf_key(k)
        if nk is not None:
            out[nk] = v.to(dtype) if dtype is not None else v
    return out


def check_checkpoint_dir(path):
    """Raise if `path` is not something the reference converter can consume; returns the converted key list."""
    from safetensors import safe_open
    files = [f for f in os.listdir(path) if f.endswith(".safetensors")]
    if not files:
        raise FileNotFoundError(f"{path}: the converter reads *.safetensors (load_original_state_dict)")
    with open(os.path.join(path, "config.json")) as f:
        cfg = json.load(f)
    if not cfg.get("mm_vision_tower"):
        raise KeyError("config.json lacks mm_vision_tower (the converter builds its processors from it)")
    keys = []
    for fn in files:
        with safe_open(os.path.join(path, fn), framework="pt", device="cpu") as sf:
            keys += list(sf.keys())
    conv = [hf_key(k) for k in keys]
    bad = [k for k, c in zip(keys, conv) if c is not None and not c.startswith(("language_model.", "vision_tower.", "multi_modal_projector.", "image_newline"))]
    if bad:
        raise KeyError(f"keys outside the HF LlavaOnevision layout after conversion: {bad[:4]}")
    return [c for c in conv if c is not None]
