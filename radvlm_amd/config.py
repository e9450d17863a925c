"""Model geometries of the hot path and the portable weight-init convention.

The hyper-parameters of the public checkpoints (CLIP-ViT-L/14-336, Vicuna-7B/13B v1.5) are external
knowledge fixed here; no weights are ever fetched (random init from ``portable_rng``).
"""

GEOMETRIES = {
    # toy geometry for golden fixtures and GPU parity tests (exercises tile edges: ffn 448, vocab 1000)
    "toy": {
        "vision": dict(d=128, heads=2, ffn=256, layers=3, image=56, patch=14),
        "lm": dict(d=256, heads=2, ffn=448, layers=2, vocab=1000),
    },
    # BASELINE.json configs[0]: ViT-B/32-224 + 125M-scale Llama-style decoder (SURVEY 8d note)
    "config1": {
        "vision": dict(d=768, heads=12, ffn=3072, layers=12, image=224, patch=32),
        "lm": dict(d=768, heads=12, ffn=3072, layers=12, vocab=32000),
    },
    # BASELINE.json configs[1..2]: LLaVA-1.5-7B = CLIP-ViT-L/14-336 + Vicuna-7B-v1.5
    "llava15_7b": {
        "vision": dict(d=1024, heads=16, ffn=4096, layers=24, image=336, patch=14),
        "lm": dict(d=4096, heads=32, ffn=11008, layers=32, vocab=32000),
    },
    # the configuration RadVLM really trains (finetune_radio_7b.sh:21-60) = LLaVA-OneVision
    # Qwen2-7B decoder (GQA 28/4 heads, q/k/v bias, rope theta 1e6) + SigLIP-so400m-patch14-384 tower (729 tokens,
    # head_dim 72, gelu_tanh MLP, last layer dropped: siglip_encoder.py:538-590)
    "llava_ov_qwen2_7b": {
        "vision": dict(kind="siglip", d=1152, heads=16, ffn=4304, layers=27, image=384, patch=14),
        "lm": dict(d=3584, heads=28, kv_heads=4, ffn=18944, layers=28, vocab=152064, qkv_bias=True, rope_theta=1e6,
                   rms_eps=1e-6),
    },
    # toy version of the above for golden fixtures: 27x27 patches (the reference asserts 729 tokens,
    # siglip_encoder.py:581/586), head_dim 24 (exercises the head padding), ffn 200 (K tail), GQA 4/2
    "toy_qwen": {
        "vision": dict(kind="siglip", d=96, heads=4, ffn=200, layers=3, image=54, patch=2),
        "lm": dict(d=256, heads=4, kv_heads=2, ffn=448, layers=2, vocab=1000, qkv_bias=True, rope_theta=1e6, rms_eps=1e-6),
    },
    # BASELINE.json configs[4]: LLaVA-1.5-13B
    "llava15_13b": {
        "vision": dict(d=1024, heads=16, ffn=4096, layers=24, image=336, patch=14),
        "lm": dict(d=5120, heads=40, ffn=13824, layers=40, vocab=32000),
    },
}


def canonical_name(key: str) -> str:
    """Map a state-dict key of any transformers version onto the 4.x-era LLaVA key layout.

    transformers 5.x drops the ``vision_model.`` level inside CLIPVisionModel.
    """
    pre = "model.vision_tower.vision_tower."
    if key.startswith(pre) and not key[len(pre):].startswith("vision_model."):
        key = pre + "vision_model." + key[len(pre):]
    return key


def init_std_for(cname: str, lm_hidden: int):
    """(kind, std) used by the portable init; kind == 'norm_weight' means value = 1 + N(0, std)."""
    leaf = cname.rsplit(".", 1)[-1]
    low = cname.lower()
    if ("norm" in low or "layrnorm" in low) and leaf == "weight":
        return "norm_weight", 0.1
    if leaf == "bias":
        return "bias", 0.02
    if cname.endswith("image_newline"):
        return "newline", lm_hidden ** -0.5
    return "matrix", 0.02
