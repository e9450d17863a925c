"""Generate golden fixtures from the REFERENCE (build container only; needs /root/reference).

    python tests/golden/make_golden.py [per_op host toy cfg1 anyres]   # writes tests/golden/*.npz / *.json

The reference's glue (llava_arch.py / llava_llama.py / mm_utils.py, imported in place through
ref_shim) runs on top of the installed transformers CLIP/Llama (eager attention, CPU fp32) with
weights from radvlm_amd.portable_rng.  Only inputs/outputs are stored (data, not code); weights are
regenerated from the seed on the consumer side.  See SURVEY.md section 8c for why this is the oracle.
"""
import json
import os
import sys
import tempfile
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)

from radvlm_amd import portable_rng as prng  # noqa: E402
from radvlm_amd.config import GEOMETRIES, canonical_name, init_std_for  # noqa: E402
import ref_shim  # noqa: E402


class _ProcAdapter:
    """transformers-5 SizeDict -> plain dict adapter (SURVEY 8c skew list)."""

    def __init__(self, size, mean):
        self.size = {"shortest_edge": size}
        self.crop_size = {"height": size, "width": size}
        self.image_mean = mean


def build_reference_model(mods, geo, seed=0, merge_type="flat", aspect="square", pinpoints=None):
    from transformers import CLIPVisionConfig, CLIPVisionModel, CLIPImageProcessor
    v = geo["vision"]
    l = geo["lm"]
    tmp = tempfile.mkdtemp(prefix="clip_local_")
    vcfg = CLIPVisionConfig(hidden_size=v["d"], intermediate_size=v["ffn"], num_hidden_layers=v["layers"],
                            num_attention_heads=v["heads"], image_size=v["image"], patch_size=v["patch"],
                            hidden_act="quick_gelu", layer_norm_eps=1e-5, projection_dim=v["d"])
    CLIPVisionModel(vcfg).save_pretrained(tmp)
    CLIPImageProcessor(size={"shortest_edge": v["image"]},
                       crop_size={"height": v["image"], "width": v["image"]}).save_pretrained(tmp)
    LlavaConfig = mods["llava_llama"].LlavaConfig
    cfg = LlavaConfig(hidden_size=l["d"], intermediate_size=l["ffn"], num_hidden_layers=l["layers"],
                      num_attention_heads=l["heads"], num_key_value_heads=l["heads"], vocab_size=l["vocab"],
                      rms_norm_eps=1e-5, max_position_embeddings=4096, rope_theta=10000.0,
                      attention_bias=False, tie_word_embeddings=False, attn_implementation="eager")
    cfg.mm_vision_tower = tmp
    cfg.mm_projector_type = "mlp2x_gelu"
    cfg.mm_hidden_size = v["d"]
    cfg.mm_vision_select_layer = -2
    cfg.mm_vision_select_feature = "patch"
    cfg.mm_patch_merge_type = merge_type
    cfg.image_aspect_ratio = aspect
    if pinpoints is not None:
        cfg.image_grid_pinpoints = pinpoints
    cfg.use_cache = False
    model = mods["llava_llama"].LlavaLlamaForCausalLM(cfg).float()
    sd = model.state_dict()
    new = {}
    for k, t in sd.items():
        ck = canonical_name(k)
        kind, std = init_std_for(ck, l["d"])
        w = prng.normal(seed, prng.name_tag(ck), tuple(t.shape), std)
        if kind == "norm_weight":
            w = 1.0 + w
        new[k] = torch.from_numpy(w).to(t.dtype) if t.dtype.is_floating_point else t
    model.load_state_dict(new)
    model.train()
    return model


def make_batch(geo, spec, seed_img=1, seed_ids=2, tiles=None, left=False):
    """spec: list of (n_ids, image_pos or None, n_ignored_prefix)."""
    v = geo["vision"]
    V = geo["lm"]["vocab"]
    T = max(s[0] for s in spec)
    ids = np.zeros((len(spec), T), dtype=np.int64)
    labels = np.full((len(spec), T), -100, dtype=np.int64)
    mask = np.zeros((len(spec), T), dtype=bool)
    images = []
    modalities = []
    for i, (n, pos, nign) in enumerate(spec):
        row = prng.integers(seed_ids, 10 + i, (n,), 3, V)
        lab = row.copy()
        lab[:nign] = -100
        shape = (3, v["image"], v["image"]) if tiles is None else (tiles[i], 3, v["image"], v["image"])
        if pos is not None:
            row[pos] = -200
            lab[pos] = -100
            images.append(prng.normal(seed_img, 100 + i, shape, 1.0))
            modalities.append("image")
        else:
            images.append(np.zeros((3, v["image"], v["image"]), dtype=np.float32))
            modalities.append("text")
        sl = slice(T - n, T) if left else slice(0, n)      # left: what the collator builds for tokenizer.padding_side == "left"
        ids[i, sl] = row
        labels[i, sl] = lab
        mask[i, sl] = True
    return ids, labels, mask, images, modalities


def build_reference_qwen_model(mods, geo, seed=0, merge_type="flat", aspect="square", pinpoints=None):
    """LlavaQwenForCausalLM (language_model/llava_qwen.py:46-58) over a SigLipVisionTower (siglip_encoder.py:538-590).

    build_vision_tower only reaches SigLipVisionTower for a hub name (builder.py:16-22) and load_model() would fetch it,
    so the tower is created with delay_load and load_model's three steps (:568-572: build SigLipVisionModel, delete the
    last encoder layer, head = Identity) are performed here on a randomly initialised toy-sized SigLipVisionModel."""
    import importlib
    lq = importlib.import_module("llava.model.language_model.llava_qwen")
    se = importlib.import_module("llava.model.multimodal_encoder.siglip_encoder")
    v, l = geo["vision"], geo["lm"]

    class _Cfg(lq.LlavaQwenConfig):
        """transformers-5 skew: there ``rope_scaling`` aliases ``rope_parameters`` (which also holds theta), so the
        reference's 4.x-era ``config.rope_scaling = None`` (llava_qwen.py:53) would delete the rope setup; ignore it
        (in 4.x it is a no-op for a config without scaling)."""

        def __setattr__(self, k, val):
            if k == "rope_scaling" and val is None:
                return
            super().__setattr__(k, val)

    cfg = _Cfg(hidden_size=l["d"], intermediate_size=l["ffn"], num_hidden_layers=l["layers"],
                             num_attention_heads=l["heads"], num_key_value_heads=l["kv_heads"], vocab_size=l["vocab"],
                             rms_norm_eps=l["rms_eps"], max_position_embeddings=8192, rope_theta=l["rope_theta"],
                             tie_word_embeddings=False, attn_implementation="eager", use_sliding_window=False)
    cfg.mm_vision_tower = "toy/siglip-27x27"      # not a path: builder.py routes the name to SigLipVisionTower
    cfg.delay_load = True
    cfg.mm_projector_type = "mlp2x_gelu"
    cfg.mm_hidden_size = v["d"]
    cfg.mm_vision_select_layer = -2
    cfg.mm_vision_select_feature = "patch"
    cfg.mm_patch_merge_type = merge_type
    cfg.image_aspect_ratio = aspect
    if pinpoints is not None:
        cfg.image_grid_pinpoints = pinpoints
    cfg.use_cache = False
    model = lq.LlavaQwenForCausalLM(cfg)
    tower = model.get_model().get_vision_tower()
    assert isinstance(tower, se.SigLipVisionTower) and not tower.is_loaded
    scfg = se.SigLipVisionConfig(hidden_size=v["d"], intermediate_size=v["ffn"], num_hidden_layers=v["layers"],
                                 num_attention_heads=v["heads"], image_size=v["image"], patch_size=v["patch"])
    scfg._attn_implementation = "eager"
    tower.config = scfg
    tower.vision_tower = se.SigLipVisionModel(scfg)
    del tower.vision_tower.vision_model.encoder.layers[-1:]
    tower.vision_tower.vision_model.head = torch.nn.Identity()
    tower.vision_tower.requires_grad_(False)
    tower.is_loaded = True
    model = model.float()
    sd = model.state_dict()
    new = {}
    for k, t in sd.items():
        ck = canonical_name(k)
        kind, std = init_std_for(ck, l["d"])
        w = prng.normal(seed, prng.name_tag(ck), tuple(t.shape), std)
        if kind == "norm_weight":
            w = 1.0 + w
        new[k] = torch.from_numpy(w).to(t.dtype) if t.dtype.is_floating_point else t
    model.load_state_dict(new)
    model.train()
    return model


def run_e2e(mods, geo_name, spec, out_name, grads_full=(), merge_type="flat", aspect="square", pinpoints=None,
            tiles=None, image_sizes=None, unfreeze_tower=False, builder=None, slices=False, padding_side="right"):
    geo = GEOMETRIES[geo_name]
    model = (builder or build_reference_model)(mods, geo, merge_type=merge_type, aspect=aspect, pinpoints=pinpoints)
    if unfreeze_tower:  # mm_tunable_parts contains mm_vision_tower (train/train.py:1658-1661)
        model.get_model().get_vision_tower().vision_tower.requires_grad_(True)
    model.config.tokenizer_padding_side = padding_side      # train/train.py copies tokenizer.padding_side here; llava_arch.py:520-524
    ids, labels, mask, images, modalities = make_batch(geo, spec, tiles=tiles, left=padding_side == "left")
    imgs = [torch.from_numpy(x) for x in images]
    sizes = image_sizes or [[geo["vision"]["image"]] * 2 for _ in images]
    with torch.no_grad():
        (_, pos_ids, attn, _, embeds, new_labels) = model.prepare_inputs_labels_for_multimodal(
            torch.from_numpy(ids), None, torch.from_numpy(mask), None, torch.from_numpy(labels), imgs, modalities, sizes)
        cat = torch.cat([x if x.ndim == 4 else x[None] for x in imgs])
        feats = model.encode_images(cat)
        tower_feats = model.get_model().get_vision_tower()(cat)
    out = model(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask), labels=torch.from_numpy(labels),
                images=imgs, image_sizes=sizes, modalities=modalities)
    out.loss.backward()
    res = {
        "input_ids": ids, "labels": labels, "attention_mask": mask,
        "modalities": np.array(modalities), "image_sizes": np.array(sizes),
        "splice_labels": new_labels.numpy(), "splice_attention_mask": attn.numpy(),
        "splice_position_ids_is_none": np.array(pos_ids is None),
        "loss": out.loss.detach().numpy().astype(np.float32),
    }
    for i, im in enumerate(images):
        res[f"image{i}"] = im
    logits = out.logits.detach().numpy().astype(np.float32)
    emb = embeds.numpy()
    if logits.size <= 2_000_000 and not slices:
        res["logits"] = logits
        res["inputs_embeds"] = emb
        res["image_features"] = feats.numpy()
        res["tower_features"] = tower_feats.numpy()
    else:
        res["logits_slice"] = logits[:, ::7, ::997].copy()
        res["logits_absmax"] = np.abs(logits).max(keepdims=True)
        res["logits_rowsum"] = logits.sum(-1)
        res["inputs_embeds_slice"] = emb[:, :, ::37].copy()
        res["image_features_slice"] = feats.numpy()[:, :, ::37].copy()
    gn = {}
    for k, p in model.named_parameters():
        ck = canonical_name(k)
        if p.grad is None:
            gn[ck] = None
        else:
            gn[ck] = float(p.grad.norm())
            if ck in grads_full:
                res["grad::" + ck] = p.grad.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, out_name + ".npz"), **res)
    with open(os.path.join(HERE, out_name + "_gradnorms.json"), "w") as f:
        json.dump({"geometry": geo_name, "spec": spec, "merge_type": merge_type, "aspect": aspect,
                   "pinpoints": pinpoints, "grad_norms": gn, "loss": float(out.loss)}, f, indent=1)
    print(out_name, "loss", float(out.loss), "logits", logits.shape, "embeds", emb.shape)


def per_op(mods):
    """Per-op I/O from the in-tree vendored modeling_llama.py (+ projector builder)."""
    ml = ref_shim.load_vendored_llama()
    res = {}
    d, h, ffn, S, b = 256, 2, 448, 40, 2
    cfg = SimpleNamespace(hidden_size=d, intermediate_size=ffn, num_attention_heads=h, num_key_value_heads=h,
                          max_position_embeddings=4096, rope_theta=10000.0, rope_scaling=None, attention_bias=False,
                          attention_dropout=0.0, pretraining_tp=1, hidden_act="silu", rms_norm_eps=1e-5,
                          _attn_implementation="eager")
    x = torch.from_numpy(prng.normal(7, 1, (b, S, d), 1.0))
    rn = ml.LlamaRMSNorm(d, eps=1e-5)
    rn.weight.data = torch.from_numpy(1.0 + prng.normal(7, 2, (d,), 0.1))
    xr = x.clone().requires_grad_(True)
    y = rn(xr)
    g = torch.from_numpy(prng.normal(7, 3, (b, S, d), 1.0))
    y.backward(g)
    res.update(rms_x=x.numpy(), rms_w=rn.weight.detach().numpy(), rms_y=y.detach().numpy(), rms_gy=g.numpy(),
               rms_gx=xr.grad.numpy(), rms_gw=rn.weight.grad.numpy())
    hd = d // h
    rope = ml.LlamaRotaryEmbedding(hd, max_position_embeddings=4096, base=10000.0)
    q = torch.from_numpy(prng.normal(7, 4, (b, h, S, hd), 1.0))
    k = torch.from_numpy(prng.normal(7, 5, (b, h, S, hd), 1.0))
    pos = torch.arange(S)[None].expand(b, S)
    cos, sin = rope(q, pos)
    qe, ke = ml.apply_rotary_pos_emb(q, k, cos, sin)
    res.update(rope_q=q.numpy(), rope_k=k.numpy(), rope_cos=cos.numpy(), rope_sin=sin.numpy(),
               rope_qe=qe.numpy(), rope_ke=ke.numpy())
    mlp = ml.LlamaMLP(cfg)
    for i, lin in enumerate((mlp.gate_proj, mlp.up_proj, mlp.down_proj)):
        lin.weight.data = torch.from_numpy(prng.normal(7, 10 + i, tuple(lin.weight.shape), 0.05))
    xr = x.clone().requires_grad_(True)
    y = mlp(xr)
    y.backward(g)
    res.update(mlp_wg=mlp.gate_proj.weight.detach().numpy(), mlp_wu=mlp.up_proj.weight.detach().numpy(),
               mlp_wd=mlp.down_proj.weight.detach().numpy(), mlp_y=y.detach().numpy(), mlp_gx=xr.grad.numpy(),
               mlp_gwg=mlp.gate_proj.weight.grad.numpy(), mlp_gwu=mlp.up_proj.weight.grad.numpy(),
               mlp_gwd=mlp.down_proj.weight.grad.numpy())
    layer = ml.LlamaDecoderLayer(cfg, 0)
    cnt = 0
    for n, p in layer.named_parameters():
        w = prng.normal(7, 20 + cnt, tuple(p.shape), 0.05)
        if p.ndim == 1:
            w = 1.0 + w
        p.data = torch.from_numpy(w)
        cnt += 1
        res["layer_w::" + n] = w
    lens = [S, 29]
    minv = torch.finfo(torch.float32).min
    m4 = torch.zeros((b, 1, S, S))
    causal = torch.triu(torch.ones(S, S, dtype=torch.bool), 1)
    for i, L in enumerate(lens):
        mm = causal.clone()
        mm[:, L:] = True
        m4[i, 0][mm] = minv
    xr = x.clone().requires_grad_(True)
    y = layer(xr, attention_mask=m4, position_ids=pos)[0]
    gm = g.clone()
    gm[1, 29:] = 0.0  # padded rows carry no loss
    y.backward(gm)
    res.update(layer_lens=np.array(lens), layer_y=y.detach().numpy(), layer_gy=gm.numpy(), layer_gx=xr.grad.numpy())
    for n, p in layer.named_parameters():
        res["layer_g::" + n] = p.grad.numpy()
    import importlib
    pb = importlib.import_module("llava.model.multimodal_projector.builder")
    pcfg = SimpleNamespace(mm_projector_type="mlp2x_gelu", mm_hidden_size=128, hidden_size=d)
    proj = pb.build_vision_projector(pcfg)
    for i, (n, p) in enumerate(proj.named_parameters()):
        p.data = torch.from_numpy(prng.normal(7, 40 + i, tuple(p.shape), 0.05))
        res["proj_w::" + n] = p.detach().numpy()
    px = torch.from_numpy(prng.normal(7, 50, (3, 16, 128), 1.0)).requires_grad_(True)
    py = proj(px)
    pg = torch.from_numpy(prng.normal(7, 51, tuple(py.shape), 1.0))
    py.backward(pg)
    res.update(proj_x=px.detach().numpy(), proj_y=py.detach().numpy(), proj_gy=pg.numpy(), proj_gx=px.grad.numpy())
    for n, p in proj.named_parameters():
        res["proj_g::" + n] = p.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "per_op.npz"), **res)
    print("per_op done", len(res))


def host_functions(mods):
    """Known answers for the pure host functions (mm_utils.py, unpad_image)."""
    mu = mods["mm_utils"]
    arch = mods["llava_arch"]
    pin = [[336, 672], [672, 336], [672, 672], [1008, 336], [336, 1008]]
    out = {"pinpoints": pin, "select_best_resolution": [], "anyres_grid_shape": [], "unpad_shape": []}
    for size in [(672, 672), (1000, 700), (500, 1200), (336, 336), (640, 480), (100, 900), (1344, 336), (337, 673)]:
        out["select_best_resolution"].append([list(size), list(mu.select_best_resolution(size, pin))])
        out["anyres_grid_shape"].append([list(size), list(mu.get_anyres_image_grid_shape(size, pin, 336))])
    for (c, hh, ww), osz in [((4, 48, 48), (1000, 700)), ((4, 48, 48), (500, 1200)), ((4, 24, 48), (640, 480)),
                             ((4, 48, 24), (300, 900)), ((2, 48, 48), (336, 336)), ((2, 72, 24), (337, 673))]:
        t = torch.arange(c * hh * ww, dtype=torch.float32).view(c, hh, ww)
        u = arch.unpad_image(t, osz)
        out["unpad_shape"].append([[c, hh, ww], list(osz), list(u.shape), float(u.sum())])

    class Ids:
        def __init__(self, ids):
            self.input_ids = ids

    class Tok:
        bos_token_id = 1

        def __call__(self, s):
            return Ids([1] + [ord(ch) for ch in s])

    out["tokenizer_image_token"] = []
    for prompt in ["ab<image>cd<image>e", "<image>\nxyz", "no image here", "a<image>"]:
        out["tokenizer_image_token"].append([prompt, mu.tokenizer_image_token(prompt, Tok(), -200)])
    with open(os.path.join(HERE, "host_functions.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("host functions done")


if __name__ == "__main__":
    mods = ref_shim.load_reference()
    which = sys.argv[1:] or ["per_op", "host", "toy", "cfg1", "anyres", "tower"]
    if "per_op" in which:
        per_op(mods)
    if "host" in which:
        host_functions(mods)
    if "toy" in which:
        run_e2e(mods, "toy", [(20, 5, 8), (12, 3, 4), (9, None, 2)], "toy_e2e",
                grads_full=("model.mm_projector.0.weight", "model.mm_projector.2.bias",
                            "model.layers.0.self_attn.q_proj.weight", "model.layers.1.mlp.down_proj.weight",
                            "model.norm.weight", "lm_head.weight", "model.layers.0.input_layernorm.weight",
                            "model.embed_tokens.weight"))
    if "anyres" in which:
        # toy anyres: 56px tiles, pinpoints in units of 56; one 112x112 image (2x2 grid + base = 5 tiles) and one
        # 150x100 image -> best resolution picked by the reference; spatial_unpad inserts image_newline columns.
        pin = [[56, 112], [112, 56], [112, 112], [168, 56], [56, 168]]
        sizes = [[112, 112], [150, 50]]
        tiles = [1 + int(np.prod(mods["mm_utils"].get_anyres_image_grid_shape(sz, pin, 56))) for sz in sizes]
        run_e2e(mods, "toy", [(14, 4, 6), (10, 2, 3)], "toy_anyres_e2e", merge_type="spatial_unpad", aspect="anyres",
                pinpoints=pin, tiles=tiles, image_sizes=sizes,
                grads_full=("model.image_newline", "model.mm_projector.2.weight"))
    if "tower" in which:
        vp = "model.vision_tower.vision_tower.vision_model."
        run_e2e(mods, "toy", [(20, 5, 8), (12, 3, 4), (9, None, 2)], "toy_tower_e2e", unfreeze_tower=True,
                grads_full=(vp + "embeddings.patch_embedding.weight", vp + "embeddings.position_embedding.weight",
                            vp + "embeddings.class_embedding", vp + "encoder.layers.0.self_attn.q_proj.weight",
                            vp + "encoder.layers.1.mlp.fc1.bias", vp + "encoder.layers.0.layer_norm1.weight",
                            vp + "pre_layrnorm.bias", "model.mm_projector.0.weight"))
    if "qwen" in which:
        vp = "model.vision_tower.vision_tower.vision_model."
        # flat merge, frozen tower: Qwen2 decoder (GQA 4/2, q/k/v bias, theta 1e6) over 729 SigLIP tokens per image
        run_e2e(mods, "toy_qwen", [(20, 5, 8), (12, 3, 4), (9, None, 2)], "toy_qwen_e2e", builder=build_reference_qwen_model,
                slices=True, grads_full=("model.layers.0.self_attn.k_proj.weight", "model.layers.1.self_attn.v_proj.bias",
                                         "model.layers.0.self_attn.q_proj.bias", "model.mm_projector.0.weight", "model.norm.weight"))
        # anyres_max_2 + spatial_unpad with the tower tunable (the RadVLM recipe, finetune_radio_7b.sh: anyres_max_9,
        # mm_tunable_parts incl. mm_vision_tower): 2x2 grid -> 54x54 features -> bilinear to 38x38 (times = 1.41 > 1.1),
        # and a 2x1 grid that stays below the limit (no interpolation)
        pin = [[54, 108], [108, 54], [108, 108]]
        sizes = [[100, 100], [100, 40]]
        tiles = [1 + int(np.prod(mods["mm_utils"].get_anyres_image_grid_shape(sz, pin, 54))) for sz in sizes]
        run_e2e(mods, "toy_qwen", [(14, 4, 6), (10, 2, 3)], "toy_qwen_anyres_max_e2e", merge_type="spatial_unpad",
                aspect="anyres_max_2", pinpoints=pin, tiles=tiles, image_sizes=sizes, unfreeze_tower=True,
                builder=build_reference_qwen_model, slices=True,
                grads_full=("model.image_newline", vp + "embeddings.patch_embedding.bias",
                            vp + "encoder.layers.0.self_attn.q_proj.bias", vp + "encoder.layers.1.mlp.fc1.bias",
                            vp + "embeddings.position_embedding.weight"))
    if "edge" in which:
        # splice edge cases (llava_arch.py:442-531): image placeholder first / last, two images in one sample, a text-only
        # sample, ragged lengths, truncation at tokenizer_model_max_length through an image span -- inputs built exactly as in
        # tests/test_e2e_gpu.py::test_edge_cases_against_oracle (torch generator seed 11)
        geo = GEOMETRIES["toy"]
        model = build_reference_model(mods, geo)
        model.config.tokenizer_model_max_length = 30
        g = torch.Generator().manual_seed(11)
        T = 14
        ids = torch.randint(3, 1000, (4, T), generator=g)
        labels = ids.clone()
        mask = torch.ones(4, T, dtype=torch.bool)
        ids[0, 0] = -200
        ids[1, 9] = -200; mask[1, 10:] = False
        ids[2, 3] = -200; ids[2, 8] = -200
        mask[3, 6:] = False
        labels[ids == -200] = -100
        labels[:, :2] = -100
        ids[~mask] = 0; labels[~mask] = -100
        images = [torch.randn(3, 56, 56, generator=g).to(torch.bfloat16).float() for _ in range(4)] + [torch.zeros(3, 56, 56)]
        modalities = ["image"] * 4 + ["text"]
        sizes = [[56, 56]] * 5
        with torch.no_grad():
            (_, _, attn, _, embeds, new_labels) = model.prepare_inputs_labels_for_multimodal(ids, None, mask, None, labels, images, modalities, sizes)
        out = model(input_ids=ids, attention_mask=mask, labels=labels, images=images, image_sizes=sizes, modalities=modalities)
        out.loss.backward()
        gn = {canonical_name(k): (None if p.grad is None else float(p.grad.norm())) for k, p in model.named_parameters()}
        np.savez_compressed(os.path.join(HERE, "toy_edge_e2e.npz"), input_ids=ids.numpy(), labels=labels.numpy(), attention_mask=mask.numpy(),
                            splice_labels=new_labels.numpy(), splice_attention_mask=attn.numpy(), loss=out.loss.detach().numpy().astype(np.float32),
                            logits=out.logits.detach().numpy().astype(np.float32)[:, :, ::5].copy(), **{f"image{i}": im.numpy() for i, im in enumerate(images)})
        with open(os.path.join(HERE, "toy_edge_e2e_gradnorms.json"), "w") as f:
            json.dump({"geometry": "toy", "max_len": 30, "grad_norms": gn, "loss": float(out.loss)}, f, indent=1)
        print("toy_edge_e2e loss", float(out.loss), "embeds", tuple(embeds.shape))
    if "edge2" in which:
        # (a) 'spatial' merge (no unpad, no image_newline) with anyres tiles; (b) a batch of text-only samples (dummy images only)
        pin = [[56, 112], [112, 56], [112, 112], [168, 56], [56, 168]]
        sizes = [[112, 112], [150, 50]]
        tiles = [1 + int(np.prod(mods["mm_utils"].get_anyres_image_grid_shape(sz, pin, 56))) for sz in sizes]
        run_e2e(mods, "toy", [(14, 4, 6), (10, 2, 3)], "toy_spatial_e2e", merge_type="spatial", aspect="anyres", pinpoints=pin, tiles=tiles,
                image_sizes=sizes, slices=True)
        run_e2e(mods, "toy", [(11, None, 2), (7, None, 1)], "toy_textonly_e2e", slices=True)
    if "pool" in which:
        # 'spatial_maxpool2x2' merge (llava_arch.py:375-379) over anyres tiles: 2x2 grid -> 8x8 tokens pooled to 4x4; 3x1 grid -> 4x12 to 2x6
        pin = [[56, 112], [112, 56], [112, 112], [168, 56], [56, 168]]
        sizes = [[112, 112], [150, 50]]
        tiles = [1 + int(np.prod(mods["mm_utils"].get_anyres_image_grid_shape(sz, pin, 56))) for sz in sizes]
        run_e2e(mods, "toy", [(14, 4, 6), (10, 2, 3)], "toy_maxpool_e2e", merge_type="spatial_maxpool2x2", aspect="anyres", pinpoints=pin,
                tiles=tiles, image_sizes=sizes, slices=True, grads_full=("model.mm_projector.2.weight",))
    if "leftpad" in which:
        # tokenizer_padding_side = "left" (llava_arch.py:520-524): short samples sit at the end of their row; position_ids are
        # discarded in training (:534-545), so every row keeps positions arange(S)
        run_e2e(mods, "toy", [(20, 5, 8), (12, 3, 4), (9, None, 2)], "toy_leftpad_e2e", padding_side="left",
                grads_full=("model.mm_projector.0.weight", "model.layers.0.self_attn.q_proj.weight", "lm_head.weight", "model.embed_tokens.weight"))
    if "cfg1" in which:
        run_e2e(mods, "config1", [(48, 35, 40)], "config1_e2e")
