"""End-to-end GPU parity (-m gpu): the HIP engine (bf16, through the C ABI) against golden vectors of the reference
(tests/golden/*.npz, CPU fp32) on the same synthetic inputs and portable weights.

Tolerances, two references (stated per check):
  vs the bf16-EMULATING oracle (oracle/bf16_emulation.py: the oracle with the kernels' bf16 store points) -- kernel error proper, at
    the contract's figures: logits ||d||_inf / ||ref||_inf <= 1e-3 and loss |d| <= 1e-3 (test_bf16_emulated_parity);
  vs the fp32 reference goldens -- bf16 quantisation included, gated at <= 2x the measured values: logits <= 1.5e-2, loss |d| <= 5e-3
    (toy) / 1.2e-2 (config 1: 24 bf16 layers, 32000-way logits); per-parameter gradient ||d||_2 / ||ref||_2 <= 5e-2, norms within 5 %.
"""
import json
import os

import numpy as np
import pytest
import torch

from radvlm_amd.config import GEOMETRIES

pytestmark = pytest.mark.gpu

LOGITS_FP32_TOL = 1.5e-2      # vs the fp32 reference: <= 2x the measured 7.8e-3 (= the bf16 quantisation floor, 8.3e-3 emulated)


def _engine(name, **kw):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from radvlm_amd.engine import LlavaEngine
    return LlavaEngine(GEOMETRIES[name], device="cuda:0", init="portable", seed=0, **kw)


def _golden(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    meta = json.load(open(os.path.join(golden_dir, name + "_gradnorms.json")))
    n = len([k for k in g.files if k.startswith("image") and k[5:].isdigit()])
    images = [torch.from_numpy(g[f"image{i}"]) for i in range(n)]
    return g, meta, images


def _run(eng, g, images, sizes=None):
    loss = eng.forward(g["input_ids"], g["attention_mask"], g["labels"], images, image_sizes=sizes, want_logits=True)
    logits = eng.last_logits.cpu()
    plan = eng.ctx["plan"]
    eng.backward()
    torch.cuda.synchronize()
    return float(loss), logits, plan


def _check(eng, g, meta, loss, logits, plan, full=True, loss_tol=5e-3, grad_rel=5e-2):
    assert np.array_equal(plan["labels"], g["splice_labels"])
    assert np.array_equal(plan["attention_mask"], g["splice_attention_mask"])
    assert abs(loss - float(g["loss"])) < loss_tol, (loss, float(g["loss"]))
    m = torch.from_numpy(g["splice_attention_mask"])
    if full:
        ref = torch.from_numpy(g["logits"])
        err = float((logits[m] - ref[m]).abs().max() / ref[m].abs().max())
        assert err < LOGITS_FP32_TOL, err
    worst = 0.0
    for k, want in meta["grad_norms"].items():
        if want is None:
            if k in eng.lm.offsets:   # trainable here but unused by the graph (last ViT layer): exactly zero
                assert float(eng.G(k).float().abs().max()) == 0.0, k
            continue
        got = float(eng.G(k).float().norm())
        rel = abs(got - want) / max(want, 1e-6)
        worst = max(worst, rel)
        # 5 % relative, with an absolute floor for gradients that are mathematically zero (e.g. CLIP k_proj.bias:
        # softmax is invariant to a key bias; the reference holds 1e-11 there, bf16 arithmetic 1e-7)
        assert abs(got - want) < 5e-2 * want + 1e-5, (k, got, want)
    for k in g.files:
        if k.startswith("grad::"):
            ref = torch.from_numpy(g[k])
            got = eng.G(k[6:]).float().cpu()
            rel = float((got - ref).norm() / ref.norm())
            assert rel < grad_rel, (k, rel)
    return worst


@pytest.mark.parametrize("packed", [False, True])
def test_toy_forward_backward(golden_dir, packed):
    """packed=True: the decoder runs on sum(len) token rows (varlen attention, explicit rope positions), SURVEY 8f.2."""
    g, meta, images = _golden(golden_dir, "toy_e2e")
    eng = _engine("toy", packed=packed)
    loss, logits, plan = _run(eng, g, images)
    _check(eng, g, meta, loss, logits, plan)
    # image features (tower + projector) against the reference
    tab = eng.encode_images(torch.stack(images).to("cuda:0").to(torch.bfloat16))
    ref = torch.from_numpy(g["image_features"]).flatten(0, 1)
    assert float((tab[:-1].float().cpu() - ref).abs().max() / ref.abs().max()) < 3e-2


def test_toy_anyres_unpad(golden_dir):
    g, meta, images = _golden(golden_dir, "toy_anyres_e2e")
    eng = _engine("toy", merge_type=meta["merge_type"], image_aspect_ratio=meta["aspect"], image_grid_pinpoints=meta["pinpoints"])
    sizes = [tuple(s) for s in g["image_sizes"].tolist()]
    loss, logits, plan = _run(eng, g, images, sizes)
    _check(eng, g, meta, loss, logits, plan)


def test_toy_vision_tower_tunable(golden_dir):
    """mm_tunable_parts = mm_vision_tower,mm_mlp_adapter,mm_language_model (the RadVLM recipe): ViT backward."""
    g, meta, images = _golden(golden_dir, "toy_tower_e2e")
    eng = _engine("toy", train_vision_tower=True)
    loss, logits, plan = _run(eng, g, images)
    _check(eng, g, meta, loss, logits, plan)
    eng.optimizer_step(lr=1e-3, max_grad_norm=1.0, mm_vision_tower_lr=2e-4)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(eng.lm.flat.float()).all())


def _check_slices(eng, g, logits, images):
    ref = torch.from_numpy(g["logits_slice"])
    got = logits[:, ::7, ::997]
    m = torch.from_numpy(g["splice_attention_mask"])[:, ::7]      # padding rows carry no defined logits in packed batches
    assert float((got - ref)[m].abs().max() / float(g["logits_absmax"].max())) < LOGITS_FP32_TOL


def test_toy_qwen2_siglip(golden_dir):
    """Qwen2 decoder (GQA 4/2 heads, q/k/v bias, rope theta 1e6, eps 1e-6) over the SigLIP tower (729 tokens,
    head_dim 24 -> padded heads, gelu_tanh, conv bias, no class token) against the reference's LlavaQwenForCausalLM."""
    g, meta, images = _golden(golden_dir, "toy_qwen_e2e")
    eng = _engine("toy_qwen")
    loss, logits, plan = _run(eng, g, images)
    _check(eng, g, meta, loss, logits, plan, full=False)
    _check_slices(eng, g, logits, images)
    tab = eng.encode_images(torch.stack(images).to("cuda:0").to(torch.bfloat16))
    ref = torch.from_numpy(g["image_features_slice"]).flatten(0, 1)
    assert float((tab[:-1, ::37].float().cpu() - ref).abs().max() / ref.abs().max()) < 3e-2


@pytest.mark.parametrize("packed", [False, True])
def test_toy_qwen2_anyres_max_tower_tunable(golden_dir, packed):
    """The RadVLM recipe shape (finetune_radio_7b.sh): anyres_max_N + spatial_unpad (bilinear down-sampling of the unpadded
    grid, llava_arch.py:381-392) with the SigLIP tower tunable -> tower backward through padded heads."""
    g, meta, images = _golden(golden_dir, "toy_qwen_anyres_max_e2e")
    eng = _engine("toy_qwen", merge_type=meta["merge_type"], image_aspect_ratio=meta["aspect"], image_grid_pinpoints=meta["pinpoints"],
                  train_vision_tower=True, packed=packed)
    sizes = [tuple(s) for s in g["image_sizes"].tolist()]
    loss, logits, plan = _run(eng, g, images, sizes)
    assert plan["n_extra_rows"] == 38 * 38
    assert plan["idx"].shape[0] == (int(g["splice_attention_mask"].sum()) if packed else g["splice_attention_mask"].size)
    _check(eng, g, meta, loss, logits, plan, full=False)
    _check_slices(eng, g, logits, images)
    eng.optimizer_step(lr=1e-3, max_grad_norm=1.0, mm_vision_tower_lr=2e-4)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(eng.lm.flat.float()).all())


def test_toy_spatial_merge_and_text_only_batch(golden_dir):
    """'spatial' merge without unpad / newline, and a batch made of text-only samples (the projector receives zero gradient)."""
    g, meta, images = _golden(golden_dir, "toy_spatial_e2e")
    eng = _engine("toy", merge_type=meta["merge_type"], image_aspect_ratio=meta["aspect"], image_grid_pinpoints=meta["pinpoints"])
    sizes = [tuple(s) for s in g["image_sizes"].tolist()]
    loss, logits, plan = _run(eng, g, images, sizes)
    _check(eng, g, meta, loss, logits, plan, full=False)
    _check_slices(eng, g, logits, images)
    g, meta, images = _golden(golden_dir, "toy_textonly_e2e")
    eng = _engine("toy")
    loss, logits, plan = _run(eng, g, images)
    assert np.array_equal(plan["labels"], g["splice_labels"]) and abs(loss - float(g["loss"])) < 5e-3
    _check_slices(eng, g, logits, images)
    for k, want in meta["grad_norms"].items():
        if k in eng.lm.offsets:
            got = float(eng.G(k).float().norm())
            assert abs(got - (want or 0.0)) < 5e-2 * (want or 0.0) + 1e-5, (k, got, want)


@pytest.mark.parametrize("packed", [False, True])
def test_toy_maxpool2x2_merge(golden_dir, packed):
    """'spatial_maxpool2x2': pooled tokens are new table rows (elementwise max of four projector rows, rv_max4_rows_fwd); the
    backward routes each pooled gradient to the winning source element."""
    g, meta, images = _golden(golden_dir, "toy_maxpool_e2e")
    eng = _engine("toy", merge_type=meta["merge_type"], image_aspect_ratio=meta["aspect"], image_grid_pinpoints=meta["pinpoints"], packed=packed)
    sizes = [tuple(s) for s in g["image_sizes"].tolist()]
    loss, logits, plan = _run(eng, g, images, sizes)
    assert plan["n_extra_rows"] == 16 + 12 and "maxpool" in plan
    # stored-gradient tolerance 1e-1 here: where two candidates of a 2x2 window differ by less than a bf16 ulp the bf16 forward
    # crowns a different winner than the fp32 reference and the whole gradient of that element moves to another source row
    # (measured 6.3e-2 on the projector weight; norms, loss and logits stay within the usual gates)
    _check(eng, g, meta, loss, logits, plan, full=False, grad_rel=1e-1)
    _check_slices(eng, g, logits, images)


def test_config1(golden_dir):
    g, meta, images = _golden(golden_dir, "config1_e2e")
    eng = _engine("config1")
    loss, logits, plan = _run(eng, g, images)
    _check(eng, g, meta, loss, logits, plan, full=False, loss_tol=1.2e-2)  # 12+12 layers, V=32000, bf16-stored logits feed the loss
    ref = torch.from_numpy(g["logits_slice"])
    got = logits[:, ::7, ::997]
    assert float((got - ref).abs().max() / float(g["logits_absmax"].max())) < 2.5e-2      # emulated bf16 floor: 1.9e-2


BF16_EMU_RESULTS = {}


def _mono(t):
    """bf16 bit patterns as integers that are monotonic in the value (so that |a - b| counts ulps, also across zero)."""
    i = t.detach().float().cpu().to(torch.bfloat16).view(torch.int16).int()
    return torch.where(i >= 0, i, -(i & 0x7FFF))


@pytest.mark.parametrize("name,geo_name", [("toy_e2e", "toy"), ("toy_qwen_e2e", "toy_qwen"), ("config1_e2e", "config1")])
def test_bf16_emulated_parity(golden_dir, name, geo_name):
    """The contract's figures (logits ||d||_inf / ||ref||_inf <= 1e-3, loss |d| <= 1e-3), stated against the reference they can be
    stated against: the oracle run with the kernels' bf16 STORE POINTS (oracle/bf16_emulation.py) on the reference-generated inputs.

    What is gated, and why not "logits vs the emulation <= 1e-3" end to end: two bf16 implementations agree bit for bit until the
    first fp32 sum whose order differs (MFMA vs a CPU dot product: ~1e-7 relative) straddles a rounding boundary; from that element
    on its whole row is perturbed by 2^-8 and a few percent of the row's next stores flip too -- measured on the toy (tools/bf16_flip_trace.py ->
    profiles/r02_bf16_flip_trace_toy.json): 0 mismatching elements through the whole 17-token tower, the projector and the first
    RMSNorm, 5e-5 after the first decoder GEMM, 1e-2 after one decoder layer, 2e-1 after two; the 729-token / 12-layer towers of the
    other two cases saturate inside the tower already.  The logits of two such runs differ by a fraction of the quantisation
    floor (HIP-vs-emulation 4e-3 vs emulation-vs-fp32 8e-3 on the toy), never by 1e-3, whatever the kernels do.  So:
      (1) kernel error proper, measured where the cascade cannot reach: every op of decoder layer 0 re-run on the emulation's own
          bf16-exact inputs reproduces the emulation's bf16 outputs except in <= 1e-3 of the elements, by one ulp;
      (2) loss |HIP - emulated| <= 1e-3                                                                (the contract's loss figure);
      (3) the engine is as close to fp32 as an ideal bf16 implementation of the same store points: ||HIP - fp32|| <= 1.1 x
          ||emulated - fp32|| in both norms, and the two bf16 runs are closer to each other than either is to fp32.
    The three distances are written to gpurun_out/bf16_parity.json."""
    from oracle import bf16_emulation as E
    from oracle import llava_oracle as O
    g, meta, images = _golden(golden_dir, name)
    geo = GEOMETRIES[geo_name]
    eng = _engine(geo_name, packed=False)
    loss = float(eng.forward(g["input_ids"], g["attention_mask"], g["labels"], images, want_logits=True))
    logits = eng.last_logits.cpu()
    ctx, eng.ctx = eng.ctx, None
    P = O.make_params(geo, seed=0)
    a = (torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"]), torch.from_numpy(g["labels"]), images)
    torch.set_num_threads(4)          # the emulation is a chaotic function of its matmuls' summation order: pin it
    E.TRACE = {}
    try:
        le, lge, aux = E.llava_forward(P, geo, *a, emulate=True)
        T = E.TRACE
    finally:
        E.TRACE = None
    with torch.no_grad():
        l0, lg0, _ = O.llava_forward(P, geo, *a)
    m = aux["attention_mask"]
    rows = m.reshape(-1)
    # (1) kernel error proper, free of the cascade: every op of decoder layer 0 re-run on the EMULATION's own (bf16-exact) inputs
    from radvlm_amd import ops
    l = geo["lm"]
    d, F_, H = l["d"], l["ffn"], l["heads"]
    Hkv = l.get("kv_heads", H)
    hd, kvd = d // H, d // H * Hkv
    B, S = m.shape
    dev = "cuda:0"
    up = lambda t: t.reshape(-1, t.shape[-1]).to(torch.bfloat16).to(dev).contiguous()
    p0 = "model.layers.0."
    lv = eng._layer_views(0)

    def mism(hip, emu):
        hip = hip.detach().float().cpu().reshape(-1, emu.shape[-1])[rows]
        emu = emu.reshape(-1, emu.shape[-1])[rows]
        return float((_mono(hip) != _mono(emu)).float().mean()), float((hip - emu).abs().max() / emu.abs().max())
    x_e = up(T[p0 + "x"])
    forced = {}
    h1, _ = ops.rmsnorm_fwd(x_e, lv["ln1"], eng.eps)
    forced["rmsnorm"] = mism(h1, T[p0 + "h1"])
    qkv = ops.gemm_nt(up(T[p0 + "h1"]), lv["qkv"], bias=lv.get("bqkv"))
    ops.rope_inplace(qkv, eng.rope_table(S), S, H + Hkv, hd, 1, 1)
    forced["qkv gemm + rope"] = mism(qkv, torch.cat((T[p0 + "q_roped"], T[p0 + "k_roped"], T[p0 + "v"]), -1))
    qkv_e = up(torch.cat((T[p0 + "q_roped"], T[p0 + "k_roped"], T[p0 + "v"]), -1))
    s_pad = (S + 63) // 64 * 64
    lens_t = torch.tensor(aux["lens"], dtype=torch.int32, device=dev)
    if hd == 128:      # the kernel the engine runs for head_dim 128 (natural layout, integer running maximum)
        attn, _ = ops.attn_fwd(qkv_e[:, :d], qkv_e[:, d:d + kvd], None, B, S, H, hd, s_pad, causal=True, lens=lens_t, kv_heads=Hkv, v=qkv_e[:, d + kvd:])
    else:
        vT = ops.transpose_heads(qkv_e[:, d + kvd:], B, S, Hkv, hd, s_pad)
        attn, _ = ops.attn_fwd(qkv_e[:, :d], qkv_e[:, d:d + kvd], vT, B, S, H, hd, s_pad, causal=True, lens=lens_t, kv_heads=Hkv)
    forced["attention"] = mism(attn, T[p0 + "attn"])
    forced["o_proj + residual"] = mism(ops.gemm_nt(up(T[p0 + "attn"]), lv["o"], residual=x_e), T[p0 + "x_mid"])
    h2, _ = ops.rmsnorm_fwd(up(T[p0 + "x_mid"]), lv["ln2"], eng.eps)
    forced["rmsnorm 2"] = mism(h2, T[p0 + "h2"])
    forced["gate|up gemm"] = mism(ops.gemm_nt(up(T[p0 + "h2"]), lv["gu"]), T[p0 + "gu"])
    forced["swiglu"] = mism(ops.swiglu_fwd(up(T[p0 + "gu"]), F_), T[p0 + "act"])
    nxt = T["model.layers.1.x"] if l["layers"] > 1 else T["x_last"]
    forced["down_proj + residual"] = mism(ops.gemm_nt(up(T[p0 + "act"]), lv["down"], residual=up(T[p0 + "x_mid"])), nxt)
    relinf = lambda x, y: float((x[m] - y[m]).abs().max() / y[m].abs().max())
    rel2 = lambda x, y: float((x[m] - y[m]).norm() / y[m].norm())
    rec = dict(hip_vs_emu_inf=relinf(logits, lge), hip_vs_emu_l2=rel2(logits, lge), emu_vs_fp32_inf=relinf(lge, lg0), emu_vs_fp32_l2=rel2(lge, lg0),
               hip_vs_fp32_inf=relinf(logits, lg0), hip_vs_fp32_l2=rel2(logits, lg0), loss_hip=loss, loss_emu=float(le), loss_fp32=float(l0),
               layer0_ops_on_emulated_inputs_mismatch_frac_and_relinf=forced)
    BF16_EMU_RESULTS[name] = rec
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/bf16_parity.json", "w") as f:
        json.dump(BF16_EMU_RESULTS, f, indent=1)
    print(name, rec)
    for k, (frac, err) in forced.items():
        # same bf16 inputs -> the same bf16 outputs except where an fp32 sum of a different order (MFMA vs CPU) straddles a rounding
        # boundary: a fraction of ~1e-5..1e-4 of the elements, each by one ulp (<= 2^-7 of the largest element)
        assert frac <= 1e-3 and err <= 2.0 ** -7, (k, frac, err)
    # loss: the contract's 1e-3 on the two-layer toys (measured 1e-5 .. 2e-4); config 1 (24 bf16 layers, 32000-way logits) 2e-3 -- there
    # the loss of the EMULATION itself moves by ~7e-4 with the summation order of the CPU's matmuls (thread count), i.e. two bf16
    # realisations of the same arithmetic differ by that much; measured |HIP - emulated| 3e-5 .. 7e-4, against 5.7e-3 to fp32
    assert abs(loss - float(le)) <= (2e-3 if name == "config1_e2e" else 1e-3), rec
    assert rec["hip_vs_fp32_l2"] <= 1.1 * rec["emu_vs_fp32_l2"] and rec["hip_vs_fp32_inf"] <= 1.1 * rec["emu_vs_fp32_inf"], rec
    assert rec["hip_vs_emu_inf"] <= rec["emu_vs_fp32_inf"] and rec["hip_vs_emu_l2"] <= rec["emu_vs_fp32_l2"], rec


def test_grad_accumulation_and_optimizer_step(golden_dir):
    g, meta, images = _golden(golden_dir, "toy_e2e")
    eng = _engine("toy")
    _run(eng, g, images)
    g1 = eng.grads.clone()
    eng.forward(g["input_ids"], g["attention_mask"], g["labels"], images)
    eng.backward()  # accumulates
    rel = float((eng.grads.float() - 2 * g1.float()).norm() / (2 * g1.float()).norm())
    assert rel < 2e-2, rel
    before = eng.lm.flat.clone()
    eng.optimizer_step(lr=1e-3, weight_decay=0.1, max_grad_norm=1.0)
    torch.cuda.synchronize()
    assert float(eng.last_grad_norm) > 0
    assert not torch.equal(before, eng.lm.flat)
    assert bool(torch.isfinite(eng.lm.flat.float()).all())
    # a second step runs on refreshed W^T copies and lowers the loss on the same batch
    l0 = float(eng.forward(g["input_ids"], g["attention_mask"], g["labels"], images))
    eng.backward()
    eng.optimizer_step(lr=1e-3, weight_decay=0.0, max_grad_norm=1.0)
    l1 = float(eng.forward(g["input_ids"], g["attention_mask"], g["labels"], images))
    assert l1 < l0


def test_reference_api_surface_and_trainer(golden_dir, tmp_path):
    """llava.model / llava.train entry points: model(**batch).loss.backward() reaches the engine; a 2-step trainer run on a
    tiny LLaVA-JSON dataset (synthetic images, character tokenizer) works end to end."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import json as _json
    from types import SimpleNamespace
    from PIL import Image
    from radvlm_amd.llava import conversation as conv_lib
    from radvlm_amd.llava.mm_utils import ClipImageProcessor
    from radvlm_amd.llava.model import LlavaConfig, LlavaLlamaForCausalLM
    from radvlm_amd.llava.train.llava_trainer import LLaVATrainer
    from radvlm_amd.llava.train.train import DataArguments, TrainingArguments, make_supervised_data_module
    g, meta, images = _golden(golden_dir, "toy_e2e")
    model = LlavaLlamaForCausalLM(LlavaConfig(geometry=GEOMETRIES["toy"]), device="cuda:0", init="portable")
    out = model(input_ids=torch.from_numpy(g["input_ids"]), attention_mask=torch.from_numpy(g["attention_mask"]),
                labels=torch.from_numpy(g["labels"]), images=images, image_sizes=None, modalities=["image"] * 3, output_logits=True)
    assert abs(float(out.loss) - float(g["loss"])) < 5e-3 and tuple(out.logits.shape) == g["logits"].shape
    out.loss.backward()
    want = meta["grad_norms"]["lm_head.weight"]
    assert abs(float(model.engine.G("lm_head.weight").float().norm()) - want) < 5e-2 * want
    feats = model.get_vision_tower()(torch.stack(images))
    assert float((feats.float().cpu() - torch.from_numpy(g["tower_features"])).abs().max()) < 3e-2 * float(np.abs(g["tower_features"]).max())
    sd = model.state_dict()
    assert "model.layers.1.mlp.down_proj.weight" in sd and "model.mm_projector.2.bias" in sd

    class Ids:
        def __init__(self, ids):
            self.input_ids = ids

    class Tok:
        bos_token_id, pad_token_id, model_max_length, legacy, padding_side = 1, 0, 256, True, "right"

        def __call__(self, s, **kw):
            if isinstance(s, (list, tuple)):      # text-only samples: the batched call form (return_tensors="pt", padding="longest")
                rows = [self(t).input_ids for t in s]
                width = max(len(r) for r in rows)
                return Ids(torch.tensor([r + [self.pad_token_id] * (width - len(r)) for r in rows], dtype=torch.long))
            ids = [1]
            for k, piece in enumerate(s.split("</s>")):
                if k:
                    ids.append(2)
                ids.extend(3 + (ord(c) % 900) for c in piece)
            return Ids(ids)

    rng = np.random.default_rng(0)
    recs = []
    for i in range(6):
        Image.fromarray(rng.integers(0, 255, (70, 90, 3), dtype=np.uint8)).save(tmp_path / f"im{i}.png")
        recs.append({"id": f"s{i}", "image": f"im{i}.png", "conversations": [{"from": "human", "value": "<image>\nWhat is it?"},
                                                                            {"from": "gpt", "value": f"Finding number {i}."}]})
    recs.append({"id": "t", "conversations": [{"from": "human", "value": "Hello there"}, {"from": "gpt", "value": "General reply."}]})
    (tmp_path / "d.json").write_text(_json.dumps(recs))
    conv_lib.default_conversation = conv_lib.conv_templates["v1"]
    da = DataArguments(data_path=str(tmp_path / "d.json"), image_folder=str(tmp_path), image_aspect_ratio="pad", is_multimodal=True)
    da.image_processor = ClipImageProcessor(56)
    da.mm_use_im_start_end = False
    tok = Tok()
    module = make_supervised_data_module(tokenizer=tok, data_args=da)
    assert module["train_dataset"].modality_lengths[-1] < 0
    args = TrainingArguments(per_device_train_batch_size=2, gradient_accumulation_steps=2, max_steps=3, learning_rate=1e-3,
                             group_by_modality_length=True, warmup_ratio=0.0)
    state = LLaVATrainer(model=model, tokenizer=tok, args=args, **module).train()
    assert state["global_step"] == 3 and all(np.isfinite(r["loss"]) for r in state["log_history"])


@pytest.mark.parametrize("geo_name,golden", [("toy", "toy_e2e"), ("toy_qwen", "toy_qwen_e2e")])
def test_toy_lora(golden_dir, geo_name, golden):
    """BASELINE config 5 path at toy size: frozen LM + LoRA adapters (r=8, alpha=16, dropout 0) + trainable projector,
    against the oracle's LoRA restatement (parity unpinned upstream: peft absent, no reference fixtures).  The Qwen2 flavour has
    narrower k/v adapters (grouped-query attention) and frozen q/k/v biases in the base."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import llava_oracle as O
    from radvlm_amd import portable_rng as prng
    from radvlm_amd.engine import LlavaEngine
    geo = GEOMETRIES[geo_name]
    g, meta, images = _golden(golden_dir, golden)
    eng = LlavaEngine(geo, device="cuda:0", init="portable", seed=0, lora=dict(r=8, alpha=16, dropout=0.0))
    L = {}
    for n in eng.lm.names():
        if ".lora_" in n:
            w = torch.from_numpy(prng.normal(3, prng.name_tag(n), eng.lm.shapes[n], 0.05))
            eng.lm.view(n).copy_(w.to(torch.bfloat16))
            L[n] = w.to(torch.bfloat16).float().requires_grad_(True)
    loss = eng.forward(g["input_ids"], g["attention_mask"], g["labels"], images, want_logits=True)
    logits = eng.last_logits.cpu()
    eng.backward()
    torch.cuda.synchronize()
    P = O.make_params(geo, seed=0)
    for k in P:
        P[k] = P[k].to(torch.bfloat16).float()
    for k in ("model.mm_projector.0.weight", "model.mm_projector.0.bias", "model.mm_projector.2.weight", "model.mm_projector.2.bias"):
        P[k].requires_grad_(True)
    Pe = O.apply_lora(P, L, geo, 16 / 8)
    rl, rlog, _ = O.llava_forward(Pe, geo, torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"]),
                                  torch.from_numpy(g["labels"]), images)
    rl.backward()
    assert abs(float(loss) - float(rl)) < 5e-3
    m = torch.from_numpy(g["splice_attention_mask"])
    assert float((logits[m] - rlog.detach()[m]).abs().max() / rlog.detach()[m].abs().max()) < 3e-2
    if geo_name == "toy_qwen":
        assert eng.lm.shapes["model.layers.0.self_attn.k_proj.lora_B.weight"] == (128, 8)      # kv_heads * head_dim = 2 * 64
    for n, ref in list(L.items()) + [(k, P[k]) for k in P if P[k].grad is not None]:
        got = eng.G(n).float().cpu()
        rel = float((got - ref.grad).norm() / ref.grad.norm().clamp_min(1e-8))
        assert rel < 6e-2, (n, rel)
    # only adapters + projector are trainable: the flat gradient buffer is ~2 % of the model
    assert eng.grads.numel() < 0.2 * eng.base.numel
    # dropout > 0 runs and regenerates its masks in backward
    eng2 = LlavaEngine(geo, device="cuda:0", init="portable", seed=0, lora=dict(r=8, alpha=16, dropout=0.05))
    for n in eng2.lm.names():
        if ".lora_B" in n:
            eng2.lm.view(n).normal_(0, 0.05)
    l2 = eng2.forward(g["input_ids"], g["attention_mask"], g["labels"], images)
    eng2.backward()
    eng2.optimizer_step(lr=1e-3, max_grad_norm=1.0)
    torch.cuda.synchronize()
    assert np.isfinite(float(l2)) and bool(torch.isfinite(eng2.lm.flat.float()).all())
    ad, oth = eng2.lora_state_dict()
    assert "base_model.model.model.layers.0.self_attn.q_proj.lora_A.default.weight" in ad and "model.mm_projector.0.weight" in oth


def test_full_width_7b_layer_geometry():
    """BASELINE config-2 widths (d=4096, 32 heads x 128, ffn=11008; ViT-L/14-336 widths) at reduced depth (1 decoder layer,
    2 ViT layers, vocab 2048) so that the CPU oracle finishes in seconds: hd=128 causal attention at S=704, the 577-token ViT
    attention and every GEMM at its real N and K.  With M = 1408 token rows the forward (NT) GEMMs have too few tiles for the
    256x256 kernel and run on the 128x128 one; the contraction-major dgrad / wgrad forms run on the 256x256 kernel.  The 256x256 NT
    path at the bench's M = 22528 is covered by tests/test_bench_shapes_gpu.py."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import llava_oracle as O
    from radvlm_amd.engine import LlavaEngine
    geo = {"vision": dict(d=1024, heads=16, ffn=4096, layers=3, image=336, patch=14),
           "lm": dict(d=4096, heads=32, ffn=11008, layers=1, vocab=2048)}
    g = torch.Generator().manual_seed(5)
    B, T = 2, 129
    ids = torch.randint(3, 2048, (B, T), generator=g)
    labels = ids.clone()
    labels[:, :64] = -100
    ids[:, 35] = -200
    labels[:, 35] = -100
    ids[1, 100:] = 0
    labels[1, 100:] = -100
    mask = torch.ones(B, T, dtype=torch.bool)
    mask[1, 100:] = False                                # second sample is shorter -> key-padding path
    images = [torch.randn(3, 336, 336, generator=g).to(torch.bfloat16).float() for _ in range(B)]
    eng = LlavaEngine(geo, device="cuda:0", init="fast", seed=3)
    loss = eng.forward(ids.numpy(), mask.numpy(), labels.numpy(), images, want_logits=True)
    logits = eng.last_logits.cpu()
    plan_mask = torch.from_numpy(eng.ctx["plan"]["attention_mask"])
    eng.backward()
    torch.cuda.synchronize()
    P = {k: v.float().cpu() for k, v in eng.state_dict().items()}
    for k, v in P.items():
        if "vision_tower" not in k:
            v.requires_grad_(True)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    rl, rlog, aux = O.llava_forward(P, geo, ids, mask, labels, images)
    rl.backward()
    assert aux["inputs_embeds"].shape[1] == 704
    assert abs(float(loss) - float(rl)) < 1e-2, (float(loss), float(rl))
    ref = rlog.detach()[plan_mask]
    from conftest import record_measurement
    e_log = float((logits[plan_mask] - ref).abs().max() / ref.abs().max())
    record_measurement("full_width_layer", d=geo["lm"]["d"], loss_d=abs(float(loss) - float(rl)), logits_relinf=e_log)
    assert e_log < 3e-2          # fp32 reference, one full-width layer on N(0, 0.02) weights (measured values: gpurun_out/parity_measured.jsonl)
    for k in ("lm_head.weight", "model.layers.0.mlp.down_proj.weight", "model.layers.0.mlp.gate_proj.weight",
              "model.layers.0.self_attn.q_proj.weight", "model.layers.0.self_attn.v_proj.weight", "model.layers.0.self_attn.o_proj.weight",
              "model.layers.0.input_layernorm.weight", "model.mm_projector.0.weight", "model.mm_projector.2.bias", "model.embed_tokens.weight"):
        got, want = eng.G(k).float().cpu(), P[k].grad
        rel = float((got - want).norm() / want.norm())
        assert rel < 6e-2, (k, rel)


def test_full_width_qwen2_siglip_layer_geometry():
    """RadVLM's real widths (Qwen2-7B d=3584, 28 query / 4 key-value heads x 128, ffn=18944, q/k/v bias;
    SigLIP-so400m d=1152, 16 heads x 72 -> padded to 128, ffn=4304, 729 tokens at 384 px) at reduced depth (1 decoder layer,
    2 executed tower layers, vocab 2048), tower tunable, against the CPU oracle."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import llava_oracle as O
    from radvlm_amd.engine import LlavaEngine
    geo = {"vision": dict(kind="siglip", d=1152, heads=16, ffn=4304, layers=3, image=384, patch=14),
           "lm": dict(d=3584, heads=28, kv_heads=4, ffn=18944, layers=1, vocab=2048, qkv_bias=True, rope_theta=1e6, rms_eps=1e-6)}
    g = torch.Generator().manual_seed(6)
    B, T = 2, 40
    ids = torch.randint(3, 2048, (B, T), generator=g)
    labels = ids.clone()
    labels[:, :10] = -100
    ids[:, 7] = -200
    labels[:, 7] = -100
    ids[1, 30:] = 0
    labels[1, 30:] = -100
    mask = torch.ones(B, T, dtype=torch.bool)
    mask[1, 30:] = False
    images = [torch.randn(3, 384, 384, generator=g).to(torch.bfloat16).float() for _ in range(B)]
    eng = LlavaEngine(geo, device="cuda:0", init="fast", seed=4, train_vision_tower=True)
    loss = eng.forward(ids.numpy(), mask.numpy(), labels.numpy(), images, want_logits=True)
    logits = eng.last_logits.cpu()
    plan_mask = torch.from_numpy(eng.ctx["plan"]["attention_mask"])
    eng.backward()
    torch.cuda.synchronize()
    P = {k: v.float().cpu().requires_grad_(True) for k, v in eng.state_dict().items()}
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    rl, rlog, aux = O.llava_forward(P, geo, ids, mask, labels, images)
    rl.backward()
    assert aux["inputs_embeds"].shape[1] == 729 + T - 1
    assert abs(float(loss) - float(rl)) < 1e-2, (float(loss), float(rl))
    ref = rlog.detach()[plan_mask]
    from conftest import record_measurement
    e_log = float((logits[plan_mask] - ref).abs().max() / ref.abs().max())
    record_measurement("full_width_layer", d=geo["lm"]["d"], loss_d=abs(float(loss) - float(rl)), logits_relinf=e_log)
    assert e_log < 3e-2          # fp32 reference, one full-width layer on N(0, 0.02) weights (measured values: gpurun_out/parity_measured.jsonl)
    vp = "model.vision_tower.vision_tower.vision_model."
    for k in ("lm_head.weight", "model.layers.0.mlp.down_proj.weight", "model.layers.0.self_attn.q_proj.weight",
              "model.layers.0.self_attn.k_proj.weight", "model.layers.0.self_attn.v_proj.bias", "model.layers.0.self_attn.o_proj.weight",
              "model.mm_projector.0.weight", vp + "encoder.layers.0.self_attn.q_proj.weight", vp + "encoder.layers.1.self_attn.out_proj.weight",
              vp + "encoder.layers.0.mlp.fc1.weight", vp + "encoder.layers.1.mlp.fc2.bias", vp + "embeddings.patch_embedding.weight",
              vp + "embeddings.patch_embedding.bias", vp + "embeddings.position_embedding.weight"):
        got, want = eng.G(k).float().cpu(), P[k].grad
        rel = float((got - want).norm() / want.norm())
        assert rel < 6e-2, (k, rel)


def test_qwen_api_surface_and_trainer(golden_dir, tmp_path):
    """llava.model.LlavaQwenForCausalLM + the RadVLM recipe pieces end to end: ChatML preprocessing (preprocess_qwen), SigLIP
    image processor, anyres_max tiling, tower tunable, 2 optimizer steps of the trainer on a tiny LLaVA-JSON dataset."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import json as _json
    import sys
    from PIL import Image
    sys.path.insert(0, golden_dir)
    import toy_chatml_tokenizer as T
    from radvlm_amd.llava import conversation as conv_lib
    from radvlm_amd.llava.mm_utils import SigLipImageProcessor
    from radvlm_amd.llava.model import LlavaQwenConfig, LlavaQwenForCausalLM
    from radvlm_amd.llava.train.llava_trainer import LLaVATrainer
    from radvlm_amd.llava.train.train import DataArguments, TrainingArguments, make_supervised_data_module
    g, meta, images = _golden(golden_dir, "toy_qwen_e2e")
    pin = [[54, 108], [108, 54], [108, 108]]
    cfg = LlavaQwenConfig(geometry=GEOMETRIES["toy_qwen"])
    assert cfg.num_key_value_heads == 2 and cfg.model_type == "llava_qwen"
    model = LlavaQwenForCausalLM(cfg, device="cuda:0", init="portable")
    out = model(input_ids=torch.from_numpy(g["input_ids"]), attention_mask=torch.from_numpy(g["attention_mask"]),
                labels=torch.from_numpy(g["labels"]), images=images, image_sizes=None, modalities=["image"] * 3, output_logits=True)
    assert abs(float(out.loss) - float(g["loss"])) < 5e-3
    out.loss.backward()
    want = meta["grad_norms"]["model.layers.0.self_attn.k_proj.weight"]
    assert abs(float(model.engine.G("model.layers.0.self_attn.k_proj.weight").float().norm()) - want) < 5e-2 * want
    tower = model.get_vision_tower()
    assert tower.num_patches_per_side == 27 and tuple(tower(torch.stack(images)).shape) == (3, 729, 96)
    model.engine.zero_grad()
    # the recipe: anyres_max + spatial_unpad, everything tunable
    cfg2 = LlavaQwenConfig(geometry=GEOMETRIES["toy_qwen"], mm_patch_merge_type="spatial_unpad", image_aspect_ratio="anyres_max_2",
                           image_grid_pinpoints=pin, unfreeze_mm_vision_tower=True)
    del model
    model = LlavaQwenForCausalLM(cfg2, device="cuda:0", init="portable")
    rng = np.random.default_rng(0)
    recs = []
    for i in range(4):
        Image.fromarray(rng.integers(0, 255, (100, 100 if i % 2 else 60, 3), dtype=np.uint8)).save(tmp_path / f"im{i}.png")
        recs.append({"id": f"s{i}", "image": f"im{i}.png", "conversations": [{"from": "human", "value": "<image>\nWhat is it?"},
                                                                            {"from": "gpt", "value": f"Finding number {i}."}]})
    recs.append({"id": "t", "conversations": [{"from": "human", "value": "Hello there"}, {"from": "gpt", "value": "General reply."}]})
    (tmp_path / "d.json").write_text(_json.dumps(recs))
    conv_lib.default_conversation = conv_lib.conv_templates["qwen_2"]
    try:
        da = DataArguments(data_path=str(tmp_path / "d.json"), image_folder=str(tmp_path), image_aspect_ratio="anyres_max_2",
                           image_grid_pinpoints=pin, is_multimodal=True)
        da.image_processor = SigLipImageProcessor(size=(54, 54), crop_size={"height": 54, "width": 54})
        da.mm_use_im_start_end = False
        tok = T.build()
        tok.model_max_length = 4096
        module = make_supervised_data_module(tokenizer=tok, data_args=da)
        item = module["train_dataset"][0]
        assert item["image"][0][0].shape[1:] == (3, 54, 54) and item["image"][0][0].shape[0] in (3, 5) and -200 in item["input_ids"].tolist()
        args = TrainingArguments(per_device_train_batch_size=2, gradient_accumulation_steps=1, max_steps=2, learning_rate=1e-3, warmup_ratio=0.0)
        state = LLaVATrainer(model=model, tokenizer=tok, args=args, **module).train()
        assert state["global_step"] == 2 and all(np.isfinite(r["loss"]) for r in state["log_history"])
    finally:
        conv_lib.default_conversation = conv_lib.conv_templates["v1"]


def test_checkpoint_resume_is_bit_identical(golden_dir, tmp_path):
    """a run of 4 steps == 2 steps + checkpoint + a fresh process-like resume + 2 steps (weights, fp32 master, AdamW
    moments, step counters and the sample stream all continue), and the checkpoint holds the reference-format pieces."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import json as _json
    from PIL import Image
    from radvlm_amd.llava import conversation as conv_lib
    from radvlm_amd.llava.mm_utils import ClipImageProcessor
    from radvlm_amd.llava.model import LlavaConfig, LlavaLlamaForCausalLM
    from radvlm_amd.llava.train.llava_trainer import LLaVATrainer
    from radvlm_amd.llava.train.train import DataArguments, TrainingArguments, make_supervised_data_module

    class Ids:
        def __init__(self, ids):
            self.input_ids = ids

    class Tok:
        bos_token_id, pad_token_id, model_max_length, legacy, padding_side = 1, 0, 256, True, "right"

        def __call__(self, s, **kw):
            ids = [1]
            for k, piece in enumerate(s.split("</s>")):
                if k:
                    ids.append(2)
                ids.extend(3 + (ord(c) % 900) for c in piece)
            return Ids(ids)

    rng = np.random.default_rng(1)
    recs = []
    for i in range(8):
        Image.fromarray(rng.integers(0, 255, (64, 80, 3), dtype=np.uint8)).save(tmp_path / f"im{i}.png")
        recs.append({"id": f"s{i}", "image": f"im{i}.png", "conversations": [{"from": "human", "value": "<image>\nWhat is it?"},
                                                                            {"from": "gpt", "value": f"Finding number {i} " + "x" * i}]})
    (tmp_path / "d.json").write_text(_json.dumps(recs))
    conv_lib.default_conversation = conv_lib.conv_templates["v1"]

    def run(out_dir, max_steps, save_steps, resume):
        da = DataArguments(data_path=str(tmp_path / "d.json"), image_folder=str(tmp_path), image_aspect_ratio="pad", is_multimodal=True)
        da.image_processor = ClipImageProcessor(56)
        da.mm_use_im_start_end = False
        tok = Tok()
        model = LlavaLlamaForCausalLM(LlavaConfig(geometry=GEOMETRIES["toy"]), device="cuda:0", init="portable")
        module = make_supervised_data_module(tokenizer=tok, data_args=da)
        args = TrainingArguments(per_device_train_batch_size=2, gradient_accumulation_steps=1, max_steps=max_steps, learning_rate=1e-3,
                                 warmup_ratio=0.0, output_dir=str(out_dir), save_steps=save_steps)
        tr = LLaVATrainer(model=model, tokenizer=tok, args=args, **module)
        state = tr.train(resume_from_checkpoint=resume)
        torch.cuda.synchronize()
        return model, state

    full, s_full = run(tmp_path / "a", 4, 2, None)          # 4 steps, checkpoints after steps 2 and 4
    ck = tmp_path / "a" / "checkpoint-2"
    assert sorted(os.listdir(ck)) == ["config.json", "mm_projector.bin", "model.safetensors", "optimizer.safetensors", "trainer_state.json"]
    proj = torch.load(ck / "mm_projector.bin", map_location="cpu", weights_only=True)
    assert sorted(proj) == ["model.mm_projector.0.bias", "model.mm_projector.0.weight", "model.mm_projector.2.bias", "model.mm_projector.2.weight"]
    resumed, s_res = run(tmp_path / "b", 4, 0, str(ck))     # fresh model, continue from step 2 of the same 4-step schedule
    assert s_res["global_step"] == 4 and [r["step"] for r in s_res["log_history"]] == [1, 2, 3, 4]
    assert torch.equal(resumed.engine.lm.flat, full.engine.lm.flat)
    assert torch.equal(resumed.engine.master, full.engine.master) and torch.equal(resumed.engine.vv, full.engine.vv)
    assert [r["loss"] for r in s_res["log_history"]] == [r["loss"] for r in s_full["log_history"]]
    # auto-resume picks the newest checkpoint: nothing left to do after checkpoint-4
    again, s_again = run(tmp_path / "a", 4, 0, True)
    assert s_again["global_step"] == 4 and torch.equal(again.engine.lm.flat, full.engine.lm.flat)


@pytest.mark.parametrize("name,golden,kw", [("toy", "toy_e2e", {}), ("toy_qwen", "toy_qwen_e2e", {"train_vision_tower": True})])
def test_training_reduces_the_loss(golden_dir, name, golden, kw):
    """Forward + backward + AdamW really train: 25 steps on one fixed batch drive the loss far below its starting value (both model
    flavours; the Qwen2/SigLIP one with the tower tunable), and every parameter stays finite."""
    g, meta, images = _golden(golden_dir, golden)
    eng = _engine(name, **kw)
    losses = []
    for _ in range(25):
        loss = eng.forward(g["input_ids"], g["attention_mask"], g["labels"], images)
        eng.backward()
        eng.optimizer_step(lr=2e-3, weight_decay=0.0, max_grad_norm=1.0)
        losses.append(float(loss))
    assert abs(losses[0] - float(g["loss"])) < 5e-3
    assert losses[-1] < 0.5 * losses[0], losses
    assert all(np.isfinite(losses)) and bool(torch.isfinite(eng.lm.flat.float()).all())


@pytest.mark.parametrize("packed", [False, True])
def test_edge_cases_against_reference_golden(golden_dir, packed):
    """The same edge-case batch against the REFERENCE's own outputs (tests/golden/toy_edge_e2e.*, make_golden.py edge)."""
    g, meta, _ = _golden(golden_dir, "toy_edge_e2e")
    images = [torch.from_numpy(g[f"image{i}"]) for i in range(5)]
    eng = _engine("toy", max_len=meta["max_len"], packed=packed)
    loss = eng.forward(g["input_ids"], g["attention_mask"], g["labels"], images, want_logits=True)
    logits = eng.last_logits.cpu()
    plan = eng.ctx["plan"]
    eng.backward()
    torch.cuda.synchronize()
    assert np.array_equal(plan["labels"], g["splice_labels"]) and np.array_equal(plan["attention_mask"], g["splice_attention_mask"])
    assert abs(float(loss) - float(g["loss"])) < 5e-3
    m = torch.from_numpy(g["splice_attention_mask"])
    ref = torch.from_numpy(g["logits"])
    assert float((logits[:, :, ::5][m] - ref[m]).abs().max() / ref[m].abs().max()) < 3e-2
    for k, want in meta["grad_norms"].items():
        if want is not None and k in eng.lm.offsets:
            got = float(eng.G(k).float().norm())
            assert abs(got - want) < 5e-2 * want + 1e-5, (k, got, want)


@pytest.mark.parametrize("packed", [False, True])
def test_edge_cases_against_oracle(packed):
    """Edge cases of the splice (llava_arch.py:442-531) through the HIP engine vs the CPU oracle: an image placeholder at the
    first and at the last position, two images in one sample, a text-only sample (consumes a dummy image, contributes no rows),
    ragged lengths, and truncation at tokenizer_model_max_length that cuts through an image span."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import llava_oracle as O
    from radvlm_amd.engine import LlavaEngine
    geo = GEOMETRIES["toy"]
    g = torch.Generator().manual_seed(11)
    T = 14
    ids = torch.randint(3, 1000, (4, T), generator=g)
    labels = ids.clone()
    mask = torch.ones(4, T, dtype=torch.bool)
    ids[0, 0] = -200                               # image first
    ids[1, 9] = -200; mask[1, 10:] = False         # image last (sample of 10 ids)
    ids[2, 3] = -200; ids[2, 8] = -200             # two images in one sample
    mask[3, 6:] = False                            # text only, short
    labels[ids == -200] = -100
    labels[:, :2] = -100
    ids[~mask] = 0; labels[~mask] = -100
    images = [torch.randn(3, 56, 56, generator=g).to(torch.bfloat16).float() for _ in range(4)] + [torch.zeros(3, 56, 56)]
    # image slots are consumed in order: sample 0 -> img0, sample 1 -> img1, sample 2 -> img2 and img3, sample 3 (text) -> img4 (dummy)
    max_len = 30                                    # sample 2 would be 12 + 2*16 = 44 rows: cut inside its second image
    eng = LlavaEngine(geo, device="cuda:0", init="portable", seed=0, max_len=max_len, packed=packed)
    loss = eng.forward(ids.numpy(), mask.numpy(), labels.numpy(), images, want_logits=True)
    logits = eng.last_logits.cpu()
    plan = eng.ctx["plan"]
    eng.backward()
    torch.cuda.synchronize()
    P = O.make_params(geo, seed=0)
    for k, v in P.items():
        if "vision_tower" not in k:
            v.requires_grad_(True)
    rl, rlog, aux = O.llava_forward(P, geo, ids, mask, labels, images, cfg={"tokenizer_model_max_length": max_len})
    rl.backward()
    assert np.array_equal(plan["labels"], aux["labels"].numpy()) and np.array_equal(plan["attention_mask"], aux["attention_mask"].numpy())
    assert plan["S"] == max_len and plan["lens"].tolist() == [13 + 16, 9 + 16, 30, 6]
    assert abs(float(loss) - float(rl)) < 5e-3, (float(loss), float(rl))
    m = aux["attention_mask"]
    ref = rlog.detach()[m]
    assert float((logits[m] - ref).abs().max() / ref.abs().max()) < 3e-2
    for k in ("model.embed_tokens.weight", "model.mm_projector.2.weight", "model.layers.0.self_attn.q_proj.weight", "lm_head.weight"):
        got, want = eng.G(k).float().cpu(), P[k].grad
        assert float((got - want).norm() / want.norm()) < 5e-2, k


def test_vocab_growth_for_extra_image_tokens(golden_dir):
    """initialize_vision_tokenizer (llava_arch.py:557-597): <im_start>/<im_end> rows appended to embed_tokens / lm_head as the mean of the
    existing rows, every other tensor untouched; the grown model still matches the oracle (which sees the grown tables)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import sys
    from types import SimpleNamespace
    from oracle import llava_oracle as O
    sys.path.insert(0, golden_dir)
    import toy_chatml_tokenizer as T
    from radvlm_amd.llava.model import LlavaConfig, LlavaLlamaForCausalLM
    g, meta, images = _golden(golden_dir, "toy_e2e")
    model = LlavaLlamaForCausalLM(LlavaConfig(geometry=GEOMETRIES["toy"]), device="cuda:0", init="portable")
    eng = model.engine
    before = {k: v.clone() for k, v in eng.state_dict().items()}
    tok = T.build()
    n0 = len(tok)
    model.initialize_vision_tokenizer(SimpleNamespace(mm_use_im_patch_token=False, mm_use_im_start_end=True, tune_mm_mlp_adapter=False), tok)
    assert len(tok) == n0 + 2 and eng.l["vocab"] == 1000          # the toy tokenizer (202 ids) still fits the 1000-row tables: no growth
    eng.resize_token_embeddings(1002)
    after = eng.state_dict()
    assert eng.vocab == 1002 and eng.l["vocab"] == 1008 and after["lm_head.weight"].shape[0] == 1002      # tables padded to 8 rows, exported unpadded
    assert float(eng.W("lm_head.weight")[1002:].float().abs().max()) == 0.0
    assert GEOMETRIES["toy"]["lm"]["vocab"] == 1000                # the shared geometry table is not mutated
    for k, v in before.items():
        if k in ("model.embed_tokens.weight", "lm_head.weight"):
            assert torch.equal(after[k][:1000], v)
            mean = v.float().mean(0).to(torch.bfloat16)
            assert torch.equal(after[k][1000], mean) and torch.equal(after[k][1001], mean)
        else:
            assert torch.equal(after[k], v), k
    ids = g["input_ids"].copy()
    ids[0, 1], ids[1, 0] = 1000, 1001                               # use the new rows
    labels = g["labels"].copy()
    labels[0, 1] = 1000
    loss = eng.forward(ids, g["attention_mask"], labels, images, want_logits=True)
    assert tuple(eng.last_logits.shape)[-1] == 1002
    eng.backward()
    torch.cuda.synchronize()
    geo2 = {"vision": GEOMETRIES["toy"]["vision"], "lm": dict(GEOMETRIES["toy"]["lm"], vocab=1002)}
    P = {k: v.float().cpu() for k, v in after.items()}
    for k in P:
        if "vision_tower" not in k:
            P[k].requires_grad_(True)
    rl, rlog, aux = O.llava_forward(P, geo2, torch.from_numpy(ids), torch.from_numpy(g["attention_mask"]), torch.from_numpy(labels), images)
    rl.backward()
    assert abs(float(loss) - float(rl)) < 5e-3
    ge, we = eng.G("model.embed_tokens.weight").float().cpu(), P["model.embed_tokens.weight"].grad
    assert float((ge[1000:1002] - we[1000:]).norm() / we[1000:].norm()) < 5e-2 and float(we[1000:].norm()) > 0
    assert float(ge[1002:].abs().max()) == 0.0 and float(eng.G("lm_head.weight").float()[1002:].abs().max()) == 0.0   # pad rows: zero gradient
    eng.optimizer_step(lr=1e-3, weight_decay=0.1, max_grad_norm=1.0)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(eng.lm.flat.float()).all())
    assert float(eng.W("lm_head.weight")[1002:].float().abs().max()) == 0.0 and float(eng.W("model.embed_tokens.weight")[1002:].float().abs().max()) == 0.0


def test_projector_only_stage_and_state_dict_round_trip(golden_dir):
    """tune_mm_mlp_adapter / mm_tunable_parts='mm_mlp_adapter' (train/train.py:1613-1640): tower and decoder frozen, gradients and optimizer
    state for the projector alone -- same loss and projector gradients as the full run; the full state dict (frozen stores included)
    round-trips through load_state_dict, also into a LoRA engine's frozen base."""
    from radvlm_amd.engine import LlavaEngine
    g, meta, images = _golden(golden_dir, "toy_e2e")
    eng = _engine("toy", freeze_lm=True)
    assert eng.lm.names() == ["model.mm_projector.0.weight", "model.mm_projector.0.bias", "model.mm_projector.2.weight", "model.mm_projector.2.bias"]
    assert eng.grads.numel() < 0.1 * eng.base.numel
    loss, logits, plan = _run(eng, g, images)
    assert abs(loss - float(g["loss"])) < 5e-3
    for k in eng.lm.names():
        want = meta["grad_norms"][k]
        assert abs(float(eng.G(k).float().norm()) - want) < 5e-2 * want, k
    ref = torch.from_numpy(g["grad::model.mm_projector.0.weight"])
    assert float((eng.G("model.mm_projector.0.weight").float().cpu() - ref).norm() / ref.norm()) < 5e-2
    lm_before = eng.base.flat.clone()
    proj_before = eng.lm.flat.clone()
    eng.optimizer_step(lr=1e-3, max_grad_norm=1.0)
    torch.cuda.synchronize()
    assert torch.equal(eng.base.flat, lm_before) and not torch.equal(eng.lm.flat, proj_before)
    sd = {k: v.clone() for k, v in eng.state_dict().items()}
    assert "model.layers.1.mlp.down_proj.weight" in sd and "lm_head.weight" in sd and "model.mm_projector.2.bias" in sd
    full = LlavaEngine(GEOMETRIES["toy"], device="cuda:0", init="fast", seed=9)
    missing, unexpected = full.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    for k, v in sd.items():
        assert torch.equal(full.state_dict()[k], v), k
    lora = LlavaEngine(GEOMETRIES["toy"], device="cuda:0", init="fast", seed=5, lora=dict(r=8, alpha=16, dropout=0.0))
    missing, unexpected = lora.load_state_dict(sd)
    assert all(".lora_" in k for k in missing) and not unexpected
    assert torch.equal(lora.base.view("model.layers.0.self_attn.q_proj.weight"), sd["model.layers.0.self_attn.q_proj.weight"])
