"""GPU tests (-m gpu) of the training-entry behaviours the reference script relies on: train() starts from the saved checkpoint it is
pointed at, gradient accumulation averages, LoRA checkpoints resume, left-padded batches, and the pretraining stage that trains the
input embeddings of the added <im_start>/<im_end> tokens (llava_arch.py:557-597)."""
import json
import os

import numpy as np
import pytest
import torch

from radvlm_amd.config import GEOMETRIES

pytestmark = pytest.mark.gpu


class _Ids:
    def __init__(self, ids):
        self.input_ids = ids


class Tok:
    """Character tokenizer with the attributes the data path reads (ids < 1000 = the toy vocabulary)."""
    bos_token_id, pad_token_id, model_max_length, legacy, padding_side = 1, 0, 256, True, "right"

    def __init__(self):
        self.extra = []

    def __call__(self, s, **kw):
        ids = [1]
        for k, piece in enumerate(s.split("</s>")):
            if k:
                ids.append(2)
            ids.extend(3 + (ord(c) % 900) for c in piece)
        return _Ids(ids)

    def add_tokens(self, toks, special_tokens=False):
        new = [t for t in toks if t not in self.extra]
        self.extra += new
        return len(new)

    def __len__(self):
        return 1000 + len(self.extra)


def _dataset(tmp_path, n=8, seed=1):
    from PIL import Image
    rng = np.random.default_rng(seed)
    recs = []
    for i in range(n):
        Image.fromarray(rng.integers(0, 255, (64, 80, 3), dtype=np.uint8)).save(tmp_path / f"im{i}.png")
        recs.append({"id": f"s{i}", "image": f"im{i}.png", "conversations": [{"from": "human", "value": "<image>\nWhat is it?"},
                                                                            {"from": "gpt", "value": f"Finding number {i} " + "x" * i}]})
    (tmp_path / "d.json").write_text(json.dumps(recs))
    return str(tmp_path / "d.json")


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _golden(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    meta = json.load(open(os.path.join(golden_dir, name + "_gradnorms.json")))
    n = len([k for k in g.files if k.startswith("image") and k[5:].isdigit()])
    return g, meta, [torch.from_numpy(g[f"image{i}"]) for i in range(n)]


def test_gradient_accumulation_averages_the_micro_batches(golden_dir):
    """HF Trainer back-propagates loss / gradient_accumulation_steps per micro-batch: two micro-batches of 2 (loss_scale 1/2) must leave
    the gradients, hence the pre-clip norm max_grad_norm acts on, of ONE batch of the same 4 samples when the micro-batches hold
    equally many label tokens (mean of means == mean)."""
    _need_gpu()
    from radvlm_amd.engine import LlavaEngine
    geo = GEOMETRIES["toy"]
    rng = np.random.default_rng(3)
    T = 18
    ids = rng.integers(3, 1000, size=(4, T), dtype=np.int64)
    labels = ids.copy()
    labels[:, :6] = -100                        # 12 label tokens per sample in every micro-batch
    ids[:, 4] = -200
    labels[:, 4] = -100
    mask = np.ones_like(ids, dtype=bool)
    images = [torch.randn(3, 56, 56, generator=torch.Generator().manual_seed(i)).to(torch.bfloat16) for i in range(4)]
    whole = LlavaEngine(geo, device="cuda:0", init="portable", seed=0)
    l4 = float(whole.forward(ids, mask, labels, images))
    whole.backward()
    acc = LlavaEngine(geo, device="cuda:0", init="portable", seed=0)
    acc.loss_scale = 0.5
    la = float(acc.forward(ids[:2], mask[:2], labels[:2], images[:2]))
    acc.backward()
    lb = float(acc.forward(ids[2:], mask[2:], labels[2:], images[2:]))
    acc.backward()
    torch.cuda.synchronize()
    assert abs(0.5 * (la + lb) - l4) < 2e-3                 # the reported losses are never scaled
    gw, ga = whole.grads.float(), acc.grads.float()
    assert float((ga - gw).norm() / gw.norm()) < 2e-2
    nw = float(whole.optimizer_step(lr=1e-3, max_grad_norm=1.0) or whole.last_grad_norm)
    na = float(acc.optimizer_step(lr=1e-3, max_grad_norm=1.0) or acc.last_grad_norm)
    assert abs(na - nw) < 2e-2 * nw
    # and the un-averaged sum the trainer used to build is twice as large (what the advisor flagged)
    summed = LlavaEngine(geo, device="cuda:0", init="portable", seed=0)
    summed.forward(ids[:2], mask[:2], labels[:2], images[:2]); summed.backward()
    summed.forward(ids[2:], mask[2:], labels[2:], images[2:]); summed.backward()
    assert abs(float(summed.grads.float().norm()) / float(gw.norm()) - 2.0) < 0.05


def _train_args(tmp_path, data, extra):
    return ["--data_path", data, "--image_folder", str(tmp_path), "--image_aspect_ratio", "pad", "--version", "v1",
            "--per_device_train_batch_size", "2", "--learning_rate", "1e-3", "--warmup_ratio", "0.0", "--mm_projector_type", "mlp2x_gelu",
            "--mm_vision_select_layer", "-2", "--mm_use_im_patch_token", "False", "--model_max_length", "256"] + extra


def test_train_starts_from_the_checkpoint_it_is_given(tmp_path):
    """train() with --model_name_or_path DIR loads DIR's weights (reference: get_model -> from_pretrained, train/train.py:1358-1427): the
    first logged loss is the loss of the saved model on that batch, not that of a random model; a hub name raises instead of silently
    training random weights."""
    _need_gpu()
    from radvlm_amd.llava import conversation as conv_lib
    from radvlm_amd.llava.model import LlavaConfig, LlavaLlamaForCausalLM
    from radvlm_amd.llava.train.train import train
    data = _dataset(tmp_path, n=2)
    saved = LlavaLlamaForCausalLM(LlavaConfig(geometry=GEOMETRIES["toy"]), device="cuda:0", init="portable", seed=0)
    ck = tmp_path / "base"
    saved.save_pretrained(str(ck))
    assert {"config.json", "model.safetensors"} <= set(os.listdir(ck))
    try:
        state = train(argv=_train_args(tmp_path, data, ["--model_name_or_path", str(ck), "--max_steps", "1", "--output_dir", str(tmp_path / "out")]),
                      tokenizer=Tok())
        first = state["log_history"][0]["loss"]
        # the same batch through the saved model directly
        from radvlm_amd.llava.mm_utils import ClipImageProcessor
        from radvlm_amd.llava.train.train import DataArguments, make_supervised_data_module
        da = DataArguments(data_path=data, image_folder=str(tmp_path), image_aspect_ratio="pad", is_multimodal=True)
        da.image_processor, da.mm_use_im_start_end = ClipImageProcessor(56), False
        mod = make_supervised_data_module(tokenizer=Tok(), data_args=da)
        batch = mod["data_collator"]([mod["train_dataset"][0], mod["train_dataset"][1]])
        want = float(saved(**batch).loss)
        assert abs(first - want) < 1e-4, (first, want)
        rand = LlavaLlamaForCausalLM(LlavaConfig(geometry=GEOMETRIES["toy"]), device="cuda:0", init="fast", seed=5)
        assert abs(float(rand(**batch).loss) - want) > 1e-2
        with pytest.raises(FileNotFoundError):
            train(argv=_train_args(tmp_path, data, ["--model_name_or_path", "lmsys/vicuna-7b-v1.5", "--vision_tower", "openai/clip-vit-large-patch14-336",
                                                    "--max_steps", "1"]), tokenizer=Tok())
        # the output directory of the run is itself a loadable checkpoint
        assert {"config.json", "model.safetensors"} <= set(os.listdir(tmp_path / "out"))
    finally:
        conv_lib.default_conversation = conv_lib.conv_templates["v1"]


def test_lora_run_saves_the_reference_files_and_resumes(tmp_path):
    """--lora_enable: the run leaves adapter_model.bin + adapter_config.json + non_lora_trainables.bin (train/train.py:1708-1717) and a
    restarted run (auto-resume finds checkpoint-*, train.py:1699-1702) continues bit-identically instead of raising."""
    _need_gpu()
    from radvlm_amd.llava import conversation as conv_lib
    from radvlm_amd.llava.train.train import train
    data = _dataset(tmp_path, n=8)
    common = ["--geometry", "toy", "--lora_enable", "True", "--lora_r", "8", "--lora_alpha", "16", "--lora_dropout", "0.05", "--save_steps", "2"]
    try:
        train(argv=_train_args(tmp_path, data, common + ["--max_steps", "4", "--output_dir", str(tmp_path / "a")]), tokenizer=Tok())
        assert {"adapter_model.bin", "adapter_config.json", "non_lora_trainables.bin", "config.json"} <= set(os.listdir(tmp_path / "a"))
        ad = torch.load(tmp_path / "a" / "adapter_model.bin", map_location="cpu", weights_only=True)
        nl = torch.load(tmp_path / "a" / "non_lora_trainables.bin", map_location="cpu", weights_only=True)
        assert "base_model.model.model.layers.0.self_attn.q_proj.lora_A.weight" in ad and ad["base_model.model.model.layers.1.mlp.down_proj.lora_B.weight"].shape == (256, 8)
        assert sorted(nl) == ["base_model.model.model.mm_projector." + k for k in ("0.bias", "0.weight", "2.bias", "2.weight")]
        cfg = json.load(open(tmp_path / "a" / "adapter_config.json"))
        assert cfg["r"] == 8 and cfg["lora_alpha"] == 16 and cfg["peft_type"] == "LORA" and "q_proj" in cfg["target_modules"]
        # a restart of the same command in a directory that holds checkpoint-2 (as if the job had died after step 2) resumes there
        import shutil
        os.makedirs(tmp_path / "b")
        shutil.copytree(tmp_path / "a" / "checkpoint-2", tmp_path / "b" / "checkpoint-2")
        st = train(argv=_train_args(tmp_path, data, common + ["--max_steps", "4", "--output_dir", str(tmp_path / "b")]), tokenizer=Tok())
        assert [r["step"] for r in st["log_history"]] == [1, 2, 3, 4]
        ad2 = torch.load(tmp_path / "b" / "adapter_model.bin", map_location="cpu", weights_only=True)
        for k in ad:
            assert torch.equal(ad[k], ad2[k]), k
    finally:
        conv_lib.default_conversation = conv_lib.conv_templates["v1"]


def test_save_only_model_checkpoints_and_their_resume(tmp_path):
    """--save_only_model True (HF TrainingArguments.save_only_model): checkpoint-N holds the weights and the trainer state but no optimizer
    state (81 GB at 7B); a run resumed from it continues at step N + 1 with a fresh AdamW (moments zero, bias correction from step 1)."""
    _need_gpu()
    from radvlm_amd.llava import conversation as conv_lib
    from radvlm_amd.llava.train.train import train
    data = _dataset(tmp_path, n=8)
    common = ["--geometry", "toy", "--save_steps", "2", "--save_only_model", "True"]
    try:
        train(argv=_train_args(tmp_path, data, common + ["--max_steps", "3", "--output_dir", str(tmp_path / "a")]), tokenizer=Tok())
        ck = set(os.listdir(tmp_path / "a" / "checkpoint-2"))
        assert "model.safetensors" in ck and "trainer_state.json" in ck and "optimizer.safetensors" not in ck
        st = train(argv=_train_args(tmp_path, data, common + ["--max_steps", "4", "--output_dir", str(tmp_path / "a")]), tokenizer=Tok())
        assert [r["step"] for r in st["log_history"]] == [1, 2, 3, 4]
        assert all(np.isfinite(r["loss"]) for r in st["log_history"])
        # with its optimizer state a checkpoint is larger and resumes the moments (the default)
        train(argv=_train_args(tmp_path, data, ["--geometry", "toy", "--save_steps", "2", "--max_steps", "2", "--output_dir", str(tmp_path / "b")]), tokenizer=Tok())
        assert "optimizer.safetensors" in os.listdir(tmp_path / "b" / "checkpoint-2")
    finally:
        conv_lib.default_conversation = conv_lib.conv_templates["v1"]


@pytest.mark.parametrize("packed", ["auto", False])
def test_left_padding_against_reference_golden(golden_dir, packed):
    """config.tokenizer_padding_side = 'left' (llava_arch.py:520-524) against a reference-generated fixture: splice labels / mask
    bit-exact, loss / logits / gradients within the bf16 gates.  Short samples sit at the end of their row and keep positions
    arange(S) (the reference discards the spliced position_ids in training)."""
    _need_gpu()
    from radvlm_amd.engine import LlavaEngine
    g, meta, images = _golden(golden_dir, "toy_leftpad_e2e")
    assert not g["attention_mask"][1, 0] and g["attention_mask"][1, -1]
    eng = LlavaEngine(GEOMETRIES["toy"], device="cuda:0", init="portable", seed=0, padding_side="left", packed=packed)
    loss = float(eng.forward(g["input_ids"], g["attention_mask"], g["labels"], images, want_logits=True))
    logits = eng.last_logits.cpu()
    plan = eng.ctx["plan"]
    eng.backward()
    torch.cuda.synchronize()
    assert np.array_equal(plan["labels"], g["splice_labels"]) and np.array_equal(plan["attention_mask"], g["splice_attention_mask"])
    assert abs(loss - float(g["loss"])) < 5e-3
    m = torch.from_numpy(g["splice_attention_mask"])
    ref = torch.from_numpy(g["logits"])
    assert float((logits[m] - ref[m]).abs().max() / ref[m].abs().max()) < 1.5e-2
    assert float(logits[~m].abs().max()) == 0.0                   # padding rows are never computed
    for k, want in meta["grad_norms"].items():
        if want is not None and k in eng.lm.offsets:
            got = float(eng.G(k).float().norm())
            assert abs(got - want) < 5e-2 * want + 1e-5, (k, got, want)
    for k in g.files:
        if k.startswith("grad::"):
            ref_g = torch.from_numpy(g[k])
            assert float((eng.G(k[6:]).float().cpu() - ref_g).norm() / ref_g.norm()) < 5e-2, k
    # a label on the first token of a left-padded row would make the reference read a padding row's logits: refused
    bad = g["labels"].copy()
    first = int(np.argmax(g["attention_mask"][1]))
    bad[1, first] = 7
    with pytest.raises(ValueError, match="left padding"):
        eng.forward(g["input_ids"], g["attention_mask"], bad, images)


def test_pretraining_stage_with_start_end_tokens_trains_the_input_embeddings(golden_dir, tmp_path):
    """tune_mm_mlp_adapter + mm_use_im_start_end (llava_arch.py:563-592): the tables grow by two mean-initialised rows, the INPUT
    embeddings train together with the projector while lm_head and the decoder stay frozen, pretrain_mm_mlp_adapter restores the two
    new rows, and the adapter-only save holds the projector + embed_tokens (llava_trainer.py:446-455 with use_im_start_end)."""
    _need_gpu()
    from types import SimpleNamespace
    from radvlm_amd.llava.model import LlavaConfig, LlavaLlamaForCausalLM
    g, meta, images = _golden(golden_dir, "toy_e2e")
    model = LlavaLlamaForCausalLM(LlavaConfig(geometry=GEOMETRIES["toy"], freeze_lm=True, train_embed_tokens=True), device="cuda:0", init="portable")
    eng = model.engine
    assert eng.lm.names()[0] == "model.embed_tokens.weight" and "lm_head.weight" not in eng.lm.offsets and "lm_head.weight" in eng.base.offsets
    rows = torch.randn(2, 256).to(torch.bfloat16)
    torch.save({"model.embed_tokens.weight": rows, "model.mm_projector.0.bias": torch.zeros(256)}, tmp_path / "pre.bin")
    tok = Tok()
    args = SimpleNamespace(mm_use_im_patch_token=False, mm_use_im_start_end=True, tune_mm_mlp_adapter=True, pretrain_mm_mlp_adapter=str(tmp_path / "pre.bin"))
    model.initialize_vision_tokenizer(args, tok)
    assert len(tok) == 1002 and eng.vocab == 1002 and model.config.vocab_size == 1002
    emb = eng.W("model.embed_tokens.weight")
    assert emb.shape[0] == 1008 and torch.equal(emb[1000:1002].cpu(), rows) and float(emb[1002:].float().abs().max()) == 0.0
    head = eng.W("lm_head.weight")
    assert torch.equal(head[1000], head[:1000].float().mean(0).to(torch.bfloat16))
    ids = g["input_ids"].copy()
    ids[0, 1], ids[0, 2] = 1000, 1001
    head_before, layer_before = head.clone(), eng.W("model.layers.0.mlp.down_proj.weight").clone()
    loss = float(eng.forward(ids, g["attention_mask"], g["labels"], images))
    eng.backward()
    ge = eng.G("model.embed_tokens.weight").float()
    assert np.isfinite(loss) and float(ge[1000:1002].norm()) > 0 and float(ge[1002:].abs().max()) == 0.0
    eng.optimizer_step(lr=1e-2, max_grad_norm=1.0)
    torch.cuda.synchronize()
    assert not torch.equal(eng.W("model.embed_tokens.weight")[1000:1002].cpu(), rows)           # trained
    assert torch.equal(eng.W("lm_head.weight"), head_before) and torch.equal(eng.W("model.layers.0.mlp.down_proj.weight"), layer_before)
    with pytest.raises(ValueError, match="train_embed_tokens"):
        frozen = LlavaLlamaForCausalLM(LlavaConfig(geometry=GEOMETRIES["toy"], freeze_lm=True), device="cuda:0", init="fast")
        frozen.initialize_vision_tokenizer(SimpleNamespace(mm_use_im_patch_token=False, mm_use_im_start_end=True, tune_mm_mlp_adapter=True,
                                                           pretrain_mm_mlp_adapter=None), Tok())


def test_device_image_normalisation_gives_the_same_training_run(tmp_path):
    """--device_image_normalize True: the dataset hands uint8 pixels over and the GPU normalises them (rv_normalize_tiles_u8); losses of a
    short run are bit-identical to the host-normalised run (same bf16 pixel tensor reaches the tower)."""
    _need_gpu()
    from radvlm_amd.llava import conversation as conv_lib
    from radvlm_amd.llava.train.train import train
    data = _dataset(tmp_path, n=4)
    common = ["--geometry", "toy", "--max_steps", "2", "--dataloader_num_workers", "0"]
    try:
        a = train(argv=_train_args(tmp_path, data, common + ["--output_dir", str(tmp_path / "h")]), tokenizer=Tok())
        b = train(argv=_train_args(tmp_path, data, common + ["--output_dir", str(tmp_path / "d"), "--device_image_normalize", "True"]), tokenizer=Tok())
        assert [r["loss"] for r in a["log_history"]] == [r["loss"] for r in b["log_history"]]
    finally:
        conv_lib.default_conversation = conv_lib.conv_templates["v1"]


def _toy_batch(golden_dir, name="toy_e2e"):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    n = len([k for k in g.files if k.startswith("image") and k[5:].isdigit()])
    return g, [torch.from_numpy(g[f"image{i}"]) for i in range(n)]


@pytest.mark.parametrize("geo_name,golden,kw", [("toy", "toy_e2e", {}), ("toy_qwen", "toy_qwen_e2e", {}),
                                                ("toy", "toy_e2e", {"lora": dict(r=8, alpha=16, dropout=0.1)})])
def test_activation_recompute_is_bit_identical(golden_dir, geo_name, golden, kw):
    """--gradient_checkpointing (reference train/train.py:164,1505-1513; modeling_llama.py:1139-1149): a layer that keeps only its input
    and re-runs its forward inside backward must leave bit-identical gradients (deterministic kernels; LoRA dropout masks are
    regenerated from the step's seed), whether all layers recompute or only the first one."""
    _need_gpu()
    from radvlm_amd.engine import LlavaEngine
    g, images = _toy_batch(golden_dir, golden)
    grads, losses = [], []
    for rc in ("none", "all", "first"):
        eng = LlavaEngine(GEOMETRIES[geo_name], device="cuda:0", init="portable", seed=0, recompute=rc != "none", **kw)
        if rc == "first":        # partial: only layer 0 recomputes
            eng._recompute_layers = lambda M: 1
        if kw.get("lora"):
            for n in eng.lm.names():      # peft zero-initialises lora_B: give it values so that the adapter path carries gradient
                if n.endswith("lora_B.weight"):
                    eng.lm.view(n).copy_(torch.randn(eng.lm.shapes[n], generator=torch.Generator().manual_seed(len(n))).to("cuda:0") * 0.05)
        loss = eng.forward(g["input_ids"], g["attention_mask"], g["labels"], images)
        if rc != "none":
            assert eng.ctx["n_recomputed"] == (eng.l["layers"] if rc == "all" else 1)
            assert all(set(a) == {"x", "lora"} for a in eng.ctx["layers"][:eng.ctx["n_recomputed"]])
        eng.backward()
        torch.cuda.synchronize()
        grads.append(eng.grads.clone())
        losses.append(float(loss))
    assert losses[0] == losses[1] == losses[2]
    assert float(grads[0].float().abs().max()) > 0
    assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2])


def test_auto_recompute_keeps_activations_when_they_fit(golden_dir):
    """recompute="auto" (what --gradient_checkpointing True selects): no layer recomputes while the batch's activations fit in free HBM,
    and the count grows with the token rows of a (hypothetical) batch."""
    _need_gpu()
    from radvlm_amd.engine import LlavaEngine
    eng = LlavaEngine(GEOMETRIES["toy"], device="cuda:0", init="portable", seed=0, recompute="auto")
    assert eng._recompute_layers(96) == 0
    free, _ = torch.cuda.mem_get_info()
    rows_too_many = int(free // eng.activation_bytes_per_row())       # one layer alone would fill the device
    assert eng._recompute_layers(rows_too_many) == eng.l["layers"]
    assert eng._recompute_layers(96) == 0 and eng._recompute_cache[96] == 0          # one decision per row count
    # HF's gradient_checkpointing_enable() selects the same policy (it used to force every layer: + 1/3 forward cost for nothing)
    from radvlm_amd.llava.model import LlavaConfig, LlavaLlamaForCausalLM
    m = LlavaLlamaForCausalLM(LlavaConfig(geometry=GEOMETRIES["toy"]), device="cuda:0", init="fast")
    m.gradient_checkpointing_enable()
    assert m.engine.recompute == "auto" and m.is_gradient_checkpointing
    m.gradient_checkpointing_disable()
    assert m.engine.recompute is False
    # 7B widths: 32 pairs x 704 tokens keep ~95 GB of activations
    big = LlavaEngine.__new__(LlavaEngine)
    big.l, big.kvd = GEOMETRIES["llava15_7b"]["lm"], 4096
    assert 90e9 < 32 * 704 * 32 * big.activation_bytes_per_row() < 100e9


@pytest.mark.parametrize("packed", [False, True])
def test_stale_softmax_statistics_are_masked(golden_dir, packed):
    """The lse / delta buffers are cached by padded shape and only the valid rows are rewritten: a batch with the same (B, s_pad) but
    shorter samples sees stale values in rows >= len.  They must not reach any gradient (the kernels mask those rows)."""
    _need_gpu()
    from radvlm_amd.engine import LlavaEngine
    g, images = _toy_batch(golden_dir)
    ids, lab = g["input_ids"].copy(), g["labels"].copy()
    am_full = np.ones_like(g["attention_mask"])
    am_short = am_full.copy()
    am_short[-1, -9:] = False           # same padded shape (B, s_pad), the last sample 9 tokens shorter
    lab_short = np.where(am_short, lab, -100)

    def run(eng, am, lb):
        eng.forward(ids, am, lb, images)
        eng.backward()
        torch.cuda.synchronize()
        eng.zero_grad()
        return eng.grads.clone()
    a = LlavaEngine(GEOMETRIES["toy"], device="cuda:0", init="portable", seed=0, packed=packed)
    run(a, am_full, lab)                                  # fills lse / delta for all rows
    # plant large (finite: stale values are old statistics) numbers where a later, shorter batch must not look
    assert a._stats
    for k, buf in a._stats.items():
        buf[-1, :, :] = torch.where(buf[-1] != 0, torch.full_like(buf[-1], 1e30), buf[-1])     # the last sample's rows: all stale now
    ga = run(a, am_short, lab_short)
    b = LlavaEngine(GEOMETRIES["toy"], device="cuda:0", init="portable", seed=0, packed=packed)
    gb = run(b, am_short, lab_short)
    assert torch.isfinite(ga.float()).all() and torch.equal(ga, gb)


def test_images_already_on_the_device(golden_dir):
    """HF Trainer._prepare_inputs moves every tensor of the batch to the device before model(**batch): device-resident images (and ids)
    must be accepted and give the same loss as host tensors."""
    _need_gpu()
    from radvlm_amd.llava.model import LlavaConfig, LlavaLlamaForCausalLM
    g, images = _toy_batch(golden_dir)
    model = LlavaLlamaForCausalLM(LlavaConfig(geometry=GEOMETRIES["toy"]), device="cuda:0")
    t = lambda a: torch.from_numpy(np.asarray(a))
    host = model(input_ids=t(g["input_ids"]), attention_mask=t(g["attention_mask"]), labels=t(g["labels"]), images=images)
    l0 = float(host.loss)
    model.engine.ctx = None
    dev = model(input_ids=t(g["input_ids"]).cuda(), attention_mask=t(g["attention_mask"]).cuda(), labels=t(g["labels"]).cuda(),
                images=torch.stack(images).cuda())
    assert float(dev.loss) == l0
    dev.loss.backward()
    torch.cuda.synchronize()


@pytest.mark.gpu
@pytest.mark.parametrize("parts", ["mm_vision_tower,mm_mlp_adapter", "mm_language_model", "mm_vision_tower", "mm_vision_tower,mm_language_model"])
def test_every_subset_of_mm_tunable_parts(golden_dir, parts):
    """mm_tunable_parts accepts ANY subset of tower / projector / language model (train/train.py:1613-1665: everything is frozen first,
    then each named part is unfrozen).  Against the all-tunable engine on the same batch: same loss, bit-identical gradients for the
    tensors that train, and after an optimizer step only those tensors have moved (image_newline belongs to the language-model group)."""
    _need_gpu()
    from radvlm_amd.engine import LlavaEngine
    g, images = _toy_batch(golden_dir, "toy_anyres_e2e")
    sizes = [tuple(s) for s in g["image_sizes"].tolist()]
    names = set(parts.split(","))
    meta = json.load(open(os.path.join(golden_dir, "toy_anyres_e2e_gradnorms.json")))
    kw = dict(merge_type=meta["merge_type"], image_aspect_ratio=meta["aspect"], image_grid_pinpoints=meta["pinpoints"])
    # the all-LM-parts-tunable engine with the SAME tower mode (a frozen tower applies fc1's activation in the GEMM epilogue: one rounding less)
    full = LlavaEngine(GEOMETRIES["toy"], device="cuda:0", init="portable", seed=0, train_vision_tower="mm_vision_tower" in names, **kw)
    eng = LlavaEngine(GEOMETRIES["toy"], device="cuda:0", init="portable", seed=0, train_vision_tower="mm_vision_tower" in names,
                      freeze_lm="mm_language_model" not in names, freeze_projector="mm_mlp_adapter" not in names, **kw)
    lf = float(full.forward(g["input_ids"], g["attention_mask"], g["labels"], images, image_sizes=sizes)); full.backward()
    le = float(eng.forward(g["input_ids"], g["attention_mask"], g["labels"], images, image_sizes=sizes)); eng.backward()
    assert lf == le

    def part_of(n):
        return "mm_vision_tower" if "vision_tower" in n else ("mm_mlp_adapter" if "mm_projector" in n else "mm_language_model")
    trainable = [n for n in eng.lm.names() if n not in eng.frozen_names]
    assert trainable and {part_of(n) for n in trainable} == names
    assert all(part_of(n) in names for n in trainable) and "model.image_newline" in full.lm.offsets
    for n in trainable:
        assert torch.equal(eng.G(n), full.G(n)), n
    before = {k: v.clone() for k, v in eng.state_dict().items()}
    eng.optimizer_step(lr=1e-2, weight_decay=0.1, max_grad_norm=1.0)
    torch.cuda.synchronize()
    after = eng.state_dict()
    moved = {k for k in before if not torch.equal(before[k], after[k])}
    assert moved and {part_of(k) for k in moved} == names, sorted(moved)[:5]
    # the clipped norm is the norm of the trainable gradients alone
    want = torch.sqrt(sum(full.G(n).float().pow(2).sum() for n in trainable))
    assert abs(float(eng.last_grad_norm) - float(want)) < 2e-3 * float(want)


@pytest.mark.gpu
def test_inputs_embeds_and_label_free_call_forms(golden_dir):
    """LlavaLlamaForCausalLM.forward beyond the training call (llava_llama.py:69-120): `inputs_embeds` given -> no multimodal splice, fp32
    logits (+ loss when labels are passed); `labels=None` with input_ids -> logits only.  Checked against the CPU oracle's decoder on the
    same embeddings (same gates as the other toy end-to-end tests), and left- against right-padded masks (rotary positions are relative)."""
    _need_gpu()
    from oracle import llava_oracle as O
    from radvlm_amd.llava.model import LlavaConfig, LlavaLlamaForCausalLM
    geo = GEOMETRIES["toy"]
    model = LlavaLlamaForCausalLM(LlavaConfig(geometry=geo), device="cuda:0", init="portable")
    model.training = False
    d, V = geo["lm"]["d"], geo["lm"]["vocab"]
    gen = torch.Generator().manual_seed(3)
    emb = (torch.randn(2, 40, d, generator=gen) * 0.05).to(torch.bfloat16)
    mask = torch.ones(2, 40, dtype=torch.bool); mask[1, 29:] = False
    labels = torch.randint(3, V, (2, 40), generator=gen); labels[:, :5] = -100
    out = model(inputs_embeds=emb, attention_mask=mask, labels=labels)
    P = {k: v.float().cpu() for k, v in model.engine.state_dict().items()}
    ref = O.llama_forward(P, geo, emb.float(), torch.tensor([40, 29]))
    ref_loss = float(O.causal_lm_loss(ref, torch.where(mask, labels, torch.full_like(labels, -100))))
    lg = out.logits.cpu()
    assert lg.dtype == torch.float32 and tuple(lg.shape) == (2, 40, V)
    valid = mask[:, :, None].expand_as(lg)
    assert float((lg - ref)[valid].abs().max() / ref[valid].abs().max()) < 1.5e-2
    assert abs(float(out.loss) - ref_loss) < 5e-3
    assert model(inputs_embeds=emb, attention_mask=mask).loss is None
    with pytest.raises(ValueError, match="both"):
        model(input_ids=torch.zeros(2, 40, dtype=torch.long), inputs_embeds=emb)
    # left padding: the same 29 tokens at columns 11..39 give the same logits (positions are the column index, attention is relative)
    lmask = torch.zeros(1, 40, dtype=torch.bool); lmask[0, 11:] = True
    lemb = torch.zeros(1, 40, d, dtype=torch.bfloat16); lemb[0, 11:] = emb[1, :29]
    left = model(inputs_embeds=lemb, attention_mask=lmask).logits.cpu()
    assert float((left[0, 11:] - lg[1, :29]).abs().max() / lg[1, :29].abs().max()) < 3e-2
    assert float(left[0, :11].abs().max()) == 0.0
    # labels=None with input_ids: logits of the spliced sequence, no loss
    g, images = _toy_batch(golden_dir)
    o2 = model(input_ids=torch.from_numpy(g["input_ids"]), attention_mask=torch.from_numpy(g["attention_mask"]), images=images)
    assert o2.loss is None and o2.logits is not None and o2.logits.dtype == torch.float32
    want, valid2 = torch.from_numpy(g["logits"]), torch.from_numpy(g["splice_attention_mask"]).bool()
    assert want.shape == o2.logits.shape
    assert float((o2.logits.cpu() - want)[valid2].abs().max() / want[valid2].abs().max()) < 1.5e-2       # the reference's own logits, valid rows


@pytest.mark.gpu
@pytest.mark.parametrize("packed,kw", [(False, {}), (True, {}), (False, {"lora": dict(r=8, alpha=16, dropout=0.0)})])
def test_head_on_labelled_rows_gives_the_same_loss_and_gradients(golden_dir, packed, kw):
    """The final norm, lm_head and cross entropy run on the rows that carry a label (ignore_index rows contribute exact zeros to the loss and
    to every gradient, modeling_llama.py:1326-1337; nothing after the last decoder layer mixes rows).  Against head_rows="all": same loss,
    same gradients up to the fp32 grouping of the weight-gradient sums, same fp32 logits for callers that ask for them."""
    _need_gpu()
    from radvlm_amd.engine import LlavaEngine
    g, images = _toy_batch(golden_dir)
    res = {}
    for mode in ("all", "labeled", "labeled_head_only"):
        eng = LlavaEngine(GEOMETRIES["toy"], device="cuda:0", init="portable", seed=0, packed=packed, **kw)
        if kw:      # adapters: lora_B starts at zero -- give it values so that every adapter gradient is exercised
            gb = torch.Generator().manual_seed(5)
            for n in eng.lm.names():
                if n.endswith("lora_B.weight"):
                    eng.lm.view(n).copy_((torch.randn(eng.lm.shapes[n], generator=gb) * 0.05).to(torch.bfloat16))
        eng.head_rows = "all" if mode == "all" else "labeled"
        eng.last_layer_rows = "all" if mode == "labeled_head_only" else "labeled"
        loss = float(eng.forward(g["input_ids"], g["attention_mask"], g["labels"], images))
        used, last_sel = eng.ctx["head_idx"], eng.ctx["last_sel"]
        n_mlp_rows = eng.ctx["layers"][-1]["act"].shape[0]
        eng.backward()
        torch.cuda.synchronize()
        eng.forward(g["input_ids"], g["attention_mask"], g["labels"], images, want_logits=True)      # callers' logits: every row, fp32
        res[mode] = (loss, eng.last_logits.clone(), eng.grads.float().clone(), used, eng, last_sel, n_mlp_rows)
        eng.ctx = None
    M = res["all"][6]
    assert res["all"][3] is None and res["labeled"][3] is not None and len(res["labeled"][3]) % 64 == 0
    # the last decoder layer's o_proj / norm / MLP ran on the labelled rows too (same dead-row argument one step earlier)
    assert res["labeled"][5] and not res["labeled_head_only"][5] and res["labeled"][6] == len(res["labeled"][3]) < M == res["labeled_head_only"][6]
    n_lab = int((np.roll(g["labels"], -1, 1)[:, :-1] != -100).sum())
    assert (res["labeled"][3] >= 0).sum() <= n_lab           # the spliced labels: text positions only (image rows carry none)
    eng = res["all"][4]
    for mode in ("labeled", "labeled_head_only"):
        assert abs(res["all"][0] - res[mode][0]) < 1e-6, mode
        assert torch.equal(res["all"][1], res[mode][1])                                   # callers' fp32 logits: the full product either way
        for n in eng.lm.names():
            a, b = eng.lm.view(n, res["all"][2]), eng.lm.view(n, res[mode][2])
            assert float((a - b).norm()) <= 2e-3 * float(a.norm()) + 1e-12, (mode, n)
