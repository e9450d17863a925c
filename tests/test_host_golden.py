"""CPU tests of the host-side rows against golden vectors produced by the
reference's own functions (tests/golden/make_golden_host.py, make_golden.py)."""
import copy
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch
from PIL import Image

from radvlm_amd import splice as SP
from radvlm_amd.llava import conversation as conv_lib
from radvlm_amd.llava import mm_utils as MU
from radvlm_amd.llava.train import llava_trainer as LT
from radvlm_amd.llava.train import train as TR


@pytest.fixture(scope="module")
def H(golden_dir):
    return json.load(open(os.path.join(golden_dir, "host_golden.json")))


class Ids:
    def __init__(self, ids):
        self.input_ids = ids


class CharTok:
    bos_token_id = 1
    pad_token_id = 0
    model_max_length = 512
    legacy = True
    padding_side = "right"

    def __call__(self, s, **kw):
        ids = [1]
        for k, piece in enumerate(s.split("</s>")):
            if k:
                ids.append(2)
            ids.extend(ord(c) for c in piece)
        return Ids(ids)


def img(seed, w, h):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 255, size=(h // 8 + 2, w // 8 + 2, 3), dtype=np.uint8)
    return Image.fromarray(base).resize((w, h), Image.BICUBIC)


def test_samplers(H):
    for c in H["samplers"]:
        L, ML, bs, ws = c["lengths"], c["modality_lengths"], c["batch_size"], c["world_size"]
        for fn, ln in (("get_length_grouped_indices", L), ("get_length_grouped_indices_auto_single", L),
                       ("get_variable_length_grouped_indices", L), ("get_modality_length_grouped_indices", ML),
                       ("get_modality_length_grouped_indices_auto", ML)):
            g = torch.Generator().manual_seed(c["seed"])
            torch.manual_seed(1000 + c["seed"])  # inner permutations of the modality variants use the global generator
            got = [int(i) for i in getattr(LT, fn)(ln, bs, ws, generator=g)]
            assert got == c[fn], fn
        srt = sorted(range(c["n"]), key=lambda i: L[i], reverse=True)[: (c["n"] // ws) * ws]
        assert LT.split_to_even_chunks(srt, L, ws) == c["split_to_even_chunks"]
    # sampler object + rank sharding: disjoint, equal-sized shards
    c = H["samplers"][0]
    s = LT.LengthGroupedSampler(c["batch_size"], c["world_size"], lengths=c["modality_lengths"], group_by_modality=True,
                                generator=torch.Generator().manual_seed(c["seed"]))
    torch.manual_seed(1000 + c["seed"])
    order = list(iter(s))
    assert order == c["get_modality_length_grouped_indices"]
    shards = [LT.shard_for_rank(order, c["batch_size"], 2, r) for r in range(2)]
    assert len(shards[0]) == len(shards[1]) and not set(shards[0]) & set(shards[1])


def test_collator(H):
    c = H["collator"]
    tok = CharTok()
    tok.model_max_length = 12
    inst = [dict(input_ids=torch.tensor([1, 5, -200, 7, 8]), labels=torch.tensor([-100, -100, -100, 7, 8]),
                 image=[(torch.zeros(3, 4, 4), (640, 480), "image")]),
            dict(input_ids=torch.arange(1, 20), labels=torch.arange(1, 20), image=[(torch.zeros(1, 3, 4, 4), (4, 4), "text")]),
            dict(input_ids=torch.tensor([1, 9]), labels=torch.tensor([-100, 9]),
                 image=[(torch.zeros(5, 3, 4, 4), (1000, 700), "image"), (torch.zeros(3, 4, 4), (10, 10), "image")])]
    b = TR.DataCollatorForSupervisedDataset(tokenizer=tok)(inst)
    assert b["input_ids"].tolist() == c["input_ids"] and b["labels"].tolist() == c["labels"]
    assert b["attention_mask"].tolist() == c["attention_mask"]
    assert [list(s) for s in b["image_sizes"]] == c["image_sizes"] and b["modalities"] == c["modalities"]
    assert [list(i.shape) for i in b["images"]] == c["image_shapes"]


def test_preprocessors(H):
    p = H["preprocess"]
    da = SimpleNamespace(is_multimodal=True, mm_use_im_start_end=False)
    pm = TR.preprocess_multimodal(copy.deepcopy(p["conversations"]), da)
    assert pm == p["multimodal"]
    tok = CharTok()
    tok.model_max_length = 2048
    conv_lib.default_conversation = conv_lib.conv_templates["v1"]
    try:
        for conv, want in zip(copy.deepcopy(pm), p["v1"]):
            d = TR.preprocess_v1([conv], tok, has_image=True)
            assert d["input_ids"][0].tolist() == want["input_ids"]
            assert d["labels"][0].tolist() == want["labels"]
            assert any(l != -100 for l in want["labels"])  # the golden case exercises real answer spans
        conv_lib.default_conversation = conv_lib.conv_templates["plain"]
        d = TR.preprocess_plain([copy.deepcopy(pm[1])], tok)
        assert d["input_ids"][0].tolist() == p["plain"]["input_ids"] and d["labels"][0].tolist() == p["plain"]["labels"]
    finally:
        conv_lib.default_conversation = conv_lib.conv_templates["v1"]


def test_llava_json_cells(H):
    from radvlm_amd.data import create_json_cell_llava, generate_llava_dataset_from_instruction_dataset
    j = H["json_cells"]
    ds = SimpleNamespace(pathologies=["Edema", "Effusion"])
    for i, (s, want) in enumerate(zip(j["samples"], j["cells"])):
        assert create_json_cell_llava(s, "pre", 10 + i, ds if i == 1 else None) == want
    cells = generate_llava_dataset_from_instruction_dataset([dict(dataset=j["samples"], id_prefix="x", num_samples=2),
                                                             dict(dataset=j["samples"], id_prefix="y")])
    assert len(cells) == 5 and [c["id"] for c in cells] == ["x_0", "x_1", "y_2", "y_3", "y_4"]
    assert all(c["conversations"][0]["value"].startswith("<image>\n") for c in cells)


def test_image_geometry(H, golden_dir):
    A = np.load(os.path.join(golden_dir, "host_images.npz"))
    for k, g in enumerate(H["image_geometry"]):
        im = img(k, *g["size"])
        rp = MU.resize_and_pad_image(im, tuple(g["target"]))
        assert int(np.asarray(rp, dtype=np.int64).sum()) == g["resize_pad_sum"]
        assert np.array_equal(np.asarray(rp)[::8, ::8], A[f"resize_pad{k}"])
        patches = MU.divide_to_patches(rp, 336)
        assert len(patches) == g["n_patches"] and np.array_equal(np.asarray(patches[-1])[::8, ::8], A[f"patch_last{k}"])
        sq = MU.expand2square(im, (122, 116, 104))
        assert list(sq.size) == g["square_size"] and int(np.asarray(sq, dtype=np.int64).sum()) == g["square_sum"]
    # CLIP preprocessing + anyres tiling: this build's processor vs HF CLIPImageProcessor driven by the reference code
    proc = MU.ClipImageProcessor(336)
    pin = [[336, 672], [672, 336], [672, 672], [1008, 336], [336, 1008]]
    t = MU.process_anyres_image(img(0, 500, 400), proc, pin)
    assert list(t.shape) == H["anyres0_shape"]
    assert np.abs(t.numpy()[:, :, ::16, ::16] - A["anyres0"]).max() < 2e-2   # 8-bit resampling rounding differences only
    assert np.abs(t.numpy().reshape(t.shape[0], -1).mean(1) - A["anyres0_mean"]).max() < 1e-3
    one = proc.preprocess(img(1, 300, 900))["pixel_values"][0]
    assert np.abs(one.numpy()[:, ::8, ::8] - A["clip_pre1"]).max() < 2e-2
    cfg = SimpleNamespace(image_aspect_ratio="pad", image_grid_pinpoints=pin)
    out = MU.process_images([img(2, 200, 100), img(3, 100, 100)], proc, cfg)
    assert tuple(out.shape) == (2, 3, 336, 336)


def test_splice_plan_against_reference(golden_dir):
    """Index plan == the reference's spliced labels / mask for flat and for anyres + spatial_unpad."""
    for name in ("toy_e2e", "toy_anyres_e2e"):
        g = np.load(os.path.join(golden_dir, name + ".npz"))
        meta = json.load(open(os.path.join(golden_dir, name + "_gradnorms.json")))
        n = len([k for k in g.files if k.startswith("image") and k[5:].isdigit()])
        rows, r0 = [], 0
        for i in range(n):
            im = g[f"image{i}"]
            t = 1 if im.ndim == 3 else im.shape[0]
            rows.append(SP.merged_feature_rows(r0, t, 4, meta["merge_type"], meta["aspect"], tuple(g["image_sizes"][i]),
                                               meta["pinpoints"], 56))
            r0 += t * 16
        plan = SP.build_splice_plan(g["input_ids"], g["attention_mask"], g["labels"], rows, r0)
        assert np.array_equal(plan["labels"], g["splice_labels"])
        assert np.array_equal(plan["attention_mask"], g["splice_attention_mask"])
        # every used projector row lands exactly once; CSR covers every text token
        used = plan["feat_pos"][plan["feat_pos"] >= 0]
        assert len(set(used.tolist())) == used.size
        assert plan["tok_off"][-1] == int((plan["idx"] >= 0).sum())
        emb = g["inputs_embeds"]
        pad = ~g["splice_attention_mask"]
        assert np.all(plan["idx"].reshape(pad.shape)[pad] == -1) and float(np.abs(emb[pad]).max()) == 0.0


def test_splice_truncation_and_text_only():
    ids = np.array([[5, -200, 6, 7], [8, 9, 0, 0]])
    mask = np.array([[1, 1, 1, 1], [1, 1, 0, 0]], dtype=bool)
    labels = np.array([[-100, -100, 6, 7], [-100, 9, -100, -100]])
    rows = [np.arange(0, 4), np.arange(4, 8)]  # second sample is text-only: its (dummy) rows stay unused
    plan = SP.build_splice_plan(ids, mask, labels, rows, 8, max_len=5)
    assert plan["S"] == 5 and plan["lens"].tolist() == [5, 2]
    assert plan["idx"].reshape(2, 5).tolist() == [[5, -2, -3, -4, -5], [8, 9, -1, -1, -1]]
    assert plan["labels"].tolist() == [[-100, -100, -100, -100, -100], [-100, 9, -100, -100, -100]]
    assert plan["feat_pos"].tolist() == [1, 2, 3, 4, -1, -1, -1, -1]
    assert SP.shifted_labels(plan["labels"]).tolist() == [[-100] * 5, [9, -100, -100, -100, -100]]


def test_optimizer_groups_and_schedule():
    from radvlm_amd.config import GEOMETRIES
    from radvlm_amd.params import FlatParams, lm_param_shapes
    fp = FlatParams(lm_param_shapes(GEOMETRIES["toy"], True), "cpu")
    eng = SimpleNamespace(lm=fp)
    from radvlm_amd.engine import LlavaEngine
    groups = LlavaEngine.param_groups(eng, lr=1e-3, weight_decay=0.1, mm_projector_lr=5e-4)
    covered = np.zeros(fp.numel, dtype=bool)
    for s, e, lr, wd in groups:
        assert not covered[s:e].any()
        covered[s:e] = True
    for name in fp.names():
        off, n = fp.offsets[name]
        (g,) = [g for g in groups if g[0] <= off and off + n <= g[1]]
        is_norm_or_bias = ("norm" in name and name.endswith("weight")) or name.endswith("bias")
        assert g[3] == (0.0 if is_norm_or_bias else 0.1), name          # llava_trainer.py:369-403: no decay on norms/biases
        assert g[2] == (5e-4 if "mm_projector" in name else 1e-3), name  # separate projector LR
    assert LT.cosine_lr(0, 100, 1.0, 10) == 0.0 and LT.cosine_lr(10, 100, 1.0, 10) == 1.0
    assert abs(LT.cosine_lr(55, 100, 1.0, 10) - 0.5) < 1e-9 and LT.cosine_lr(100, 100, 1.0, 10) < 1e-9


def test_arg_parsing_and_tunable_parts():
    m, d, t = TR.parse_args_into_dataclasses(["--model_name_or_path", "lmsys/vicuna-7b-v1.5", "--version", "v1", "--bf16", "True",
                                              "--mm_projector_type", "mlp2x_gelu", "--mm_vision_select_layer", "-2",
                                              "--per_device_train_batch_size", "4", "--deepspeed", "zero3.json",
                                              "--image_aspect_ratio", "pad", "--learning_rate", "2e-5", "--group_by_modality_length", "True"])
    assert m.version == "v1" and m.mm_vision_select_layer == -2 and d.image_aspect_ratio == "pad"
    assert t.per_device_train_batch_size == 4 and t.deepspeed == "zero3.json" and t.group_by_modality_length is True
    assert TR.tunable_parts(m) == {"mm_mlp_adapter", "mm_language_model"}
    m.mm_tunable_parts = "mm_vision_tower,mm_mlp_adapter,mm_language_model"
    assert "mm_vision_tower" in TR.tunable_parts(m)


def test_preprocess_qwen_and_siglip_processor(golden_dir):
    """SURVEY 8f.1 host functions against the reference's outputs (tests/golden/make_golden_host_qwen.py)."""
    import sys
    sys.path.insert(0, golden_dir)
    import toy_chatml_tokenizer as T
    from radvlm_amd.llava.mm_utils import SigLipImageProcessor
    H = json.load(open(os.path.join(golden_dir, "host_golden_qwen.json")))
    A = np.load(os.path.join(golden_dir, "host_images_qwen.npz"))
    tok = T.build()                                   # plain transformers-5 tokenizer: no 4.x adapter needed on this side
    k = 0
    for c in H["conversations"]:
        for has_image in (True, False):
            want = H["qwen"][k]
            k += 1
            assert want["has_image"] == has_image
            d = TR.preprocess_qwen([copy.deepcopy(c)], tok, has_image=has_image)
            assert d["input_ids"][0].tolist() == want["input_ids"]
            assert d["labels"][0].tolist() == want["labels"]
    assert -200 in H["qwen"][0]["input_ids"] and -200 not in H["qwen"][1]["input_ids"]
    proc = SigLipImageProcessor()
    for i, (w, h) in enumerate([(500, 400), (300, 900), (384, 384)]):
        px = proc.preprocess(img(i, w, h))["pixel_values"][0]
        assert list(px.shape) == H["siglip_shapes"][i]
        assert np.abs(px.numpy()[:, ::8, ::8] - A[f"siglip_pre{i}"]).max() < 1e-5
        assert np.abs(px.numpy().reshape(3, -1).mean(1) - A[f"siglip_pre{i}_mean"]).max() < 1e-5


def test_highres_and_crop_split_image_modes(golden_dir):
    """image_aspect_ratio = 'highres' / 'crop_split' (mm_utils.py:12-118) against the reference's outputs."""
    from types import SimpleNamespace
    from radvlm_amd.llava.mm_utils import ClipImageProcessor, process_highres_image, process_highres_image_crop_split
    H = json.load(open(os.path.join(golden_dir, "host_golden_qwen.json")))
    A = np.load(os.path.join(golden_dir, "host_images_qwen.npz"))
    proc = ClipImageProcessor(112)
    for i, (w, h) in enumerate([(500, 400), (300, 900)]):
        t = process_highres_image(img(i, w, h), proc, "224,336,448")
        assert list(t.shape) == H["highres_shapes"][i]
        assert np.abs(t.numpy()[:, :, ::8, ::8] - A[f"highres{i}"]).max() < 2e-2        # PIL-vs-HF bicubic rounding, as for the CLIP processor
        assert np.abs(t.numpy().reshape(t.shape[0], -1).mean(1) - A[f"highres{i}_mean"]).max() < 2e-3
        da = SimpleNamespace(image_crop_resolution=448, image_split_resolution=112, image_processor=proc)
        t = process_highres_image_crop_split(img(i, w, h), da)
        assert list(t.shape) == H["crop_split_shapes"][i]
        assert np.abs(t.numpy()[:, :, ::8, ::8] - A[f"crop_split{i}"]).max() < 2e-2
        assert np.abs(t.numpy().reshape(t.shape[0], -1).mean(1) - A[f"crop_split{i}_mean"]).max() < 2e-3


def test_onevision_to_hf_converter_contract(golden_dir):
    """the key renaming of the reference's LLaVA-OneVision -> HF converter (its table and function, executed from their source
    lines on this package's Qwen2 + SigLIP key set -> host_convert_keys.json) == radvlm_amd.convert_hf, key for key; the converted names
    are the HF LlavaOnevision parameter names."""
    import json as _json
    import torch
    from oracle import llava_oracle as O
    from radvlm_amd.config import GEOMETRIES
    from radvlm_amd.convert_hf import convert_state_dict_to_hf, hf_key
    g = _json.load(open(os.path.join(golden_dir, "host_convert_keys.json")))
    for k, want in g["mapping"].items():
        assert hf_key(k) == want, (k, hf_key(k), want)
    for k in g["dropped"]:
        assert hf_key(k) is None
    keys = list(O.param_shapes(GEOMETRIES["toy_qwen"], with_newline=True))
    assert set(keys) == set(g["mapping"])                               # the fixture covers every tensor this package writes
    assert g["mapping"]["model.image_newline"] == "image_newline"
    assert g["mapping"]["model.mm_projector.2.bias"] == "multi_modal_projector.linear_2.bias"
    assert g["mapping"]["model.vision_tower.vision_tower.vision_model.encoder.layers.1.mlp.fc1.weight"] == "vision_tower.vision_model.encoder.layers.1.mlp.fc1.weight"
    sd = {k: torch.ones(1) for k in keys if k != "lm_head.weight"}      # tied head: cloned from the embeddings like the reference does
    out = convert_state_dict_to_hf(sd, dtype=torch.float16)
    assert "language_model.lm_head.weight" in out and all(v.dtype == torch.float16 for v in out.values()) and len(out) == len(keys)


def test_saved_checkpoint_is_consumable_by_the_converter(golden_dir, tmp_path):
    """A directory written by save_pretrained holds what convert_llava_onevision_weights_to_hf.py reads: *.safetensors under LLaVA names that
    convert onto the HF layout, and config.json naming the tower (no GPU needed: the engine's stores live on the CPU here)."""
    import json as _json
    from radvlm_amd.config import GEOMETRIES
    from radvlm_amd.convert_hf import check_checkpoint_dir
    from radvlm_amd.llava.model import LlavaQwenConfig, LlavaQwenForCausalLM
    model = LlavaQwenForCausalLM(LlavaQwenConfig(geometry=GEOMETRIES["toy_qwen"], mm_patch_merge_type="spatial_unpad"), device="cpu", init="portable")
    with pytest.raises(KeyError, match="mm_vision_tower"):
        model.save_pretrained(str(tmp_path / "a"))
        check_checkpoint_dir(str(tmp_path / "a"))
    model.config.mm_vision_tower = "google/siglip-so400m-patch14-384"
    model.save_pretrained(str(tmp_path / "b"))
    got = check_checkpoint_dir(str(tmp_path / "b"))
    want = _json.load(open(os.path.join(golden_dir, "host_convert_keys.json")))["mapping"]
    assert sorted(got) == sorted(want.values())


def test_instruction_generators_replay_the_reference(golden_dir):
    """radvlm_amd.data.create_instructions against outputs of the reference's own generators (radvlm/data/create_instructions.py:9-26,
    :120-529; fixture from tests/golden/make_golden_instructions.py): same text for the same `random` seed, and the generator left in
    the same state (the next `random.random()` agrees), i.e. the draws happen in the reference's order."""
    import random
    from radvlm_amd.data import create_instructions as ci
    G = json.load(open(os.path.join(golden_dir, "instruction_generators.json"), encoding="utf-8"))
    n = 0
    for fn, recs in G.items():
        if fn == "grouped_length_mismatch_error":
            continue
        for r in recs:
            random.seed(r["seed"])
            got = getattr(ci, fn)(*r["args"], **r["kwargs"])
            assert got == r["result"], (fn, r["args"], r["seed"], got, r["result"])
            assert random.random() == r["next_random"], (fn, r["seed"])
            n += 1
    assert n > 300
    with pytest.raises(ValueError, match=G["grouped_length_mismatch_error"][:20]):
        ci.generate_instruction_abnormalities_grouped([[0, 0, 1, 1]], ["a", "b"])
    with pytest.raises(IndexError):
        ci.format_boxes([])


def test_dropin_radvlm_package_extends_instead_of_shadowing(tmp_path):
    """`dropin/radvlm` must not hide the reference's own `radvlm` package (INTEGRATION.md section 1 puts dropin/ first on PYTHONPATH):
    it is a namespace extension -- submodules this build does not restate (radvlm.data.datasets, radvlm.evaluation, ...) still resolve
    to whatever `radvlm` tree follows on the path.  Checked in a child process against a stand-in tree."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    other = tmp_path / "site"
    (other / "radvlm" / "data").mkdir(parents=True)
    (other / "radvlm" / "evaluation").mkdir()
    (other / "radvlm" / "__init__.py").write_text("raise RuntimeError('the stand-in package __init__ must not be needed')\n")
    (other / "radvlm" / "data" / "__init__.py").write_text("")
    # like the reference's datasets.py:24 / :1088: names such as defaultdict reach it through the star import of create_instructions
    (other / "radvlm" / "data" / "utils.py").write_text("def custom_collate_fn(batch):\n    return batch\n")
    (other / "radvlm" / "data" / "datasets.py").write_text(
        "from radvlm.data.create_instructions import *\nMARK = 'datasets of the other tree'\n"
        "GROUPS = defaultdict(list)\nGROUPS['a'].append(Counter('aab')['a'] + int(np.int64(1)) + len(custom_collate_fn([1])))\n"
        "assert DataLoader is not None and random.random() < 1 and GROUPS['a'] == [4]\n")
    (other / "radvlm" / "evaluation" / "__init__.py").write_text("MARK = 'evaluation of the other tree'\n")
    code = ("import radvlm.data.datasets as d, radvlm.evaluation as e, radvlm.data.create_instructions as c, radvlm.data as rd;"
            "from radvlm import DATA_DIR;"
            "print(d.MARK, '|', e.MARK, '|', c.__name__, '|', callable(rd.create_json_cell_llava), '|', DATA_DIR)")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([root, os.path.join(root, "dropin"), str(other)]), DATA_DIR="/data/cxr")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert p.returncode == 0, p.stderr[-800:]
    assert p.stdout.strip() == "datasets of the other tree | evaluation of the other tree | radvlm.data.create_instructions | True | /data/cxr"
    env.pop("DATA_DIR")
    p = subprocess.run([sys.executable, "-c", "from radvlm import DATA_DIR"], capture_output=True, text=True, env=env)
    assert p.returncode != 0 and "DATA_DIR" in p.stderr          # same error behaviour as the reference's radvlm/__init__.py:5-7
