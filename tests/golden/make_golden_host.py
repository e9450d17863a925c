"""Golden vectors for the HOST-side rows of the hot path (samplers, collator, preprocessors, image geometry,
LLaVA-JSON cells), produced by running the reference's own functions (build container only; needs /root/reference).

The reference modules that hold these functions cannot be imported whole here (absent third-party packages:
deepspeed, trl, openai, ... -- SURVEY.md section 8c), so the pure functions are executed from their source line
ranges, in a namespace that provides the few names they use.  Nothing of the reference's text is stored: only
inputs and outputs go to tests/golden/host_golden.json / host_images.npz.
"""
import json
import os
import sys
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

REF = "/root/reference"


def exec_slice(path, first, last, ns):
    with open(path) as f:
        lines = f.readlines()[first - 1:last]
    exec(compile("".join(lines), path, "exec"), ns)
    return ns


class Ids:
    def __init__(self, ids):
        self.input_ids = ids


class CharTok:
    """Character-level stand-in tokenizer (BOS=1, ids = ord)."""
    bos_token_id = 1
    pad_token_id = 0
    model_max_length = 512
    legacy = True
    padding_side = "right"

    def __call__(self, s, **kw):
        ids = [1]
        for k, piece in enumerate(s.split("</s>")):   # "</s>" is one token (id 2), like a sentencepiece EOS
            if k:
                ids.append(2)
            ids.extend(ord(c) for c in piece)
        return Ids(ids)


def img(seed, w, h):
    rng = np.random.default_rng(seed)
    # smooth-ish random image so that resampling differences are meaningful but not noise-dominated
    base = rng.integers(0, 255, size=(h // 8 + 2, w // 8 + 2, 3), dtype=np.uint8)
    return Image.fromarray(base).resize((w, h), Image.BICUBIC)


def main():
    mods = ref_shim.load_reference()
    out = {}
    # ---------------- samplers (llava_trainer.py:51-237)
    from transformers.trainer_pt_utils import get_length_grouped_indices as hf_lgi
    from torch.utils.data import Sampler
    ns = exec_slice(REF + "/finetuning/llava/train/llava_trainer.py", 51, 237,
                    dict(torch=torch, Sampler=Sampler, List=List, Optional=Optional, get_length_grouped_indices_hf=hf_lgi))
    rng = np.random.default_rng(5)
    cases = []
    for n, bs, ws, seed in [(64, 4, 2, 0), (50, 3, 4, 1), (33, 2, 2, 7), (128, 8, 2, 3)]:
        lengths = rng.integers(5, 400, size=n).tolist()
        mod = [(l if i % 3 else -l) for i, l in enumerate(lengths)]
        rec = dict(n=n, batch_size=bs, world_size=ws, seed=seed, lengths=lengths, modality_lengths=mod)
        for fn, ln in (("get_length_grouped_indices", lengths), ("get_length_grouped_indices_auto_single", lengths),
                       ("get_variable_length_grouped_indices", lengths), ("get_modality_length_grouped_indices", mod),
                       ("get_modality_length_grouped_indices_auto", mod)):
            g = torch.Generator().manual_seed(seed)
            torch.manual_seed(1000 + seed)  # the modality variants draw their inner permutations from the GLOBAL generator
            rec[fn] = [int(i) for i in ns[fn](ln, bs, ws, generator=g)]
        srt = sorted(range(n), key=lambda i: lengths[i], reverse=True)[: (n // ws) * ws]
        rec["split_to_even_chunks"] = ns["split_to_even_chunks"](srt, lengths, ws)
        cases.append(rec)
    out["samplers"] = cases
    # ---------------- collator (train.py:1243-1286)
    import transformers
    ns = exec_slice(REF + "/finetuning/llava/train/train.py", 1242, 1286,
                    dict(dataclass=dataclass, transformers=transformers, torch=torch, Sequence=Sequence, Dict=Dict, IGNORE_INDEX=-100))
    tok = CharTok()
    tok.model_max_length = 12
    inst = [dict(input_ids=torch.tensor([1, 5, -200, 7, 8]), labels=torch.tensor([-100, -100, -100, 7, 8]),
                 image=[(torch.zeros(3, 4, 4), (640, 480), "image")]),
            dict(input_ids=torch.arange(1, 20), labels=torch.arange(1, 20), image=[(torch.zeros(1, 3, 4, 4), (4, 4), "text")]),
            dict(input_ids=torch.tensor([1, 9]), labels=torch.tensor([-100, 9]),
                 image=[(torch.zeros(5, 3, 4, 4), (1000, 700), "image"), (torch.zeros(3, 4, 4), (10, 10), "image")])]
    b = ns["DataCollatorForSupervisedDataset"](tokenizer=tok)(inst)
    out["collator"] = dict(input_ids=b["input_ids"].tolist(), labels=b["labels"].tolist(), attention_mask=b["attention_mask"].tolist(),
                           image_sizes=[list(s) for s in b["image_sizes"]], modalities=b["modalities"],
                           image_shapes=[list(i.shape) for i in b["images"]])
    # ---------------- preprocess_multimodal / preprocess_v1 / preprocess_plain (train.py:378-403, 722-798, 882-901)
    import copy
    import re
    from radvlm_amd.llava import conversation as conv_lib  # templates are this build's restated strings
    mm = mods["mm_utils"]
    base = dict(torch=torch, transformers=transformers, Dict=Dict, Sequence=Sequence, copy=copy, re=re, conversation_lib=conv_lib,
                tokenizer_image_token=mm.tokenizer_image_token, IGNORE_INDEX=-100, DEFAULT_IMAGE_TOKEN="<image>",
                DEFAULT_IM_START_TOKEN="<im_start>", DEFAULT_IM_END_TOKEN="<im_end>", IS_TOKENIZER_GREATER_THAN_0_14=True,
                DataArguments=object)
    path = REF + "/finetuning/llava/train/train.py"
    ns = dict(base)
    exec_slice(path, 378, 403, ns)
    exec_slice(path, 722, 798, ns)
    exec_slice(path, 882, 901, ns)
    convs = [[{"from": "human", "value": "What is shown? <image>"}, {"from": "gpt", "value": "A chest X-ray."},
              {"from": "human", "value": "Any finding?"}, {"from": "gpt", "value": "No acute disease."}],
             [{"from": "human", "value": "<image>\nDescribe."}, {"from": "gpt", "value": "Cardiomegaly is present."}]]
    from types import SimpleNamespace
    da = SimpleNamespace(is_multimodal=True, mm_use_im_start_end=False)
    pm = ns["preprocess_multimodal"](copy.deepcopy(convs), da)
    tok = CharTok()
    tok.model_max_length = 2048
    conv_lib.default_conversation = conv_lib.conv_templates["v1"]
    v1 = [ns["preprocess_v1"]([c], tok, has_image=True) for c in copy.deepcopy(pm)]
    conv_lib.default_conversation = conv_lib.conv_templates["plain"]
    plain = ns["preprocess_plain"]([copy.deepcopy(pm[1])], tok)
    conv_lib.default_conversation = conv_lib.conv_templates["v1"]
    out["preprocess"] = dict(conversations=convs, multimodal=pm,
                             v1=[dict(input_ids=d["input_ids"][0].tolist(), labels=d["labels"][0].tolist()) for d in v1],
                             plain=dict(input_ids=plain["input_ids"][0].tolist(), labels=plain["labels"][0].tolist()))
    # ---------------- LLaVA JSON cells (radvlm/data/create_instructions.py:29-71)
    ns = exec_slice(REF + "/radvlm/data/create_instructions.py", 29, 71, {})
    samples = [dict(img_path="a/b.jpg", instr=dict(question="Q1?", answer="A1.")),
               dict(img_path="c.jpg", conversation=[{"from": "human", "value": "h1"}, {"from": "gpt", "value": "g1"},
                                                    {"from": "human", "value": "h2"}, {"from": "gpt", "value": "g2"}], labels=[0, 1]),
               dict(img_path="d.jpg", conversation=[dict(question="q1", answer="a1"), dict(question="q2", answer="a2")])]
    ds = SimpleNamespace(pathologies=["Edema", "Effusion"])
    out["json_cells"] = dict(samples=samples, cells=[ns["create_json_cell_llava"](s, "pre", 10 + i, ds if i == 1 else None)
                                                     for i, s in enumerate(samples)])
    # ---------------- image geometry (mm_utils.py:152-311) through the reference functions
    arrs = {}
    geo = []
    for k, (w, h, res) in enumerate([(500, 400, (672, 672)), (300, 900, (336, 1008)), (1000, 250, (1008, 336)), (336, 336, (672, 336))]):
        im = img(k, w, h)
        rp = mm.resize_and_pad_image(im, res)
        arrs[f"resize_pad{k}"] = np.asarray(rp)[::8, ::8].copy()
        patches = mm.divide_to_patches(rp, 336)
        arrs[f"patch_last{k}"] = np.asarray(patches[-1])[::8, ::8].copy()
        sq = mm.expand2square(im, (122, 116, 104))
        arrs[f"square{k}"] = np.asarray(sq)[::8, ::8].copy()
        geo.append(dict(size=[w, h], target=list(res), n_patches=len(patches), square_size=list(sq.size),
                        resize_pad_sum=int(np.asarray(rp, dtype=np.int64).sum()), square_sum=int(np.asarray(sq, dtype=np.int64).sum())))
    out["image_geometry"] = geo
    # anyres end-to-end + CLIP preprocessing with the HF processor (PIL backend) behind the size-dict adapter
    from transformers import CLIPImageProcessor
    hf = CLIPImageProcessor(size={"shortest_edge": 336}, crop_size={"height": 336, "width": 336})

    class Adapter:
        size = {"shortest_edge": 336}
        crop_size = {"height": 336, "width": 336}
        image_mean = hf.image_mean

        def preprocess(self, image, return_tensors="pt"):
            return hf.preprocess(image, return_tensors=return_tensors)

    pin = [[336, 672], [672, 336], [672, 672], [1008, 336], [336, 1008]]
    t = mm.process_anyres_image(img(0, 500, 400), Adapter(), pin)
    arrs["anyres0"] = t.numpy().astype(np.float32)[:, :, ::16, ::16].copy()
    arrs["anyres0_mean"] = t.numpy().reshape(t.shape[0], -1).mean(1)
    out["anyres0_shape"] = list(t.shape)
    one = hf.preprocess(img(1, 300, 900), return_tensors="pt")["pixel_values"][0]
    arrs["clip_pre1"] = one.numpy()[:, ::8, ::8].copy()
    arrs["clip_pre1_mean"] = one.numpy().reshape(3, -1).mean(1)
    with open(os.path.join(HERE, "host_golden.json"), "w") as f:
        json.dump(out, f)
    np.savez_compressed(os.path.join(HERE, "host_images.npz"), **arrs)
    print("host golden written:", list(out))


if __name__ == "__main__":
    main()
