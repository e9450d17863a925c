"""Golden vectors for the Qwen-side host functions, from the REFERENCE (build container only):
preprocess_qwen (train/train.py:560-633, run from its source lines with the 4.x-API adapter of toy_chatml_tokenizer.Tok4)
and SigLipImageProcessor.preprocess (multimodal_encoder/siglip_encoder.py:34-67).  Writes host_golden_qwen.json/.npz."""
import copy
import json
import os
import sys
from typing import Dict, Sequence

import numpy as np
import torch
import transformers

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
import toy_chatml_tokenizer as T  # noqa: E402
from make_golden_host import exec_slice, img, REF  # noqa: E402


def main():
    ref_shim.load_reference()
    ns = dict(torch=torch, transformers=transformers, Dict=Dict, Sequence=Sequence, copy=copy, IGNORE_INDEX=-100, IMAGE_TOKEN_INDEX=-200)
    exec_slice(REF + "/finetuning/llava/train/train.py", 560, 633, ns)
    tok = T.build(T.Tok4)
    convs = [[{"from": "human", "value": "<image>\nWhat is shown?"}, {"from": "gpt", "value": "A chest X-ray."},
              {"from": "human", "value": "Any finding?\nBe brief."}, {"from": "gpt", "value": "No acute disease."}],
             [{"from": "gpt", "value": "(dropped leading answer)"}, {"from": "human", "value": "Describe. <image>"},
              {"from": "gpt", "value": "Cardiomegaly."}]]
    out = {"conversations": convs, "qwen": []}
    for c in convs:
        for has_image in (True, False):
            d = ns["preprocess_qwen"]([copy.deepcopy(c)], tok, has_image=has_image)
            out["qwen"].append(dict(has_image=has_image, input_ids=d["input_ids"][0].tolist(), labels=d["labels"][0].tolist()))
    import importlib
    se = importlib.import_module("llava.model.multimodal_encoder.siglip_encoder")
    proc = se.SigLipImageProcessor()
    arrs = {}
    for k, (w, h) in enumerate([(500, 400), (300, 900), (384, 384)]):
        px = proc.preprocess(img(k, w, h), return_tensors="pt")["pixel_values"][0]
        arrs[f"siglip_pre{k}"] = px.numpy()[:, ::8, ::8].copy()
        arrs[f"siglip_pre{k}_mean"] = px.numpy().reshape(3, -1).mean(1)
        out.setdefault("siglip_shapes", []).append(list(px.shape))
    # highres / crop_split image modes (mm_utils.py:12-118) through the reference functions; Pillow >= 10 renamed the
    # ANTIALIAS filter the reference names to LANCZOS (same filter): alias it for the run
    from PIL import Image as _I
    if not hasattr(_I, "ANTIALIAS"):
        _I.ANTIALIAS = _I.LANCZOS
    mmu = importlib.import_module("llava.mm_utils")
    from types import SimpleNamespace
    from transformers import CLIPImageProcessor
    hf = CLIPImageProcessor(size={"shortest_edge": 112}, crop_size={"height": 112, "width": 112})

    class Adapter:
        size = {"shortest_edge": 112}
        crop_size = {"height": 112, "width": 112}
        image_mean = hf.image_mean

        def preprocess(self, image, return_tensors="pt"):
            return hf.preprocess(image, return_tensors=return_tensors)

    out["highres_shapes"], out["crop_split_shapes"] = [], []
    for k, (w, h) in enumerate([(500, 400), (300, 900)]):
        t = mmu.process_highres_image(img(k, w, h), Adapter(), "224,336,448")
        arrs[f"highres{k}"] = t.numpy()[:, :, ::8, ::8].copy()
        arrs[f"highres{k}_mean"] = t.numpy().reshape(t.shape[0], -1).mean(1)
        out["highres_shapes"].append(list(t.shape))
        da = SimpleNamespace(image_crop_resolution=448, image_split_resolution=112, image_processor=Adapter())
        t = mmu.process_highres_image_crop_split(img(k, w, h), da)
        arrs[f"crop_split{k}"] = t.numpy()[:, :, ::8, ::8].copy()
        arrs[f"crop_split{k}_mean"] = t.numpy().reshape(t.shape[0], -1).mean(1)
        out["crop_split_shapes"].append(list(t.shape))
    with open(os.path.join(HERE, "host_golden_qwen.json"), "w") as f:
        json.dump(out, f)
    np.savez_compressed(os.path.join(HERE, "host_images_qwen.npz"), **arrs)
    print("written", [len(q["input_ids"]) for q in out["qwen"]], out["siglip_shapes"], out["highres_shapes"], out["crop_split_shapes"])


if __name__ == "__main__":
    main()
