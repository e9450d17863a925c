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
    with open(os.path.join(HERE, "host_golden_qwen.json"), "w") as f:
        json.dump(out, f)
    np.savez_compressed(os.path.join(HERE, "host_images_qwen.npz"), **arrs)
    print("written", [len(q["input_ids"]) for q in out["qwen"]], out["siglip_shapes"])


if __name__ == "__main__":
    main()
