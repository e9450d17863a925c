"""Host input pipeline rate, no GPU needed: samples/s of LazySupervisedDataset + collator through the
BatchPrefetcher at N worker threads, on synthetic 1024x1024 PNG radiographs, for the LLaVA-1.5 'pad' path (336 px) and the
anyres_max_9 SigLIP path (384 px, 10 tiles).  The GPU consumes ~35 samples/s (7B, 336 px) or ~2/s (RadVLM recipe) per GPU."""
import json
import os
import sys
import tempfile
import time

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radvlm_amd.llava import conversation as conv_lib  # noqa: E402
from radvlm_amd.llava.mm_utils import ClipImageProcessor, SigLipImageProcessor  # noqa: E402
from radvlm_amd.llava.train.llava_trainer import BatchPrefetcher  # noqa: E402
from radvlm_amd.llava.train.train import DataArguments, make_supervised_data_module  # noqa: E402


class Ids:
    def __init__(self, ids):
        self.input_ids = ids


class Tok:
    bos_token_id, pad_token_id, model_max_length, legacy, padding_side = 1, 0, 2048, True, "right"

    def __call__(self, s, **kw):
        ids = [1]
        for k, piece in enumerate(s.split("</s>")):
            if k:
                ids.append(2)
            ids.extend(3 + (ord(c) % 900) for c in piece)
        return Ids(ids)


def main():
    n = 64
    tmp = tempfile.mkdtemp(prefix="rv_host_")
    rng = np.random.default_rng(0)
    recs = []
    for i in range(n):
        Image.fromarray(rng.integers(0, 255, (1024, 1024), dtype=np.uint8)).save(os.path.join(tmp, f"im{i}.png"))
        recs.append({"id": i, "image": f"im{i}.png", "conversations": [{"from": "human", "value": "<image>\nDescribe the radiograph."},
                                                                       {"from": "gpt", "value": "No acute cardiopulmonary disease."}]})
    json.dump(recs, open(os.path.join(tmp, "d.json"), "w"))
    conv_lib.default_conversation = conv_lib.conv_templates["v1"]
    out = {}
    def u8(proc):      # device-side normalisation: the host stops after resize / crop / pad and hands uint8 pixels over
        proc.device_normalize = True
        return proc
    for name, proc, aspect, pin in (("llava15_pad_336", ClipImageProcessor(336), "pad", None),
                                    ("llava15_pad_336_u8", u8(ClipImageProcessor(336)), "pad", None),
                                    ("radvlm_anyres_max_9_384", SigLipImageProcessor(), "anyres_max_9", "(1x1),...,(6x6)"),
                                    ("radvlm_anyres_max_9_384_u8", u8(SigLipImageProcessor()), "anyres_max_9", "(1x1),...,(6x6)")):
        da = DataArguments(data_path=os.path.join(tmp, "d.json"), image_folder=tmp, image_aspect_ratio=aspect, image_grid_pinpoints=pin,
                           is_multimodal=True)
        da.image_processor, da.mm_use_im_start_end = proc, False
        mod = make_supervised_data_module(tokenizer=Tok(), data_args=da)
        for workers in (0, 8, 16):
            batches = [list(range(i, i + 8)) for i in range(0, n, 8)] * 2
            p = BatchPrefetcher(mod["train_dataset"], mod["data_collator"], batches, num_workers=workers, pin=False)
            t = time.perf_counter()
            k = sum(len(b["images"]) for b in p)
            dt = time.perf_counter() - t
            p.close()
            out[f"{name}/workers={workers}"] = round(k / dt, 1)
            print(f"{name:28s} workers={workers:2d}: {k / dt:7.1f} samples/s", flush=True)
    out["cores"] = os.cpu_count()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
