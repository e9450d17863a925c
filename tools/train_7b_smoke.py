"""The trainer entry at full scale (python tools/train_7b_smoke.py, on the GPU box): train() on the LLaVA-1.5-7B geometry (random
init: --geometry), synthetic 336-px images, a few optimizer steps with gradient accumulation, periodic checkpoint, final
save_pretrained, and a second run that resumes from the checkpoint -- the memory plan and the file formats at 7B, not parity
(parity is tested at toy / config-1 sizes)."""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from PIL import Image
from test_train_features_gpu import Tok
from radvlm_amd.llava import conversation as conv_lib
from radvlm_amd.llava.train.train import train

tmp = tempfile.mkdtemp(prefix="rv7b_")
rng = np.random.default_rng(0)
recs = []
for i in range(64):
    Image.fromarray(rng.integers(0, 255, (400, 360, 3), dtype=np.uint8)).save(os.path.join(tmp, f"im{i}.png"))
    recs.append({"id": f"s{i}", "image": f"im{i}.png", "conversations": [{"from": "human", "value": "<image>\nDescribe the radiograph."},
                                                                        {"from": "gpt", "value": "No acute cardiopulmonary disease. " * (1 + i % 3)}]})
json.dump(recs, open(os.path.join(tmp, "d.json"), "w"))
conv_lib.default_conversation = conv_lib.conv_templates["v1"]
tok = Tok(); tok.model_max_length = 2048
common = ["--data_path", os.path.join(tmp, "d.json"), "--image_folder", tmp, "--image_aspect_ratio", "pad", "--version", "v1", "--geometry", "llava15_7b",
          "--per_device_train_batch_size", "8", "--gradient_accumulation_steps", "2", "--learning_rate", "2e-5", "--warmup_ratio", "0.03",
          "--mm_projector_type", "mlp2x_gelu", "--mm_vision_select_layer", "-2", "--mm_use_im_patch_token", "False", "--model_max_length", "2048",
          "--dataloader_num_workers", "4", "--logging_steps", "1", "--save_steps", "2", "--save_only_model", "True", "--output_dir", os.path.join(tmp, "out")]
t = time.time()
st = train(argv=common + ["--max_steps", "3"], tokenizer=tok)
print("run 1:", {k: st[k] for k in st if k in ("global_step", "log")} if isinstance(st, dict) else st, f"{time.time() - t:.0f} s", flush=True)
print("files:", sorted((f, os.path.getsize(os.path.join(tmp, "out", f)) >> 20) for f in os.listdir(os.path.join(tmp, "out")) if os.path.isfile(os.path.join(tmp, "out", f))))
print("peak GPU memory GiB:", torch.cuda.max_memory_allocated() >> 30, flush=True)
t = time.time()
st = train(argv=common + ["--max_steps", "4"], tokenizer=tok)          # resumes from checkpoint-2
print("run 2 (resumed):", {k: st[k] for k in st if k in ("global_step", "log")} if isinstance(st, dict) else st, f"{time.time() - t:.0f} s", flush=True)
import shutil; shutil.rmtree(tmp, ignore_errors=True)
