"""One 32k-token sample through the full 7B geometry (the reference recipe's --model_max_length 32768, finetune_radio_7b.sh:79) with
recompute="auto": how many decoder layers shed their activations, peak memory, step time.  python tools/long_sample_probe.py [tokens]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from radvlm_amd.config import GEOMETRIES
from radvlm_amd.engine import LlavaEngine

T = int(sys.argv[1]) if len(sys.argv) > 1 else 32768 - 575
geo = GEOMETRIES["llava15_7b"]
eng = LlavaEngine(geo, device="cuda:0", init="fast", seed=0, recompute="auto")
eng.init_optimizer()
rng = np.random.default_rng(0)
ids = rng.integers(3, geo["lm"]["vocab"], size=(1, T), dtype=np.int64)
labels = ids.copy(); labels[:, :64] = -100
ids[:, 35] = -200; labels[:, 35] = -100
mask = np.ones_like(ids, dtype=bool)
images = [torch.randn(3, 336, 336).to(torch.bfloat16)]
for step in range(3):
    torch.cuda.reset_peak_memory_stats()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss = eng.forward(ids, mask, labels, images)
    n_re, S = eng.ctx["n_recomputed"], eng.ctx["S"]
    eng.backward()
    eng.optimizer_step(lr=2e-5, weight_decay=0.0, max_grad_norm=1.0)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"step {step}: S = {S}, loss {float(loss):.4f}, {n_re} of {geo['lm']['layers']} layers recomputed, peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB, "
          f"{dt * 1e3:.0f} ms ({S / dt:.0f} tokens/s)", flush=True)
