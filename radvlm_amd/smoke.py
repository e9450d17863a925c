"""smoke(): one tiny LLaVA training step through the HIP path on cuda:0, checked against the CPU oracle."""
import os

import numpy as np
import torch


def load_golden_batch(name="toy_e2e"):
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    g = np.load(os.path.join(here, name + ".npz"))
    n = len([k for k in g.files if k.startswith("image") and k[5:].isdigit()])
    images = [torch.from_numpy(g[f"image{i}"]) for i in range(n)]
    return g, images


def run_smoke():
    from radvlm_amd import lib
    from radvlm_amd.config import GEOMETRIES
    from radvlm_amd.engine import LlavaEngine
    lib.load()  # fails loudly if the HIP library is missing
    assert torch.cuda.is_available(), "smoke() needs cuda:0"
    torch.cuda.set_device(0)
    geo = GEOMETRIES["toy"]
    g, images = load_golden_batch()
    eng = LlavaEngine(geo, device="cuda:0", init="portable", seed=0)
    loss = eng.forward(g["input_ids"], g["attention_mask"], g["labels"], images, want_logits=True)
    eng.backward()
    eng.optimizer_step(lr=1e-3, weight_decay=0.0, max_grad_norm=1.0)
    torch.cuda.synchronize()
    # checker: CPU oracle on the same inputs (test infrastructure, not the product path)
    from oracle import llava_oracle as O
    P = O.make_params(geo, seed=0)
    ref_loss, ref_logits, _ = O.llava_forward(P, geo, torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"]),
                                              torch.from_numpy(g["labels"]), images)
    m = torch.from_numpy(g["splice_attention_mask"])
    got = eng.last_logits.cpu()[m]
    ref = ref_logits.detach()[m]
    err = float((got - ref).abs().max() / ref.abs().max())
    dl = abs(float(loss) - float(ref_loss))
    print(f"smoke: loss {float(loss):.5f} (oracle {float(ref_loss):.5f}), logits rel-inf err {err:.3e}")
    assert dl < 2e-2 and err < 5e-2, (dl, err)
    assert bool(torch.isfinite(eng.lm.flat.float()).all())
