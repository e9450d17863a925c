"""End-to-end GPU parity (-m gpu): the HIP engine (bf16, through the C ABI) against golden vectors of the reference
(tests/golden/*.npz, CPU fp32) on the same synthetic inputs and portable weights.

Tolerances (bf16 compute vs an fp32 reference; stated per check):
  loss |d| <= 5e-3 (toy) / 2e-2 (config 1: 24 bf16 layers, 32000-way bf16 logits); logits ||d||_inf / ||ref||_inf <= 3e-2 over valid positions;
  per-parameter gradient: ||d||_2 / ||ref||_2 <= 5e-2 on the fully stored tensors, norms within 5 %.
"""
import json
import os

import numpy as np
import pytest
import torch

from radvlm_amd.config import GEOMETRIES

pytestmark = pytest.mark.gpu


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


def _check(eng, g, meta, loss, logits, plan, full=True, loss_tol=5e-3):
    assert np.array_equal(plan["labels"], g["splice_labels"])
    assert np.array_equal(plan["attention_mask"], g["splice_attention_mask"])
    assert abs(loss - float(g["loss"])) < loss_tol, (loss, float(g["loss"]))
    m = torch.from_numpy(g["splice_attention_mask"])
    if full:
        ref = torch.from_numpy(g["logits"])
        err = float((logits[m] - ref[m]).abs().max() / ref[m].abs().max())
        assert err < 3e-2, err
    worst = 0.0
    for k, want in meta["grad_norms"].items():
        if want is None:
            continue
        got = float(eng.G(k).float().norm())
        rel = abs(got - want) / max(want, 1e-6)
        worst = max(worst, rel)
        assert rel < 5e-2, (k, got, want)
    for k in g.files:
        if k.startswith("grad::"):
            ref = torch.from_numpy(g[k])
            got = eng.G(k[6:]).float().cpu()
            rel = float((got - ref).norm() / ref.norm())
            assert rel < 5e-2, (k, rel)
    return worst


def test_toy_forward_backward(golden_dir):
    g, meta, images = _golden(golden_dir, "toy_e2e")
    eng = _engine("toy")
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


def test_config1(golden_dir):
    g, meta, images = _golden(golden_dir, "config1_e2e")
    eng = _engine("config1")
    loss, logits, plan = _run(eng, g, images)
    _check(eng, g, meta, loss, logits, plan, full=False, loss_tol=2e-2)  # 12+12 layers, V=32000, bf16 logits
    ref = torch.from_numpy(g["logits_slice"])
    got = logits[:, ::7, ::997]
    assert float((got - ref).abs().max() / float(g["logits_absmax"].max())) < 3e-2


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
