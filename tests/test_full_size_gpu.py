"""Full-size BASELINE config-2 parity (-m gpu): the WHOLE LLaVA-1.5-7B model (24-layer ViT-L/14-336 tower, projector, 32 full-width
decoder layers, 32000-way head; 6.76 B trainable parameters) through forward, backward and two AdamW steps on the HIP engine, against
the CPU oracle in fp32 on the same weights and inputs (b = 1, one 336 px image + 129 ids -> S = 704).

Why it exists: toy / one-layer parity cannot see whole-model faults.  Round 2's 32-bit work-item overflow zeroed every parameter behind
decoder layer 12 from the second optimizer step on and no parity test noticed.  Here every one of the 291 trainable
tensors is compared after the backward (gradients) and after the second optimizer step (fp32 master copy and bf16 parameters).

Reference text followed by the oracle: language_model/modeling_llama.py:1083-1185 (model), :1304-1337 (loss), llava_arch.py:251-555
(splice), clip_encoder.py:46-79, multimodal_projector/builder.py:41-48; optimizer = torch.optim.AdamW semantics (optim adamw_torch,
train/train.py:140) with the global-norm clip of HF Trainer (max_grad_norm) and the decay / no-decay groups of
LLaVATrainer.create_optimizer (train/llava_trainer.py:369-418).

Host memory: working weights + gradients + fp32 masters + two AdamW moments of 6.76 B parameters = 135 GB (the GPU box allows ~270 GiB).
If the host offers less than 180 GB the test runs 8 full-width layers + head instead and says so in its record (RV_FULLSIZE_LAYERS
overrides).
"""
import copy
import json
import math
import os
import time

import pytest
import torch

from radvlm_amd.config import GEOMETRIES

pytestmark = pytest.mark.gpu

# Gates: <= 1.25 x the values measured on MI355X (profiles/r03_full_size_parity.json holds the measurements)
LOSS_TOL = 1e-2               # |loss_hip - loss_fp32|, both steps
GRAD_REL_L2 = 0.22            # per tensor ||g_hip - g_fp32||_2 / ||g_fp32||_2: measured 0.066 typical, 0.13 (step 1) / 0.174 (step 2) worst
                              # (q/k projections of layer 30) -- the bf16 noise of a 32-layer random-init model, whose logits the
                              # bf16-EMULATING oracle itself misses by 5.7 % (emu_vs_fp32_l2); kernel errors proper are gated per op in
                              # tests/test_backward_parity_gpu.py
GRAD_NORM_REL = 5e-3          # global gradient norm, step 1 (measured 9e-5 on the 7B model, 1.05e-3 on the recipe model)
GRAD_NORM_REL_STEP2 = 1e-2    # step 2: the two models have taken one AdamW step each (first step ~ lr * sign(g): gradient elements inside the bf16
                              # noise take opposite signs).  Measured: 1.4e-3 (7B); recipe model 3.2e-3 with the head on every row and 5.3e-3 with the
                              # head on the labelled rows -- two equally valid bf16 realisations of the SAME kernels (the run with round 3's kernel
                              # library gives the same digits; the gathered-row lm_head GEMMs differ from the all-row ones in 2e-5 .. 1.4e-3 of their
                              # output elements by one ulp, tools/probe/headrows_probe.py).  Round 3's single 5e-3 gate sat inside that spread:
                              # profiles/r04_full_size_parity_qwen2_siglip.json.  Kernel errors proper are gated per op.
UPDATE_MEAN = 0.21            # mean |master_hip - master_fp32| / lr per tensor after two steps (an AdamW update is <= ~1 lr per step; a slice
                              # that was zeroed, skipped or updated with a wrong gradient reads ~1 .. 1e3 here); measured worst 0.164 (layer 31 k_proj)
UPDATE_FRAC_BAD = 0.09        # fraction of a tensor's elements whose two-step update differs by more than 0.5 lr (measured worst 0.071)


def _avail_gb():
    try:
        import psutil
        return psutil.virtual_memory().available / 1e9
    except Exception:
        return 0.0


def _no_decay(name, shape):
    return len(shape) == 1 and ("norm" in name or name.endswith("bias"))


def _planned_layers():
    """What the test will run, decided at collection so that the test id says it (`...[32_layers]`): the full depth, or the explicit
    RV_FULLSIZE_LAYERS override.  A host that cannot hold the fp32 oracle of the full model (< 180 GB free) FAILS the full-depth test
    instead of silently shrinking it -- run it knowingly reduced with RV_FULLSIZE_LAYERS=8."""
    want = os.environ.get("RV_FULLSIZE_LAYERS")
    gname = os.environ.get("RV_FULLSIZE_GEOMETRY", "llava15_7b")
    return int(want) if want else GEOMETRIES[gname]["lm"]["layers"]


@pytest.mark.parametrize("layers", [pytest.param(_planned_layers(), id=f"{_planned_layers()}_layers")])
def test_config2_full_size_forward_backward_two_adamw_steps(layers):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import bf16_emulation as E
    from oracle import llava_oracle as O
    from radvlm_amd.engine import LlavaEngine
    from conftest import ROOT, record_measurement
    t_start = time.time()
    want = os.environ.get("RV_FULLSIZE_LAYERS")
    # RV_FULLSIZE_GEOMETRY=llava_ov_qwen2_7b runs the same comparison on the configuration RadVLM really trains (Qwen2-7B, 28:4
    # grouped-query heads, q/k/v bias, rope theta 1e6, 152k vocabulary, SigLIP-so400m tower): not part of the suite's default run (another
    # 2.5 minutes), its record is profiles/r03_full_size_parity_qwen2_siglip.json
    gname = os.environ.get("RV_FULLSIZE_GEOMETRY", "llava15_7b")
    full_layers = GEOMETRIES[gname]["lm"]["layers"]
    if not want and _avail_gb() < 180:
        pytest.fail(f"{_avail_gb():.0f} GB of host memory available: the fp32 oracle of all {full_layers} layers needs ~180 GB.  Set "
                    "RV_FULLSIZE_LAYERS=8 to run the reduced model knowingly (the test id then reads [8_layers]).")
    geo = copy.deepcopy(GEOMETRIES[gname])
    geo["lm"]["layers"] = layers
    V = geo["lm"]["vocab"]
    g = torch.Generator().manual_seed(17)
    ids = torch.randint(3, V, (1, 129), generator=g)
    labels = ids.clone()
    labels[:, :64] = -100
    ids[:, 35] = -200
    labels[:, 35] = -100
    mask = torch.ones(1, 129, dtype=torch.bool)
    side = geo["vision"]["image"]
    images = [torch.randn(3, side, side, generator=g).to(torch.bfloat16).float()]
    lr, wd, clip, b1, b2, eps = 2e-5, 0.05, 1.0, 0.9, 0.999, 1e-8      # the recipe's learning rate (finetune_radio_7b.sh)

    eng = LlavaEngine(geo, device="cuda:0", init="fast", seed=11)
    if os.environ.get("RV_FULLSIZE_HEAD_ROWS") == "all":     # A/B of round 4's head-on-the-labelled-rows (the default): every row through the head
        eng.head_rows = eng.last_layer_rows = "all"
    names = eng.lm.names()
    n_params = sum(eng.lm.offsets[n][1] for n in names)
    if layers == full_layers:
        assert n_params > 6.7e9, n_params
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    P = {k: v.float().cpu() for k, v in eng.state_dict().items()}         # bf16-exact fp32 copies (27 GB at 32 layers)
    for k in names:
        P[k].requires_grad_(True)
    a = (ids, mask, labels, images)
    rec = dict(geometry=gname, layers=layers, trainable_params=n_params, host_threads=torch.get_num_threads())

    def dump():
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "full_size_parity.json"), "w") as f:
            json.dump(rec, f, indent=1)

    def hip_step():
        loss = float(eng.forward(ids.numpy(), mask.numpy(), labels.numpy(), images, want_logits=True))
        logits = eng.last_logits.cpu()
        pm = torch.from_numpy(eng.ctx["plan"]["attention_mask"])
        eng.backward()
        torch.cuda.synchronize()
        return loss, logits, pm

    def oracle_step():
        for k in names:
            P[k].grad = None
        rl, rlog, aux = O.llava_forward(P, geo, *a)
        rl.backward()
        return float(rl), rlog.detach(), aux

    def compare_grads(tag):
        worst, worst_name, sq_h, sq_r = 0.0, None, 0.0, 0.0
        per_kind = {}
        for k in names:
            ref = P[k].grad.to("cuda:0")
            got = eng.G(k).float()
            rn = float(ref.norm())
            sq_h += float(got.double().pow(2).sum())
            sq_r += float(ref.double().pow(2).sum())
            rel = float((got - ref).norm()) / max(rn, 1e-30)
            kind = k.split(".")[-2] if k.startswith("model.layers.") else k
            per_kind[kind] = max(per_kind.get(kind, 0.0), rel)
            if rel > worst:
                worst, worst_name = rel, k
            if not (rn > 0 and rel < GRAD_REL_L2):
                rec[tag + "_failed"] = dict(tensor=k, rel_l2=rel, ref_norm=rn, per_kind_worst_rel_l2=per_kind)
                dump()
                raise AssertionError((tag, k, rel, rn, rec))
        gn_h, gn_r = math.sqrt(sq_h), math.sqrt(sq_r)
        rec[tag] = dict(worst_rel_l2=worst, worst_tensor=worst_name, per_kind_worst_rel_l2=per_kind, grad_norm_hip=gn_h, grad_norm_fp32=gn_r)
        dump()
        assert abs(gn_h - gn_r) <= (GRAD_NORM_REL_STEP2 if tag.endswith("step2") else GRAD_NORM_REL) * gn_r, (tag, gn_h, gn_r)
        return gn_r

    M, Vv, MASTER = {}, {}, {}

    def oracle_adamw(step, total_norm):
        """fp32 AdamW on fp32 master weights; the weights the next forward multiplies with are the masters rounded to bf16 -- the
        storage scheme under test (the reference's run: bf16 model, fp32 optimizer state, zero3.json `bf16.enabled`).  Without the
        rounding the two models would differ after one update by the weights' bf16 rounding noise (~0.5 ulp = 6e-5 at |w| = 0.02, the
        size of the update itself), which is not an error of the kernels."""
        coef = min(1.0, clip / (total_norm + 1e-6))           # torch.nn.utils.clip_grad_norm_
        with torch.no_grad():
            for k in names:
                p = P[k]
                if k not in M:
                    M[k], Vv[k], MASTER[k] = torch.zeros_like(p), torch.zeros_like(p), p.detach().clone()
                O.adamw_step(MASTER[k], p.grad * coef, M[k], Vv[k], step, lr, b1, b2, eps, 0.0 if _no_decay(k, p.shape) else wd)
                p.copy_(MASTER[k].to(torch.bfloat16).float())

    # ---- step 1: forward (vs fp32 oracle and vs the bf16-emulating oracle), backward, optimizer
    loss1, logits, pm = hip_step()
    t0 = time.time()
    rl1, rlog, aux = oracle_step()
    rec["oracle_fwd_bwd_s"] = time.time() - t0
    assert aux["inputs_embeds"].shape[1] == (geo["vision"]["image"] // geo["vision"]["patch"]) ** 2 + 128
    le, lge, _ = E.llava_forward({k: v.detach() for k, v in P.items()}, geo, *a, emulate=True)
    m = pm
    relinf = lambda x, y: float((x[m] - y[m]).abs().max() / y[m].abs().max())
    rel2 = lambda x, y: float((x[m] - y[m]).norm() / y[m].norm())
    rec.update(loss_hip=loss1, loss_fp32=rl1, loss_emu=float(le), hip_vs_fp32_inf=relinf(logits, rlog), hip_vs_fp32_l2=rel2(logits, rlog),
               emu_vs_fp32_inf=relinf(lge, rlog), emu_vs_fp32_l2=rel2(lge, rlog), hip_vs_emu_inf=relinf(logits, lge), hip_vs_emu_l2=rel2(logits, lge))
    del lge, rlog
    dump()
    assert abs(loss1 - rl1) <= LOSS_TOL, rec
    assert abs(loss1 - float(le)) <= LOSS_TOL, rec
    # the same three inequalities as test_bf16_emulated_parity: as close to fp32 as an ideal bf16 implementation of the same store points
    assert rec["hip_vs_fp32_l2"] <= 1.1 * rec["emu_vs_fp32_l2"] and rec["hip_vs_fp32_inf"] <= 1.25 * rec["emu_vs_fp32_inf"], rec
    assert rec["hip_vs_emu_l2"] <= rec["emu_vs_fp32_l2"] * 1.25, rec      # two bf16 realisations are as far apart as either is from fp32 (measured 1.06x)
    gn1 = compare_grads("grads_step1")
    eng.optimizer_step(lr, weight_decay=wd, betas=(b1, b2), eps=eps, max_grad_norm=clip)
    rec["grad_norm_engine_step1"] = float(eng.last_grad_norm)
    assert abs(rec["grad_norm_engine_step1"] - gn1) <= GRAD_NORM_REL * gn1, rec
    oracle_adamw(1, gn1)

    # ---- step 2 (from the second optimizer step on, a wrong master copy / a slice that was never updated shows)
    loss2, _, _ = hip_step()
    rl2, _, _ = oracle_step()
    rec.update(loss2_hip=loss2, loss2_fp32=rl2)
    assert abs(loss2 - rl2) <= LOSS_TOL, rec
    gn2 = compare_grads("grads_step2")
    eng.optimizer_step(lr, weight_decay=wd, betas=(b1, b2), eps=eps, max_grad_norm=clip)
    oracle_adamw(2, gn2)
    torch.cuda.synchronize()

    # ---- every trainable tensor after two updates: fp32 master vs the oracle's fp32 parameters, bf16 parameters = round(master)
    worst_mean, worst_bad = (0.0, None), (0.0, None)
    for k in names:
        off, n = eng.lm.offsets[k]
        ref = MASTER[k].to("cuda:0").view(-1)
        mas = eng.master[off:off + n]
        par = eng.lm.flat[off:off + n]
        assert torch.equal(par, mas.to(torch.bfloat16)), k                      # the bf16 parameters ARE the rounded master copy
        du = (mas - ref).abs() / lr
        mean, bad = float(du.mean()), float((du > 0.5).float().mean())
        if k.endswith("k_proj.bias"):
            # Qwen2's key bias: softmax is invariant to a constant key offset, so this gradient is ~0 (only the rotary embedding makes it
            # position dependent) and AdamW's normalised direction m / sqrt(v) of a ~0 gradient is rounding noise on both sides (measured
            # mean 0.45 lr); the gradient itself is gated above like every other tensor
            rec.setdefault("update_key_bias_mean_over_lr", []).append(round(mean, 3))
            continue
        if mean > worst_mean[0]:
            worst_mean = (mean, k)
        if bad > worst_bad[0]:
            worst_bad = (bad, k)
    rec.update(update_worst_mean_over_lr=worst_mean, update_worst_frac_gt_half_lr=worst_bad,
               seconds=time.time() - t_start, lr=lr, weight_decay=wd, max_grad_norm=clip)
    record_measurement("config2_full_size", **rec)
    dump()
    print(json.dumps(rec))
    assert worst_mean[0] <= UPDATE_MEAN and worst_bad[0] <= UPDATE_FRAC_BAD, (worst_mean, worst_bad)
    # 108 GB of device memory and 135 GB of host memory: hand them back before the next test of this process
    del eng, P, M, Vv, MASTER
    import gc
    gc.collect()
    torch.cuda.empty_cache()
