"""Diagnostic (not product, not a test): where do the HIP engine and the bf16-emulating oracle stop being bit-identical?

    python tools/bf16_flip_trace.py [toy_e2e toy]

For every stored activation of the toy forward pass: the fraction of elements whose bf16 bit pattern differs between the engine and
oracle/bf16_emulation.py, and the max difference in bf16 ulps.  A kernel that follows the stated store points differs only where an
fp32 pre-rounding value sits within summation-order noise of a rounding boundary (~1e-5 of the elements); every such flip perturbs
its whole row downstream (relative 2^-8 on one element -> the row's next GEMM outputs move by ~2^-8/sqrt(d), i.e. a few percent
of THEIR elements flip), so agreement decays layer by layer -- the reason an end-to-end 1e-3 gate cannot be met by any two bf16
implementations that are not bit-identical."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle import bf16_emulation as E  # noqa: E402
from oracle import llava_oracle as O  # noqa: E402
from radvlm_amd.config import GEOMETRIES  # noqa: E402
from radvlm_amd.engine import LlavaEngine  # noqa: E402
from radvlm_amd.smoke import load_golden_batch  # noqa: E402

name, geo_name = (sys.argv[1:3] + ["toy_e2e", "toy"])[:2] if len(sys.argv) >= 3 else ("toy_e2e", "toy")
geo = GEOMETRIES[geo_name]
g, images = load_golden_batch(name)
eng = LlavaEngine(geo, device="cuda:0", init="portable", seed=0, packed=False)
eng.forward(g["input_ids"], g["attention_mask"], g["labels"], images, want_logits=True)
ctx = eng.ctx
E.TRACE = {}
P = O.make_params(geo, seed=0)
le, lge, aux = E.llava_forward(P, geo, torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"]), torch.from_numpy(g["labels"]), images)
T = E.TRACE
m = aux["attention_mask"].reshape(-1)
d = geo["lm"]["d"]
kvd = d // geo["lm"]["heads"] * geo["lm"].get("kv_heads", geo["lm"]["heads"])
out = []


def cmp(tag, hip, emu, rows=None):
    hip = hip.detach().float().cpu().reshape(emu.shape if rows is None else (-1, emu.shape[-1]))
    emu = emu.reshape(hip.shape)
    if rows is not None:
        hip, emu = hip[rows], emu[rows]
    mono = lambda t: (lambda i: torch.where(i >= 0, i, -(i & 0x7FFF)))(t.to(torch.bfloat16).view(torch.int16).int())   # monotonic in the value
    hb, eb = mono(hip), mono(emu)
    frac = float((hb != eb).float().mean())
    ulp = int((hb - eb).abs().max())
    rel = float((hip - emu).abs().max() / emu.abs().max())
    out.append(dict(tensor=tag, mismatch_frac=frac, max_ulp=ulp, rel_inf=rel))
    print(f"{tag:40s} mismatch {frac:9.2e}  max {ulp:3d} ulp  rel-inf {rel:.2e}")


cmp("tower features f0", ctx["f0"], T["f0"])
cmp("projector z1", ctx["z1"], T["z1"])
cmp("projector a1 (gelu)", ctx["a1"], T["a1"])
for i, a in enumerate(ctx["layers"]):
    p = f"model.layers.{i}."
    cmp(p + "x", a["x"], T[p + "x"], m)
    cmp(p + "h1 (rmsnorm)", a["h1"], T[p + "h1"], m)
    cmp(p + "q (roped)", a["qkv"][:, :d], T[p + "q_roped"], m)
    cmp(p + "k (roped)", a["qkv"][:, d:d + kvd], T[p + "k_roped"], m)
    cmp(p + "v", a["qkv"][:, d + kvd:], T[p + "v"], m)
    cmp(p + "attn", a["attn"], T[p + "attn"], m)
    cmp(p + "x_mid", a["x_mid"], T[p + "x_mid"], m)
    cmp(p + "h2 (rmsnorm)", a["h2"], T[p + "h2"], m)
    cmp(p + "gate|up", a["gu"], T[p + "gu"], m)
    cmp(p + "act (swiglu)", a["act"], T[p + "act"], m)
cmp("x_last", ctx["x_last"], T["x_last"], m)
cmp("hN", ctx["hN"], T["hN"], m)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"bf16_flip_trace_{name}.json"), "w"), indent=1)
