"""CPU checks of the anyres_max host planning: tap lists against torch's bilinear interpolate, and the splice
plan (labels / mask / created-row bookkeeping) against the reference's golden vectors."""
import json
import os

import numpy as np
import torch
import torch.nn.functional as F

from radvlm_amd.config import GEOMETRIES
from radvlm_amd.splice import bilinear_taps, build_splice_plan, merged_feature_rows


def test_bilinear_taps_match_interpolate():
    g = torch.Generator().manual_seed(0)
    for (h, w, oh, ow) in [(54, 54, 38, 38), (54, 40, 38, 28), (27, 81, 19, 57), (10, 7, 3, 2)]:
        x = torch.randn(5, h, w, generator=g)
        want = F.interpolate(x[None], [oh, ow], mode="bilinear")[0]
        idx, wt = bilinear_taps(h, w, oh, ow)
        got = (x.reshape(5, -1)[:, torch.from_numpy(idx)] * torch.from_numpy(wt)).sum(-1).view(5, oh, ow)
        assert float((got - want).abs().max()) < 1e-5
        assert np.allclose(wt.sum(-1), 1.0, atol=1e-6)


def test_anyres_max_plan_against_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "toy_qwen_anyres_max_e2e.npz"))
    meta = json.load(open(os.path.join(golden_dir, "toy_qwen_anyres_max_e2e_gradnorms.json")))
    v = GEOMETRIES["toy_qwen"]["vision"]
    side = v["image"] // v["patch"]
    P = side * side
    tiles = [g[f"image{i}"].shape[0] for i in range(2)]
    n_proj = sum(tiles) * P
    extra = dict(next=n_proj, src=[], w=[])
    rows, r0 = [], 0
    for i, t in enumerate(tiles):
        rows.append(merged_feature_rows(r0, t, side, meta["merge_type"], meta["aspect"], tuple(g["image_sizes"][i].tolist()),
                                        meta["pinpoints"], v["image"], extra=extra))
        r0 += t * P
    n_extra = extra["next"] - n_proj
    assert n_extra == 38 * 38                     # 54x54 grid, times = sqrt(2916 / 1458) = 1.414 > 1.1 -> 38x38
    assert len(rows[0]) == P + 38 * 39            # base tile + (38 rows of 38 created tokens + newline)
    assert len(extra["src"]) == 1                 # the second image (2x1 grid, 54x20 after unpad) stays below the limit
    plan = build_splice_plan(g["input_ids"], g["attention_mask"], g["labels"], rows, n_proj + n_extra)
    assert np.array_equal(plan["labels"], g["splice_labels"])
    assert np.array_equal(plan["attention_mask"], g["splice_attention_mask"])
    src = np.concatenate(extra["src"]).reshape(-1)
    assert (plan["feat_pos"][np.unique(src)] < 0).all()
    assert (plan["feat_pos"][n_proj:] >= 0).all()  # every created row is spliced somewhere
