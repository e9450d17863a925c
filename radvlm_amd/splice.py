"""Host-side index planning for the multimodal splice (int64 index arithmetic only; no device work).

Restates steps (iii)-(ix) of ``LlavaMetaForCausalLM.prepare_inputs_labels_for_multimodal``
(reference finetuning/llava/model/llava_arch.py:251-555) as index arrays, so that the device side is ONE
row gather (rv_gather_rows) instead of the reference's per-sample Python loop with host syncs (:450,:461):

  * ``merged_feature_rows``: which projector-output row (or the image_newline marker) sits at each merged
    image-token position -- 'flat' (:298-299) and 'spatial[_unpad]' + anyres (:301-413) geometry, obtained by
    running the reference's view/permute/unpad/cat sequence on an index tensor;
  * ``build_splice_plan``: strip padding by mask (:442-443), split at IMAGE_TOKEN_INDEX, interleave, labels -100
    over image spans (:449-493), truncate (:496-500), right-pad with exact-zero rows (:507-531).
"""
import math
import re

import numpy as np

IGNORE_INDEX = -100       # constants.py:7-12
IMAGE_TOKEN_INDEX = -200  # constants.py:7-12
NEWLINE = -1              # marker inside merged feature index arrays


def select_best_resolution(original_size, possible_resolutions):
    """mm_utils.py:119-149."""
    ow, oh = original_size
    best, max_eff, min_waste = None, 0, float("inf")
    for w, h in possible_resolutions:
        scale = min(w / ow, h / oh)
        dw, dh = int(ow * scale), int(oh * scale)
        eff = min(dw * dh, ow * oh)
        waste = w * h - eff
        if eff > max_eff or (eff == max_eff and waste < min_waste):
            max_eff, min_waste, best = eff, waste, (w, h)
    return best


def get_anyres_image_grid_shape(image_size, grid_pinpoints, patch_size):
    """mm_utils.py:213-240 (list-form pinpoints, and the '(1x1),...,(NxN)' range string form :225-236)."""
    if isinstance(grid_pinpoints, str) and "x" in grid_pinpoints:
        m = re.findall(r"\((\d+)x(\d+)\)", grid_pinpoints)
        (a0, b0), (a1, b1) = tuple(map(int, m[0])), tuple(map(int, m[-1]))
        grid_pinpoints = [(i * patch_size, j * patch_size) for i in range(a0, a1 + 1) for j in range(b0, b1 + 1)]
    w, h = select_best_resolution(image_size, [tuple(p) for p in grid_pinpoints])
    return w // patch_size, h // patch_size


def unpad_index(t, original_size):
    """unpad_image (llava_arch.py:127-159) on an index array t [H, W]; original_size = (W, H)."""
    ow, oh = original_size
    ch, cw = t.shape
    if ow / oh > cw / ch:
        nh = int(oh * (cw / ow))
        pad = (ch - nh) // 2
        return t[pad:ch - pad, :]
    nw = int(ow * (ch / oh))
    pad = (cw - nw) // 2
    return t[:, pad:cw - pad]


def bilinear_taps(h, w, oh, ow):
    """The 4 taps of every output pixel of nn.functional.interpolate(x[None], [oh, ow], mode="bilinear") (align_corners
    False, no antialias; llava_arch.py:389-391) as (idx int64 [oh*ow, 4] into the h*w source grid, weight fp32 [oh*ow, 4]).
    Source coordinate = (in/out) * (dst + 0.5) - 0.5 clamped at 0, in fp32 like ATen's area_pixel_compute_source_index."""
    def axis(n_in, n_out):
        scale = np.float32(n_in) / np.float32(n_out)
        src = np.maximum(scale * (np.arange(n_out, dtype=np.float32) + np.float32(0.5)) - np.float32(0.5), np.float32(0))
        i0 = np.minimum(np.floor(src).astype(np.int64), n_in - 1)
        i1 = np.minimum(i0 + 1, n_in - 1)
        l1 = (src - i0.astype(np.float32)).astype(np.float32)
        return i0, i1, (np.float32(1) - l1).astype(np.float32), l1
    y0, y1, wy0, wy1 = axis(h, oh)
    x0, x1, wx0, wx1 = axis(w, ow)
    idx = np.stack([y0[:, None] * w + x0[None, :], y0[:, None] * w + x1[None, :],
                    y1[:, None] * w + x0[None, :], y1[:, None] * w + x1[None, :]], axis=-1).reshape(-1, 4)
    wt = np.stack([wy0[:, None] * wx0[None, :], wy0[:, None] * wx1[None, :],
                   wy1[:, None] * wx0[None, :], wy1[:, None] * wx1[None, :]], axis=-1).reshape(-1, 4).astype(np.float32)
    return idx, wt


def merged_feature_rows(row0, n_tiles, side, merge_type="flat", aspect="square", image_size=None, pinpoints=None,
                        tower_image_size=None, extra=None):
    """Index (into the feature row table) of every merged image token of ONE sample; NEWLINE = -1.

    ``extra`` (dict with ``next``, ``src``, ``w``) collects the rows that anyres_max creates by bilinear down-sampling
    (llava_arch.py:381-392): they get fresh table row ids starting at extra["next"], and their 4 (source row, weight)
    taps are appended to extra["src"] / extra["w"] for the device-side weighted gather."""
    P = side * side
    rows = np.arange(row0, row0 + n_tiles * P, dtype=np.int64).reshape(n_tiles, P)
    if merge_type == "flat":
        return rows.reshape(-1)
    if not merge_type.startswith("spatial"):
        raise ValueError(f"Unexpected mm_patch_merge_type: {merge_type}")
    if n_tiles > 1:
        base, rest = rows[0], rows[1:]
        if aspect == "anyres" or "anyres_max" in aspect:
            gw, gh = get_anyres_image_grid_shape(image_size, pinpoints, tower_image_size)
        else:
            gw, gh = 2, 2
        rest = rest.reshape(gh, gw, side, side)
        if "maxpool2x2" in merge_type:
            # llava_arch.py:375-379: 2x2 max pooling (stride 2, floor) of the tile grid; no image_newline.  Every pooled token is
            # a NEW table row whose four source rows are recorded in extra["pool_src"] (the device takes the elementwise max).
            if extra is None:
                raise ValueError("maxpool2x2 merge needs the `extra` collector")
            grid = rest.transpose(0, 2, 1, 3).reshape(gh * side, gw * side)
            oh, ow = grid.shape[0] // 2, grid.shape[1] // 2
            win = grid[:2 * oh, :2 * ow].reshape(oh, 2, ow, 2).transpose(0, 2, 1, 3).reshape(oh * ow, 4)   # (dy, dx) scan order
            extra.setdefault("pool_src", []).append(win)
            pooled = extra["next"] + np.arange(oh * ow, dtype=np.int64)
            extra.setdefault("pool_out", []).append(pooled)
            extra["next"] += oh * ow
            return pooled if "nobase" in merge_type else np.concatenate([base, pooled])
        if "unpad" in merge_type:
            grid = rest.transpose(0, 2, 1, 3).reshape(gh * side, gw * side)
            grid = unpad_index(grid, image_size)
            mx = re.match(r"anyres_max_(\d+)", aspect)
            if mx:
                hh, ww = grid.shape
                times = math.sqrt(hh * ww / (int(mx.group(1)) * side ** 2))
                if times > 1.1:
                    if extra is None:
                        raise ValueError("anyres_max down-sampling needs the `extra` collector")
                    oh, ow = int(hh // times), int(ww // times)
                    idx, wt = bilinear_taps(hh, ww, oh, ow)
                    extra["src"].append(grid.reshape(-1)[idx])
                    extra["w"].append(wt)
                    grid = (extra["next"] + np.arange(oh * ow, dtype=np.int64)).reshape(oh, ow)
                    extra["next"] += oh * ow
            nl = np.full((grid.shape[0], 1), NEWLINE, dtype=np.int64)
            rest = np.concatenate([grid, nl], axis=1).reshape(-1)
        else:
            rest = rest.transpose(0, 2, 1, 3).reshape(-1)
        if "nobase" in merge_type:
            return rest
        return np.concatenate([base, rest])
    out = rows[0]
    if "unpad" in merge_type:
        out = np.concatenate([out, np.array([NEWLINE], dtype=np.int64)])
    return out


def build_splice_plan(input_ids, attention_mask, labels, feature_rows, n_feat_rows, max_len=None, padding_side="right"):
    """input_ids/labels int64 [B,T], attention_mask bool [B,T] (numpy); feature_rows: per-sample merged index arrays.

    Returns dict with
      S, lens [B], idx int32 [B*S] (>=0 token id | -1 zero row | <=-2 feature-table row -idx-2, where row
      n_feat_rows is image_newline), labels [B,S], attention_mask [B,S], feat_pos int32 [n_feat_rows] (sequence row
      where each projector row landed, -1 if unused), newline_pos int32 [...], and the embedding CSR
      (tok_ids, tok_off, tok_pos) grouping sequence rows by token id (for the no-atomics embedding gradient).
    padding_side "left" (config.tokenizer_padding_side, llava_arch.py:520-524): shorter samples sit at the END of their row.
    """
    B = input_ids.shape[0]
    seqs, labs = [], []
    img_i = 0
    for b in range(B):
        ids = input_ids[b][attention_mask[b]]
        lab = labels[b][attention_mask[b]]
        pos = np.nonzero(ids == IMAGE_TOKEN_INDEX)[0].tolist()
        if not pos:  # text-only sample still consumes one (dummy) image slot, contributing zero rows (:452-459)
            seqs.append(ids.astype(np.int64))
            labs.append(lab)
            img_i += 1
            continue
        bounds = [-1] + pos + [ids.shape[0]]
        ps, pl = [], []
        for i in range(len(bounds) - 1):
            seg = slice(bounds[i] + 1, bounds[i + 1])
            ps.append(ids[seg].astype(np.int64))
            pl.append(lab[seg])
            if i < len(pos):
                fr = feature_rows[min(img_i, len(feature_rows) - 1)]
                img_i += 1
                enc = np.where(fr == NEWLINE, -(n_feat_rows) - 2, -fr - 2)
                ps.append(enc)
                pl.append(np.full(fr.shape[0], IGNORE_INDEX, dtype=lab.dtype))
        seqs.append(np.concatenate(ps)[:max_len])
        labs.append(np.concatenate(pl)[:max_len])
    S = max(s.shape[0] for s in seqs)
    idx = np.full((B, S), -1, dtype=np.int64)
    L = np.full((B, S), IGNORE_INDEX, dtype=np.int64)
    M = np.zeros((B, S), dtype=bool)
    lens = np.zeros(B, dtype=np.int32)
    for b, (s, l) in enumerate(zip(seqs, labs)):
        n = s.shape[0]
        lens[b] = n
        if padding_side == "left":
            if n:
                idx[b, S - n:], L[b, S - n:], M[b, S - n:] = s, l, True
        else:
            idx[b, :n], L[b, :n], M[b, :n] = s, l, True
    flat = idx.reshape(-1)
    feat_pos = np.full(n_feat_rows, -1, dtype=np.int32)
    is_feat = flat <= -2
    fr = -flat[is_feat] - 2
    where = np.nonzero(is_feat)[0]
    nl = fr == n_feat_rows
    feat_pos[fr[~nl]] = where[~nl]
    newline_pos = where[nl].astype(np.int32)
    tok_where = np.nonzero(flat >= 0)[0]
    tok = flat[tok_where]
    order = np.argsort(tok, kind="stable")
    tok_sorted = tok[order]
    tok_ids, starts = np.unique(tok_sorted, return_index=True)
    tok_off = np.concatenate([starts, [tok_sorted.shape[0]]]).astype(np.int32)
    return dict(S=S, lens=lens, idx=flat.astype(np.int32), labels=L, attention_mask=M, feat_pos=feat_pos,
                newline_pos=newline_pos, tok_ids=tok_ids.astype(np.int32), tok_off=tok_off,
                tok_pos=tok_where[order].astype(np.int32))


def shifted_labels(L):
    """Target of row s is labels[s+1] (modeling_llama.py:1326-1331); last position ignored."""
    out = np.full_like(L, IGNORE_INDEX)
    out[:, :-1] = L[:, 1:]
    return out
