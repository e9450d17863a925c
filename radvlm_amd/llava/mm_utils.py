"""Host image geometry + prompt tokenisation of the LLaVA input pipeline (CPU, dataloader workers).

Restates reference finetuning/llava/mm_utils.py: select_best_resolution :119-149, resize_and_pad_image :152-188,
divide_to_patches :191-210, get_anyres_image_grid_shape :213-240, process_anyres_image :243-293,
expand2square :300-311, process_images :314-338, tokenizer_image_token :341-360.  A small dependency-free CLIP
image processor (``ClipImageProcessor``) stands in for HF's CLIPImageProcessor (resize shortest edge bicubic,
centre crop, 1/255 rescale, mean/std normalise).
"""
import ast
import math
import re

import numpy as np
import torch
from PIL import Image

from ..splice import get_anyres_image_grid_shape, select_best_resolution  # noqa: F401  (same functions, one source)
from .constants import IMAGE_TOKEN_INDEX

OPENAI_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class ClipImageProcessor:
    def __init__(self, size=336, crop_size=None, image_mean=OPENAI_CLIP_MEAN, image_std=OPENAI_CLIP_STD):
        self.size = {"shortest_edge": size}
        c = crop_size or size
        self.crop_size = {"height": c, "width": c}
        self.image_mean = list(image_mean)
        self.image_std = list(image_std)

    def _one(self, img):
        img = img.convert("RGB")
        w, h = img.size
        s = self.size["shortest_edge"]
        if w <= h:
            nw, nh = s, int(s * h / w)
        else:
            nw, nh = int(s * w / h), s
        if (nw, nh) != (w, h):
            img = img.resize((nw, nh), resample=Image.BICUBIC)
        ch, cw = self.crop_size["height"], self.crop_size["width"]
        left, top = (nw - cw) // 2, (nh - ch) // 2
        img = img.crop((left, top, left + cw, top + ch))
        a = np.asarray(img, dtype=np.float32) / 255.0
        a = (a - np.asarray(self.image_mean, dtype=np.float32)) / np.asarray(self.image_std, dtype=np.float32)
        return torch.from_numpy(a.transpose(2, 0, 1).copy())

    def preprocess(self, images, return_tensors="pt"):
        if not isinstance(images, (list, tuple)):
            images = [images]
        if self.device_normalize:
            return {"pixel_values": torch.stack([self._one_u8(im) for im in images], 0)}
        return {"pixel_values": torch.stack([self._one(im) for im in images], 0)}

    __call__ = preprocess

    # device-side normalisation: the host stops after resize + crop and hands over uint8 HWC pixels; rescale, normalise,
    # channel-first layout and the bf16 cast run on the GPU (rv_normalize_tiles_u8, bit-identical to _one() + cast)
    device_normalize = False
    normalize_mode = 0

    def _one_u8(self, img):
        img = img.convert("RGB")
        w, h = img.size
        s = self.size["shortest_edge"]
        nw, nh = (s, int(s * h / w)) if w <= h else (int(s * w / h), s)
        if (nw, nh) != (w, h):
            img = img.resize((nw, nh), resample=Image.BICUBIC)
        ch, cw = self.crop_size["height"], self.crop_size["width"]
        left, top = (nw - cw) // 2, (nh - ch) // 2
        return torch.from_numpy(np.asarray(img.crop((left, top, left + cw, top + ch)), dtype=np.uint8).copy())


class SigLipImageProcessor:
    """SigLipImageProcessor (multimodal_encoder/siglip_encoder.py:34-67): RGB, direct bicubic resize to `size` (no aspect
    preservation, no crop), rescale 1/255, normalise with mean = std = 0.5, channels first.  `size` is a (h, w) tuple like
    the reference's, which process_anyres_image reads as ``processor.size[0]`` (mm_utils.py:257-260)."""

    def __init__(self, image_mean=(0.5, 0.5, 0.5), image_std=(0.5, 0.5, 0.5), size=(384, 384), crop_size=None, rescale_factor=1 / 255):
        self.image_mean = image_mean
        self.image_std = image_std
        self.size = size
        self.rescale_factor = rescale_factor
        self.crop_size = crop_size if crop_size is not None else {"height": 384, "width": 384}

    def _one(self, img):
        img = img.convert("RGB")
        h, w = self.size
        img = img.resize((w, h), resample=Image.BICUBIC)
        a = np.asarray(img).astype(np.float64) * self.rescale_factor
        a = (a.astype(np.float32) - np.asarray(self.image_mean, dtype=np.float32)) / np.asarray(self.image_std, dtype=np.float32)
        return torch.from_numpy(a.transpose(2, 0, 1).copy())

    def preprocess(self, images, return_tensors="pt"):
        if not isinstance(images, (list, tuple)):
            images = [images]
        if self.device_normalize:
            return {"pixel_values": torch.stack([self._one_u8(im) for im in images], 0)}
        return {"pixel_values": torch.stack([self._one(im) for im in images], 0)}

    __call__ = preprocess

    device_normalize = False       # see ClipImageProcessor: uint8 HWC hand-over, normalisation on the GPU
    normalize_mode = 1

    def _one_u8(self, img):
        h, w = self.size
        return torch.from_numpy(np.asarray(img.convert("RGB").resize((w, h), resample=Image.BICUBIC), dtype=np.uint8).copy())


def resize_and_pad_image(image, target_resolution):
    """Aspect-preserving resize to fit `target_resolution` (w, h), centred on a black canvas."""
    ow, oh = image.size
    tw, th = target_resolution
    sw, sh = tw / ow, th / oh
    if sw < sh:
        nw, nh = tw, min(math.ceil(oh * sw), th)
    else:
        nh, nw = th, min(math.ceil(ow * sh), tw)
    canvas = Image.new("RGB", (tw, th), (0, 0, 0))
    canvas.paste(image.resize((nw, nh)), ((tw - nw) // 2, (th - nh) // 2))
    return canvas


def divide_to_patches(image, patch_size):
    w, h = image.size
    return [image.crop((x, y, x + patch_size, y + patch_size)) for y in range(0, h, patch_size) for x in range(0, w, patch_size)]


def _pinpoints_list(grid_pinpoints, patch_size):
    if isinstance(grid_pinpoints, str) and "x" in grid_pinpoints:
        m = re.findall(r"\((\d+)x(\d+)\)", grid_pinpoints)
        (a0, b0), (a1, b1) = tuple(map(int, m[0])), tuple(map(int, m[-1]))
        return [[i * patch_size, j * patch_size] for i in range(a0, a1 + 1) for j in range(b0, b1 + 1)]
    return grid_pinpoints if isinstance(grid_pinpoints, list) else ast.literal_eval(grid_pinpoints)


def process_anyres_image(image, processor, grid_pinpoints):
    """-> [1 + tiles, 3, s, s]: globally resized base tile first, then the row-major tiles of the padded image."""
    edge = processor.size["shortest_edge"] if isinstance(processor.size, dict) else min(processor.size)
    res = select_best_resolution(image.size, [tuple(p) for p in _pinpoints_list(grid_pinpoints, edge)])
    tiles = divide_to_patches(resize_and_pad_image(image, res), processor.crop_size["height"])
    tiles = [image.resize((edge, edge))] + tiles
    return torch.stack([processor.preprocess(t, return_tensors="pt")["pixel_values"][0] for t in tiles], 0)


def _center_crop_to_short_edge(image, edge):
    """Resize so the SHORT side equals `edge` (long side by the aspect ratio, truncated), then centre-crop an edge x edge
    square (mm_utils.py:12-30; the reference's Image.ANTIALIAS filter is Pillow's LANCZOS, its name since Pillow 10)."""
    ar = float(image.width) / float(image.height)
    nw, nh = (int(edge * ar), edge) if ar > 1 else (edge, int(edge / ar))
    im = image.resize((nw, nh), Image.LANCZOS)
    left, top = (nw - edge) / 2, (nh - edge) / 2
    return im.crop((left, top, left + edge, top + edge))


def extract_patches(image, patch_size, overlap_ratio):
    """Row-major grid of patch_size squares with the given overlap, centred in the image (mm_utils.py:62-85)."""
    assert patch_size > 0 and 0 <= overlap_ratio < 1
    W, H = image.size
    stride = int(patch_size * (1 - overlap_ratio))
    ny, nx = (H - patch_size) // stride + 1, (W - patch_size) // stride + 1
    y0, x0 = (H - (ny - 1) * stride - patch_size) // 2, (W - (nx - 1) * stride - patch_size) // 2
    return [image.crop((x, y, x + patch_size, y + patch_size))
            for y in range(y0, y0 + ny * stride, stride) for x in range(x0, x0 + nx * stride, stride)]


def process_highres_image_crop_split(image, data_args, processor=None):
    """image_aspect_ratio='crop_split' (mm_utils.py:88-97): centre crop at image_crop_resolution, split into
    image_split_resolution tiles, preprocess each -> [tiles, 3, s, s] (no base tile)."""
    processor = processor or data_args.image_processor
    crop = _center_crop_to_short_edge(image, data_args.image_crop_resolution)
    tiles = extract_patches(crop, data_args.image_split_resolution, 0)
    return torch.stack([processor.preprocess(t, return_tensors="pt")["pixel_values"][0] for t in tiles], 0)


def process_highres_image(image, processor, grid_pinpoints):
    """image_aspect_ratio='highres' (mm_utils.py:100-118): the largest size of the comma-separated `grid_pinpoints` is always
    selected (the reference's FIXME), the image is padded to a square with the processor mean, resized to that size and cut
    into shortest_edge tiles; the globally resized image comes first -> [1 + tiles, 3, s, s]."""
    select = max(int(x) for x in grid_pinpoints.split(","))
    edge = processor.size["shortest_edge"]
    padded = expand2square(image, tuple(int(x * 255) for x in processor.image_mean)).resize((select, select))
    tiles = [image.resize((edge, edge))] + extract_patches(padded, edge, 0)
    return torch.stack([processor.preprocess(t, return_tensors="pt")["pixel_values"][0] for t in tiles], 0)


def expand2square(pil_img, background_color):
    w, h = pil_img.size
    if w == h:
        return pil_img
    side = max(w, h)
    out = Image.new(pil_img.mode, (side, side), background_color)
    out.paste(pil_img, ((side - w) // 2, (side - h) // 2))
    return out


def process_images(images, image_processor, model_cfg):
    aspect = getattr(model_cfg, "image_aspect_ratio", None)
    if aspect == "anyres" or (aspect and "anyres_max" in aspect):
        out = [process_anyres_image(im, image_processor, model_cfg.image_grid_pinpoints) for im in images]
    elif aspect == "pad":
        bg = tuple(int(x * 255) for x in image_processor.image_mean)
        out = [image_processor.preprocess(expand2square(im, bg), return_tensors="pt")["pixel_values"][0] for im in images]
    elif aspect == "highres":
        out = [process_highres_image(im, image_processor, model_cfg.image_grid_pinpoints) for im in images]
    elif aspect == "crop_split":
        out = [process_highres_image_crop_split(im, model_cfg, image_processor) for im in images]
    else:
        return image_processor.preprocess(images, return_tensors="pt")["pixel_values"]
    if all(x.shape == out[0].shape for x in out):
        out = torch.stack(out, 0)
    return out


def tokenizer_image_token(prompt, tokenizer, image_token_index=IMAGE_TOKEN_INDEX, return_tensors=None):
    """Tokenise the text between '<image>' markers and join the pieces with `image_token_index`, keeping one BOS."""
    pieces = [tokenizer(c).input_ids for c in prompt.split("<image>")]
    has_bos = bool(pieces) and bool(pieces[0]) and pieces[0][0] == tokenizer.bos_token_id
    ids = [pieces[0][0]] if has_bos else []
    skip = 1 if has_bos else 0
    for n, piece in enumerate(pieces):
        if n:
            ids.append(image_token_index)
        ids.extend(piece[skip:])
    if return_tensors is None:
        return ids
    if return_tensors == "pt":
        return torch.tensor(ids, dtype=torch.long)
    raise ValueError(f"Unsupported tensor type: {return_tensors}")
