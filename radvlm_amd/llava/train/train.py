"""Entry point + host data path of LLaVA fine-tuning on the MI355X engine.

Mirrors the surface of reference finetuning/llava/train/train.py: the three argument groups
(:58-166), preprocess_multimodal (:378-403), preprocess_v1 (:722-798), preprocess_plain (:882-901),
preprocess_qwen (:560-633, ChatML masking), preprocess (:904-952), LazySupervisedDataset (:955-1239),
DataCollatorForSupervisedDataset (:1243-1286), make_supervised_data_module (:1289-1293), find_all_linear_names
(:242-255), the tunable-parts policy (:1613-1665) and train() (:1449-1725).  Flags that configure machinery this
build replaces (DeepSpeed, torch.compile, bits/quantisation) are accepted and ignored; --gradient_checkpointing selects the
engine's activation-recompute policy ("auto": layers recompute only when their activations would not fit in HBM).
"""
import argparse
import copy
import dataclasses
import json
import math
import os
import random
import re
import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import torch
from PIL import Image
from torch.utils.data import Dataset

from .. import conversation as conversation_lib
from ..constants import DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IMAGE_TOKEN, IGNORE_INDEX, IMAGE_TOKEN_INDEX
from ..mm_utils import (ClipImageProcessor, expand2square, process_anyres_image, process_highres_image,
                        process_highres_image_crop_split, tokenizer_image_token)
from .llava_trainer import LLaVATrainer


@dataclass
class ModelArguments:
    model_name_or_path: Optional[str] = "facebook/opt-125m"
    model_class_name: Optional[str] = None
    mm_tunable_parts: Optional[str] = None
    version: Optional[str] = "v0"
    freeze_backbone: bool = False
    tune_mm_mlp_adapter: bool = False
    tune_mm_vision_resampler: bool = False
    vision_tower: Optional[str] = None
    vision_tower_pretrained: Optional[str] = None
    unfreeze_mm_vision_tower: bool = False
    unfreeze_language_model: bool = False
    mm_vision_select_layer: Optional[int] = -1
    pretrain_mm_mlp_adapter: Optional[str] = None
    mm_projector_type: Optional[str] = "linear"
    mm_use_im_start_end: bool = False
    mm_use_im_patch_token: bool = True
    mm_patch_merge_type: Optional[str] = "flat"
    mm_vision_select_feature: Optional[str] = "patch"
    mm_resampler_type: Optional[str] = None
    mm_spatial_pool_stride: Optional[int] = None
    mm_spatial_pool_mode: str = "bilinear"
    rope_scaling_factor: Optional[float] = None
    rope_scaling_type: Optional[str] = None
    use_pos_skipping: Optional[bool] = False
    pos_skipping_range: Optional[int] = 4096
    mm_newline_position: Optional[str] = "grid"
    delay_load: Optional[bool] = True
    add_faster_video: Optional[bool] = False
    faster_token_stride: Optional[int] = 10
    geometry: Optional[str] = None  # build-specific: named geometry of radvlm_amd.config.GEOMETRIES (random init)


@dataclass
class DataArguments:
    data_path: Optional[str] = None
    lazy_preprocess: bool = False
    is_multimodal: bool = False
    early_mix_text: bool = False
    image_folder: Optional[str] = None
    image_aspect_ratio: str = "square"
    image_grid_pinpoints: Optional[str] = None
    image_crop_resolution: Optional[int] = None
    image_split_resolution: Optional[int] = None
    video_folder: Optional[str] = None
    video_fps: Optional[int] = 1
    frames_upbound: Optional[int] = 0
    add_time_instruction: Optional[bool] = False
    force_sample: Optional[bool] = False
    device_image_normalize: bool = False   # build-specific: the dataset hands uint8 pixels over, rescale / normalise / layout run on the GPU


@dataclass
class TrainingArguments:
    """The subset of transformers.TrainingArguments the reference script uses (finetune_radio_7b.sh:45-89) plus the
    reference's own additions (train.py:137-166)."""
    output_dir: str = "./checkpoints"
    cache_dir: Optional[str] = None
    optim: str = "adamw_torch"
    remove_unused_columns: bool = False
    freeze_mm_mlp_adapter: bool = False
    freeze_mm_vision_resampler: bool = False
    model_max_length: int = 4096
    bits: int = 16
    double_quant: bool = True
    quant_type: str = "nf4"
    lora_enable: bool = False
    lora_r: int = 64
    lora_alpha: int = 16
    lora_dropout: float = 0.05
    lora_weight_path: str = ""
    lora_bias: str = "none"
    mm_projector_lr: Optional[float] = None
    mm_vision_tower_lr: Optional[float] = None
    group_by_length: bool = False
    group_by_varlen: bool = False
    group_by_modality_length: bool = False
    group_by_modality_length_auto: bool = False
    gradient_checkpointing: bool = True
    verbose_logging: bool = False
    attn_implementation: str = "flash_attention_2"
    bf16: bool = True
    tf32: bool = True
    fp16: bool = False
    num_train_epochs: float = 1.0
    max_steps: int = -1
    per_device_train_batch_size: int = 1
    per_device_eval_batch_size: int = 1
    gradient_accumulation_steps: int = 1
    learning_rate: float = 2e-5
    weight_decay: float = 0.0
    adam_beta1: float = 0.9
    adam_beta2: float = 0.999
    adam_epsilon: float = 1e-8
    max_grad_norm: float = 1.0
    warmup_ratio: float = 0.03
    warmup_steps: int = 0
    lr_scheduler_type: str = "linear"    # HF TrainingArguments default; finetune_radio_7b.sh passes cosine
    logging_steps: int = 1
    save_strategy: str = "steps"
    save_steps: int = 500
    save_total_limit: Optional[int] = None
    save_only_model: bool = False        # HF TrainingArguments.save_only_model: checkpoints without optimizer state (a 7B AdamW state is 81 GB)
    evaluation_strategy: str = "no"
    dataloader_num_workers: int = 4
    dataloader_drop_last: bool = False
    report_to: str = "none"
    run_name: Optional[str] = None
    seed: int = 42
    deepspeed: Optional[str] = None      # accepted, ignored: ZeRO-3 sharding is replaced by plain DP
    torch_compile: bool = False          # accepted, ignored
    torch_compile_backend: Optional[str] = None
    local_rank: int = -1
    world_size: int = 1
    process_index: int = 0


def parse_args_into_dataclasses(argv=None):
    """HfArgumentParser-style CLI: every dataclass field is a --flag; unknown flags are an error like the reference."""
    groups = (ModelArguments, DataArguments, TrainingArguments)
    ap = argparse.ArgumentParser()
    for g in groups:
        for f in dataclasses.fields(g):
            t = f.type
            base = str(t)
            if "bool" in base:
                ap.add_argument(f"--{f.name}", type=lambda s: str(s).lower() in ("1", "true", "yes"), nargs="?", const=True, default=f.default)
            elif "int" in base and "float" not in base:
                ap.add_argument(f"--{f.name}", type=int, default=f.default)
            elif "float" in base:
                ap.add_argument(f"--{f.name}", type=float, default=f.default)
            else:
                ap.add_argument(f"--{f.name}", type=str, default=f.default)
    ns = vars(ap.parse_args(argv))
    return tuple(g(**{f.name: ns[f.name] for f in dataclasses.fields(g)}) for g in groups)


def find_all_linear_names(model):
    """LoRA targets: every LM linear except lm_head and the multimodal modules (train.py:242-255)."""
    skip = ("mm_projector", "vision_tower", "vision_resampler")
    names = set()
    for name in model.engine.lm.names():
        if any(k in name for k in skip) or not name.endswith("_proj.weight"):
            continue
        names.add(name.split(".")[-2])
    names.discard("lm_head")
    return sorted(names)


# ---------------------------------------------------------------------------------------------- preprocessing
def preprocess_multimodal(sources, data_args):
    """Move a single '<image>' to the front of its turn as '<image>\\n...' (train.py:378-403)."""
    if not data_args.is_multimodal:
        return sources
    for source in sources:
        for s in source:
            v = s["value"]
            if len(re.findall(DEFAULT_IMAGE_TOKEN, v)) == 1 and not v.startswith(DEFAULT_IMAGE_TOKEN):
                v = (DEFAULT_IMAGE_TOKEN + "\n" + v.replace(DEFAULT_IMAGE_TOKEN, "").strip()).strip()
            if getattr(data_args, "mm_use_im_start_end", False):
                v = v.replace(DEFAULT_IMAGE_TOKEN, DEFAULT_IM_START_TOKEN + DEFAULT_IMAGE_TOKEN + DEFAULT_IM_END_TOKEN)
            s["value"] = v.replace("QA_GT_caption_based_noisy", "")
    return sources


def _tok_len(text, tokenizer, has_image):
    return len(tokenizer_image_token(text, tokenizer)) if has_image else len(tokenizer(text).input_ids)


def preprocess_v1(sources, tokenizer, has_image=False):
    """Vicuna-v1 template; loss only on assistant replies (instruction spans and BOS masked)."""
    conv = conversation_lib.default_conversation.copy()
    roles = {"human": conv.roles[0], "gpt": conv.roles[1]}
    prompts = []
    for source in sources:
        if roles[source[0]["from"]] != conv.roles[0]:
            source = source[1:]
        conv.messages = []
        for j, s in enumerate(source):
            assert roles[s["from"]] == conv.roles[j % 2]
            conv.append_message(roles[s["from"]], s["value"])
        prompts.append(conv.get_prompt())
    if has_image:
        input_ids = torch.stack([tokenizer_image_token(p, tokenizer, return_tensors="pt") for p in prompts], 0)
    else:
        input_ids = tokenizer(prompts, return_tensors="pt", padding="longest", max_length=tokenizer.model_max_length, truncation=True).input_ids
    targets = input_ids.clone()
    assert conv.sep_style == conversation_lib.SeparatorStyle.TWO
    sep = conv.sep + conv.roles[1] + ": "
    shrink = (not getattr(tokenizer, "legacy", True))
    for prompt, target in zip(prompts, targets):
        total = int(target.ne(tokenizer.pad_token_id).sum())
        cur = 1
        target[:cur] = IGNORE_INDEX
        for i, rnd in enumerate(prompt.split(conv.sep2)):
            if rnd == "":
                break
            parts = rnd.split(sep)
            if len(parts) != 2:
                break
            round_len = _tok_len(rnd, tokenizer, has_image)
            instr_len = _tok_len(parts[0] + sep, tokenizer, has_image) - 2
            if i != 0 and shrink:
                round_len -= 1
                instr_len -= 1
            target[cur:cur + instr_len] = IGNORE_INDEX
            cur += round_len
        target[cur:] = IGNORE_INDEX
        if cur < tokenizer.model_max_length and cur != total:
            target[:] = IGNORE_INDEX
            print(f"WARNING: tokenization mismatch: {cur} vs. {total}. (ignored)")
    return dict(input_ids=input_ids, labels=targets)


def preprocess_plain(sources, tokenizer):
    """Pretraining template: '<image>' + caption + sep; the image token is masked."""
    prompts = []
    for source in sources:
        assert len(source) == 2 and DEFAULT_IMAGE_TOKEN in source[0]["value"]
        source[0]["value"] = DEFAULT_IMAGE_TOKEN
        prompts.append(source[0]["value"] + source[1]["value"] + conversation_lib.default_conversation.sep)
    input_ids = [tokenizer_image_token(p, tokenizer, return_tensors="pt") for p in prompts]
    targets = copy.deepcopy(input_ids)
    for t, source in zip(targets, sources):
        t[:len(tokenizer_image_token(source[0]["value"], tokenizer))] = IGNORE_INDEX
    return dict(input_ids=input_ids, labels=targets)


def preprocess_qwen(sources, tokenizer, has_image=False, max_len=2048, system_message="You are a helpful assistant."):
    """ChatML supervision of train/train.py:560-633: every turn is rendered as '<|im_start|>{role}\n{content}<|im_end|>\n'
    (the chat template the reference installs, :579) and tokenised without further special tokens; system and user turns are
    masked, assistant turns supervised in full (header included); then -- as the reference does (:575, :620-622) -- every
    token whose id is 198 ('\n' in the Qwen2 vocabulary), <|im_start|> or <|im_end|> is un-masked wherever it occurs, and
    '<image>' ids become IMAGE_TOKEN_INDEX.  A leading non-human turn is dropped (:592-593)."""
    roles = {"human": "user", "gpt": "assistant"}
    tokenizer = copy.deepcopy(tokenizer)
    if has_image:
        tokenizer.add_tokens(["<image>"], special_tokens=True)
    image_token_index = tokenizer.convert_tokens_to_ids("<image>")
    extra = getattr(tokenizer, "additional_special_tokens_ids", None)    # transformers 4.x attribute (:573)
    if extra is None:
        extra = tokenizer.convert_tokens_to_ids(["<|im_start|>", "<|im_end|>"])
    im_start, im_end = extra
    unmask = (198, im_start, im_end)

    def encode(role, content):
        return list(tokenizer("<|im_start|>" + role + "\n" + content + "<|im_end|>" + "\n", add_special_tokens=False).input_ids)

    all_ids, all_tgt = [], []
    for source in sources:
        if roles[source[0]["from"]] != roles["human"]:
            source = source[1:]
        ids = encode("system", system_message)
        tgt = [IGNORE_INDEX] * len(ids)
        for conv in source:
            try:
                role, content = conv["role"], conv["content"]
            except KeyError:
                role, content = conv["from"], conv["value"]
            role = roles.get(role, role)
            enc = encode(role, content)
            ids += enc
            tgt += [IGNORE_INDEX] * len(enc) if role in ("user", "system") else enc
        assert len(ids) == len(tgt), f"{len(ids)} != {len(tgt)}"
        for i, e in enumerate(ids):
            if e in unmask:
                tgt[i] = e
            if e == image_token_index:
                ids[i] = IMAGE_TOKEN_INDEX
        all_ids.append(ids)
        all_tgt.append(tgt)
    return dict(input_ids=torch.tensor(all_ids, dtype=torch.long), labels=torch.tensor(all_tgt, dtype=torch.long))


def preprocess(sources, tokenizer, has_image=False):
    conv = conversation_lib.default_conversation
    if conv.sep_style == conversation_lib.SeparatorStyle.PLAIN:
        return preprocess_plain(sources, tokenizer)
    if conv.version.startswith("v1"):
        return preprocess_v1(sources, tokenizer, has_image=has_image)
    if conv.version == "qwen":
        return preprocess_qwen(sources, tokenizer, has_image=has_image)
    raise NotImplementedError(f"conversation version {conv.version!r} is outside the hot path (SURVEY.md section 8)")


# ---------------------------------------------------------------------------------------------- dataset + collator
def _load_records(data_path, data_args):
    """json | brace-glob '/p/{a,b}.json' | yaml {datasets: [{json_path, sampling_strategy}]} (train.py:961-1030)."""
    def read(p):
        with open(p) as f:
            return [json.loads(l) for l in f] if p.endswith(".jsonl") else json.load(f)
    m = re.match(r"^(.*)\{(.*)\}\.json$", data_path) if ("{" in data_path and "}" in data_path) else None
    if m:
        paths = [f"{m.group(1)}{n}.json" for n in m.group(2).split(",")]
        data_args.dataset_paths = paths
        return [r for p in paths for r in read(p)]
    if data_path.endswith(".yaml"):
        import yaml
        with open(data_path) as f:
            sets = yaml.safe_load(f).get("datasets")
        data_args.dataset_paths = [d.get("json_path") for d in sets]
        out = []
        for d in sets:
            cur = read(d["json_path"])
            strat, num = d.get("sampling_strategy", "all"), None
            if ":" in strat:
                strat, num = strat.split(":")
                num = math.ceil(int(num.split("%")[0]) * len(cur) / 100) if "%" in num else int(num)
            if num is not None:
                if strat == "first":
                    cur = cur[:num]
                elif strat == "end":
                    cur = cur[-num:]
                elif strat == "random":
                    random.shuffle(cur)
                    cur = cur[:num]
            out.extend(cur)
        return out
    data_args.dataset_paths = [data_path]
    return read(data_path)


class LazySupervisedDataset(Dataset):
    """LLaVA-format records {image?, conversations:[{from,value}], id} -> {input_ids, labels, image:[(tensor,size,modality)], id}."""

    def __init__(self, data_path, tokenizer, data_args):
        self.tokenizer, self.data_args = tokenizer, data_args
        self.list_data_dict = _load_records(data_path, data_args)

    def __len__(self):
        return len(self.list_data_dict)

    @staticmethod
    def _words(sample):
        return sum(len(c["value"].split()) for c in sample["conversations"])

    @property
    def lengths(self):
        return [self._words(s) + (128 if "image" in s else 0) for s in self.list_data_dict]

    @property
    def modality_lengths(self):
        out = []
        for s in self.list_data_dict:
            n = self._words(s)
            assert n > 0, f"Conversation length is 0 for {s}"
            out.append(n if ("image" in s or "video" in s or self.data_args.early_mix_text) else -n)
        return out

    def process_image(self, image_file, overwrite_image_aspect_ratio=None):
        proc = self.data_args.image_processor
        image = Image.open(os.path.join(self.data_args.image_folder or "", image_file)).convert("RGB")
        size = image.size
        aspect = overwrite_image_aspect_ratio or self.data_args.image_aspect_ratio
        if aspect == "anyres" or "anyres_max" in aspect:
            t = process_anyres_image(image, proc, self.data_args.image_grid_pinpoints)
        elif aspect == "pad":
            t = proc.preprocess(expand2square(image, tuple(int(x * 255) for x in proc.image_mean)), return_tensors="pt")["pixel_values"][0]
        elif aspect == "highres":
            t = process_highres_image(image, proc, self.data_args.image_grid_pinpoints)
        elif aspect == "crop_split":
            t = process_highres_image_crop_split(image, self.data_args)
        else:
            t = proc.preprocess(image, return_tensors="pt")["pixel_values"][0]
        return t, size, "image"

    def __getitem__(self, i):
        """Retry policy of the reference (train.py:1101-1132): 3 tries (1 s apart), 3 tries on the next sample, then raise."""
        for _ in range(3):
            try:
                return self._get_item(i)
            except Exception as e:  # noqa: BLE001
                print(f"Failed to fetch sample {i}. Exception:", e)
                time.sleep(1)
        nxt = min(i + 1, len(self.list_data_dict) - 1)
        for _ in range(3):
            try:
                return self._get_item(nxt)
            except Exception as e:  # noqa: BLE001
                print(f"Failed to fetch sample {nxt}. Exception:", e)
        return self._get_item(i)

    def _get_item(self, i):
        rec = self.list_data_dict[i]
        image = None
        if "image" in rec:
            f = rec["image"]
            if isinstance(f, list):
                image = [self.process_image(x, "pad" if len(f) > 1 else None) for x in f]
            else:
                image = [self.process_image(f)]
            sources = preprocess_multimodal(copy.deepcopy([rec["conversations"]]), self.data_args)
        else:
            sources = copy.deepcopy([rec["conversations"]])
        d = preprocess(sources, self.tokenizer, has_image="image" in rec)
        out = dict(input_ids=d["input_ids"][0], labels=d["labels"][0])
        if image is not None:
            out["image"] = image
        elif self.data_args.is_multimodal:
            cs = self.data_args.image_processor.crop_size
            if getattr(self.data_args.image_processor, "device_normalize", False):      # uint8 hand-over: the dummy image too
                out["image"] = [(torch.zeros(1, cs["height"], cs["width"], 3, dtype=torch.uint8), (cs["width"], cs["height"]), "text")]
            else:
                out["image"] = [(torch.zeros(1, 3, cs["height"], cs["width"]), (cs["width"], cs["height"]), "text")]
        out["id"] = rec.get("id", i)
        return out


@dataclass
class DataCollatorForSupervisedDataset:
    tokenizer: object

    def _pad(self, seqs, value):
        left = getattr(self.tokenizer, "padding_side", "right") == "left"
        n = max(s.shape[0] for s in seqs)
        out = torch.full((len(seqs), n), value, dtype=seqs[0].dtype)
        for r, s in enumerate(seqs):
            if left:
                out[r, n - s.shape[0]:] = s
            else:
                out[r, :s.shape[0]] = s
        return out

    def __call__(self, instances: Sequence[Dict]) -> Dict:
        L = self.tokenizer.model_max_length
        ids = [x["input_ids"][:L] for x in instances]
        labs = [x["labels"][:L] for x in instances]
        if self.tokenizer.pad_token_id is None:
            self.tokenizer.pad_token_id = 0
        input_ids = self._pad(ids, self.tokenizer.pad_token_id)
        labels = self._pad(labs, IGNORE_INDEX).long()
        batch = dict(input_ids=input_ids, labels=labels, attention_mask=input_ids.ne(self.tokenizer.pad_token_id))
        if "image" in instances[0]:
            flat = [im for x in instances for im in x["image"]]
            batch["image_sizes"] = [im[1] for im in flat]
            batch["modalities"] = [im[2] for im in flat]
            batch["images"] = [im[0] for im in flat]
        if "prompt" in instances[0]:
            batch["prompts"] = [x["prompt"] for x in instances]
        return batch


def make_supervised_data_module(tokenizer, data_args):
    ds = LazySupervisedDataset(tokenizer=tokenizer, data_path=data_args.data_path, data_args=data_args)
    return dict(train_dataset=ds, eval_dataset=None, data_collator=DataCollatorForSupervisedDataset(tokenizer=tokenizer))


# ---------------------------------------------------------------------------------------------- train()
def tunable_parts(model_args):
    """Which parts receive gradients (train.py:1613-1665). Returns a set of {'mm_mlp_adapter','mm_language_model',
    'mm_vision_tower'}; default (mm_tunable_parts unset, nothing frozen) = projector + language model."""
    if model_args.mm_tunable_parts:
        return {p.strip() for p in model_args.mm_tunable_parts.split(",")}
    parts = {"mm_mlp_adapter", "mm_language_model"}
    if model_args.tune_mm_mlp_adapter:
        parts = {"mm_mlp_adapter"}
    if model_args.freeze_backbone:
        parts.discard("mm_language_model")
    if model_args.unfreeze_mm_vision_tower:
        parts.add("mm_vision_tower")
    return parts


def resolve_model_sources(model_args, verbose=False):
    """Where the weights come from and which geometry they imply (host-only; no device is touched).

    get_model -> from_pretrained (train/train.py:1358-1427) and CLIPVisionTower.load_model (clip_encoder.py:35-44): the LM and the
    tower start from SAVED weights.  There is no network here, so --model_name_or_path and --vision_tower must be local HF-format
    directories (config.json + safetensors / bin shards); anything else raises.  The build-specific --geometry NAME switch instead
    selects a named geometry with RANDOM weights (benchmarks, smoke runs) and loads nothing.
    Returns (geometry, lm_dir, tower_dir, is_qwen, true_vocab); true_vocab != geometry vocab when the checkpoint's tables have a row
    count that is not a multiple of 8 (extra tokens were added): the engine is built at the multiple below and then grown."""
    from ...config import GEOMETRIES
    mname = (model_args.model_name_or_path or "").lower()
    # model-class routing of get_model (train/train.py:1366-1436): 'qwen' in the name -> LlavaQwenForCausalLM, else Llama
    is_qwen = "qwen" in mname or (model_args.geometry or "").find("qwen") >= 0
    lm_dir = tower_dir = None
    if model_args.geometry:
        geometry = GEOMETRIES[model_args.geometry]
        if verbose:
            print(f"[radvlm_amd] --geometry {model_args.geometry}: random-init weights, no checkpoint is loaded", flush=True)
    else:
        from ...checkpoint_io import lm_geometry_from_config, read_config, vision_geometry_from_config
        lm_dir = model_args.model_name_or_path
        if not (lm_dir and os.path.isdir(lm_dir)):
            raise FileNotFoundError(f"--model_name_or_path {lm_dir!r} is not a local checkpoint directory (nothing is downloaded; "
                                    "pass --geometry NAME for a random-init run)")
        lm_cfg = read_config(lm_dir)
        is_qwen = is_qwen or "qwen" in (lm_cfg.get("model_type") or "").lower()
        tower_dir = model_args.vision_tower or lm_cfg.get("mm_vision_tower")
        if "mm_vision_geometry" in lm_cfg:            # a checkpoint written by this package carries its tower inside
            vgeo = dict(lm_cfg["mm_vision_geometry"])
            tower_dir = tower_dir if (tower_dir and os.path.isdir(tower_dir)) else None
        else:
            if not (tower_dir and os.path.isdir(tower_dir)):
                raise FileNotFoundError(f"--vision_tower {tower_dir!r} is not a local checkpoint directory (nothing is downloaded)")
            vgeo = vision_geometry_from_config(read_config(tower_dir))
        geometry = {"vision": vgeo, "lm": lm_geometry_from_config(lm_cfg)}
    true_vocab = geometry["lm"]["vocab"]
    if true_vocab % 8:
        geometry = {"vision": geometry["vision"], "lm": dict(geometry["lm"], vocab=true_vocab // 8 * 8)}
    return geometry, lm_dir, tower_dir, is_qwen, true_vocab


def train(attn_implementation=None, argv=None, tokenizer=None):
    from ..model import LlavaConfig, LlavaLlamaForCausalLM, LlavaQwenConfig, LlavaQwenForCausalLM
    from ..mm_utils import SigLipImageProcessor
    from ...config import GEOMETRIES
    model_args, data_args, training_args = parse_args_into_dataclasses(argv)
    world, rank, local = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
    training_args.world_size, training_args.process_index, training_args.local_rank = world, rank, local
    torch.cuda.set_device(local)
    pg = None
    if world > 1:
        # the reference lifts the collective timeout (llava_trainer.py:247-248, InitProcessGroupKwargs(timeout=timedelta(weeks=52))): rank 0
        # writes checkpoints (13.5 GB of weights + 81 GB of optimizer state for the 7B model) while the other ranks wait in a collective
        import datetime
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local),
                                             timeout=datetime.timedelta(weeks=int(os.environ.get("RV_PG_TIMEOUT_WEEKS", 52))))
        pg = torch.distributed.group.WORLD
    if rank == 0 and str(getattr(training_args, "report_to", "none")).lower() not in ("none", "[]", ""):
        # the recipe passes --report_to wandb (finetune_radio_7b.sh:83); there is no wandb (and no network) here
        print(f"[train] --report_to {training_args.report_to}: no experiment tracker is attached; the per-step record (loss, learning_rate, "
              "grad_norm, epoch, step_time_s, data_wait_s) is printed by rank 0 and kept in trainer_state.json", flush=True)
    parts = tunable_parts(model_args)
    known = {"mm_mlp_adapter", "mm_language_model", "mm_vision_tower"}
    if "mm_vision_resampler" in parts:
        raise NotImplementedError("mm_vision_resampler: no resampler is on the hot path (SURVEY section 2: out of scope)")
    if not parts & known:
        raise ValueError(f"mm_tunable_parts {sorted(parts)}: nothing to train (expected a subset of {sorted(known)})")
    # any subset of the three parts (train/train.py:1613-1665): the LM either lives in the trainable flat buffer or is a frozen store; the
    # tower joins the buffer when tunable; a projector that is not named stays in the buffer but frozen
    frozen_lm = "mm_language_model" not in parts
    projector_only = parts == {"mm_mlp_adapter"}      # tune_mm_mlp_adapter / mm_tunable_parts="mm_mlp_adapter": the pretraining stage
    lora = None
    if training_args.lora_enable:   # peft LoraConfig(r, lora_alpha, lora_dropout, bias="none") on every LM linear (train.py:1515-1532)
        if training_args.lora_bias != "none" or "mm_vision_tower" in parts:
            raise NotImplementedError("lora_bias != 'none' / LoRA together with a tunable vision tower")
        lora = dict(r=training_args.lora_r, alpha=training_args.lora_alpha, dropout=training_args.lora_dropout)
    geometry, lm_dir, tower_dir, is_qwen, true_vocab = resolve_model_sources(model_args, verbose=rank == 0)
    Config, Model = (LlavaQwenConfig, LlavaQwenForCausalLM) if is_qwen else (LlavaConfig, LlavaLlamaForCausalLM)
    cfg = Config(geometry=geometry, mm_patch_merge_type=model_args.mm_patch_merge_type,
                      image_aspect_ratio=data_args.image_aspect_ratio, image_grid_pinpoints=data_args.image_grid_pinpoints,
                      tokenizer_model_max_length=training_args.model_max_length,
                      unfreeze_mm_vision_tower="mm_vision_tower" in parts, lora=lora, freeze_lm=frozen_lm and not lora,
                      freeze_mm_mlp_adapter="mm_mlp_adapter" not in parts,
                      train_embed_tokens=projector_only and model_args.mm_use_im_start_end)
    cfg._name_or_path = model_args.model_name_or_path
    model = Model(cfg, device=f"cuda:{local}", process_group=pg, init="fast")
    # --gradient_checkpointing (reference default True, train.py:164; it wraps every decoder layer in torch checkpointing, :1505-1513):
    # here the engine re-runs a layer's forward inside backward only for as many layers as the batch's activations exceed free HBM
    model.engine.recompute = "auto" if training_args.gradient_checkpointing else False
    if rank == 0 and training_args.gradient_checkpointing:
        print("[train] gradient_checkpointing: activation recompute is decided per batch from free device memory "
              "(288 GB HBM keep a 32 x 704-token batch of the 7B model resident: 0 layers recomputed)", flush=True)
    if true_vocab != geometry["lm"]["vocab"]:
        model.engine.resize_token_embeddings(true_vocab)
    if lm_dir:
        from ...checkpoint_io import load_pretrained
        load_pretrained(model.engine, lm_path=lm_dir, tower_path=tower_dir)     # raises if any LM / tower tensor is missing
    # DDP broadcasts rank 0's parameters when it wraps the model (HF Trainer / accelerate): here every rank initialised or
    # loaded its own copy -- one broadcast makes the replicas identical by construction, the check below proves it from then on
    model.engine.broadcast_parameters()
    model.config.use_cache = False
    model.get_model().initialize_vision_modules(model_args)
    if model_args.version in conversation_lib.conv_templates:
        conversation_lib.default_conversation = conversation_lib.conv_templates[model_args.version]
    else:
        conversation_lib.default_conversation = conversation_lib.conv_templates["vicuna_v1"]
    if tokenizer is None:
        import transformers
        tokenizer = transformers.AutoTokenizer.from_pretrained(model_args.model_name_or_path, cache_dir=training_args.cache_dir,
                                                                model_max_length=training_args.model_max_length, padding_side="right", use_fast=False)
    model.config.tokenizer_padding_side = model.engine.padding_side = getattr(tokenizer, "padding_side", "right")   # train/train.py:1600
    # train/train.py:1678-1679: optional extra tokens grow the embedding tables (needs a tokenizer with add_tokens / __len__;
    # the RadVLM script passes --mm_use_im_patch_token False and no start/end tokens, so this is normally a no-op)
    model.config.mm_use_im_patch_token = model_args.mm_use_im_patch_token
    if hasattr(tokenizer, "add_tokens") and (model_args.mm_use_im_start_end or model_args.mm_use_im_patch_token):
        model.initialize_vision_tokenizer(model_args, tokenizer=tokenizer)
    side = cfg.geometry["vision"]["image"]
    if cfg.geometry["vision"].get("kind") == "siglip":
        data_args.image_processor = SigLipImageProcessor(size=(side, side), crop_size={"height": side, "width": side})
    else:
        data_args.image_processor = ClipImageProcessor(size=side)
    data_args.image_processor.device_normalize = bool(data_args.device_image_normalize)
    model.engine.image_mean, model.engine.image_std = tuple(data_args.image_processor.image_mean), tuple(data_args.image_processor.image_std)
    data_args.is_multimodal = True
    data_args.mm_use_im_start_end = model_args.mm_use_im_start_end
    module = make_supervised_data_module(tokenizer=tokenizer, data_args=data_args)
    # the switches the trainer's checkpoint policy reads (train/train.py:1578-1611 copies them onto training_args)
    training_args.tune_mm_mlp_adapter = model_args.tune_mm_mlp_adapter
    training_args.mm_tunable_parts = model_args.mm_tunable_parts
    training_args.use_im_start_end = model_args.mm_use_im_start_end
    trainer = LLaVATrainer(model=model, tokenizer=tokenizer, args=training_args, **module)
    # auto-resume like the reference (train/train.py:1699-1702): continue from the newest checkpoint-* of output_dir
    has_ckpt = bool(training_args.output_dir) and os.path.isdir(training_args.output_dir) and any(
        d.startswith("checkpoint-") for d in os.listdir(training_args.output_dir))
    state = trainer.train(resume_from_checkpoint=True if has_ckpt else None)
    if rank == 0 and training_args.output_dir:
        os.makedirs(training_args.output_dir, exist_ok=True)
        if trainer._adapter_only() and not lora:
            # safe_save_model_for_hf_trainer (train/train.py:258-289): projector-only runs save mm_projector.bin alone
            keys = ["mm_projector", "vision_resampler"] + (["embed_tokens", "embed_in"] if model_args.mm_use_im_start_end else [])
            torch.save({k: v.detach().clone().cpu() for k, v in model.state_dict().items() if any(m in k for m in keys)},
                       os.path.join(training_args.output_dir, "mm_projector.bin"))
            model.save_config(training_args.output_dir)
        else:
            model.save_pretrained(training_args.output_dir)      # LoRA: adapter + non_lora_trainables.bin (train/train.py:1708-1717)
    if world > 1:
        torch.distributed.destroy_process_group()
    return state


if __name__ == "__main__":
    train()
