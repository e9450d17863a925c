"""``LlavaConfig`` / ``LlavaLlamaModel`` / ``LlavaLlamaForCausalLM`` over the MI355X engine.

Mirrors the boundary class of reference finetuning/llava/model/language_model/llava_llama.py:35-156 and the
LlavaMetaModel / LlavaMetaForCausalLM glue of model/llava_arch.py:36-124, 192-196, 251-555:
same ``forward`` keyword list, ``.loss`` (fp32 scalar, mean CE over non-ignored shifted labels) and ``.logits``
(fp32 [b,S,V]) on the output, ``get_model()``, ``get_vision_tower()``, ``initialize_vision_modules`` and
``initialize_vision_tokenizer``, reference state-dict names.  ``out.loss.backward()`` runs the engine's hand-written
backward (one autograd node for the whole step), so an HF-Trainer-style ``training_step`` drives it unchanged.
"""
import os
from types import SimpleNamespace

import numpy as np
import torch

from ...engine import LlavaEngine


class LlavaConfig:
    model_type = "llava_llama"

    def __init__(self, geometry=None, mm_patch_merge_type="flat", image_aspect_ratio="square", image_grid_pinpoints=None,
                 tokenizer_model_max_length=None, rms_norm_eps=1e-5, rope_theta=10000.0, unfreeze_mm_vision_tower=False, lora=None, **kw):
        from ...config import GEOMETRIES
        self.geometry = geometry or GEOMETRIES["llava15_7b"]
        l, v = self.geometry["lm"], self.geometry["vision"]
        self.hidden_size, self.intermediate_size = l["d"], l["ffn"]
        self.num_hidden_layers, self.num_attention_heads, self.num_key_value_heads = l["layers"], l["heads"], l["heads"]
        self.vocab_size = l["vocab"]
        self.rms_norm_eps, self.rope_theta = rms_norm_eps, rope_theta
        self.mm_hidden_size = v["d"]
        self.mm_projector_type = "mlp2x_gelu"
        self.mm_vision_select_layer = -2
        self.mm_vision_select_feature = "patch"
        self.mm_patch_merge_type = mm_patch_merge_type
        self.image_aspect_ratio = image_aspect_ratio
        self.image_grid_pinpoints = image_grid_pinpoints
        self.tokenizer_model_max_length = tokenizer_model_max_length
        self.tokenizer_padding_side = "right"
        self.use_cache = False
        self.use_mm_proj = True
        self.unfreeze_mm_vision_tower = unfreeze_mm_vision_tower
        self.lora = lora   # {'r', 'alpha', 'dropout'} or None
        for k, val in kw.items():
            setattr(self, k, val)


class CausalLMOutputWithPast(SimpleNamespace):
    def __getitem__(self, i):
        return (self.loss, self.logits)[i]


class _StepFunction(torch.autograd.Function):
    """One autograd node for the whole training step: forward saved its context inside the engine."""

    @staticmethod
    def forward(ctx, flat_params, engine, loss):
        ctx.engine = engine
        return loss.clone().view(())

    @staticmethod
    def backward(ctx, grad_out):
        eng = ctx.engine
        # d(loss)/d(params) is accumulated into engine.grads by the hand-written backward; the incoming scalar
        # multiplier (1 for plain loss.backward()) must be 1 -- loss scaling goes through forward(loss_scale=...).
        eng.backward()
        return None, None, None


class CLIPVisionTower:
    """Handle with the attributes the reference reads off its tower (clip_encoder.py:81-122)."""

    def __init__(self, engine):
        self._e = engine
        self.is_loaded = True
        self.select_layer, self.select_feature = -2, "patch"

    @property
    def config(self):
        v = self._e.v
        return SimpleNamespace(hidden_size=v["d"], image_size=v["image"], patch_size=v["patch"], num_hidden_layers=v["layers"])

    hidden_size = property(lambda s: s._e.v["d"])
    image_size = property(lambda s: s._e.v["image"])
    num_patches_per_side = property(lambda s: s._e.side)
    num_patches = property(lambda s: s._e.P)
    dtype = torch.bfloat16
    device = property(lambda s: s._e.device)

    def load_model(self, device_map=None):
        return None

    def __call__(self, images):
        """[n,3,H,W] -> [n,P,dv] patch features of hidden_states[-2] (clip_encoder.py:68-79)."""
        from ... import ops
        e = self._e
        x = images.to(e.device)
        x = x if x.dtype == torch.bfloat16 else ops.to_bf16(x.float())
        h = e.vision_forward(x.contiguous()).view(x.shape[0], e.P + 1, e.v["d"])
        return h[:, 1:]


class LlavaLlamaModel:
    def __init__(self, engine, config):
        self.engine, self.config = engine, config
        self.vision_tower = CLIPVisionTower(engine)

    def get_vision_tower(self):
        return self.vision_tower

    @property
    def embed_tokens(self):
        return self.engine.W("model.embed_tokens.weight")

    @property
    def mm_projector(self):
        e = self.engine
        return {n: e.W(f"model.mm_projector.{n}") for n in ("0.weight", "0.bias", "2.weight", "2.bias")}

    def initialize_vision_modules(self, model_args, fsdp=None):
        """Record the mm_* settings on the config (llava_arch.py:54-124); the tower/projector already live in the engine."""
        c = self.config
        c.mm_vision_tower = getattr(model_args, "vision_tower", None)
        c.mm_projector_type = getattr(model_args, "mm_projector_type", "mlp2x_gelu") or "mlp2x_gelu"
        if c.mm_projector_type not in ("mlp2x_gelu",):
            raise NotImplementedError(f"mm_projector_type={c.mm_projector_type}: only mlp2x_gelu is on the hot path")
        c.mm_vision_select_layer = getattr(model_args, "mm_vision_select_layer", -2)
        if c.mm_vision_select_layer != -2:
            raise NotImplementedError("mm_vision_select_layer must be -2 (LLaVA-1.5)")
        c.mm_vision_select_feature = getattr(model_args, "mm_vision_select_feature", "patch")
        c.mm_patch_merge_type = getattr(model_args, "mm_patch_merge_type", c.mm_patch_merge_type)
        p = getattr(model_args, "pretrain_mm_mlp_adapter", None)
        if p:
            w = torch.load(p, map_location="cpu", weights_only=True)
            self.engine.load_state_dict({("model." + k if not k.startswith("model.") else k): v for k, v in w.items() if "mm_projector" in k})


class LlavaLlamaForCausalLM:
    config_class = LlavaConfig
    model_class = LlavaLlamaModel

    def __init__(self, config, device="cuda", process_group=None, init="portable", seed=0):
        self.config = config
        self.engine = LlavaEngine(config.geometry, device=device, merge_type=config.mm_patch_merge_type,
                                  image_aspect_ratio=config.image_aspect_ratio, image_grid_pinpoints=config.image_grid_pinpoints,
                                  max_len=config.tokenizer_model_max_length, init=init, seed=seed, rms_eps=config.rms_norm_eps,
                                  rope_theta=config.rope_theta, process_group=process_group,
                                  train_vision_tower=getattr(config, "unfreeze_mm_vision_tower", False),
                                  lora=getattr(config, "lora", None), freeze_lm=getattr(config, "freeze_lm", False),
                                  train_embed_tokens=getattr(config, "train_embed_tokens", False),
                                  freeze_projector=getattr(config, "freeze_mm_mlp_adapter", False),
                                  padding_side=getattr(config, "tokenizer_padding_side", "right"),
                                  recompute=getattr(config, "activation_recompute", False))
        self.model = self.model_class(self.engine, config)
        self.training = True
        self.is_gradient_checkpointing = bool(self.engine.recompute)
        # a leaf that makes loss require grad so that `.backward()` reaches the engine
        self._anchor = torch.zeros(1, device=self.engine.device, requires_grad=True)

    def get_model(self):
        return self.model

    def gradient_checkpointing_enable(self, gradient_checkpointing_kwargs=None):
        """HF PreTrainedModel.gradient_checkpointing_enable (what Trainer calls for --gradient_checkpointing, reference
        train/train.py:1505-1513).  The reference re-runs EVERY decoder layer's forward in backward; here the flag selects the engine's
        "auto" policy -- the first n layers are recomputed, n taken per batch from free HBM (0 for the BASELINE batch on 288 GB), with
        bit-identical gradients either way.  LlavaEngine(recompute=True) is the explicit recompute-everything switch."""
        if self.engine.recompute is not True:
            self.engine.recompute = "auto"
        self.is_gradient_checkpointing = True

    def gradient_checkpointing_disable(self):
        self.engine.recompute = False
        self.is_gradient_checkpointing = False

    def get_vision_tower(self):
        return self.model.get_vision_tower()

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def parameters(self):
        return [self.engine.lm.flat]

    def state_dict(self):
        return self.engine.state_dict()

    def load_state_dict(self, sd, strict=False):
        return self.engine.load_state_dict(sd, strict=strict)

    def encode_images(self, images):
        e = self.engine
        from ... import ops
        x = images.to(e.device)
        x = x if x.dtype == torch.bfloat16 else ops.to_bf16(x.float())
        return e.encode_images(x.contiguous())[:x.shape[0] * e.P].view(x.shape[0], e.P, e.l["d"])

    def initialize_vision_tokenizer(self, model_args, tokenizer):
        """llava_arch.py:557-597: optional <im_patch> / <im_start>, <im_end> tokens are added to the tokenizer and the embedding
        tables grow by as many rows, initialised to the mean of the existing rows (:563-575).  In the pretraining stage
        (tune_mm_mlp_adapter) the INPUT embeddings then train with the projector while lm_head stays frozen (:577-581: build the
        model with config.train_embed_tokens=True, as train() does), and with pretrain_mm_mlp_adapter the two new embedding rows
        are restored from that file (:583-592)."""
        from ..constants import DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IMAGE_PATCH_TOKEN
        e = self.engine
        if getattr(model_args, "mm_use_im_patch_token", False):
            tokenizer.add_tokens([DEFAULT_IMAGE_PATCH_TOKEN], special_tokens=True)
            e.resize_token_embeddings(max(len(tokenizer), e.vocab))
        if getattr(model_args, "mm_use_im_start_end", False):
            num_new = tokenizer.add_tokens([DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN], special_tokens=True)
            e.resize_token_embeddings(max(len(tokenizer), e.vocab))
            if getattr(model_args, "tune_mm_mlp_adapter", False) and "model.embed_tokens.weight" not in e.lm.offsets:
                raise ValueError("tune_mm_mlp_adapter + mm_use_im_start_end trains the input embeddings (llava_arch.py:577-581): "
                                 "build the model with config.train_embed_tokens=True")
            p = getattr(model_args, "pretrain_mm_mlp_adapter", None)
            if p:
                w = torch.load(p, map_location="cpu", weights_only=True)["model.embed_tokens.weight"]
                assert num_new == 2
                emb = e.W("model.embed_tokens.weight")
                if w.shape[0] in (e.vocab, emb.shape[0]) and w.shape[1] == emb.shape[1]:
                    emb[e.vocab - num_new:e.vocab].copy_(w[e.vocab - num_new:e.vocab].to(emb.dtype))
                elif w.shape[0] == num_new:
                    emb[e.vocab - num_new:e.vocab].copy_(w.to(emb.dtype))
                else:
                    raise ValueError(f"Unexpected embed_tokens_weight shape. Pretrained: {tuple(w.shape)}. Current: {(e.vocab, emb.shape[1])}. "
                                     f"Numer of new tokens: {num_new}.")
                if e.master is not None:
                    from ... import ops
                    e.master.copy_(ops.to_f32(e.lm.flat))
        self.config.vocab_size = e.vocab
        self.config.mm_use_im_start_end = bool(getattr(model_args, "mm_use_im_start_end", False))

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None, labels=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, images=None, image_sizes=None, return_dict=None,
                modalities=("image",), dpo_forward=None, cache_position=None, output_logits=None):
        if past_key_values is not None or dpo_forward:
            raise NotImplementedError("key/value caches and the DPO forward are serving / preference-tuning paths (SURVEY section 2: out of scope)")
        if inputs_embeds is not None:
            # llava_llama.py:83-120 with inputs_embeds given: no multimodal splice, the decoder runs on the embeddings (eval loss, the
            # per-step call of a generation loop without cache); fp32 logits, loss when labels are passed; no backward
            if input_ids is not None:
                raise ValueError("You cannot specify both input_ids and inputs_embeds at the same time")
            loss, logits = self.engine.forward_embeds(inputs_embeds, attention_mask=attention_mask, labels=labels)
            return CausalLMOutputWithPast(loss=loss, logits=logits)
        if images is None:
            raise ValueError("images is required (text-only samples carry a dummy zero image, train.py:1227-1232)")
        imgs = list(images) if not torch.is_tensor(images) else [im for im in images]
        ids = input_ids.cpu().numpy() if torch.is_tensor(input_ids) else input_ids
        am = attention_mask.cpu().numpy() if torch.is_tensor(attention_mask) else attention_mask
        lab = labels.cpu().numpy() if torch.is_tensor(labels) else labels
        if lab is None:          # labels=None (llava_llama.py:69-120 returns logits only): nothing to score
            lab = np.full(np.asarray(ids).shape, -100, dtype=np.int64)
        want_logits = (labels is None or not self.training) if output_logits is None else output_logits
        loss = self.engine.forward(ids, am, lab, imgs, image_sizes=image_sizes, want_logits=want_logits)
        if self.training and labels is not None:
            loss = _StepFunction.apply(self._anchor, self.engine, loss)
        else:
            self.engine.ctx = None
        return CausalLMOutputWithPast(loss=loss if labels is not None else None, logits=self.engine.last_logits)

    __call__ = forward

    def save_config(self, out_dir):
        """config.json in HF's key vocabulary (what model.config.save_pretrained leaves next to the weights), plus the tower
        geometry under 'mm_vision_geometry' so that the directory is loadable as --model_name_or_path on its own."""
        import json
        os.makedirs(out_dir, exist_ok=True)
        c, l = self.config, self.engine.l
        d = {"model_type": c.model_type, "architectures": [type(self).__name__], "hidden_size": l["d"], "intermediate_size": l["ffn"],
             "num_hidden_layers": l["layers"], "num_attention_heads": l["heads"], "num_key_value_heads": l.get("kv_heads", l["heads"]),
             "vocab_size": self.engine.vocab, "rms_norm_eps": self.engine.eps, "rope_theta": self.engine.theta, "torch_dtype": "bfloat16",
             "use_cache": False, "mm_vision_geometry": dict(self.engine.v)}
        for k, v in vars(c).items():
            if k.startswith(("mm_", "image_", "tokenizer_")) and isinstance(v, (str, int, float, bool, list, type(None))):
                d[k] = v
        with open(os.path.join(out_dir, "config.json"), "w") as f:
            json.dump(d, f, indent=2)

    def save_pretrained(self, out_dir):
        """Weights in the reference's formats.  Full fine-tune: model.safetensors under the reference state-dict names.  LoRA
        (train/train.py:1708-1717): what peft's save_pretrained(state_dict=get_peft_state_maybe_zero_3(...)) leaves --
        adapter_model.bin (torch.save; keys base_model.model.<module>.lora_{A,B}.weight, adapter name stripped) and
        adapter_config.json -- plus non_lora_trainables.bin (torch.save of the trainable non-LoRA tensors under their
        base_model.model.* names).  peft itself is not installed here: the two adapter files restate its published layout."""
        from safetensors.torch import save_file
        self.save_config(out_dir)
        cpu = lambda d: {k: v.detach().clone().contiguous().cpu() for k, v in d.items()}
        if self.engine.lora:
            import json
            adapters, others = self.engine.lora_state_dict()
            torch.save(cpu({k.replace(".default.weight", ".weight"): v for k, v in adapters.items()}), os.path.join(out_dir, "adapter_model.bin"))
            torch.save(cpu({"base_model.model." + k: v for k, v in others.items()}), os.path.join(out_dir, "non_lora_trainables.bin"))
            lo = self.engine.lora
            with open(os.path.join(out_dir, "adapter_config.json"), "w") as f:
                json.dump({"peft_type": "LORA", "task_type": "CAUSAL_LM", "base_model_name_or_path": getattr(self.config, "_name_or_path", None),
                           "r": lo["r"], "lora_alpha": lo.get("alpha", 16), "lora_dropout": lo.get("dropout", 0.0), "bias": "none",
                           "target_modules": ["q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj"],
                           "fan_in_fan_out": False, "inference_mode": True, "init_lora_weights": True, "modules_to_save": None,
                           "layers_to_transform": None, "layers_pattern": None, "revision": None}, f, indent=2)
            return
        save_file(cpu(self.engine.state_dict()), os.path.join(out_dir, "model.safetensors"))

    def load_adapter(self, path):
        """Inverse of the LoRA branch of save_pretrained (resume / continued training): adapter_model.bin + non_lora_trainables.bin."""
        e = self.engine
        assert e.lora, "load_adapter needs a LoRA engine"
        strip = lambda k: k[len("base_model.model."):] if k.startswith("base_model.model.") else k
        sd = {}
        for name in ("adapter_model.bin", "non_lora_trainables.bin"):
            f = os.path.join(path, name)
            if os.path.exists(f):
                sd.update({strip(k).replace(".default.weight", ".weight"): v for k, v in torch.load(f, map_location="cpu", weights_only=True).items()})
        if not sd:
            raise FileNotFoundError(f"no adapter_model.bin / non_lora_trainables.bin under {path}")
        from ...params import load_named
        missing, unexpected = load_named(e.lm, sd)
        if missing or unexpected:
            raise KeyError(f"adapter checkpoint does not match the model: missing={missing[:4]} unexpected={unexpected[:4]}")
        if e.master is not None:
            from ... import ops
            e.master.copy_(ops.to_f32(e.lm.flat))
