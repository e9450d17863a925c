"""LlavaEngine: the MI355X-native forward/backward/optimizer of the LLaVA training step.

No autograd graph and no tracing compiler: forward and backward are explicit sequences of C-ABI kernel launches
(radvlm_amd.ops) over flat bf16 buffers, on the current HIP stream; data-parallel gradient buckets are contiguous
slices of the flat gradient buffer, all-reduced with RCCL on a side stream while backward continues.

Reference semantics followed (paths under /root/reference/finetuning/llava):
  forward    : LlavaLlamaForCausalLM.forward model/language_model/llava_llama.py:69-120 ->
               prepare_inputs_labels_for_multimodal model/llava_arch.py:251-555 -> HF LlamaForCausalLM
               (text mirror model/language_model/modeling_llama.py:1083-1185, :1304-1337), CLIPVisionTower
               model/multimodal_encoder/clip_encoder.py:46-79, mm_projector model/multimodal_projector/builder.py:41-48
  freeze     : tower frozen (clip_encoder.py:42), projector + LM trainable (train/train.py:1613-1665)
  optimizer  : AdamW groups of LLaVATrainer.create_optimizer train/llava_trainer.py:356-433
  grad sync  : what DDP would do for the reference: mean over ranks of per-rank mean losses
"""
import math

import numpy as np
import torch

from . import ops
from .params import (LORA_TARGETS, VP, FlatParams, fast_random_init_, lm_param_shapes, lora_trainable_shapes, portable_init_,
                     vision_param_shapes)
from .splice import build_splice_plan, merged_feature_rows, shifted_labels

BF16 = torch.bfloat16


def _ru(x, m):
    return (x + m - 1) // m * m


class LlavaEngine:
    def __init__(self, geo, device="cuda", merge_type="flat", image_aspect_ratio="square", image_grid_pinpoints=None,
                 max_len=None, init="portable", seed=0, rms_eps=1e-5, rope_theta=10000.0, process_group=None,
                 bucket_layers=1, train_vision_tower=False, lora=None, packed="auto", freeze_lm=False, train_embed_tokens=False,
                 freeze_projector=False,
                 padding_side="right", force_grad_sync=False, recompute=False):
        self.geo = geo
        assert recompute in (False, True, "auto")
        self.recompute = recompute            # activation recompute policy of the decoder layers (_recompute_layers)
        self.head_rows = "labeled"            # final norm + lm_head + cross entropy on the rows that carry a label ("all": every row)
        self.last_layer_rows = "labeled"      # ... and the last decoder layer's o_proj / norm / MLP (needs head_rows == "labeled")
        self._recompute_cache, self._recompute_forced = {}, False
        # packed (varlen) decoder batches: True / False / "auto" (pack when the samples of a batch differ in length): the decoder
        # then runs on sum(len_b) token rows instead of B * max(len_b) -- no padding rows through GEMMs, norms, CE
        self.packed = packed
        self.v, self.l = geo["vision"], geo["lm"]
        self.device = torch.device(device)
        self.merge_type = merge_type
        self.aspect = image_aspect_ratio
        self.pinpoints = image_grid_pinpoints
        self.max_len = max_len
        self.eps = self.l.get("rms_eps", rms_eps)
        self.theta = self.l.get("rope_theta", rope_theta)
        self.vocab = self.l["vocab"]                                # logical vocabulary (the tables may hold up to 7 zero pad rows)
        assert self.vocab % 8 == 0, "built-in geometries keep the vocabulary a multiple of 8; grow it with resize_token_embeddings"
        self.hd = self.l["d"] // self.l["heads"]
        self.Hkv = self.l.get("kv_heads", self.l["heads"])       # grouped-query attention (Qwen2)
        self.kvd = self.Hkv * self.hd
        self.siglip = self.v.get("kind") == "siglip"
        self.has_cls = not self.siglip
        self.ln_eps = 1e-6 if self.siglip else 1e-5               # siglip_encoder.py:83 / CLIP config default
        # per-channel statistics of the image processor (uint8 inputs are normalised on the device with them)
        self.image_mean = (0.5, 0.5, 0.5) if self.siglip else (0.48145466, 0.4578275, 0.40821073)
        self.image_std = (0.5, 0.5, 0.5) if self.siglip else (0.26862954, 0.26130258, 0.27577711)
        vhd = self.v["d"] // self.v["heads"]
        self.vhd_pad = vhd if vhd in (64, 128) else (64 if vhd < 64 else 128)
        assert vhd <= 128
        self._vis_pad = {}
        self.with_newline = "unpad" in merge_type
        self.side = self.v["image"] // self.v["patch"]
        self.P = self.side * self.side
        self.N_vis = self.P + (1 if self.has_cls else 0)
        self.kp = _ru(3 * self.v["patch"] ** 2, 8)
        self.train_tower = train_vision_tower
        self.lora = dict(lora) if lora else None   # {"r": 64, "alpha": 16, "dropout": 0.05}
        self.freeze_lm = bool(freeze_lm) and not self.lora
        self.base = None
        if self.lora:
            # LoRA (BASELINE config 5): the language model is a frozen bf16 store; only adapters + projector are
            # trainable, so gradients / AdamW state / the DP all-reduce cover ~2 % of the parameters
            assert not train_vision_tower
            r = self.lora["r"]
            self.lora_scale = self.lora.get("alpha", 16) / r
            self.lora_p = float(self.lora.get("dropout", 0.0))
            self.base = FlatParams(lm_param_shapes(geo, False), self.device)
            self.lm = FlatParams(lora_trainable_shapes(geo, r, self.with_newline), self.device)
            self.vis = FlatParams(vision_param_shapes(geo), self.device)
        elif freeze_lm:
            # projector-only stage (tune_mm_mlp_adapter / mm_tunable_parts="mm_mlp_adapter", train/train.py:1613-1640): tower and
            # language model are frozen stores; the trainable flat buffer holds the projector (+ image_newline) alone -- the
            # backward computes input gradients through the frozen decoder and no weight gradients for it.  With train_vision_tower
            # (mm_tunable_parts="mm_vision_tower,mm_mlp_adapter": tower + projector, frozen LM) the tower joins that buffer, in front.
            from collections import OrderedDict
            full = lm_param_shapes(geo, self.with_newline)
            self.base = FlatParams(lm_param_shapes(geo, False), self.device)
            # train_embed_tokens: tune_mm_mlp_adapter + mm_use_im_start_end makes the INPUT embeddings trainable as well
            # (llava_arch.py:577-581: get_input_embeddings requires_grad True, get_output_embeddings False)
            keep = lambda k: "mm_projector" in k or k == "model.image_newline" or (train_embed_tokens and k == "model.embed_tokens.weight")
            shapes = OrderedDict(vision_param_shapes(geo)) if train_vision_tower else OrderedDict()
            shapes.update((k, v) for k, v in full.items() if keep(k))
            self.lm = FlatParams(shapes, self.device)
            self.vis = self.lm if train_vision_tower else FlatParams(vision_param_shapes(geo), self.device)
        elif train_vision_tower:
            # mm_tunable_parts contains mm_vision_tower (train/train.py:1658-1661): the tower joins the trainable flat
            # buffer, in front (forward order), so its gradients are the last bucket of the backward pass
            from collections import OrderedDict
            shapes = OrderedDict(vision_param_shapes(geo))
            shapes.update(lm_param_shapes(geo, self.with_newline))
            self.lm = FlatParams(shapes, self.device)
            self.vis = self.lm
        else:
            self.lm = FlatParams(lm_param_shapes(geo, self.with_newline), self.device)
            self.vis = FlatParams(vision_param_shapes(geo), self.device)
        # Tensors of the trainable buffer that stay FROZEN (any subset of mm_tunable_parts, train/train.py:1613-1665): their gradients are
        # still computed where a kernel produces them anyway, then zeroed before the norm / clip, and AdamW skips them.
        #   * mm_mlp_adapter absent  -> the projector;
        #   * mm_language_model absent -> image_newline too (a parameter of the LM group there: no 'mm_projector' / 'vision_tower' in its name)
        self.frozen_names = set()
        if freeze_projector:
            self.frozen_names |= {n for n in self.lm.names() if "mm_projector" in n}
        if self.freeze_lm and "model.image_newline" in self.lm.offsets:
            self.frozen_names.add("model.image_newline")
        if not [n for n in self.lm.names() if n not in self.frozen_names]:
            raise ValueError("nothing is trainable: mm_tunable_parts must name at least one of mm_vision_tower, mm_mlp_adapter, mm_language_model")
        self.grads = self.lm.like(BF16)
        stores = [self.lm] + ([] if train_vision_tower else [self.vis]) + ([self.base] if self.base is not None else [])
        for k, st in enumerate(stores):
            if init == "portable":
                portable_init_(st, self.l["d"], seed)
            elif init == "fast":
                fast_random_init_(st, self.l["d"], seed + k)
        if self.lora and init in ("portable", "fast"):
            # the projector and the frozen base keep the names (hence values) of the full model; adapters: peft init
            ga = torch.Generator(device=self.device).manual_seed(seed + 7919)      # seeded: two runs of one command start identically
            for n in self.lm.names():
                if n.endswith("lora_B.weight"):
                    self.lm.view(n).zero_()
                elif n.endswith("lora_A.weight") and init == "fast":
                    bound = 1.0 / math.sqrt(self.lm.shapes[n][1])   # kaiming_uniform(a=sqrt(5)) of peft's lora_A
                    self.lm.view(n).copy_((torch.rand(self.lm.shapes[n], device=self.device, generator=ga) * 2 - 1) * bound)
        self.lora_step = 0
        self._patch_w = None
        self._rope = {}
        self._stats = {}
        self.master = self.m = self.vv = None
        self.opt_step = 0
        self.pg = process_group
        self.world = torch.distributed.get_world_size(process_group) if process_group is not None else 1
        self.bucket_layers = bucket_layers
        from .ddp import FlatGradSync
        # per-layer buckets (~0.4 GB bf16 for the 7B geometry): large transfers suit point-to-point xGMI links
        # force_grad_sync: a one-rank group still issues every bucket's collective (single-GPU rehearsal of the RCCL path)
        self.force_grad_sync = bool(force_grad_sync) and process_group is not None
        self.sync = FlatGradSync(self.grads, process_group, force=self.force_grad_sync) if (self.world > 1 or self.force_grad_sync) else None
        self._select_gemm_launch_shape()
        self.ctx = None
        self.grad_accum_started = False
        self.loss_scale = 1.0
        import os
        # A/B switch (measurement only): RV_FUSED=0 runs the unfused sequences (GEMM, then rope / swiglu kernels); results are bit-identical
        self.fused = os.environ.get("RV_FUSED", "1") != "0"
        assert padding_side in ("right", "left")
        self.padding_side = padding_side     # config.tokenizer_padding_side (llava_arch.py:520-524)

    def _select_gemm_launch_shape(self):
        """The all-reduce kernels of a data-parallel run share the CUs with backward's GEMMs: one-tile-per-block launches lose part of a
        round to them, a persistent block that cannot start would delay its whole share of the tiles (rv_gemm_select_kernel 40 / 41,
        include/radvlm_hip.h).  The switch is process-wide, so it is set at the top of every forward / backward of THIS engine: a second
        engine in the process (an eval or reference model with or without gradient collectives) cannot leave the wrong shape behind."""
        if self.device.type == "cuda":
            from . import lib
            lib.load().rv_gemm_select_kernel(40 if self.sync is not None else 41)

    # ------------------------------------------------------------------ vocabulary
    def resize_token_embeddings(self, new_vocab):
        """resize_token_embeddings + the mean initialisation of initialize_vision_tokenizer (llava_arch.py:563-575): embed_tokens and
        lm_head grow to `new_vocab` rows, the added rows start as the mean of the existing ones.  Every store that holds one of the
        two tables (the trainable buffer, or the frozen base of LoRA / projector-only runs) is rebuilt with every other tensor copied
        bit for bit; the optimizer state restarts."""
        old_v = self.vocab
        if new_vocab == old_v:
            return
        assert new_vocab > old_v, "the vocabulary only grows"
        import copy
        from collections import OrderedDict
        tables = ("model.embed_tokens.weight", "lm_head.weight")
        phys = _ru(int(new_vocab), 8)        # table rows: a multiple of 8 (GEMM / CE vector width); rows >= new_vocab stay exactly zero
        geo = copy.deepcopy(self.geo)
        geo["lm"]["vocab"] = phys

        def grow(store):
            if not any(t in store.offsets for t in tables):
                return store
            shapes = OrderedDict((n, ((phys, shp[1]) if n in tables else shp)) for n, shp in store.shapes.items())
            new = FlatParams(shapes, self.device)
            for name in store.names():
                src, dst = store.view(name), new.view(name)
                if name in tables:
                    dst[:old_v].copy_(src[:old_v])
                    dst[old_v:new_vocab].copy_(src[:old_v].float().mean(dim=0, keepdim=True).to(BF16).expand(new_vocab - old_v, -1))
                else:
                    dst.copy_(src)
            return new

        tower_shared = self.vis is self.lm
        new_lm = grow(self.lm)
        if self.base is not None:
            self.base = grow(self.base)
        self.vocab = int(new_vocab)
        self.geo, self.l, self.v = geo, geo["lm"], geo["vision"]
        if new_lm is not self.lm:
            self.lm = new_lm
            if tower_shared:
                self.vis = new_lm
            self.grads = new_lm.like(BF16)
            self.master = self.m = self.vv = None
            self.opt_step = 0
            self.grad_accum_started = False
            if self.sync is not None:
                from .ddp import FlatGradSync
                self.sync = FlatGradSync(self.grads, self.pg, force=self.force_grad_sync)
        self.weights_changed()

    # ------------------------------------------------------------------ weights
    def W(self, name):
        if name in self.lm.offsets:
            return self.lm.view(name)
        return self.base.view(name)

    def G(self, name):
        return self.lm.view(name, self.grads)

    def _layer_views(self, i, flat=None):
        d, F = self.l["d"], self.l["ffn"]
        p = f"model.layers.{i}."
        f = self.lm
        if self.base is not None:
            if flat is not None:
                return {}          # the base weights of a LoRA run have no gradients
            f = self.base
        out = {}
        if self.l.get("qkv_bias"):
            out["bqkv"] = f.fused(p + "self_attn.q_proj.bias", p + "self_attn.v_proj.bias", 1, d + 2 * self.kvd, flat).view(-1)
        return dict(
            out,
            ln1=f.view(p + "input_layernorm.weight", flat),
            qkv=f.fused(p + "self_attn.q_proj.weight", p + "self_attn.v_proj.weight", d + 2 * self.kvd, d, flat),
            o=f.view(p + "self_attn.o_proj.weight", flat),
            ln2=f.view(p + "post_attention_layernorm.weight", flat),
            gu=f.fused(p + "mlp.gate_proj.weight", p + "mlp.up_proj.weight", 2 * F, d, flat),
            down=f.view(p + "mlp.down_proj.weight", flat),
        )

    def weights_changed(self, tower=True):
        """Call after any in-place edit of the flat parameters (load_state_dict, optimizer step): drops the derived
        copies of tower weights (padded patch / attention projections) when the tower may have changed."""
        if tower:
            self._patch_w = None
            self._vis_pad = {}

    def _stat_buffer(self, key, B, H, s_pad):
        """fp32 [B, H, s_pad] softmax-statistics buffers (lse per layer, delta), allocated once (zeroed) and reused every step: the
        kernels overwrite the entries of valid rows only -- no fill kernel per layer.  With ragged data a later batch of the same padded
        shape but shorter samples therefore sees STALE lse / delta in rows >= len; that is harmless by construction, not by zeroing:
        the dK/dV edge tiles select p = 0 for query rows >= q_end before any use of lse / delta, and the dQ pass reads them only for
        rows < len (tests/test_train_features_gpu.py::test_stale_softmax_statistics_are_masked)."""
        k = (key, B, H, s_pad)
        buf = self._stats.get(k)
        if buf is None:
            if len(self._stats) > 4 * (self.l["layers"] + self.v["layers"]) + 8:      # shapes change from batch to batch with ragged data: do not hoard
                self._stats.clear()
            buf = self._stats[k] = torch.zeros(B, H, s_pad, dtype=torch.float32, device=self.device)
        return buf

    def _dev(self, a):
        """Host array -> device tensor through pinned memory, asynchronously.  A copy from pageable memory makes the runtime wait for the
        stream first, i.e. the host could never enqueue ahead of the GPU (each of the ~10 index uploads of a step was such a wait: any host
        hiccup then idled the GPU); torch's pinned-memory cache recycles a block only after the copy that used it has finished."""
        t = torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a
        if self.device.type != "cuda" or t.device.type != "cpu":
            return t.to(self.device)
        return (t if t.is_pinned() else t.pin_memory()).to(self.device, non_blocking=True)

    def rope_table(self, S):
        if S not in self._rope:
            self._rope[S] = ops.rope_table(S, self.l["d"] // self.l["heads"], self.theta, self.device)
        return self._rope[S]

    # ------------------------------------------------------------------ vision tower
    def _vision_prepare(self):
        if self._patch_w is None:
            w = self.vis.view(VP + "embeddings.patch_embedding.weight").reshape(self.v["d"], -1)
            pw = torch.zeros(self.v["d"], self.kp, dtype=BF16, device=self.device)
            pw[:, :w.shape[1]] = w
            self._patch_w = pw

    def _vis_attn_weights(self, i):
        """(wqkv [3*H*hp, dv], bqkv [3*H*hp], wo [dv, H*hp]) of tower layer i.  The attention kernels take head_dim 64 or
        128; for any other head_dim (SigLIP-so400m: 72, siglip_encoder.py:185) every head is zero-padded to hp in derived
        copies of the projection weights, so q/k/v come out of the fused GEMM already padded: the padded lanes are exact
        zeros in QK^T and PV and the softmax scale stays head_dim^-0.5."""
        v, f = self.v, self.vis
        dv, H = v["d"], v["heads"]
        hd, hp = dv // H, self.vhd_pad
        p = VP + f"encoder.layers.{i}."
        wqkv = f.fused(p + "self_attn.q_proj.weight", p + "self_attn.v_proj.weight", 3 * dv, dv)
        bqkv = f.fused(p + "self_attn.q_proj.bias", p + "self_attn.v_proj.bias", 1, 3 * dv).view(-1)
        wo = f.view(p + "self_attn.out_proj.weight")
        if hp == hd:
            return wqkv, bqkv, wo
        c = self._vis_pad.get(i)
        if c is None:
            wp = torch.zeros(3, H, hp, dv, dtype=BF16, device=self.device)
            wp[:, :, :hd] = wqkv.view(3, H, hd, dv)
            bp = torch.zeros(3, H, hp, dtype=BF16, device=self.device)
            bp[:, :, :hd] = bqkv.view(3, H, hd)
            op = torch.zeros(dv, H, hp, dtype=BF16, device=self.device)
            op[:, :, :hd] = wo.view(dv, H, hd)
            c = (wp.view(3 * H * hp, dv), bp.view(-1), op.view(dv, H * hp))
            self._vis_pad[i] = c
        return c

    def vision_forward(self, pixels, save=None):
        """pixels bf16 [n,3,H,W] -> the tower's selected hidden state as rows [n*N, dv]:
          CLIP   (clip_encoder.py:68-79): hidden_states[-2], N = P+1 (class token first);
          SigLIP (siglip_encoder.py:568-590): hidden_states[-1] of the encoder WITHOUT its last layer, N = P = 729
        -- both are 'run all layers but the last'.  With `save` (tower tunable) the activations vision_backward needs are kept."""
        v, f = self.v, self.vis
        self._vision_prepare()
        n = pixels.shape[0]
        dv, H = v["d"], v["heads"]
        hd, hp = dv // H, self.vhd_pad
        dvp = H * hp
        N = self.N_vis
        n_pad = _ru(N, 64)
        keep = save is not None
        eps = self.ln_eps
        cols = ops.im2col_patches(pixels, v["patch"], self.kp)
        ln = lambda t, w, b: ops.layernorm_fwd(t, f.view(w), f.view(b), eps=eps, save_stats=keep)
        if self.siglip:
            e = ops.gemm_nt(cols, self._patch_w, bias=f.view(VP + "embeddings.patch_embedding.bias"))
            x = ops.add_pos_rows(e, f.view(VP + "embeddings.position_embedding.weight"), n, self.P)
            st0 = None
        else:
            po = ops.gemm_nt(cols, self._patch_w)
            e = ops.clip_embed(po, f.view(VP + "embeddings.class_embedding"), f.view(VP + "embeddings.position_embedding.weight"),
                               n, self.P, dv)
            r = ln(e, VP + "pre_layrnorm.weight", VP + "pre_layrnorm.bias")
            x, st0 = r if keep else (r, None)
        act_code = ops.ACT_GELU_TANH if self.siglip else ops.ACT_QUICK_GELU
        act_fwd = ops.gelu_tanh_fwd if self.siglip else ops.quick_gelu_fwd
        layers = []
        for i in range(v["layers"] - 1):
            p = VP + f"encoder.layers.{i}."
            r = ln(x, p + "layer_norm1.weight", p + "layer_norm1.bias")
            h, st1 = r if keep else (r, None)
            wqkv, bqkv, wo = self._vis_attn_weights(i)
            qkv = ops.gemm_nt(h, wqkv, bias=bqkv)
            # softmax statistics: one reused buffer for a frozen tower (nobody reads them), one per layer when backward will
            vlse = self._stat_buffer(("vlse", i if keep else -1), n, H, n_pad)
            if hp == 128:     # natural-layout kernel: no V^T copy
                a, lse = ops.attn_fwd(qkv[:, :dvp], qkv[:, dvp:2 * dvp], None, n, N, H, hp, n_pad, causal=False, scale=hd ** -0.5, v=qkv[:, 2 * dvp:], lse=vlse)
            else:
                vT = ops.transpose_heads(qkv[:, 2 * dvp:], n, N, H, hp, n_pad)
                a, lse = ops.attn_fwd(qkv[:, :dvp], qkv[:, dvp:2 * dvp], vT, n, N, H, hp, n_pad, causal=False, scale=hd ** -0.5, lse=vlse)
            x1 = ops.gemm_nt(a, wo, bias=f.view(p + "self_attn.out_proj.bias"), residual=x)
            r = ln(x1, p + "layer_norm2.weight", p + "layer_norm2.bias")
            h2, st2 = r if keep else (r, None)
            if keep:
                z = ops.gemm_nt(h2, f.view(p + "mlp.fc1.weight"), bias=f.view(p + "mlp.fc1.bias"))
                g = act_fwd(z)
            else:
                z = None
                g = ops.gemm_nt(h2, f.view(p + "mlp.fc1.weight"), bias=f.view(p + "mlp.fc1.bias"), act=act_code)
            x2 = ops.gemm_nt(g, f.view(p + "mlp.fc2.weight"), bias=f.view(p + "mlp.fc2.bias"), residual=x1)
            if keep:
                layers.append(dict(x=x, st1=st1, h=h, qkv=qkv, a=a, lse=lse, x1=x1, st2=st2, h2=h2, z=z, g=g))
            x = x2
        if keep:
            save.update(v_cols=cols, v_e=e, v_st0=st0, v_layers=layers, v_n=n)
        return x

    @staticmethod
    def _put(dst, src, acc):
        dst.add_(src) if acc else dst.copy_(src)

    def vision_backward(self, dx, c):
        """Backward of vision_forward: dx = d(selected hidden state) [n*N, dv]; parameter grads -> flat grad views."""
        v, f, G = self.v, self.vis, self.G
        n = c["v_n"]
        dv, H = v["d"], v["heads"]
        hd, hp = dv // H, self.vhd_pad
        dvp = H * hp
        padded = hp != hd
        N = self.N_vis
        n_pad = _ru(N, 64)
        acc = self.grad_accum_started
        W = lambda name: f.view(name)
        act_bwd = ops.gelu_tanh_bwd if self.siglip else ops.quick_gelu_bwd
        for i in reversed(range(v["layers"] - 1)):
            a = c["v_layers"][i]
            p = VP + f"encoder.layers.{i}."
            ops.bias_grad(dx, out=G(p + "mlp.fc2.bias"), accumulate=acc)
            dg = self._linear_bwd(dx, a["g"], W(p + "mlp.fc2.weight"), G(p + "mlp.fc2.weight"))
            dz = act_bwd(dg, a["z"])
            ops.bias_grad(dz, out=G(p + "mlp.fc1.bias"), accumulate=acc)
            dh2 = self._linear_bwd(dz, a["h2"], W(p + "mlp.fc1.weight"), G(p + "mlp.fc1.weight"))
            ops.layernorm_bwd(dh2, a["x1"], W(p + "layer_norm2.weight"), a["st2"], G(p + "layer_norm2.weight"),
                              G(p + "layer_norm2.bias"), dx=dx, dx_add=True, accumulate=acc)
            ops.bias_grad(dx, out=G(p + "self_attn.out_proj.bias"), accumulate=acc)
            wqkv, bqkv, wo = self._vis_attn_weights(i)
            gqkv = f.fused(p + "self_attn.q_proj.weight", p + "self_attn.v_proj.weight", 3 * dv, dv, self.grads)
            gb = f.fused(p + "self_attn.q_proj.bias", p + "self_attn.v_proj.bias", 1, 3 * dv, self.grads).view(-1)
            if padded:   # gradients of the padded copies, valid lanes folded back into the flat views
                gwo = ops.gemm(dx, a["a"], ta=True, tb=True)
                self._put(G(p + "self_attn.out_proj.weight").view(dv, H, hd), gwo.view(dv, H, hp)[:, :, :hd], acc)
                da = ops.gemm(dx, wo, tb=True)
            else:
                da = self._linear_bwd(dx, a["a"], wo, G(p + "self_attn.out_proj.weight"))
            qkv = a["qkv"]
            dqkv = torch.empty_like(qkv)
            ops.attn_bwd(qkv[:, :dvp], qkv[:, dvp:2 * dvp], qkv[:, 2 * dvp:], a["a"], da, a["lse"], n, N, H, hp, n_pad, False,
                         scale=hd ** -0.5, dq=dqkv[:, :dvp], dk=dqkv[:, dvp:2 * dvp], dv=dqkv[:, 2 * dvp:])
            if padded:
                gbp = ops.bias_grad(dqkv)
                self._put(gb.view(3, H, hd), gbp.view(3, H, hp)[:, :, :hd], acc)
                gwp = ops.gemm(dqkv, a["h"], ta=True, tb=True)
                self._put(gqkv.view(3, H, hd, dv), gwp.view(3, H, hp, dv)[:, :, :hd], acc)
                dh = ops.gemm(dqkv, wqkv, tb=True)
            else:
                ops.bias_grad(dqkv, out=gb, accumulate=acc)
                dh = self._linear_bwd(dqkv, a["h"], wqkv, gqkv)
            ops.layernorm_bwd(dh, a["x"], W(p + "layer_norm1.weight"), a["st1"], G(p + "layer_norm1.weight"),
                              G(p + "layer_norm1.bias"), dx=dx, dx_add=True, accumulate=acc)
            c["v_layers"][i] = None
            self._bucket_done(p + "layer_norm1.weight", p + "mlp.fc2.bias")
        if self.siglip:
            de = dx
        else:
            de = ops.layernorm_bwd(dx, c["v_e"], W(VP + "pre_layrnorm.weight"), c["v_st0"], G(VP + "pre_layrnorm.weight"),
                                   G(VP + "pre_layrnorm.bias"), accumulate=acc)
        # embeddings: position table = sum over images, class token = rows t = 0, patch conv = wgrad over im2col rows
        de2 = de.view(n, N * dv)
        ops.bias_grad(de2, out=G(VP + "embeddings.position_embedding.weight").view(-1), accumulate=acc)
        if self.siglip:
            ops.bias_grad(de, out=G(VP + "embeddings.patch_embedding.bias"), accumulate=acc)
            dpo = de
        else:
            ops.bias_grad(de2[:, :dv], out=G(VP + "embeddings.class_embedding"), accumulate=acc)
            drop_cls = (torch.arange(n * self.P, device=self.device, dtype=torch.int32)
                        + torch.arange(n, device=self.device, dtype=torch.int32).repeat_interleave(self.P) + 1)
            dpo = ops.gather_rows(drop_cls, dv, de)
        dwp = ops.gemm(dpo, c["v_cols"], ta=True, tb=True)
        gp = G(VP + "embeddings.patch_embedding.weight").view(dv, -1)
        self._put(gp, dwp[:, :gp.shape[1]], acc)
        # the last encoder layer and post_layernorm do not feed the selected hidden state: their gradients are zero
        last = VP + f"encoder.layers.{v['layers'] - 1}."
        s0, _ = self.lm.span(last + "layer_norm1.weight", last + "layer_norm1.weight")
        _, e1 = self.lm.span(VP + "post_layernorm.bias", VP + "post_layernorm.bias")
        if not acc:
            self.grads[s0:e1].zero_()
        first = VP + ("embeddings.patch_embedding.weight" if self.siglip else "embeddings.class_embedding")
        self._bucket_done(first, VP + ("embeddings.position_embedding.weight" if self.siglip else "pre_layrnorm.bias"))
        self._bucket_done(last + "layer_norm1.weight", VP + "post_layernorm.bias")

    def encode_images(self, pixels, save=None, plan=None):
        """encode_images (llava_arch.py:192-196): tower (patch features, CLS dropped) then mlp2x_gelu projector.
        Returns the feature table [n*P + n_extra + 1, d]: projector rows, then the rows anyres_max creates by bilinear
        down-sampling (llava_arch.py:381-392; a 4-tap weighted gather of projector rows), then image_newline."""
        n = pixels.shape[0]
        rs = (plan or {}).get("resample")
        mp = (plan or {}).get("maxpool")
        n_extra = plan["n_extra_rows"] if (rs or mp) else 0
        d = self.l["d"]
        hid = self.vision_forward(pixels, save=save if (self.train_tower and save is not None) else None)
        if self.has_cls:
            drop_cls = (torch.arange(n * self.P, device=self.device, dtype=torch.int32)
                        + torch.arange(n, device=self.device, dtype=torch.int32).repeat_interleave(self.P) + 1)
            f0 = ops.gather_rows(drop_cls, self.v["d"], hid)
        else:
            f0 = hid
        z1 = ops.gemm_nt(f0, self.W("model.mm_projector.0.weight"), bias=self.W("model.mm_projector.0.bias"))
        a1 = ops.gelu_fwd(z1)
        table = torch.empty(n * self.P + n_extra + 1, d, dtype=BF16, device=self.device)
        ops.gemm_nt(a1, self.W("model.mm_projector.2.weight"), bias=self.W("model.mm_projector.2.bias"), out=table[:n * self.P])
        if rs:
            t = lambda k: self._dev(rs[k])
            ops.weighted_segment_sum_rows(table, t("fwd_off"), t("fwd_pos"), t("fwd_w"), t("fwd_out"), table)
        if mp:
            which = ops.max4_rows_fwd(table, self._dev(mp["idx4"]), self._dev(mp["out"]), table)
            if save is not None:
                save["pool_which"] = which
        if self.with_newline:
            table[n * self.P + n_extra].copy_(self.W("model.image_newline"))
        else:
            table[n * self.P + n_extra].zero_()
        if save is not None:
            save.update(f0=f0, z1=z1, a1=a1)
        return table

    # ------------------------------------------------------------------ forward
    def plan(self, input_ids, attention_mask, labels, images, image_sizes=None):
        """Host-side index planning (numpy); images: list of [3,H,W] or [T,3,H,W] tensors."""
        tiles = [1 if im.ndim == 3 else im.shape[0] for im in images]
        n_proj = sum(tiles) * self.P
        extra = dict(next=n_proj, src=[], w=[])     # rows created by anyres_max down-sampling follow the projector rows
        rows, r0 = [], 0
        for i, t in enumerate(tiles):
            rows.append(merged_feature_rows(r0, t, self.side, self.merge_type, self.aspect,
                                            tuple(image_sizes[i]) if image_sizes is not None else None, self.pinpoints,
                                            self.v["image"], extra=extra))
            r0 += t * self.P
        n_extra = extra["next"] - n_proj
        ids = np.asarray(input_ids)
        am = np.asarray(attention_mask).astype(bool) if attention_mask is not None else np.ones_like(ids, dtype=bool)
        lab = np.asarray(labels) if labels is not None else np.full_like(ids, -100)
        plan = build_splice_plan(ids, am, lab, rows, n_proj + n_extra, self.max_len, self.padding_side)
        plan["n_feat_rows"] = n_proj + n_extra
        plan["n_proj_rows"] = n_proj
        plan["n_extra_rows"] = n_extra
        if extra.get("pool_src"):   # maxpool2x2: created rows = elementwise max of four projector rows
            win = np.concatenate(extra["pool_src"]).astype(np.int32)
            assert (plan["feat_pos"][np.unique(win)] < 0).all()
            plan["maxpool"] = dict(idx4=win.reshape(-1), out=np.concatenate(extra["pool_out"]).astype(np.int32))
        if extra["src"]:
            src = np.concatenate(extra["src"]).reshape(-1)            # [4 * n_resampled] projector rows
            w = np.concatenate(extra["w"]).reshape(-1).astype(np.float32)
            assert (plan["feat_pos"][np.unique(src)] < 0).all()       # a down-sampled grid's sources are never spliced directly
            order = np.argsort(src, kind="stable")
            usrc, starts = np.unique(src[order], return_index=True)
            n_rs = src.shape[0] // 4
            assert n_rs == n_extra or not extra.get("pool_src"), "anyres_max resampling and maxpool2x2 are mutually exclusive merges"
            plan["resample"] = dict(
                fwd_off=np.arange(0, 4 * n_rs + 1, 4, dtype=np.int32), fwd_pos=src.astype(np.int32), fwd_w=w,
                fwd_out=(n_proj + np.arange(n_rs)).astype(np.int32),
                # adjoint: for every source row, the (created row, weight) pairs it feeds
                adj_off=np.concatenate([starts, [src.shape[0]]]).astype(np.int32),
                adj_pos=(n_proj + order // 4).astype(np.int32), adj_w=w[order], adj_out=usrc.astype(np.int32))
        return plan

    # ------------------------------------------------------------------ decoder layer
    def activation_bytes_per_row(self):
        """bf16 bytes one decoder layer keeps per token row for backward: x, h1, q|k|v, attn, x_mid, h2, gate|up, act (+ fp32 statistics)."""
        d, F = self.l["d"], self.l["ffn"]
        n = 2 * (6 * d + 2 * self.kvd + 3 * F) + 4 * (2 + self.l["heads"])
        if getattr(self, "lora", None):     # the seven adapted modules keep their r-wide down-projections
            n += 2 * 7 * self.lora["r"]
        if getattr(self, "hd", 128) != 128:  # head_dim 64 kernels read a V^T copy
            n += 2 * self.kvd
        return n

    def _recompute_layers(self, M):
        """How many decoder layers (the first n) re-run their forward in backward.  False: none -- the activations of all layers stay
        resident (~95 GB for 32 pairs x 704 tokens of the 7B model: what 288 GB of HBM are for); True: all; "auto": as few as the free
        device memory requires (measured on the 7B geometry: one 32k-token sample of the reference recipe, finetune_radio_7b.sh:79, still
        fits whole -- 236 GiB peak; a 50k-token sample recomputes 12 of 32 layers)."""
        L = self.l["layers"]
        if self.recompute is True:
            return L
        if not self.recompute or self.device.type != "cuda":
            return 0
        if self._recompute_forced:           # a step ran out of memory under "auto": the retry (and the same shape later) keeps layer inputs only
            return L
        hit = self._recompute_cache.get(M)
        if hit is not None:
            return hit
        per_layer = M * self.activation_bytes_per_row()
        free, _ = torch.cuda.mem_get_info(self.device)
        free += torch.cuda.memory_reserved(self.device) - torch.cuda.memory_allocated(self.device)     # torch's cached, unused blocks
        # head room: logits (bf16 + dlogits in place) + the lm_head input gradient, one layer's recomputed activations and gradients
        budget = 0.85 * free - (4 * M * self.l["vocab"] + 3 * per_layer)
        keep = int(max(0.0, budget) // max(per_layer + 2 * M * self.l["d"], 1))
        self._recompute_cache[M] = max(0, L - keep)     # one decision per row count: later steps see less free memory only because of this one's caches
        return self._recompute_cache[M]

    def _layer_forward(self, i, x, g, rows=None):
        """One decoder layer (modeling_llama.py:852-911) on token rows x -> (x_out, the activations backward needs).
        rows (LAST layer only; int32 device index, -1 = zero pad row): the rows whose output anything downstream reads -- the labelled rows
        (forward()).  Attention still runs on every row (all rows are keys / values), but o_proj, the residual, the second norm and the
        MLP are evaluated on the gathered rows alone: x_out and the saved x_mid, rstd2, h2, gu, act then have rows.numel() rows."""
        d, F, H = self.l["d"], self.l["ffn"], self.l["heads"]
        hd, Hkv, kvd = self.hd, self.Hkv, self.kvd
        B, S, s_pad, lens, cu, pos, cs = g["B"], g["S"], g["s_pad"], g["lens"], g["cu"], g["pos"], g["cs"]
        lv = self._layer_views(i)
        h1, rstd1 = ops.rmsnorm_fwd(x, lv["ln1"], self.eps)
        sv = {}
        if self.lora:
            qkv = self._lora_linear(h1, lv["qkv"], i, self._MODS_QKV(d), sv, bias=lv.get("bqkv"))   # frozen q/k/v biases (Qwen2)
            ops.rope_inplace(qkv, cs, S, H + Hkv, hd, 1, 1, positions=pos)   # q heads then k heads: one run of H + Hkv heads
        elif not self.fused:
            qkv = ops.gemm_nt(h1, lv["qkv"], bias=lv.get("bqkv"))
            ops.rope_inplace(qkv, cs, S, H + Hkv, hd, 1, 1, positions=pos)
        else:       # q|k|v projection with the rotary embedding in the GEMM epilogue
            qkv = ops.gemm_rope(h1, lv["qkv"], cs, S, H + Hkv, hd, bias=lv.get("bqkv"), positions=pos)
        if hd == 128:     # natural-layout kernel: K and V tiles are staged as they lie in memory (no V^T copy)
            attn, lse = ops.attn_fwd(qkv[:, :d], qkv[:, d:d + kvd], None, B, S, H, hd, s_pad, causal=True, lens=lens, kv_heads=Hkv, cu=cu,
                                     v=qkv[:, d + kvd:], lse=self._stat_buffer(("lse", i), B, H, s_pad))
        else:
            vT = ops.transpose_heads(qkv[:, d + kvd:], B, S, Hkv, hd, s_pad, cu=cu)
            attn, lse = ops.attn_fwd(qkv[:, :d], qkv[:, d:d + kvd], vT, B, S, H, hd, s_pad, causal=True, lens=lens, kv_heads=Hkv, cu=cu)
        attn_sel = None
        if rows is not None:
            attn_sel = ops.gather_rows(rows, d, attn)
            if self.lora:
                x_mid = self._lora_linear(attn_sel, lv["o"], i, (("self_attn.o_proj", 0, d),), sv, residual=ops.gather_rows(rows, d, x))
            else:
                x_mid = ops.gemm_nt(attn_sel, lv["o"], residual=ops.gather_rows(rows, d, x))
        elif self.lora:
            x_mid = self._lora_linear(attn, lv["o"], i, (("self_attn.o_proj", 0, d),), sv, residual=x)
        else:
            x_mid = ops.gemm_nt(attn, lv["o"], residual=x)
        h2, rstd2 = ops.rmsnorm_fwd(x_mid, lv["ln2"], self.eps)
        if self.lora:
            gu = self._lora_linear(h2, lv["gu"], i, (("mlp.gate_proj", 0, F), ("mlp.up_proj", F, 2 * F)), sv)
            act = ops.swiglu_fwd(gu, F)
        elif not self.fused:
            gu = ops.gemm_nt(h2, lv["gu"])
            act = ops.swiglu_fwd(gu, F)
        else:       # gate|up projection and silu(gate) * up in one launch
            gu, act = ops.gemm_swiglu_fwd(h2, lv["gu"], F)
        if self.lora:
            x_out = self._lora_linear(act, lv["down"], i, (("mlp.down_proj", 0, d),), sv, residual=x_mid)
        else:
            x_out = ops.gemm_nt(act, lv["down"], residual=x_mid)
        acts = dict(x=x, rstd1=rstd1, h1=h1, qkv=qkv, attn=attn, lse=lse, x_mid=x_mid, rstd2=rstd2, h2=h2, gu=gu, act=act, lora=sv)
        if attn_sel is not None:
            acts["attn_sel"] = attn_sel
        return x_out, acts

    def forward(self, input_ids, attention_mask, labels, images, image_sizes=None, want_logits=False, loss_scale=None):
        """One training forward. Returns loss (fp32 device tensor [1], never scaled); keeps the context for backward().
        loss_scale multiplies the GRADIENTS only (default: self.loss_scale; a trainer sets 1 / gradient_accumulation_steps, which is
        what HF Trainer.training_step back-propagates, HF:trainer.py training_step `loss / gradient_accumulation_steps`)."""
        dev = self.device
        l = self.l
        d, F, H, V, L = l["d"], l["ffn"], l["heads"], l["vocab"], l["layers"]
        hd, Hkv, kvd = self.hd, self.Hkv, self.kvd
        self._select_gemm_launch_shape()
        plan = self.plan(input_ids, attention_mask, labels, images, image_sizes)
        self.lora_step += 1
        pix = torch.cat([(im if im.ndim == 4 else im[None]) for im in images], 0)
        if pix.device.type == "cpu" and not pix.is_pinned() and dev.type == "cuda":
            pix = pix.pin_memory()                                                          # pinned: the copy does not stall the stream
        pix = pix.to(dev, non_blocking=True)          # images already on the device (HF Trainer._prepare_inputs moves them) pass through
        if pix.dtype == torch.uint8:
            # device-side normalisation: uint8 HWC tiles from the host (resize / crop / pad only), rescale + (x - mean) / std
            # + channel-first layout + bf16 cast here, in the processors' own fp32 arithmetic
            pix = ops.normalize_tiles_u8(pix.contiguous(), self.v["image"], self.image_mean, self.image_std, mode=1 if self.siglip else 0)
        else:
            pix = pix if pix.dtype == BF16 else ops.to_bf16(pix.float())
        ctx = dict(plan=plan)
        table = self.encode_images(pix.contiguous(), save=ctx, plan=plan)
        B = int(np.asarray(input_ids).shape[0])   # samples (a sample may hold several images: images are consumed in order)
        S = plan["S"]
        s_pad = _ru(S, 64)
        lens_np = plan["lens"]
        ragged = bool((lens_np != S).any())
        packed = bool(self.packed) if self.packed != "auto" else ragged
        if self.padding_side == "left" and ragged:
            # left padding (llava_arch.py:520-524): the reference discards the spliced position_ids in training (:534-545), so
            # every row keeps positions arange(S) and a short sample starts at position S - len.  That is exactly the packed
            # layout with explicit positions = padded column index: valid tokens only, nothing computed for the pad rows.
            packed = True
            first = S - lens_np
            bad = [b for b in range(len(lens_np)) if 0 < lens_np[b] < S and plan["labels"][b, first[b]] != -100]
            if bad:
                raise ValueError(f"left padding: sample {bad[0]} has a label on its first token; the reference's loss would then read the "
                                 "logits of a padding row (shifted CE), which is not defined here")
        cu = pos = valid_idx = None
        if packed:
            valid = plan["attention_mask"].reshape(-1)
            p2k = np.cumsum(valid) - 1                                        # padded flat row -> packed row
            remap = lambda a: np.where(a >= 0, p2k[np.maximum(a, 0)], -1).astype(np.int32)
            plan = dict(plan, idx=plan["idx"][valid], feat_pos=remap(plan["feat_pos"]), newline_pos=remap(plan["newline_pos"]),
                        tok_pos=remap(plan["tok_pos"]))
            ctx["plan"] = plan
            M = int(valid.sum())
            cu = self._dev(np.concatenate([[0], np.cumsum(lens_np)]).astype(np.int32))
            pos = self._dev((np.arange(B * S, dtype=np.int32) % S)[valid])
            valid_idx = np.nonzero(valid)[0]
            lens = None
        else:
            M = B * S
            lens = self._dev(lens_np)
        idx = self._dev(plan["idx"])
        x = ops.gather_rows(idx, d, self.W("model.embed_tokens.weight"), table)
        cs = self.rope_table(S)
        geom = dict(B=B, S=S, s_pad=s_pad, lens=lens, cu=cu, pos=pos, cs=cs)
        # activation recompute (the reference's --gradient_checkpointing, train/train.py:164,1505-1513): layers [0, n_re) keep only their
        # input and re-run their forward inside backward (bit-identical: the kernels are deterministic, LoRA dropout masks regenerable)
        n_re = self._recompute_layers(M)
        tgt = shifted_labels(plan["labels"])
        if tgt.size and int(tgt.max()) >= self.vocab:     # torch's cross_entropy raises on an out-of-range target too
            raise IndexError(f"label {int(tgt.max())} is out of range for a vocabulary of {self.vocab}")
        count = int((tgt != -100).sum())
        inv = (1.0 / count) if count > 0 else float("nan")
        tgt = tgt.reshape(-1)
        if packed:
            tgt = tgt[valid_idx]
        # The loss reads the logits of the rows that carry a label and of no other row (modeling_llama.py:1326-1337: ignore_index rows
        # contribute neither to the loss nor -- their dlogits being exactly zero -- to any gradient), and nothing after the last decoder
        # layer mixes rows.  So the final norm, the lm_head product, the cross entropy and their backward run on the LABELLED rows only
        # (an instruction sample labels its answer tokens: 64 of 704 rows in the BASELINE workload, 9 % of a 3 x 5.9 TFLOP head).  Same
        # loss, same gradients (the dropped terms are exact zeros); HF's `logits_to_keep` is the same idea.  head_rows = "all" restores
        # the full product (A/B, tests); callers that want logits get the full fp32 product either way.
        head_idx = None
        lab_rows = np.nonzero(tgt != -100)[0]
        if self.head_rows == "labeled" and 0 < lab_rows.size <= 0.75 * M:
            n_sel = _ru(int(lab_rows.size), 64)           # whole 64-row groups: the weight gradient contracts over these rows
            head_idx = np.full(n_sel, -1, dtype=np.int32)
            head_idx[:lab_rows.size] = lab_rows
            t_sel = np.full(n_sel, -100, dtype=tgt.dtype)
            t_sel[:lab_rows.size] = tgt[lab_rows]
            tgt = t_sel
        # The same holds one step earlier: a row of the LAST decoder layer's output feeds its own logits row and nothing else, so that layer's
        # o_proj, second norm and MLP (304 MFLOP per row at 7B widths, x 3 with backward) are dead for the unlabelled rows as well; its
        # attention still runs on every row.  Not taken when the caller wants every row's logits.  (LoRA: the adapters' dropout masks are
        # indexed by the element position in the gathered tensor -- another mask than the all-rows run draws, regenerated identically in
        # backward.)
        last_sel = head_idx is not None and self.last_layer_rows == "labeled" and not want_logits
        layers = []
        for i in range(L):
            x_out, acts = self._layer_forward(i, x, geom, rows=self._dev(head_idx) if (last_sel and i == L - 1) else None)
            layers.append(dict(x=x, lora={}) if i < n_re else acts)
            x = x_out
        ctx["geom"], ctx["n_recomputed"] = geom, n_re
        if head_idx is None or last_sel:
            x_head = x
        else:
            x_head = ops.gather_rows(self._dev(head_idx), d, x)          # pad rows are zero and carry no label
        hN, rstdN = ops.rmsnorm_fwd(x_head, self.W("model.norm.weight"), self.eps)
        logits = ops.gemm_nt(hN, self.W("lm_head.weight"))
        tgt_t = self._dev(tgt)
        logits_out = None
        if want_logits:
            # callers get the reference's fp32 [B, S, V] (llava_llama.py:69-120 -> logits.float()): the lm_head GEMM once more with
            # fp32 output, i.e. the accumulators without the bf16 store the training path's loss reads (eval / tests only)
            hN_all = hN if head_idx is None else ops.rmsnorm_fwd(x, self.W("model.norm.weight"), self.eps)[0]
            lf = ops.gemm_nt(hN_all, self.W("lm_head.weight"), out_dtype=torch.float32)
            if packed:          # padded shape, padding rows zero
                logits_out = torch.zeros(B * S, V, dtype=torch.float32, device=dev)
                logits_out[self._dev(valid_idx)] = lf
            else:
                logits_out = lf
            logits_out = logits_out.view(B, S, V)[..., :self.vocab]
        # CE writes dlogits (scaled by loss_scale/count/world) over the logits buffer
        gscale = (self.loss_scale if loss_scale is None else loss_scale) / self.world
        loss, _ = self._cross_entropy(logits, tgt_t, self.vocab, inv, gscale)   # V = table rows (pad rows included), CE sees the logical vocabulary
        ctx.update(B=B, S=S, M=M, s_pad=s_pad, lens=lens, cu=cu, pos=pos, layers=layers, x_last=x_head, rstdN=rstdN, hN=hN, dlogits=logits,
                   table_rows=table.shape[0], count=count, head_idx=head_idx, head_rows=int(lab_rows.size), last_sel=last_sel)
        self.ctx = ctx
        self.last_logits = logits_out
        return loss

    def forward_embeds(self, inputs_embeds, attention_mask=None, labels=None):
        """The decoder + head on caller-supplied input embeddings -- the call form of LlavaLlamaForCausalLM.forward with `inputs_embeds`
        given (llava_llama.py:69-120: the multimodal splice is skipped and super().forward runs on the embeddings; what generate() and
        eval-loss loops use).  inputs_embeds [B, S, d]; attention_mask [B, S] (None = all valid; right or left padded); labels [B, S] or
        None.  Returns (loss or None, fp32 logits [B, S, vocab]); nothing is kept for a backward pass."""
        dev, l = self.device, self.l
        d, V, L = l["d"], l["vocab"], l["layers"]
        x3 = torch.as_tensor(inputs_embeds)
        B, S = int(x3.shape[0]), int(x3.shape[1])
        assert x3.shape[2] == d, f"inputs_embeds last dimension {x3.shape[2]} != hidden size {d}"
        am = np.ones((B, S), dtype=bool) if attention_mask is None else np.asarray(torch.as_tensor(attention_mask).cpu()).astype(bool)
        lens_np = am.sum(1).astype(np.int32)
        right_padded = all(am[b, :lens_np[b]].all() for b in range(B))
        x = x3.to(dev).reshape(B * S, d)
        x = (x if x.dtype == BF16 else ops.to_bf16(x.float().contiguous())).contiguous()
        s_pad = _ru(S, 64)
        cu = pos = valid_idx = lens = None
        if right_padded:
            M, lens = B * S, self._dev(lens_np)
        else:   # any other mask: valid tokens only, positions = padded column index (HF: position_ids = arange(S) when none are passed)
            valid = am.reshape(-1)
            valid_idx = np.nonzero(valid)[0]
            M = int(valid.sum())
            cu = self._dev(np.concatenate([[0], np.cumsum(lens_np)]).astype(np.int32))
            pos = self._dev((np.arange(B * S, dtype=np.int32) % S)[valid])
            x = x[self._dev(valid_idx.astype(np.int64))].contiguous()
        geom = dict(B=B, S=S, s_pad=s_pad, lens=lens, cu=cu, pos=pos, cs=self.rope_table(S))
        saved = (self.lora_step, getattr(self, "lora_p", 0.0))
        self.lora_p = 0.0                    # eval form: the adapters' input dropout is off (module.eval() in the reference)
        try:
            for i in range(L):
                x, _ = self._layer_forward(i, x, geom)
        finally:
            self.lora_step, self.lora_p = saved
        hN, _ = ops.rmsnorm_fwd(x, self.W("model.norm.weight"), self.eps)
        lf = ops.gemm_nt(hN, self.W("lm_head.weight"), out_dtype=torch.float32)
        if valid_idx is not None:
            full = torch.zeros(B * S, lf.shape[1], dtype=torch.float32, device=dev)
            full[self._dev(valid_idx.astype(np.int64))] = lf
            lf = full
        logits = lf.view(B, S, -1)[..., :self.vocab]
        loss = None
        if labels is not None:
            lab = np.asarray(torch.as_tensor(labels).cpu()).astype(np.int64)
            tgt = shifted_labels(np.where(am, lab, -100))
            if tgt.size and int(tgt.max()) >= self.vocab:
                raise IndexError(f"label {int(tgt.max())} is out of range for a vocabulary of {self.vocab}")
            count = int((tgt != -100).sum())
            tgt = tgt.reshape(-1)
            if valid_idx is not None:
                tgt = tgt[valid_idx]
            logits_bf = ops.gemm_nt(hN, self.W("lm_head.weight"))     # the loss reads bf16 logits, as the training path does
            loss, _ = self._cross_entropy(logits_bf, self._dev(tgt), self.vocab, (1.0 / count) if count else float("nan"), 0.0)
        self.last_logits = logits
        return loss, logits

    def _cross_entropy(self, logits, tgt, V, inv, gscale):
        from . import lib
        rows = logits.shape[0]
        loss_rows = torch.empty(rows, dtype=torch.float32, device=logits.device)
        lib.call("rv_cross_entropy", logits, logits.stride(0), tgt, loss_rows, logits, logits.stride(0), rows, V, inv * gscale)
        loss = torch.empty(1, dtype=torch.float32, device=logits.device)
        lib.call("rv_sum_f32", loss_rows, rows, inv, loss)
        return loss, loss_rows

    # ------------------------------------------------------------------ backward
    def _linear_bwd(self, dy, x, w, gw, need_dx=True, dx_out=None):
        """Backward of y = x W^T:  dW[N,K] = dY^T X  (both operands read contraction-major, in place) into the flat
        grad view (skipped when gw is None: frozen weight), and dX[M,K] = dY W (W read contraction-major) -- no
        transposed copies of weights or activations."""
        acc = self.grad_accum_started
        if gw is not None:
            ops.gemm(dy, x, ta=True, tb=True, out=gw, residual=gw if acc else None)
        if need_dx:
            return ops.gemm(dy, w, tb=True, out=dx_out)
        return None

    # ------------------------------------------------------------------ LoRA (peft LoraLayer semantics, train.py:1515-1532)
    def _lora_seed(self, i, lname):
        k = [t for t, _, _ in LORA_TARGETS].index(lname)
        return (self.lora_step * 1000003 + i * 131 + k) & 0x7FFFFFFFFFFFFFFF

    def _MODS_QKV(self, d):
        kvd = self.kvd      # == d without grouped-query attention
        return (("self_attn.q_proj", 0, d), ("self_attn.k_proj", d, d + kvd), ("self_attn.v_proj", d + kvd, d + 2 * kvd))

    def _lora_ws(self):
        if getattr(self, "_ws", None) is None:
            self._ws = torch.empty(32 << 20, dtype=torch.float32, device=self.device)   # split-K scratch (128 MiB)
        return self._ws

    def _lora_linear(self, x, w, i, mods, saved, residual=None, bias=None):
        """y[:, c0:c1] = x W[c0:c1]^T + t B^T (+ residual) with t = (alpha/r) dropout(x) A^T: ONE launch per adapted module,
        the adapter rides the main GEMM as a second operand pair (K + r), no second pass over y."""
        y = torch.empty(x.shape[0], w.shape[0], dtype=BF16, device=self.device)
        for lname, c0, c1 in mods:
            pre = f"model.layers.{i}.{lname}."
            A, B = self.W(pre + "lora_A.weight"), self.W(pre + "lora_B.weight")
            # t = (alpha / r) * dropout(x) A^T: one pass over x, the mask applied to the operand fragments (rv_lora_down_bf16)
            t = ops.lora_down(x, A, self.lora_scale, self.lora_p, self._lora_seed(i, lname))
            ops.gemm(x, w[c0:c1], out=y[:, c0:c1], bias=None if bias is None else bias[c0:c1],
                     residual=None if residual is None else residual[:, c0:c1], a2=t, b2=B)
            saved[lname] = t
        return y

    def _lora_linear_bwd(self, dy, x, w, i, mods, saved):
        """dx = sum_j dy_j W_j + (adapter path); dA, dB into the flat grad views (split-K: r-wide outputs, token-long K)."""
        acc = self.grad_accum_started
        ws = self._lora_ws()
        dx = None
        if self.lora_p > 0 and len(mods) > 1:
            # base path of a fused projection (q|k|v, gate|up): ONE input-gradient GEMM over the stacked frozen weight, dx = dy W, instead of
            # one per module accumulated through the residual (two fewer passes over dx, one long K loop); the adapters add into it below
            dx = ops.gemm(dy[:, mods[0][1]:mods[-1][2]], w[mods[0][1]:mods[-1][2]], tb=True)
        base_done = dx is not None
        for lname, c0, c1 in mods:
            pre = f"model.layers.{i}.{lname}."
            A, B = self.W(pre + "lora_A.weight"), self.W(pre + "lora_B.weight")
            gA, gB = self.G(pre + "lora_A.weight"), self.G(pre + "lora_B.weight")
            dyj = dy[:, c0:c1]
            t = saved[lname]                                        # already scaled by alpha/r
            # d(dropout(x) A^T) = (alpha / r) dy B: an r-wide product over a token-long stream -- the skinny one-pass kernel (B^T is 64 rows)
            dts = ops.lora_down(dyj, ops.transpose(B), self.lora_scale, 0.0, 0) if B.shape[1] <= 64 else ops.gemm(dyj, B, tb=True, alpha=self.lora_scale)
            ops.gemm(dyj, t, ta=True, tb=True, out=gB, residual=gB if acc else None, workspace=ws)
            first = dx is None
            if first:
                dx = torch.empty(x.shape[0], w.shape[1], dtype=BF16, device=self.device)
            if self.lora_p > 0:
                seed = self._lora_seed(i, lname)
                ops.lora_a_grad(dts, x, gA, self.lora_p, seed, acc, ws)     # gA (+)= dts^T dropout(x): the mask is re-created in registers
                if not base_done:
                    ops.gemm(dyj, w[c0:c1], tb=True, out=dx, residual=None if first else dx)
                ops.gemm_dropout_add(dts, A, dx, self.lora_p, seed)   # dx += dropout'(dts A): the mask is applied in the GEMM epilogue
            else:
                ops.gemm(dts, x, ta=True, tb=True, out=gA, residual=gA if acc else None, workspace=ws)
                ops.gemm(dyj, w[c0:c1], tb=True, out=dx, residual=None if first else dx, a2=dts, b2=A)
        return dx

    def _lm_linear_bwd(self, dy, x, w, gw, i, mods, saved):
        if self.lora:
            return self._lora_linear_bwd(dy, x, w, i, mods, saved)
        return self._linear_bwd(dy, x, w, gw)

    def backward(self):
        """Backward of the last forward(); gradients land in self.grads (bf16, flat)."""
        c = self.ctx
        assert c is not None, "forward() first"
        self._select_gemm_launch_shape()
        l = self.l
        d, F, H, V, L = l["d"], l["ffn"], l["heads"], l["vocab"], l["layers"]
        hd, Hkv, kvd = self.hd, self.Hkv, self.kvd
        B, S, M, s_pad, lens, cu, pos = c["B"], c["S"], c["M"], c["s_pad"], c["lens"], c["cu"], c["pos"]
        acc = self.grad_accum_started
        cs = self.rope_table(S)
        # head
        frozen_lm = self.base is not None          # LoRA or projector-only: the decoder weights receive no gradients
        dhN = self._linear_bwd(c["dlogits"], c["hN"], self.W("lm_head.weight"), None if frozen_lm else self.G("lm_head.weight"))
        dx, _ = ops.rmsnorm_bwd(dhN, c["x_last"], self.W("model.norm.weight"), c["rstdN"],
                                dw=None if frozen_lm else self.G("model.norm.weight"), dw_accumulate=acc)
        back = None
        if c["head_idx"] is not None:       # the head ran on the labelled rows: every other row of the last layer's output has zero gradient
            back = np.full(M, -1, dtype=np.int32)
            back[c["head_idx"][:c["head_rows"]]] = np.arange(c["head_rows"], dtype=np.int32)
            back = self._dev(back)
            if not c["last_sel"]:
                dx = ops.gather_rows(back, d, dx)
        if not frozen_lm:
            self._bucket_done("lm_head.weight", "lm_head.weight")
            self._bucket_done("model.norm.weight", "model.norm.weight")
        for i in reversed(range(L)):
            a = c["layers"][i]
            if "h1" not in a:      # activation recompute: this layer kept only its input
                _, a = self._layer_forward(i, a["x"], c["geom"], rows=self._dev(c["head_idx"]) if (c["last_sel"] and i == L - 1) else None)
            lv, gv = self._layer_views(i), self._layer_views(i, self.grads)
            sv = a["lora"]
            if self.lora or not self.fused:
                dact = self._lm_linear_bwd(dx, a["act"], lv["down"], gv.get("down"), i, (("mlp.down_proj", 0, d),), sv)
                dgu = ops.swiglu_bwd(dact, a["gu"], F)
            else:       # down_proj: weight gradient, then the input gradient with the SwiGLU backward in its epilogue (d(act) never stored)
                self._linear_bwd(dx, a["act"], lv["down"], gv.get("down"), need_dx=False)
                dgu = ops.gemm_swiglu_bwd(dx, lv["down"], a["gu"], F)
            dh2 = self._lm_linear_bwd(dgu, a["h2"], lv["gu"], gv.get("gu"), i, (("mlp.gate_proj", 0, F), ("mlp.up_proj", F, 2 * F)), sv)
            ops.rmsnorm_bwd(dh2, a["x_mid"], lv["ln2"], a["rstd2"], dx=dx, dx_add=True, dw=gv.get("ln2"), dw_accumulate=acc)
            dattn = self._lm_linear_bwd(dx, a.get("attn_sel", a["attn"]), lv["o"], gv.get("o"), i, (("self_attn.o_proj", 0, d),), sv)
            if "attn_sel" in a:       # last layer on the labelled rows: back to all token rows (zero elsewhere) for the attention backward
                dattn, dx = ops.gather_rows(back, d, dattn), ops.gather_rows(back, d, dx)
            qkv = a["qkv"]
            dqkv = torch.empty_like(qkv)
            # the rotary embedding's adjoint runs in the dQ / dK epilogues: dqkv = gradient of the un-rotated q|k|v projection
            ops.attn_bwd(qkv[:, :d], qkv[:, d:d + kvd], qkv[:, d + kvd:], a["attn"], dattn, a["lse"], B, S, H, hd, s_pad, True,
                         lens=lens, dq=dqkv[:, :d], dk=dqkv[:, d:d + kvd], dv=dqkv[:, d + kvd:], kv_heads=Hkv, cu=cu,
                         rope=(cs, pos) if self.fused else None, delta=self._stat_buffer("delta", B, H, s_pad))
            if not self.fused:
                ops.rope_inplace(dqkv, cs, S, H + Hkv, hd, 1, -1, positions=pos)
            if "bqkv" in gv:
                ops.bias_grad(dqkv, out=gv["bqkv"], accumulate=acc)
            dh1 = self._lm_linear_bwd(dqkv, a["h1"], lv["qkv"], gv.get("qkv"), i, self._MODS_QKV(d), sv)
            ops.rmsnorm_bwd(dh1, a["x"], lv["ln1"], a["rstd1"], dx=dx, dx_add=True, dw=gv.get("ln1"), dw_accumulate=acc)
            c["layers"][i] = None  # free this layer's activations
            p = f"model.layers.{i}."
            if self.freeze_lm:
                pass                        # nothing trainable inside the layer
            elif frozen_lm:
                self._bucket_done(p + "self_attn.q_proj.lora_A.weight", p + "mlp.down_proj.lora_B.weight")
            else:
                self._bucket_done(p + "input_layernorm.weight", p + "mlp.down_proj.weight")
        # dx = gradient of inputs_embeds [M, d]
        plan = c["plan"]
        dev = self.device
        # projector: rows of dx at the positions where projector outputs were spliced in (unused rows -> zero)
        n_rows = plan["n_feat_rows"]
        fpos = self._dev(plan["feat_pos"])
        dfeat = ops.gather_rows(fpos, d, dx)
        rs = plan.get("resample")
        mp = plan.get("maxpool")
        if mp:   # adjoint of the 2x2 max pooling: each pooled row's gradient goes to the winning source element
            ops.max4_rows_bwd(dfeat, self._dev(mp["idx4"]), self._dev(mp["out"]), c["pool_which"], dfeat)
            dfeat = dfeat[:plan["n_proj_rows"]]
        if rs:   # adjoint of the bilinear down-sampling: gradients of the created rows flow to their 4 source rows
            t = lambda k: self._dev(rs[k])
            ops.weighted_segment_sum_rows(dfeat, t("adj_off"), t("adj_pos"), t("adj_w"), t("adj_out"), dfeat)
            dfeat = dfeat[:plan["n_proj_rows"]]
        g = self.G
        ops.bias_grad(dfeat, out=g("model.mm_projector.2.bias"), accumulate=acc)
        da1 = self._linear_bwd(dfeat, c["a1"], self.W("model.mm_projector.2.weight"), g("model.mm_projector.2.weight"))
        dz1 = ops.gelu_bwd(da1, c["z1"])
        ops.bias_grad(dz1, out=g("model.mm_projector.0.bias"), accumulate=acc)
        df0 = self._linear_bwd(dz1, c["f0"], self.W("model.mm_projector.0.weight"), g("model.mm_projector.0.weight"),
                               need_dx=self.train_tower)
        if self.with_newline:
            gn = g("model.image_newline").view(1, d)
            if not acc:
                gn.zero_()
            npos = plan["newline_pos"]
            if npos.size:
                tmp = torch.zeros(1, d, dtype=BF16, device=dev)
                ops.segment_sum_rows(dx, self._dev(np.array([0, npos.size], dtype=np.int32)),
                                     self._dev(npos), torch.zeros(1, dtype=torch.int32, device=dev), tmp)
                gn.add_(tmp) if acc else gn.copy_(tmp)
        # embedding rows: segment sums by token id (no atomics)
        train_embed = "model.embed_tokens.weight" in self.lm.offsets     # full fine-tune, or the pretraining stage with extra tokens
        ge = g("model.embed_tokens.weight") if train_embed else None
        if not train_embed:
            pass
        elif acc:
            tmp = torch.zeros_like(ge)
            ops.segment_sum_rows(dx, self._dev(plan["tok_off"]), self._dev(plan["tok_pos"]),
                                 self._dev(plan["tok_ids"]), tmp)
            ge.add_(tmp)
        else:
            ge.zero_()
            ops.segment_sum_rows(dx, self._dev(plan["tok_off"]), self._dev(plan["tok_pos"]),
                                 self._dev(plan["tok_ids"]), ge)
        last = "model.image_newline" if self.with_newline else "model.mm_projector.2.bias"
        self._bucket_done("model.embed_tokens.weight" if train_embed else "model.mm_projector.0.weight", last)
        if self.train_tower:
            if self.has_cls:
                n_img = c["v_n"]
                put = torch.full((n_img, self.N_vis), -1, dtype=torch.int32, device=dev)
                put[:, 1:] = torch.arange(n_img * self.P, device=dev, dtype=torch.int32).view(n_img, self.P)
                dhid = ops.gather_rows(put.view(-1), self.v["d"], df0)   # CLS rows receive zero
            else:
                dhid = df0
            self.vision_backward(dhid, c)
        self.ctx = None
        self.grad_accum_started = True

    # ------------------------------------------------------------------ data-parallel gradient sync
    def _bucket_done(self, first, last):
        """Gradients of tensors first..last are final: start their all-reduce on the comm stream (RCCL), in place."""
        if self.sync is None or not self.sync_this_backward:
            return
        s, e = self.lm.span(first, last)
        self.sync.bucket_done(s, e)

    sync_this_backward = True

    def finish_grad_sync(self):
        if self.sync is not None:
            self.sync.finish()

    def zero_grad(self):
        self.grad_accum_started = False

    def _replica_buffers(self):
        bufs = [("parameters", self.lm.flat)]
        if self.vis is not self.lm:
            bufs.append(("vision_tower", self.vis.flat))
        if self.base is not None:
            bufs.append(("frozen_language_model", self.base.flat))
        return bufs + [("fp32_master", self.master), ("exp_avg", self.m), ("exp_avg_sq", self.vv)]

    def broadcast_parameters(self):
        """Rank 0's parameters (and optimizer state, once it exists) become every rank's: the broadcast DistributedDataParallel performs
        when it wraps the model (HF Trainer / accelerate in the reference).  No-op for one rank."""
        if self.world > 1:
            from .ddp import broadcast_from_rank0
            broadcast_from_rank0(self._replica_buffers(), self.pg)
            self.weights_changed(tower=True)

    def check_replicas(self, where=""):
        """Raise unless every rank holds bit-identical parameters, frozen stores and optimizer state (all-reduced checksums)."""
        if self.world > 1:
            from .ddp import assert_replicas_equal
            assert_replicas_equal(self._replica_buffers(), self.pg, where)

    def barrier(self):
        if self.world > 1:
            torch.distributed.barrier(group=self.pg)

    # ------------------------------------------------------------------ optimizer
    def init_optimizer(self):
        if self.master is None:
            self.master = ops.to_f32(self.lm.flat)
            self.m = self.lm.like(torch.float32)
            self.vv = self.lm.like(torch.float32)

    def param_groups(self, lr, weight_decay=0.0, mm_projector_lr=None, no_decay_1d=True, mm_vision_tower_lr=None):
        """Contiguous (start, end, lr, wd) slices following LLaVATrainer.create_optimizer (llava_trainer.py:369-418):
        no weight decay for norm weights and biases; optional separate LR for the projector."""
        groups = []
        fresh = True             # a frozen tensor in between: the next group must not be merged across its slice
        frozen = getattr(self, "frozen_names", ())
        for name in self.lm.names():
            if name in frozen:
                fresh = True
                continue
            off, n = self.lm.offsets[name]
            shp = self.lm.shapes[name]
            nodecay = no_decay_1d and (len(shp) == 1 and ("norm" in name or name.endswith("bias")))
            glr = mm_projector_lr if (mm_projector_lr is not None and "mm_projector" in name) else lr
            if mm_vision_tower_lr is not None and "vision_tower" in name:
                glr = mm_vision_tower_lr
            gwd = 0.0 if nodecay else weight_decay
            if groups and not fresh and groups[-1][2] == glr and groups[-1][3] == gwd and groups[-1][1] <= off:
                groups[-1][1] = off + n  # merge (alignment gaps hold zeros and stay zero)
            else:
                groups.append([off, off + n, glr, gwd])
            fresh = False
        return [tuple(g) for g in groups]

    def optimizer_step(self, lr, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=None, mm_projector_lr=None,
                       mm_vision_tower_lr=None):
        """AdamW over the flat buffers (fp32 master + moments), optional global-norm clipping without host sync."""
        self.init_optimizer()
        self.finish_grad_sync()
        self.opt_step += 1
        coef = None
        self.last_grad_norm = None
        for name in self.frozen_names:          # frozen tensors of the buffer take no part in the gradient norm
            self.G(name).zero_()
        if max_grad_norm is not None and max_grad_norm > 0:
            nc = ops.grad_norm_clip_coef(self.grads, max_grad_norm)
            self.last_grad_norm = nc[0:1]
            coef = nc[1:2]
        for s, e, glr, gwd in self.param_groups(lr, weight_decay, mm_projector_lr, mm_vision_tower_lr=mm_vision_tower_lr):
            ops.adamw(self.lm.flat[s:e], self.master[s:e], self.grads[s:e], self.m[s:e], self.vv[s:e], glr, betas[0], betas[1],
                      eps, gwd, self.opt_step, gscale=coef)
        self.weights_changed(tower=self.train_tower)
        self.zero_grad()

    # ------------------------------------------------------------------ state dict (reference names)
    def state_dict(self):
        """Every tensor under its reference name: trainable buffer, frozen tower and -- for LoRA / projector-only runs -- the frozen
        language-model store (adapters keep their .lora_A/.lora_B names; lora_state_dict() gives peft's layout)."""
        out = {}
        stores = [self.lm] + ([] if self.vis is self.lm else [self.vis]) + ([self.base] if self.base is not None else [])
        for fp in stores:
            for n in fp.names():
                if n in out:
                    continue        # the projector lives in the trainable buffer; the base store's copy of it is unused
                out[n] = fp.view(n)[:self.vocab] if n in ("model.embed_tokens.weight", "lm_head.weight") else fp.view(n)
        return out

    def lora_state_dict(self):
        """Adapter tensors under peft's key layout (base_model.model.<module>.lora_{A,B}.default.weight) + the
        non-LoRA trainables (the reference saves them as non_lora_trainables.bin, train/train.py:1708-1717)."""
        assert self.lora
        adapters, others = {}, {}
        for n in self.lm.names():
            if ".lora_" in n:
                adapters["base_model.model." + n.replace(".weight", ".default.weight")] = self.lm.view(n)
            else:
                others[n] = self.lm.view(n)
        return adapters, others

    def load_state_dict(self, sd, strict=False):
        from .params import load_named
        missing, rest = load_named(self.lm, sd)
        for fp in ([] if self.vis is self.lm else [self.vis]) + ([self.base] if self.base is not None else []):
            m, rest2 = load_named(fp, {k: v for k, v in sd.items() if k in rest})
            if fp is self.base:     # the base store also has projector slots (unused): not "missing"
                m = [k for k in m if k not in self.lm.offsets]
            missing, rest = missing + m, rest2
        self.weights_changed()
        if self.master is not None:
            self.master.copy_(ops.to_f32(self.lm.flat))
        if strict and (missing or rest):
            raise KeyError(f"missing={missing[:4]} unexpected={rest[:4]}")
        return missing, rest
