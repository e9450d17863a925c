"""``LlavaQwenConfig`` / ``LlavaQwenModel`` / ``LlavaQwenForCausalLM`` over the MI355X engine.

Mirrors reference finetuning/llava/model/language_model/llava_qwen.py:35-149: the same class names, ``forward`` keyword
list and outputs as the Llama flavour; the decoder is Qwen2 (grouped-query attention, q/k/v biases, rope theta 1e6,
rms eps 1e-6) and the tower handle is ``SigLipVisionTower`` (multimodal_encoder/siglip_encoder.py:538-620: 729 tokens,
no class token, last encoder layer dropped).  The arithmetic lives in ``radvlm_amd.engine``; this file is the interface.
"""
import torch

from .llava_llama import CLIPVisionTower, LlavaConfig, LlavaLlamaForCausalLM, LlavaLlamaModel


class LlavaQwenConfig(LlavaConfig):
    model_type = "llava_qwen"

    def __init__(self, geometry=None, **kw):
        from ...config import GEOMETRIES
        geometry = geometry or GEOMETRIES["llava_ov_qwen2_7b"]
        l = geometry["lm"]
        kw.setdefault("rms_norm_eps", l.get("rms_eps", 1e-6))
        kw.setdefault("rope_theta", l.get("rope_theta", 1e6))
        super().__init__(geometry=geometry, **kw)
        self.num_key_value_heads = l.get("kv_heads", l["heads"])
        self.rope_scaling = None     # llava_qwen.py:53


class SigLipVisionTower(CLIPVisionTower):
    """Attributes the reference reads off its SigLIP tower (siglip_encoder.py:592-620)."""

    def __init__(self, engine):
        super().__init__(engine)
        from ..mm_utils import SigLipImageProcessor
        s = engine.v["image"]
        self.image_processor = SigLipImageProcessor(size=(s, s), crop_size={"height": s, "width": s})

    def __call__(self, images):
        """[n,3,H,W] -> [n,729,dv]: hidden_states[-1] of the truncated encoder (siglip_encoder.py:576-590)."""
        from ... import ops
        e = self._e
        x = images.to(e.device)
        x = x if x.dtype == torch.bfloat16 else ops.to_bf16(x.float())
        return e.vision_forward(x.contiguous()).view(x.shape[0], e.P, e.v["d"])


class LlavaQwenModel(LlavaLlamaModel):
    config_class = LlavaQwenConfig

    def __init__(self, engine, config):
        super().__init__(engine, config)
        if engine.siglip:
            self.vision_tower = SigLipVisionTower(engine)


class LlavaQwenForCausalLM(LlavaLlamaForCausalLM):
    config_class = LlavaQwenConfig
    model_class = LlavaQwenModel
