"""Host-side mirror of the reference's ``llava`` package for the training hot path.

Same module names, symbols and argument meaning as /root/reference/finetuning/llava so the reference's fine-tune
scripts can import it; the arithmetic underneath is radvlm_amd.engine (HIP, MI355X).
"""
from .model import LlavaConfig, LlavaLlamaForCausalLM  # noqa: F401
