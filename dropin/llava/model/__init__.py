from radvlm_amd.llava.model import LlavaConfig, LlavaLlamaForCausalLM, LlavaLlamaModel  # noqa: F401
from radvlm_amd.llava.model import LlavaQwenConfig, LlavaQwenForCausalLM, LlavaQwenModel  # noqa: F401
