from .llava_llama import LlavaConfig, LlavaLlamaForCausalLM, LlavaLlamaModel  # noqa: F401
from .llava_qwen import LlavaQwenConfig, LlavaQwenForCausalLM, LlavaQwenModel  # noqa: F401
