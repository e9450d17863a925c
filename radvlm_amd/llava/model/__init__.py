from .llava_llama import LlavaConfig, LlavaLlamaForCausalLM, LlavaLlamaModel  # noqa: F401
