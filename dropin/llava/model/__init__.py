from radvlm_amd.llava.model import LlavaConfig, LlavaLlamaForCausalLM, LlavaLlamaModel  # noqa: F401
