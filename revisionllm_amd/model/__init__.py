from .revision_llama import ReVisionLlamaForCausalLM, VTimeLLMLlamaForCausalLM  # noqa: F401
