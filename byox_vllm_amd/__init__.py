"""MI355X-native paged-KV decode path behind nanovllm's Python surface.

    from byox_vllm_amd import LLM, SamplingParams

mirrors ``from nanovllm import LLM, SamplingParams`` (nanovllm/__init__.py:2-4).
The engine modules import torch lazily through their own imports; importing
this package does not require a GPU, running a model does.
"""

from byox_vllm_amd.sampling_params import SamplingParams  # noqa: F401
from byox_vllm_amd.utils.logging import set_global_log_level  # noqa: F401


def __getattr__(name):
    if name == "LLM":
        from byox_vllm_amd.llm import LLM
        return LLM
    raise AttributeError(name)
