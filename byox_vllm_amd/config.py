"""Engine configuration (mirrors nanovllm/config.py:7-27).

Differences from the reference, all supersets:
* ``hf_config`` may be passed in (any object with the HF attribute names); the
  model directory is then optional.  Needed because synthetic-weight runs have
  no checkpoint directory.
* ``kvcache_block_size`` may be any multiple of 16 (the reference demands 256
  because of flash-attn; the HIP kernels take the block size at run time).
* ``state_dict`` supplies an in-memory checkpoint (packed parameter names) instead
  of safetensors files; with neither, synthetic weights are drawn from ``seed``.
* ``xgmi_allreduce`` (or NV_XGMI_ALLREDUCE=1) routes the tensor-parallel all-reduces through
  ``nv_allreduce_add_rmsnorm`` (peer-mapped device memory) instead of torch.distributed.
* ``quantization`` selects the weight format of the layer linears:
  ``None`` (dense), ``"w8a16"`` or ``"w4a16"``.
"""

import os
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Any

import torch


def _load_hf_config(path: str):
    """Read config.json defensively (transformers 5.x moved rope/dtype fields)."""
    import json

    with open(os.path.join(path, "config.json")) as f:
        raw = json.load(f)
    rope = raw.get("rope_parameters") or {}
    rope_theta = raw.get("rope_theta", rope.get("rope_theta", 1000000))
    rope_scaling = raw.get("rope_scaling")
    if isinstance(rope_scaling, dict) and rope_scaling.get("rope_type", rope_scaling.get("type")) in (None, "default"):
        rope_scaling = None
    dtype = raw.get("torch_dtype", raw.get("dtype", "bfloat16"))
    cfg = SimpleNamespace(**raw)
    cfg.rope_theta = rope_theta
    cfg.rope_scaling = rope_scaling
    cfg.dtype = getattr(torch, dtype) if isinstance(dtype, str) else dtype
    if not hasattr(cfg, "tie_word_embeddings"):
        cfg.tie_word_embeddings = False
    return cfg


@dataclass
class Config:
    model: str
    max_num_batched_tokens: int = 16384
    max_num_seqs: int = 512
    max_model_len: int = 4096
    gpu_memory_utilization: float = 0.9
    tensor_parallel_size: int = 1
    enforce_eager: bool = False
    hf_config: Any = None
    eos: int = -1
    kvcache_block_size: int = 256
    num_kvcache_blocks: int = -1
    quantization: str | None = None
    seed: int = 0
    xgmi_allreduce: bool = False    # tensor parallel: the library's own all-reduce fused with add-RMSNorm instead of RCCL
    graph_head: bool = True         # single rank: LM head + sampler inside the decode graph (False: forward only)
    graph_batch_sizes: Any = None   # optional list: capture decode graphs only for these padded batch sizes
    state_dict: Any = None   # optional in-memory checkpoint keyed by the packed parameter names
    tp_port: int = 0         # rendezvous port of this engine's tensor-parallel ranks (0: LLMEngine picks a free one)
    shm_name: str = ""       # name of this engine's control-message shared memory ("": LLMEngine derives a unique one)

    def __post_init__(self):
        assert self.kvcache_block_size % 16 == 0 and self.kvcache_block_size > 0
        assert 1 <= self.tensor_parallel_size <= 8
        assert self.quantization in (None, "w8a16", "w4a16")
        if self.hf_config is None:
            assert os.path.isdir(self.model)
            self.hf_config = _load_hf_config(self.model)
            if self.quantization is None:       # a checkpoint written by utils/quantize.py names its format
                self.quantization = getattr(self.hf_config, "quantization", None)
        if getattr(self.hf_config, "dtype", None) is None:
            self.hf_config.dtype = torch.bfloat16
        self.max_model_len = min(self.max_model_len, self.hf_config.max_position_embeddings)
        assert self.max_num_batched_tokens >= self.max_model_len


def qwen2_config(name: str, dtype=torch.bfloat16, **overrides):
    """Public Qwen2 architecture shapes (no checkpoint needed)."""
    shapes = {
        "qwen2-0.5b": dict(hidden_size=896, intermediate_size=4864, num_hidden_layers=24,
                           num_attention_heads=14, num_key_value_heads=2, vocab_size=151936,
                           tie_word_embeddings=True),
        "qwen2-7b": dict(hidden_size=3584, intermediate_size=18944, num_hidden_layers=28,
                         num_attention_heads=28, num_key_value_heads=4, vocab_size=152064,
                         tie_word_embeddings=False),
        "qwen3-0.6b": dict(hidden_size=1024, intermediate_size=3072, num_hidden_layers=28,
                           num_attention_heads=16, num_key_value_heads=8, vocab_size=151936,
                           head_dim=128, tie_word_embeddings=True, attention_bias=False),
    }[name]
    base = dict(max_position_embeddings=32768, rms_norm_eps=1e-6, hidden_act="silu",
                attention_bias=True, rope_theta=1000000.0, rope_scaling=None, dtype=dtype)
    base.update(shapes)
    base.update(overrides)
    if "head_dim" not in base:
        base["head_dim"] = base["hidden_size"] // base["num_attention_heads"]
    return SimpleNamespace(**base)
