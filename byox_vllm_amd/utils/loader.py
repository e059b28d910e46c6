"""Checkpoint loading (contract of nanovllm/utils/loader.py:13-29) and synthetic weights.

``load_model`` feeds every safetensors entry to the owning parameter's
``weight_loader`` (which performs the tensor-parallel slicing), resolving the
HF names q_proj/k_proj/v_proj and gate_proj/up_proj onto the packed modules.
``load_state_dict_packed`` does the same from an in-memory dict whose keys are
already the packed names (synthetic checkpoints and tests).
"""

import os
from glob import glob

import torch
from torch import nn


def default_weight_loader(param: nn.Parameter, loaded_weight: torch.Tensor):
    param.data.copy_(loaded_weight)


def _load_quantized(model, packed, pending, name, tensor, quantization):
    """A ``.qweight`` / ``.scales`` pair of an offline-quantised checkpoint (utils/quantize.py): once both halves of a
    matrix have been read they go to the owning layer's load_quantized, which shards them like a dense weight."""
    base, kind = name.rsplit(".", 1)
    pair = pending.setdefault(base, {})
    pair[kind] = tensor
    if len(pair) < 2:
        return
    del pending[base]
    for hf_key, (packed_key, shard_id) in packed.items():
        if hf_key in base:
            model.get_submodule(base.replace(hf_key, packed_key)).load_quantized(pair["qweight"], pair["scales"],
                                                                                 quantization, shard_id)
            return
    model.get_submodule(base).load_quantized(pair["qweight"], pair["scales"], quantization)


def _load_one(model, packed, name, tensor):
    for hf_key, (packed_key, shard_id) in packed.items():
        if hf_key in name:
            param = model.get_parameter(name.replace(hf_key, packed_key))
            param.weight_loader(param, tensor, shard_id)
            return
    param = model.get_parameter(name)
    getattr(param, "weight_loader", default_weight_loader)(param, tensor)


def load_model(model: nn.Module, path: str, quantization: str | None = None):
    from safetensors import safe_open
    packed = getattr(model, "packed_modules_mapping", {})
    pending: dict = {}
    for file in sorted(glob(os.path.join(path, "*.safetensors"))):
        with safe_open(file, "pt", "cpu") as f:
            for name in f.keys():  # noqa: SIM118
                if name.endswith((".qweight", ".scales")):
                    assert quantization is not None, f"{name}: quantised checkpoint, but no quantisation was configured"
                    _load_quantized(model, packed, pending, name, f.get_tensor(name), quantization)
                else:
                    _load_one(model, packed, name, f.get_tensor(name))
    assert not pending, f"qweight/scales without their other half: {sorted(pending)}"


def load_state_dict_packed(model: nn.Module, weights: dict, cfg) -> None:
    """Load a dict keyed by the packed names (qkv_proj / gate_up_proj hold the FULL, unsharded tensors)."""
    d = cfg.head_dim
    nq, nkv = cfg.num_attention_heads * d, cfg.num_key_value_heads * d
    tied = getattr(cfg, "tie_word_embeddings", False)
    for name, w in weights.items():
        if name == "lm_head.weight" and tied:
            continue
        param = model.get_parameter(name)
        if "qkv_proj" in name:
            for part, sid in zip(w.split([nq, nkv, nkv], dim=0), ("q", "k", "v")):
                param.weight_loader(param, part, sid)
        elif "gate_up_proj" in name:
            for sid, part in enumerate(w.chunk(2, 0)):
                param.weight_loader(param, part, sid)
        else:
            getattr(param, "weight_loader", default_weight_loader)(param, w)


@torch.no_grad()
def init_random_weights(model: nn.Module, seed: int = 0) -> None:
    """Synthetic checkpoint drawn on the device: N(0, 0.02^2) matrices/biases, unit norm weights.

    Each parameter (already sharded for this rank) gets its own stream keyed by
    its name, so the draw does not depend on parameter order.  Used for
    throughput runs when the model directory holds no safetensors files; parity
    tests load an explicit state dict instead."""
    gens: dict = {}
    for name, param in model.named_parameters():
        if name == "lm_head.weight" and param.data_ptr() == model.model.embed_tokens.weight.data_ptr():
            continue
        if "norm" in name:
            param.data.fill_(1.0)
            continue
        dev = param.device
        g = gens.setdefault(dev, torch.Generator(device=dev).manual_seed(seed))
        # per-shard draw keyed by name keeps memory bounded for 7B-class models
        g.manual_seed(seed * 1000003 + (hash_name(name) % 1000003))
        param.data.copy_(torch.randn(param.shape, generator=g, device=dev, dtype=torch.float32) * 0.02)
    # the vocabulary matrices were rewritten without their weight_loader: layouts derived from them are stale
    from byox_vllm_amd.layers.embed_head import VocabParallelEmbedding
    VocabParallelEmbedding.loads += 1


def hash_name(name: str) -> int:
    h = 1469598103934665603
    for ch in name.encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h
