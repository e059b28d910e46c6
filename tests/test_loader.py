"""Checkpoint loading (nanovllm/utils/loader.py:13-29): a tiny checkpoint with the HF parameter names
(q_proj / k_proj / v_proj, gate_proj / up_proj, tied LM head) written as safetensors and loaded through
utils.loader.load_model must fill the packed modules exactly as the in-memory packed state dict does."""

import json
import os

import pytest
import torch

from oracle import model as OM


def hf_named(cfg, packed):
    """Packed-name checkpoint -> the names a HuggingFace Qwen2/Qwen3 checkpoint uses."""
    d = cfg.head_dim
    nq, nkv = cfg.num_attention_heads * d, cfg.num_key_value_heads * d
    out = {}
    for name, w in packed.items():
        if name == "lm_head.weight" and cfg.tie_word_embeddings:
            continue                                   # tied checkpoints do not store the head
        if "qkv_proj" in name:
            for part, tag in zip(w.split([nq, nkv, nkv], dim=0), ("q_proj", "k_proj", "v_proj")):
                out[name.replace("qkv_proj", tag)] = part.contiguous()
        elif "gate_up_proj" in name:
            for part, tag in zip(w.chunk(2, 0), ("gate_proj", "up_proj")):
                out[name.replace("gate_up_proj", tag)] = part.contiguous()
        else:
            out[name] = w.contiguous()
    return out


def build(cfg, tp_rank=0, tp_size=1):
    from byox_vllm_amd.layers.linear import set_tp
    from byox_vllm_amd.layers.rotary_embedding import get_rope
    from byox_vllm_amd.models.qwen3 import Qwen3ForCausalLM
    get_rope.cache_clear()
    set_tp(tp_rank, tp_size)
    torch.set_default_dtype(cfg.dtype)
    try:
        return Qwen3ForCausalLM(cfg)
    finally:
        torch.set_default_dtype(torch.float32)
        set_tp(0, 1)


@pytest.mark.parametrize("style", ["qwen2", "qwen3"])
@pytest.mark.parametrize("tp", [(0, 1), (0, 2), (1, 2)])
def test_safetensors_checkpoint_loads_like_packed_state_dict(tmp_path, style, tp):
    from safetensors.torch import save_file

    from byox_vllm_amd.utils.loader import load_model, load_state_dict_packed
    cfg = OM.tiny_config(style, torch.bfloat16)
    packed = OM.random_weights(cfg, seed=3)
    named = hf_named(cfg, packed)
    assert any("q_proj" in k for k in named) and any("up_proj" in k for k in named)
    assert ("lm_head.weight" in named) == (not cfg.tie_word_embeddings)
    # two shards, as large checkpoints come
    keys = sorted(named)
    save_file({k: named[k] for k in keys[::2]}, str(tmp_path / "model-00001-of-00002.safetensors"))
    save_file({k: named[k] for k in keys[1::2]}, str(tmp_path / "model-00002-of-00002.safetensors"))
    with open(tmp_path / "config.json", "w") as f:
        json.dump({"architectures": ["Qwen2ForCausalLM"]}, f)

    a, b = build(cfg, *tp), build(cfg, *tp)
    for p in list(a.parameters()) + list(b.parameters()):
        p.data.fill_(float("nan"))                     # whatever the loaders do not write stays visible
    load_model(a, str(tmp_path))
    load_state_dict_packed(b, packed, cfg)
    pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
    assert pa.keys() == pb.keys()
    for name in pa:
        assert torch.isfinite(pa[name].float()).all(), name
        assert torch.equal(pa[name], pb[name]), name
    if cfg.tie_word_embeddings:
        assert a.lm_head.weight.data_ptr() == a.model.embed_tokens.weight.data_ptr()
    # the reference's shards for this rank (oracle restatement of its weight loaders, pinned by tests/golden/sharding.npz)
    rank, size = tp
    if cfg.num_key_value_heads % size:
        return          # more ranks than KV heads: replication beyond the reference's rules (tests/test_tp_gloo.py)
    q = pa["model.layers.0.self_attn.qkv_proj.weight"]
    assert torch.equal(q, OM.shard_qkv(packed["model.layers.0.self_attn.qkv_proj.weight"], cfg, rank, size))
    assert torch.equal(pa["model.layers.1.mlp.down_proj.weight"],
                       OM.shard_cols(packed["model.layers.1.mlp.down_proj.weight"], rank, size))


def test_missing_checkpoint_is_an_error_not_random_weights(tmp_path):
    """A directory without *.safetensors must not silently become a random model (the reference fails on it too)."""
    from byox_vllm_amd.engine.model_runner import ModelRunner
    runner = ModelRunner.__new__(ModelRunner)
    from types import SimpleNamespace
    runner.config = SimpleNamespace(model=str(tmp_path), state_dict=None, seed=0, hf_config=None)
    runner.model = None
    with pytest.raises(FileNotFoundError):
        runner._load_weights()
