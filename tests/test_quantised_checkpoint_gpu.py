"""Offline-quantised checkpoints end to end on the GPU: HF-named safetensors -> `python -m byox_vllm_amd.utils.quantize`
-> `LLM(directory)` (utils.loader.load_model inside ModelRunner.__init__, the reference's model_runner.py:33 and
utils/loader.py:13-29) -> greedy tokens, on one rank and on two ranks sharing the GPU.

The oracle runs on the dequantised weights.  On two ranks the geometry (6 query heads of 64 over 2 ranks: 192 o_proj
input columns per rank) makes rank 1's first 128-column K-block straddle two quantisation groups of the stored matrix;
the format's rule for such a block - re-quantise it from the dequantised values where the two groups' scales differ - is
restated below for the oracle's weights.
"""

import json
import os
import subprocess
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import model as OM
from oracle import quant as Q
from oracle_tokens import check_tokens_against_oracle
from test_loader import hf_named

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BS = 16

HF = dict(vocab_size=256, hidden_size=384, intermediate_size=512, num_hidden_layers=2, num_attention_heads=6,
          num_key_value_heads=2, head_dim=64, max_position_embeddings=128, rms_norm_eps=1e-6, hidden_act="silu",
          attention_bias=True, rope_theta=1000000.0, tie_word_embeddings=True, torch_dtype="bfloat16")


def _cfg():
    return SimpleNamespace(rope_scaling=None, dtype=torch.bfloat16, **{k: v for k, v in HF.items() if k != "torch_dtype"})


@pytest.fixture(scope="module")
def checkpoints(tmp_path_factory):
    """A dense HF-named checkpoint directory and its w4a16 / w8a16 images written by the offline quantiser's CLI."""
    from safetensors.torch import save_file
    cfg = _cfg()
    weights = OM.random_weights(cfg, seed=7)
    src = tmp_path_factory.mktemp("dense")
    named = hf_named(cfg, weights)
    keys = sorted(named)
    save_file({k: named[k] for k in keys[::2]}, str(src / "model-00001-of-00002.safetensors"))
    save_file({k: named[k] for k in keys[1::2]}, str(src / "model-00002-of-00002.safetensors"))
    with open(src / "config.json", "w") as f:
        json.dump(dict(HF, architectures=["Qwen2ForCausalLM"]), f)
    dirs = {None: str(src)}
    for quant in ("w4a16", "w8a16"):
        dst = tmp_path_factory.mktemp(quant)
        out = subprocess.run([sys.executable, "-m", "byox_vllm_amd.utils.quantize", str(src), str(dst), quant],
                             cwd=ROOT, capture_output=True, text=True, timeout=300,
                             env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
        assert out.returncode == 0, out.stderr[-2000:]
        assert json.load(open(dst / "config.json"))["quantization"] == quant
        dirs[quant] = str(dst)
    return cfg, weights, dirs


def _generate(model, tp=1, eager=True, **kw):
    from byox_vllm_amd import LLM, SamplingParams
    from byox_vllm_amd.layers.rotary_embedding import get_rope
    get_rope.cache_clear()
    rng = np.random.RandomState(31)
    prompts = [rng.randint(0, 256, size=n).tolist() for n in (9, 23, 17)]
    llm = LLM(model, kvcache_block_size=BS, num_kvcache_blocks=16, enforce_eager=eager, max_model_len=128,
              max_num_batched_tokens=256, max_num_seqs=8, tensor_parallel_size=tp, **kw)
    try:
        quantization = llm.config.quantization
        packed = [m.packed for m in llm.model_runner.model.modules() if getattr(m, "packed", None) is not None]
        assert packed and all(p.data.is_cuda and (p.scales is None or p.scales.is_cuda) for p in packed)
        outs = llm.generate(prompts, SamplingParams(temperature=0.0, max_tokens=8, ignore_eos=True), use_tqdm=False)
    finally:
        llm._exit()
    return prompts, [o["token_ids"] for o in outs], quantization


@pytest.mark.parametrize("quant", ["w4a16", "w8a16"])
@pytest.mark.parametrize("eager", [True, False])
def test_offline_quantised_checkpoint_on_one_rank(checkpoints, quant, eager):
    """LLM(quantised directory): the format is read from config.json, every kernel-side weight is device memory, and
    the tokens are (a) the teacher-forced oracle's on the dequantised weights and (b) identical to load-time
    quantisation of the dense checkpoint (the same integers and scales reach the kernel either way)."""
    cfg, weights, dirs = checkpoints
    prompts, got, seen = _generate(dirs[quant], eager=eager)
    assert seen == quant
    check_tokens_against_oracle(cfg, OM.fake_quantize(weights, 4 if quant == "w4a16" else 8), prompts, got,
                                block_size=BS, num_blocks=16, label=f"checkpoint {quant} {'eager' if eager else 'graph'}")
    _, load_time, _ = _generate(dirs[None], eager=eager, quantization=quant)
    assert got == load_time


def _requantise_straddling(cfg, weights, bits, tp):
    """The oracle's weights for a tensor-parallel load of the quantised checkpoint: the dequantised matrices, with
    every local 128-column block of o_proj that spans two stored groups whose scales differ re-quantised as one
    group from its dequantised (model-dtype) values.  Restates the format's rule for unaligned K shards; rows whose two
    groups share one scale keep their stored integers."""
    quantize, dequantize = (Q.quantize_w4, Q.dequantize_w4) if bits == 4 else (Q.quantize_w8, Q.dequantize_w8)
    out = OM.fake_quantize(weights, bits)
    d, heads = cfg.head_dim, cfg.num_attention_heads
    per = heads // tp * d                                  # real input columns per rank
    for name, w in weights.items():
        if not name.endswith("o_proj.weight"):
            continue
        _, scales = quantize(w)                            # stored scales [N, K/128]
        deq = out[name].clone()
        for rank in range(tp):
            for lo in range(rank * per, (rank + 1) * per, 128):       # local K-blocks of this rank, in full columns
                hi = min(lo + 128, (rank + 1) * per)
                g0, g1 = lo // 128, (hi - 1) // 128
                if g0 == g1:
                    continue
                block = torch.zeros(w.size(0), 128, dtype=w.dtype)
                block[:, : hi - lo] = deq[:, lo:hi]
                requantised = dequantize(*quantize(block))[:, : hi - lo]
                mixed = scales[:, g0] != scales[:, g1]
                deq[mixed, lo:hi] = requantised[mixed]
        out[name] = deq
    return out


def test_offline_quantised_checkpoint_on_two_ranks(checkpoints, monkeypatch):
    """Two ranks sharing the GPU load the w4a16 directory: values and scales are sharded by the dense weight loaders,
    rank 1's o_proj shard starts at column 192 of 384, inside a quantisation group."""
    monkeypatch.setenv("NV_DIST_BACKEND", "gloo")
    cfg, weights, dirs = checkpoints
    want_weights = _requantise_straddling(cfg, weights, 4, 2)
    plain = OM.fake_quantize(weights, 4)
    changed = sum(int((want_weights[k] != plain[k]).any(dim=1).sum()) for k in plain if k.endswith("o_proj.weight"))
    assert changed > 0, "the geometry was chosen so that some rows are re-quantised"
    prompts, got, seen = _generate(dirs["w4a16"], tp=2)
    assert seen == "w4a16"
    check_tokens_against_oracle(cfg, want_weights, prompts, got, block_size=BS, num_blocks=16, label="checkpoint w4a16 tp2")
