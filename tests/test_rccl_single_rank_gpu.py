"""The tensor-parallel decode path over RCCL on a one-GPU machine.

RCCL wants one device per rank, so more than one rank cannot run here; what can run is the whole code path with a
process group of ONE rank (NV_TP_SINGLE_RANK_RCCL=1): init_process_group("nccl") as the reference does
(nanovllm/engine/model_runner.py:27), every dist.all_reduce of the row-parallel layers and of the embedding
(layers/linear.py:145-146, embed_head.py:38-40) issued and captured into the decode graphs, dist.gather of the logits
(embed_head.py:62-65) and the sampler outside the graph.  Sums over one rank change nothing, so the tokens must be those
of the plain single-GPU runner.  Runs in a child process: a process group is process-wide state."""

import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, random, sys, torch
import torch.distributed as dist
from byox_vllm_amd.config import Config, qwen2_config
from byox_vllm_amd.engine.model_runner import ModelRunner
from byox_vllm_amd.engine.sequence import Sequence
from byox_vllm_amd.sampling_params import SamplingParams
from byox_vllm_amd.layers.rotary_embedding import get_rope

BLOCK, BATCH, CTX = 256, 16, 700
hf = qwen2_config("qwen2-0.5b", dtype=torch.bfloat16, num_hidden_layers=3, vocab_size=8192)

def seqs():
    rng = random.Random(0)
    out = []
    per = CTX // BLOCK + 1
    for i in range(BATCH):
        s = Sequence([rng.randrange(8000) for _ in range(CTX - 31 * (i % 5))], BLOCK,
                     SamplingParams(temperature=0.0, max_tokens=1 << 20, ignore_eos=True))
        s.block_table = list(range(i * per, i * per + s.num_blocks))
        out.append(s)
    return out

def run(single, eager):
    os.environ["NV_TP_SINGLE_RANK_RCCL"] = "1" if single else "0"
    get_rope.cache_clear()
    cfg = Config("synthetic", hf_config=hf, quantization="w4a16", kvcache_block_size=BLOCK,
                 num_kvcache_blocks=BATCH * (CTX // BLOCK + 1) + 1, max_model_len=1024, max_num_seqs=BATCH,
                 max_num_batched_tokens=4096, enforce_eager=eager, graph_batch_sizes=[BATCH], seed=3)
    r = ModelRunner(cfg, 0, None)
    g = torch.Generator(device="cuda").manual_seed(5)
    r.kv_cache.copy_(torch.randn(r.kv_cache.shape, generator=g, device="cuda", dtype=torch.bfloat16))
    replays = []
    if not eager:
        real = r._replay
        r._replay = lambda bs: (replays.append(bs), real(bs))[1]
    state = dict(single=r.single_rank_rccl, head=getattr(r, "graph_has_head", None),
                 backend=dist.get_backend() if dist.is_initialized() else None)
    toks = [r.run(seqs(), False), r.run(seqs()[: BATCH - 3], False)]
    r.exit()
    return toks, replays, state

plain, _, st0 = run(False, False)
assert st0["backend"] is None and st0["head"] is True, st0
rccl_graph, replays, st1 = run(True, False)
assert st1 == dict(single=True, head=False, backend="nccl"), st1
assert replays == [BATCH, BATCH], replays
assert not dist.is_initialized(), "the runner must tear down the process group it created"
rccl_eager, _, st2 = run(True, True)
assert st2["backend"] == "nccl"
assert rccl_graph == plain, "captured RCCL path differs from the single-GPU graph"
assert rccl_eager[0] == plain[0], "eager RCCL path differs"
assert len(set(plain[0])) > 1
print("SINGLE-RANK-RCCL-OK")
"""


def test_tensor_parallel_path_over_rccl_with_one_rank():
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-u", "-c", CHILD], env=env, cwd=ROOT, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0 and "SINGLE-RANK-RCCL-OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])
