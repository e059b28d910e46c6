"""Generate golden vectors by running the reference (nanovllm) on CPU.

Run ONLY in the build container, where /root/reference exists:
    TORCHDYNAMO_DISABLE=1 python tests/golden/make_goldens.py
The outputs (tests/golden/*.npz, *.json) are committed; the reference itself
never travels.  What had to be stubbed to import it here:
* ``flash_attn`` is not installed -> a stub module is pre-seeded; the two entry
  points and the Triton ``store_kvcache`` (GPU only) are replaced by the naive
  restatements in oracle/attention.py, so the whole-model goldens pin the
  reference's wiring and layer arithmetic, not flash-attn's internals.
* ``@torch.compile`` sites run eager (TORCHDYNAMO_DISABLE=1).
* a world-size-1 gloo process group stands in for NCCL.
"""

import json
import os
import sys
import tempfile
import types
from types import SimpleNamespace

os.environ["TORCHDYNAMO_DISABLE"] = "1"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

stub = types.ModuleType("flash_attn")
stub.flash_attn_varlen_func = None
stub.flash_attn_with_kvcache = None
sys.modules["flash_attn"] = stub

import nanovllm.layers.attention as ref_attn  # noqa: E402
from nanovllm.engine.block_manager import BlockManager  # noqa: E402
from nanovllm.layers.activation import SiluAndMul  # noqa: E402
from nanovllm.layers.layernorm import RMSNorm  # noqa: E402
from nanovllm.layers.rotary_embedding import RotaryEmbedding  # noqa: E402
from nanovllm.layers.sampler import Sampler  # noqa: E402
from nanovllm.utils.context import reset_context, set_context  # noqa: E402

from oracle import attention as OA  # noqa: E402
from oracle import model as OM  # noqa: E402

DT = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}


def f32(t):
    return t.detach().float().numpy()


def gen_hash():
    cases = []
    rng = np.random.RandomState(0)
    for n, prefix in [(16, -1), (16, 8040990501003754519), (256, -1), (256, 123456789), (3, -1)]:
        toks = [int(x) for x in rng.randint(0, 151936, size=n)]
        cases.append({"tokens": toks, "prefix": prefix, "hash": BlockManager._compute_hash(toks, prefix)})
    cases.append({"tokens": [0] * 16, "prefix": -1, "hash": BlockManager._compute_hash([0] * 16, -1)})
    with open(os.path.join(HERE, "block_hash.json"), "w") as f:
        json.dump(cases, f)


def gen_layers():
    out = {}
    g = torch.Generator().manual_seed(1)
    for name, dt in DT.items():
        for h in (896, 128, 64):
            x = (torch.randn(5, h, generator=g) * 1.5).to(dt)
            r = torch.randn(5, h, generator=g).to(dt)
            wt = (1.0 + 0.1 * torch.randn(h, generator=g)).to(dt)
            norm = RMSNorm(h, eps=1e-6)
            norm.weight.data = wt.clone()
            y = norm(x.clone())
            y2, r2 = norm(x.clone(), r.clone())
            out[f"rms_{name}_{h}_x"], out[f"rms_{name}_{h}_r"], out[f"rms_{name}_{h}_w"] = f32(x), f32(r), f32(wt)
            out[f"rms_{name}_{h}_y"], out[f"rms_{name}_{h}_y2"], out[f"rms_{name}_{h}_r2"] = f32(y), f32(y2), f32(r2)
        for d in (64, 128):
            rope = RotaryEmbedding(d, d, 512, 1000000.0)
            pos = torch.tensor([0, 1, 7, 300, 511])
            q = torch.randn(5, 3, d, generator=g).to(dt)
            k = torch.randn(5, 2, d, generator=g).to(dt)
            q2, k2 = rope(pos, q.clone(), k.clone())
            out[f"rope_{name}_{d}_pos"] = pos.numpy()
            out[f"rope_{name}_{d}_q"], out[f"rope_{name}_{d}_k"] = f32(q), f32(k)
            out[f"rope_{name}_{d}_q2"], out[f"rope_{name}_{d}_k2"] = f32(q2), f32(k2)
            if name == "fp32":
                out[f"rope_table_{d}"] = f32(rope.cos_sin_cache)
        x = (torch.randn(4, 2 * 96, generator=g) * 2).to(dt)
        out[f"silu_{name}_x"], out[f"silu_{name}_y"] = f32(x), f32(SiluAndMul()(x.clone()))
    np.savez_compressed(os.path.join(HERE, "layers.npz"), **out)


def gen_sampler():
    out = {}
    g = torch.Generator().manual_seed(2)
    logits = (torch.randn(6, 1000, generator=g) * 3).to(torch.bfloat16)
    temps = torch.tensor([1e-5, 1e-5, 0.6, 1.0, 0.6, 1.3])
    torch.manual_seed(1234)
    toks = Sampler()(logits.clone(), temps)
    torch.manual_seed(1234)
    noise = torch.empty(6, 1000).exponential_(1)
    out["logits"], out["temps"], out["noise"], out["tokens"] = f32(logits), temps.numpy(), noise.numpy(), toks.numpy()
    np.savez_compressed(os.path.join(HERE, "sampler.npz"), **out)


def gen_sharding():
    """Outputs of the reference weight loaders for tp in {1,2,4} on index-valued weights."""
    import nanovllm.layers.embed_head as EH
    import nanovllm.layers.linear as LIN
    out = {}
    h, nh, nkv, d, inter, vocab = 32, 8, 4, 8, 48, 64
    wq = torch.arange(nh * d * h, dtype=torch.float32).view(nh * d, h)
    wk = 10000 + torch.arange(nkv * d * h, dtype=torch.float32).view(nkv * d, h)
    wv = 20000 + torch.arange(nkv * d * h, dtype=torch.float32).view(nkv * d, h)
    bq = torch.arange(nh * d, dtype=torch.float32)
    wgate = torch.arange(inter * h, dtype=torch.float32).view(inter, h)
    wup = 50000 + wgate
    wo = torch.arange(h * nh * d, dtype=torch.float32).view(h, nh * d)
    emb = torch.arange(vocab * h, dtype=torch.float32).view(vocab, h)
    out.update(wq=wq.numpy(), wk=wk.numpy(), wv=wv.numpy(), bq=bq.numpy(), wgate=wgate.numpy(), wup=wup.numpy(),
               wo=wo.numpy(), emb=emb.numpy())
    real_rank, real_ws = dist.get_rank, dist.get_world_size
    for tp in (1, 2, 4):
        for rank in range(tp):
            dist.get_rank = lambda *a, rank=rank: rank
            dist.get_world_size = lambda *a, tp=tp: tp
            qkv = LIN.QKVParallelLinear(h, d, nh, nkv, bias=True)
            for t, sid in ((wq, "q"), (wk, "k"), (wv, "v")):
                qkv.weight_loader(qkv.weight, t, sid)
            qkv.weight_loader(qkv.bias, bq, "q")
            mg = LIN.MergedColumnParallelLinear(h, [inter, inter])
            mg.weight_loader(mg.weight, wgate, 0)
            mg.weight_loader(mg.weight, wup, 1)
            row = LIN.RowParallelLinear(nh * d, h)
            row.weight_loader(row.weight, wo)
            ve = EH.VocabParallelEmbedding(vocab, h)
            ve.weight_loader(ve.weight, emb)
            key = f"tp{tp}_r{rank}_"
            out[key + "qkv"] = f32(qkv.weight.data)
            out[key + "qkv_bias_q"] = f32(qkv.bias.data[: nh * d // tp])
            out[key + "merged"] = f32(mg.weight.data)
            out[key + "row"] = f32(row.weight.data)
            out[key + "emb"] = f32(ve.weight.data)
    dist.get_rank, dist.get_world_size = real_rank, real_ws
    np.savez_compressed(os.path.join(HERE, "sharding.npz"), **out)


tiny_cfg = OM.tiny_config


def gen_model():
    """Reference Qwen3ForCausalLM (CPU, eager) through a paged cache: prefill, prefix-cache prefill, decode."""
    from nanovllm.layers.rotary_embedding import get_rope
    from nanovllm.models.qwen3 import Qwen3ForCausalLM

    ref_attn.store_kvcache = OA.store_kvcache
    ref_attn.flash_attn_varlen_func = (
        lambda q, k, v, max_seqlen_q, cu_seqlens_q, max_seqlen_k, cu_seqlens_k, softmax_scale, causal, block_table:
        OA.attn_varlen(q, k, v, cu_seqlens_q, cu_seqlens_k, softmax_scale, block_table))
    ref_attn.flash_attn_with_kvcache = (
        lambda q, kc, vc, cache_seqlens, block_table, softmax_scale, causal:
        OA.paged_attn_decode(q.squeeze(1), kc, vc, cache_seqlens, block_table, softmax_scale).unsqueeze(1))

    out = {}
    bs, nb = 16, 12
    for style in ("qwen2", "qwen3"):
        for dname in ("fp32", "bf16", "fp16"):
            dt = DT[dname]
            cfg = tiny_cfg(style, dt)
            weights = OM.random_weights(cfg, seed=7)
            get_rope.cache_clear()
            torch.set_default_dtype(dt)
            model = Qwen3ForCausalLM(cfg)
            torch.set_default_dtype(torch.float32)
            params = dict(model.named_parameters())
            for name, w in weights.items():
                if name == "lm_head.weight" and cfg.tie_word_embeddings:
                    continue
                params[name].data.copy_(w)
            kv = torch.zeros(2, cfg.num_hidden_layers, nb, bs, cfg.num_key_value_heads, cfg.head_dim, dtype=dt)
            li = 0
            for m in model.modules():
                if hasattr(m, "k_cache") and hasattr(m, "v_cache"):
                    m.k_cache, m.v_cache = kv[0, li], kv[1, li]
                    li += 1
            key = f"{style}_{dname}_"
            for name, w in weights.items():
                out[key + "w:" + name] = f32(w)

            g = torch.Generator().manual_seed(11)
            prompts = [torch.randint(0, 256, (n,), generator=g).tolist() for n in (5, 19)]
            tables = [[3], [7, 1]]
            lens = [5, 19]
            # step 0: prefill both
            ctx, pos = OM.prefill_ctx(lens, tables, bs)
            ids = torch.tensor(prompts[0] + prompts[1])
            set_context(True, ctx.cu_seqlens_q, ctx.cu_seqlens_k, ctx.max_seqlen_q, ctx.max_seqlen_k,
                        ctx.slot_mapping, None, None)
            with torch.inference_mode():
                logits = model.compute_logits(model(ids, pos))
            reset_context()
            out[key + "prompt0"], out[key + "prompt1"] = np.array(prompts[0]), np.array(prompts[1])
            out[key + "logits_prefill"] = f32(logits)
            toks = logits.float().argmax(-1).tolist()
            # steps 1..3: decode (seq 1 crosses into a new block: 19 -> 20..22; give it block 9 when needed)
            tables = [[3], [7, 1]]
            for step in range(3):
                lens = [n + 1 for n in lens]
                for t, n in zip(tables, lens):
                    if len(t) * bs < n:
                        t.append(9)
                ctx, pos = OM.decode_ctx(lens, tables, bs)
                set_context(False, slot_mapping=ctx.slot_mapping, context_lens=ctx.context_lens,
                            block_tables=ctx.block_tables)
                with torch.inference_mode():
                    logits = model.compute_logits(model(torch.tensor(toks), pos))
                reset_context()
                out[key + f"decode{step}_in"] = np.array(toks)
                out[key + f"logits_decode{step}"] = f32(logits)
                toks = logits.float().argmax(-1).tolist()
            # prefix-cache prefill: a third request shares seq 1's first full block (block 7)
            p2 = prompts[1][:16] + torch.randint(0, 256, (6,), generator=g).tolist()
            ctx, pos = OM.prefill_ctx([22], [[7, 5]], bs, cached=[16])
            set_context(True, ctx.cu_seqlens_q, ctx.cu_seqlens_k, ctx.max_seqlen_q, ctx.max_seqlen_k,
                        ctx.slot_mapping, None, ctx.block_tables)
            with torch.inference_mode():
                logits = model.compute_logits(model(torch.tensor(p2[16:]), pos))
            reset_context()
            out[key + "prompt2"] = np.array(p2)
            out[key + "logits_prefix_prefill"] = f32(logits)
    np.savez_compressed(os.path.join(HERE, "model.npz"), **out)


if __name__ == "__main__":
    store = tempfile.mktemp()
    dist.init_process_group("gloo", init_method=f"file://{store}", world_size=1, rank=0)
    gen_hash()
    gen_layers()
    gen_sampler()
    gen_sharding()
    gen_model()
    dist.destroy_process_group()
    print("goldens written to", HERE)
