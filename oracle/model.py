"""Whole-model forward of the reference (Qwen2/Qwen3 decoder) on CPU (test oracle).

Graph wiring follows nanovllm/models/qwen3.py:14-208; the step structure
follows nanovllm/engine/model_runner.py:57-65,237-254.  Tensor parallelism is
simulated in one process: every rank's shard is evaluated in turn and the
all-reduce points (linear.py:145-146, embed_head.py:38-40) become an ordered sum.
"""

from types import SimpleNamespace

import torch

from . import attention as A
from . import layers as L
from . import quant as Q


def random_weights(cfg, seed=0, dtype=None):
    """Synthetic checkpoint: N(0, 0.02^2) matrices and biases, unit norm weights."""
    dtype = dtype or cfg.dtype
    g = torch.Generator().manual_seed(seed)
    h, inter = cfg.hidden_size, cfg.intermediate_size
    nh, nkv, d = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    bias = getattr(cfg, "attention_bias", True)

    def mat(*shape):
        return (torch.randn(*shape, generator=g) * 0.02).to(dtype)

    w = {"model.embed_tokens.weight": mat(cfg.vocab_size, h)}
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        w[p + "input_layernorm.weight"] = torch.ones(h, dtype=dtype)
        w[p + "post_attention_layernorm.weight"] = torch.ones(h, dtype=dtype)
        w[p + "self_attn.qkv_proj.weight"] = mat((nh + 2 * nkv) * d, h)
        if bias:
            w[p + "self_attn.qkv_proj.bias"] = mat((nh + 2 * nkv) * d)
        else:
            w[p + "self_attn.q_norm.weight"] = torch.ones(d, dtype=dtype)
            w[p + "self_attn.k_norm.weight"] = torch.ones(d, dtype=dtype)
        w[p + "self_attn.o_proj.weight"] = mat(h, nh * d)
        w[p + "mlp.gate_up_proj.weight"] = mat(2 * inter, h)
        w[p + "mlp.down_proj.weight"] = mat(h, inter)
    w["model.norm.weight"] = torch.ones(h, dtype=dtype)
    if cfg.tie_word_embeddings:
        w["lm_head.weight"] = w["model.embed_tokens.weight"]
    else:
        w["lm_head.weight"] = mat(cfg.vocab_size, h)
    return w


def tiny_config(style, dtype):
    """Small Qwen2-style (QKV bias, D=64, GQA 2) / Qwen3-style (head norms, D=128) shapes for fixtures."""
    qwen3 = style == "qwen3"
    return SimpleNamespace(vocab_size=256, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                           num_attention_heads=2 if qwen3 else 4, num_key_value_heads=1 if qwen3 else 2,
                           head_dim=128 if qwen3 else 64, max_position_embeddings=128, rms_norm_eps=1e-6,
                           hidden_act="silu", attention_bias=not qwen3, rope_theta=1000000.0, rope_scaling=None,
                           tie_word_embeddings=not qwen3, dtype=dtype)


QUANT_SUFFIXES = ("qkv_proj.weight", "o_proj.weight", "gate_up_proj.weight", "down_proj.weight")


def fake_quantize(weights, bits):
    """Replace every layer-linear weight by its quantise->dequantise image."""
    out = dict(weights)
    for name, w in weights.items():
        if name.endswith(QUANT_SUFFIXES):
            if bits == 4:
                out[name] = Q.dequantize_w4(*Q.quantize_w4(w))
            else:
                out[name] = Q.dequantize_w8(*Q.quantize_w8(w))
    return out


# -- tensor-parallel sharding (restates the reference weight loaders) ---------
def shard_rows(w, rank, size):
    """ColumnParallelLinear.weight_loader, linear.py:62-67 (dim 0 narrow)."""
    per = w.size(0) // size
    return w.narrow(0, rank * per, per)


def shard_cols(w, rank, size):
    """RowParallelLinear.weight_loader, linear.py:136-141 (dim 1 narrow)."""
    per = w.size(1) // size
    return w.narrow(1, rank * per, per)


def shard_qkv(w, cfg, rank, size):
    """QKVParallelLinear.weight_loader, linear.py:109-123: per-rank q||k||v."""
    d = cfg.head_dim
    q, k, v = w.split([cfg.num_attention_heads * d, cfg.num_key_value_heads * d,
                       cfg.num_key_value_heads * d], dim=0)
    return torch.cat([t.chunk(size, 0)[rank] for t in (q, k, v)], dim=0)


def shard_merged(w, rank, size):
    """MergedColumnParallelLinear.weight_loader, linear.py:83-89: per-rank gate||up."""
    gate, up = w.chunk(2, 0)
    return torch.cat([gate.chunk(size, 0)[rank], up.chunk(size, 0)[rank]], dim=0)


class _Upcast:
    """Read-only view of a checkpoint dict that converts each tensor to ``dtype`` when it is looked up, so a
    high-precision run never holds a second copy of the whole model."""

    def __init__(self, weights, dtype):
        self.weights, self.dtype = weights, dtype

    def __getitem__(self, name):
        return self.weights[name].to(self.dtype)


class OracleQwen:
    def __init__(self, cfg, weights, block_size, num_blocks, tp_size=1, compute_dtype=None):
        """compute_dtype=torch.float64: the same graph on the same (16-bit-valued) weights with every intermediate
        kept in high precision: the yardstick two 16-bit implementations are measured against.  (The reference's
        own fp32 run is no such yardstick: in fp32 its add_rms_forward aliases the residual, see oracle/layers.py.)"""
        self.cfg, self.w = cfg, weights if compute_dtype is None else _Upcast(weights, compute_dtype)
        self.tp = tp_size
        self.block_size = block_size
        self.eps = cfg.rms_norm_eps
        self.h = cfg.num_attention_heads // tp_size
        self.hkv = cfg.num_key_value_heads // tp_size
        assert cfg.num_attention_heads % tp_size == 0 and cfg.num_key_value_heads % tp_size == 0
        self.d = cfg.head_dim
        self.bias = getattr(cfg, "attention_bias", True)
        dtype = compute_dtype or weights["model.norm.weight"].dtype
        self.cos_sin = L.build_cos_sin_cache(self.d, cfg.max_position_embeddings, cfg.rope_theta)
        # model_runner.py:137-144: [2, L, NB, bs, Hkv/tp, D], one per simulated rank
        self.kv = [torch.zeros(2, cfg.num_hidden_layers, num_blocks, block_size, self.hkv, self.d, dtype=dtype)
                   for _ in range(tp_size)]

    def _allreduce(self, parts):
        y = parts[0]
        for p in parts[1:]:
            y = y + p
        return y

    def _attn(self, li, r, q, k, v, ctx):
        kc, vc = self.kv[r][0, li], self.kv[r][1, li]
        A.store_kvcache(k, v, kc, vc, ctx.slot_mapping)
        scale = self.d ** -0.5
        if ctx.is_prefill:
            if ctx.block_tables is not None:
                return A.attn_varlen(q, kc, vc, ctx.cu_seqlens_q, ctx.cu_seqlens_k, scale, ctx.block_tables)
            return A.attn_varlen(q, k, v, ctx.cu_seqlens_q, ctx.cu_seqlens_k, scale)
        return A.paged_attn_decode(q, kc, vc, ctx.context_lens, ctx.block_tables, scale)

    def forward(self, input_ids, positions, ctx):
        """Qwen3Model.forward, qwen3.py:168-178 -> hidden states [T, h]."""
        cfg, w, tp = self.cfg, self.w, self.tp
        emb = w["model.embed_tokens.weight"]
        hidden = self._allreduce([L.vocab_embedding(input_ids, shard_rows(emb, r, tp), r, tp) for r in range(tp)])
        residual = None
        for li in range(cfg.num_hidden_layers):
            p = f"model.layers.{li}."
            if residual is None:                      # qwen3.py:148-151
                normed = L.rms_norm(hidden, w[p + "input_layernorm.weight"], self.eps)
                # fp32 only: layernorm.py:22-25 works in place on its input, which the caller
                # keeps as the residual, so the reference's residual is the normalised tensor.
                residual = normed if hidden.dtype == torch.float32 else hidden
                hidden = normed
            else:
                hidden, residual = L.add_rms_norm(hidden, residual, w[p + "input_layernorm.weight"], self.eps)
            parts = []
            for r in range(tp):                       # qwen3.py:70-86
                wqkv = shard_qkv(w[p + "self_attn.qkv_proj.weight"], cfg, r, tp)
                bqkv = shard_qkv(w[p + "self_attn.qkv_proj.bias"], cfg, r, tp) if self.bias else None
                qkv = L.linear(hidden, wqkv, bqkv)
                q, k, v = qkv.split([self.h * self.d, self.hkv * self.d, self.hkv * self.d], dim=-1)
                q = q.reshape(-1, self.h, self.d)
                k = k.reshape(-1, self.hkv, self.d)
                v = v.reshape(-1, self.hkv, self.d)
                if not self.bias:
                    q = L.rms_norm(q, w[p + "self_attn.q_norm.weight"], self.eps)
                    k = L.rms_norm(k, w[p + "self_attn.k_norm.weight"], self.eps)
                q, k = L.rope_qk(positions, q, k, self.cos_sin)
                o = self._attn(li, r, q, k, v, ctx)
                parts.append(L.linear(o.flatten(1, -1), shard_cols(w[p + "self_attn.o_proj.weight"], r, tp)))
            hidden = self._allreduce(parts)
            hidden, residual = L.add_rms_norm(hidden, residual, w[p + "post_attention_layernorm.weight"], self.eps)
            parts = []
            for r in range(tp):                       # qwen3.py:110-114
                gu = L.linear(hidden, shard_merged(w[p + "mlp.gate_up_proj.weight"], r, tp))
                parts.append(L.linear(L.silu_and_mul(gu), shard_cols(w[p + "mlp.down_proj.weight"], r, tp)))
            hidden = self._allreduce(parts)
        hidden, _ = L.add_rms_norm(hidden, residual, w["model.norm.weight"], self.eps)
        return hidden

    def compute_logits(self, hidden, ctx):
        """ParallelLMHead.forward, embed_head.py:54-66 (gather + cat over ranks)."""
        cu = ctx.cu_seqlens_q if ctx.is_prefill else None
        head = self.w["lm_head.weight"]
        return torch.cat([L.lm_head(hidden, shard_rows(head, r, self.tp), cu) for r in range(self.tp)], dim=-1)


def make_ctx(**kw):
    base = dict(is_prefill=False, cu_seqlens_q=None, cu_seqlens_k=None, max_seqlen_q=0, max_seqlen_k=0,
                slot_mapping=None, context_lens=None, block_tables=None)
    base.update(kw)
    return SimpleNamespace(**base)


def prefill_ctx(seq_lens, block_tables, block_size, cached=None):
    """Integer metadata of a prefill step (model_runner.py:160-204) from plain lists."""
    cached = cached or [0] * len(seq_lens)
    cu_q, cu_k, slots, pos = [0], [0], [], []
    for n, c, bt in zip(seq_lens, cached, block_tables):
        cu_q.append(cu_q[-1] + n - c)
        cu_k.append(cu_k[-1] + n)
        pos.extend(range(c, n))
        nblk = -(-n // block_size)
        for i in range(c // block_size, nblk):
            start = bt[i] * block_size
            end = start + (block_size if i != nblk - 1 else n - (nblk - 1) * block_size)
            slots.extend(range(start, end))
    bt_t = None
    if cu_k[-1] > cu_q[-1]:
        width = max(len(b) for b in block_tables)
        bt_t = torch.tensor([b + [-1] * (width - len(b)) for b in block_tables], dtype=torch.int32)
    ctx = make_ctx(is_prefill=True, cu_seqlens_q=torch.tensor(cu_q, dtype=torch.int32),
                   cu_seqlens_k=torch.tensor(cu_k, dtype=torch.int32),
                   max_seqlen_q=max(n - c for n, c in zip(seq_lens, cached)), max_seqlen_k=max(seq_lens),
                   slot_mapping=torch.tensor(slots, dtype=torch.int32), block_tables=bt_t)
    return ctx, torch.tensor(pos, dtype=torch.int64)


def decode_ctx(seq_lens, block_tables, block_size):
    """Integer metadata of a decode step (model_runner.py:206-226); seq_lens include the new token."""
    slots = [bt[(n - 1) // block_size] * block_size + (n - 1) % block_size for n, bt in zip(seq_lens, block_tables)]
    width = max(len(b) for b in block_tables)
    ctx = make_ctx(is_prefill=False, slot_mapping=torch.tensor(slots, dtype=torch.int32),
                   context_lens=torch.tensor(seq_lens, dtype=torch.int32),
                   block_tables=torch.tensor([b + [-1] * (width - len(b)) for b in block_tables], dtype=torch.int32))
    return ctx, torch.tensor([n - 1 for n in seq_lens], dtype=torch.int64)
