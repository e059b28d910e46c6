"""Attention over the paged KV cache (interface of nanovllm/layers/attention.py:52-97).

``forward(q, k, v)`` keeps the reference's contract: store k/v at the step's
slots, then causal varlen attention (prefill; through the cache when the step
has a prefix hit) or paged single-token attention (decode).  The model uses
``forward_fused`` instead, where the store already happened inside the RoPE
kernel.
"""

import torch
from torch import nn

from byox_vllm_amd import kernels
from byox_vllm_amd.utils.context import get_context


def store_kvcache(key, value, k_cache, v_cache, slot_mapping):
    kernels.kv_store(key, value, k_cache, v_cache, slot_mapping)


class Attention(nn.Module):
    def __init__(self, num_heads, head_dim, scale, num_kv_heads):
        super().__init__()
        self.num_heads, self.head_dim, self.scale, self.num_kv_heads = num_heads, head_dim, scale, num_kv_heads
        self.k_cache = self.v_cache = torch.tensor([])
        self.decode_splits = 1

    def forward(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor):
        context = get_context()
        if self.k_cache.numel() and self.v_cache.numel():
            store_kvcache(k, v, self.k_cache, self.v_cache, context.slot_mapping)
        return self.forward_fused(q, k, v)

    def forward_fused(self, q, k, v, out=None):
        """Attention only; k/v of this step are already in the cache (if there is one)."""
        context = get_context()
        if context.is_prefill:
            if context.block_tables is not None:  # prefix cache: read keys through the block table
                return kernels.attn_prefill(q, self.k_cache, self.v_cache, context.cu_seqlens_q, context.cu_seqlens_k,
                                            context.max_seqlen_q, self.scale, block_tables=context.block_tables, out=out)
            return kernels.attn_prefill(q, k, v, context.cu_seqlens_q, context.cu_seqlens_k, context.max_seqlen_q,
                                        self.scale, out=out)
        return kernels.paged_attn_decode(q, self.k_cache, self.v_cache, context.context_lens, context.block_tables,
                                         self.scale, out=out, num_splits=self.decode_splits)
