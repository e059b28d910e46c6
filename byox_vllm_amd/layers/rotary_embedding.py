"""NeoX rotary embedding (interface of nanovllm/layers/rotary_embedding.py:18-61).

The fp32 cos/sin table is built exactly as the reference builds it
(rotary_embedding.py:29-35); the rotation itself runs in nv_rope_kv_store.
"""

from functools import lru_cache

import torch
from torch import nn

from byox_vllm_amd import kernels


class RotaryEmbedding(nn.Module):
    def __init__(self, head_size: int, rotary_dim: int, max_position_embeddings: int, base: float) -> None:
        super().__init__()
        assert rotary_dim == head_size
        self.head_size = head_size
        inv_freq = 1.0 / (base ** (torch.arange(0, rotary_dim, 2, dtype=torch.float) / rotary_dim))
        t = torch.arange(max_position_embeddings, dtype=torch.float)
        freqs = torch.einsum("i,j -> ij", t, inv_freq)
        cache = torch.cat((freqs.cos(), freqs.sin()), dim=-1).unsqueeze_(1)
        self.register_buffer("cos_sin_cache", cache, persistent=False)

    def forward(self, positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor):
        """Stand-alone form: rotates contiguous query [T, H, D] and key [T, Hkv, D] in place."""
        d = self.head_size
        for x in (query, key):
            assert x.is_contiguous()
            kernels.rope_kv_store(x.view(x.size(0), -1), positions, self.cos_sin_cache, x.size(1), 0, d)
        return query, key


@lru_cache(1)
def get_rope(head_size: int, rotary_dim: int, max_position: int, base: float, rope_scaling: dict | None = None):
    assert rope_scaling is None
    return RotaryEmbedding(head_size, rotary_dim, max_position, base)
