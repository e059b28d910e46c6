"""The oracle's attention restatement against an independent implementation.

flash-attn (the reference's attention backend) is neither vendored nor installed, so its two entry points cannot be
pinned by reference outputs.  PyTorch's own scaled_dot_product_attention on the CPU is an independent statement of
the same mathematics: the oracle must agree with it for GQA, ragged batches, a bottom-right aligned causal mask and
a paged cache.  This does not pin flash-attn's rounding, only the semantics the oracle claims to restate.
"""

import numpy as np
import torch
import torch.nn.functional as F

from oracle import attention as OA


def _sdpa(q, k, v, scale, causal_offset):
    """q [Tq, H, D], k/v [Tk, Hkv, D] in fp32; query i sees keys <= i + causal_offset."""
    H, Hkv = q.size(1), k.size(1)
    kk = k.repeat_interleave(H // Hkv, dim=1).transpose(0, 1)     # [H, Tk, D]
    vv = v.repeat_interleave(H // Hkv, dim=1).transpose(0, 1)
    Tq, Tk = q.size(0), k.size(0)
    mask = torch.arange(Tk)[None, :] <= (torch.arange(Tq)[:, None] + causal_offset)
    return F.scaled_dot_product_attention(q.transpose(0, 1), kk, vv, attn_mask=mask, scale=scale).transpose(0, 1)


def test_varlen_causal_matches_torch_sdpa():
    g = torch.Generator().manual_seed(2)
    H, Hkv, D = 6, 2, 32
    q_lens, k_lens = [5, 1, 17], [5, 9, 40]          # equal, decode-like, prefix-cached
    q = torch.randn(sum(q_lens), H, D, generator=g)
    k = torch.randn(sum(k_lens), Hkv, D, generator=g)
    v = torch.randn(sum(k_lens), Hkv, D, generator=g)
    cu_q = torch.tensor([0] + list(np.cumsum(q_lens)), dtype=torch.int32)
    cu_k = torch.tensor([0] + list(np.cumsum(k_lens)), dtype=torch.int32)
    got = OA.attn_varlen(q, k, v, cu_q, cu_k, D ** -0.5)
    for i, (ql, kl) in enumerate(zip(q_lens, k_lens)):
        qs, ks = int(cu_q[i]), int(cu_k[i])
        want = _sdpa(q[qs:qs + ql], k[ks:ks + kl], v[ks:ks + kl], D ** -0.5, kl - ql)
        assert torch.allclose(got[qs:qs + ql], want, atol=2e-6, rtol=1e-5), i


def test_paged_decode_matches_torch_sdpa():
    g = torch.Generator().manual_seed(3)
    B, H, Hkv, D, bs = 3, 8, 2, 16, 4
    lens = [1, 6, 11]
    nb = 10
    kc = torch.randn(nb, bs, Hkv, D, generator=g)
    vc = torch.randn(nb, bs, Hkv, D, generator=g)
    tables = torch.tensor([[7, -1, -1], [2, 9, -1], [5, 0, 3]], dtype=torch.int32)   # scattered physical blocks
    q = torch.randn(B, H, D, generator=g)
    got = OA.paged_attn_decode(q, kc, vc, torch.tensor(lens, dtype=torch.int32), tables, D ** -0.5)
    for i, n in enumerate(lens):
        rows = [int(tables[i, t // bs]) * bs + t % bs for t in range(n)]
        k = kc.view(-1, Hkv, D)[rows]
        v = vc.view(-1, Hkv, D)[rows]
        want = _sdpa(q[i:i + 1], k, v, D ** -0.5, n - 1)
        assert torch.allclose(got[i:i + 1], want, atol=2e-6, rtol=1e-5), i
