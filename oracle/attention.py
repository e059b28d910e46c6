"""KV-cache store and the two flash-attn entry points, restated naively (test oracle).

flash-attn itself is a third-party dependency that is not vendored in the
reference (pyproject.toml:18, unpinned) and not installed here.  These
functions follow its documented semantics at the reference's call sites:
scores q.k^T * scale, fp32 softmax and accumulation, output in q's dtype; GQA
maps q head j to kv head j // (H / Hkv); the causal mask is aligned to the
bottom-right corner when seqlen_q < seqlen_k; a paged cache is
[num_blocks, block_size, Hkv, D] and logical token t of sequence i lives in
block block_table[i, t // block_size] at offset t % block_size.
"""

import torch


def store_kvcache(key, value, k_cache, v_cache, slot_mapping):
    """nanovllm/layers/attention.py:10-49.  key/value [T,Hkv,D]; caches [NB,bs,Hkv,D].

    The reference skips only slot == -1; other negative slots would write out
    of bounds there, so every negative slot is skipped here."""
    T, hkv, d = key.shape
    kc = k_cache.view(-1, hkv * d)
    vc = v_cache.view(-1, hkv * d)
    for i in range(T):
        s = int(slot_mapping[i])
        if s < 0:
            continue
        kc[s] = key[i].reshape(-1)
        vc[s] = value[i].reshape(-1)


def _gather_paged(cache, block_table_row, n_tokens):
    bs = cache.size(1)
    nblk = -(-n_tokens // bs)
    blocks = cache[block_table_row[:nblk].long()]  # [nblk, bs, Hkv, D]
    return blocks.reshape(nblk * bs, cache.size(2), cache.size(3))[:n_tokens]


def _attend(q, k, v, scale, causal_offset):
    """q [Sq,H,D], k/v [Sk,Hkv,D] -> [Sq,H,D]; query i sees keys j <= i + causal_offset."""
    sq, h, d = q.shape
    sk, hkv, _ = k.shape
    g = h // hkv
    qf = q.float().permute(1, 0, 2)                      # [H,Sq,D]
    kf = k.float().permute(1, 0, 2).repeat_interleave(g, dim=0)
    vf = v.float().permute(1, 0, 2).repeat_interleave(g, dim=0)
    s = torch.matmul(qf, kf.transpose(1, 2)) * scale     # [H,Sq,Sk]
    i = torch.arange(sq).unsqueeze(1)
    j = torch.arange(sk).unsqueeze(0)
    s = s.masked_fill(j > i + causal_offset, float("-inf"))
    p = torch.softmax(s, dim=-1)
    o = torch.matmul(p, vf)                              # [H,Sq,D]
    return o.permute(1, 0, 2).to(q.dtype)


def paged_attn_decode(q, k_cache, v_cache, context_lens, block_tables, scale):
    """flash_attn_with_kvcache as called at nanovllm/layers/attention.py:88-96.

    q [B,H,D] (the reference passes q.unsqueeze(1)); returns [B,H,D].  Rows with
    context_len 0 (graph padding) return zeros."""
    out = torch.zeros_like(q)
    for b in range(q.size(0)):
        n = int(context_lens[b])
        if n == 0:
            continue
        k = _gather_paged(k_cache, block_tables[b], n)
        v = _gather_paged(v_cache, block_tables[b], n)
        out[b] = _attend(q[b: b + 1], k, v, scale, n - 1)[0]
    return out


def attn_varlen(q, k, v, cu_seqlens_q, cu_seqlens_k, scale, block_tables=None):
    """flash_attn_varlen_func as called at nanovllm/layers/attention.py:75-86.

    Without block_tables k/v are packed [Tk,Hkv,D]; with them k/v are the paged
    caches and sequence i reads cu_seqlens_k[i+1]-cu_seqlens_k[i] tokens through
    block_tables[i].  Causal, bottom-right aligned."""
    out = torch.empty_like(q)
    n = cu_seqlens_q.numel() - 1
    for i in range(n):
        q0, q1 = int(cu_seqlens_q[i]), int(cu_seqlens_q[i + 1])
        k0, k1 = int(cu_seqlens_k[i]), int(cu_seqlens_k[i + 1])
        if block_tables is None:
            ki, vi = k[k0:k1], v[k0:k1]
        else:
            ki = _gather_paged(k, block_tables[i], k1 - k0)
            vi = _gather_paged(v, block_tables[i], k1 - k0)
        out[q0:q1] = _attend(q[q0:q1], ki, vi, scale, (k1 - k0) - (q1 - q0))
    return out


# ---------------------------------------------------------------------------------------------------------------
# How far a correct 16-bit kernel may sit from the functions above.  flash-attn (and the HIP kernels) round the
# softmax numerators P to the model dtype before P.V while the restatement keeps them in fp32, and both round the
# output once.  With u = half a unit in the last place of the model dtype (2^-9 bf16, 2^-12 fp16) that gives
#     |dO| <= u * sum_j p_j |v_j|   (P rounding)  +  2 u |O|   (two output roundings)
# The tests allow three times the first term and one full ulp (4 u |O|) for the second.
UNIT_ROUNDOFF = {torch.bfloat16: 2.0 ** -9, torch.float16: 2.0 ** -12}


def error_bound(out: torch.Tensor, out_abs_v: torch.Tensor) -> torch.Tensor:
    """Elementwise tolerance for an attention output: ``out`` is the oracle's result, ``out_abs_v`` the oracle's
    result with V replaced by |V| (= sum_j p_j |v_j|)."""
    u = UNIT_ROUNDOFF[out.dtype]
    return 3.0 * u * out_abs_v.float() + 4.0 * u * out.float().abs()


def check_close(got: torch.Tensor, want: torch.Tensor, want_abs_v: torch.Tensor, what: str = "attention") -> None:
    """Assert ``got`` (kernel) against ``want`` (oracle): the elementwise bound above, and an rms error below 1 % of
    the output's rms (bf16; 0.15 % fp16) so that a whole missing 32-key chunk of a 1k-3k context cannot hide."""
    got, wf = got.float().cpu(), want.float()
    assert torch.isfinite(got).all(), f"{what}: non-finite output"
    err = (got - wf).abs()
    tol = error_bound(want, want_abs_v)
    bad = err > tol
    assert not bad.any(), (f"{what}: {int(bad.sum())} of {bad.numel()} elements beyond the rounding bound; worst "
                           f"{err.max().item():.3e} vs bound {tol.flatten()[err.argmax()].item():.3e}")
    rms_o = wf.pow(2).mean().sqrt().item()
    if rms_o > 0:
        ratio = err.pow(2).mean().sqrt().item() / rms_o
        assert ratio < (1e-2 if want.dtype == torch.bfloat16 else 1.5e-3), f"{what}: rms error {ratio:.2e} of rms(O)"
