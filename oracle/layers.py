"""Layer arithmetic of the reference, restated on CPU tensors (test oracle)."""

import torch
import torch.nn.functional as F


def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """nanovllm/layers/layernorm.py:16-26: fp32 normalise, downcast, THEN scale."""
    dtype = x.dtype
    xf = x.float()
    var = xf.pow(2).mean(dim=-1, keepdim=True)
    xf = xf * torch.rsqrt(var + eps)
    return xf.to(dtype) * weight


def add_rms_norm(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float):
    """nanovllm/layers/layernorm.py:28-40: the norm sees the unrounded fp32 sum,
    the residual that is carried forward is the rounded one.

    Reference quirk, reproduced: with fp32 inputs `.to(orig_dtype)` at
    layernorm.py:36 is a no-op alias, so the in-place ops at :38-39 also
    overwrite the returned residual, which then equals the normalised output.
    bf16/fp16 (the dtypes the HIP path serves) are unaffected."""
    dtype = x.dtype
    s = x.float() + residual.float()
    new_residual = s.to(dtype)
    var = s.pow(2).mean(dim=-1, keepdim=True)
    s = s * torch.rsqrt(var + eps)
    y = s.to(dtype) * weight
    if dtype == torch.float32:
        new_residual = y
    return y, new_residual


def build_cos_sin_cache(head_dim: int, max_position: int, base: float) -> torch.Tensor:
    """nanovllm/layers/rotary_embedding.py:29-35 -> fp32 [max_position, 1, head_dim] = cat(cos, sin)."""
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2, dtype=torch.float) / head_dim))
    t = torch.arange(max_position, dtype=torch.float)
    freqs = torch.einsum("i,j -> ij", t, inv_freq)
    return torch.cat((freqs.cos(), freqs.sin()), dim=-1).unsqueeze_(1)


def apply_rope(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """nanovllm/layers/rotary_embedding.py:7-15 (NeoX halves, fp32 arithmetic)."""
    x1, x2 = torch.chunk(x.float(), 2, dim=-1)
    y1 = x1 * cos - x2 * sin
    y2 = x2 * cos + x1 * sin
    return torch.cat((y1, y2), dim=-1).to(x.dtype)


def rope_qk(positions, q, k, cos_sin_cache):
    """nanovllm/layers/rotary_embedding.py:37-48; q [T,H,D], k [T,Hkv,D]."""
    cos, sin = cos_sin_cache[positions].chunk(2, dim=-1)
    return apply_rope(q, cos, sin), apply_rope(k, cos, sin)


def silu_and_mul(x: torch.Tensor) -> torch.Tensor:
    """nanovllm/layers/activation.py:10-13 (eager: silu rounds, then the product rounds)."""
    g, u = x.chunk(2, -1)
    return F.silu(g) * u


def linear(x, weight, bias=None):
    """nanovllm/layers/linear.py:49,70,144 — F.linear with W [out, in]."""
    return F.linear(x, weight, bias)


def vocab_embedding(ids, weight, tp_rank=0, tp_size=1):
    """nanovllm/layers/embed_head.py:33-41 without the all-reduce (caller sums ranks)."""
    if tp_size > 1:
        per = weight.size(0)
        lo = per * tp_rank
        mask = (ids >= lo) & (ids < lo + per)
        ids = mask * (ids - lo)
    y = F.embedding(ids, weight)
    if tp_size > 1:
        y = mask.unsqueeze(1) * y
    return y


def lm_head(x, weight, cu_seqlens_q=None):
    """nanovllm/layers/embed_head.py:54-66: last token of each sequence in prefill."""
    if cu_seqlens_q is not None:
        x = x[(cu_seqlens_q[1:] - 1).long()].contiguous()
    return F.linear(x, weight)


def sample(logits, temperatures, noise=None, generator=None):
    """nanovllm/layers/sampler.py:11-15: argmax(softmax(l/T) / Exp(1)).

    ``noise`` injects the Exp(1) draws.  Rows with temperature 0 take the plain
    argmax of the logits (greedy extension; the reference forbids T=0)."""
    lf = logits.float()
    greedy = temperatures == 0
    t = torch.where(greedy, torch.ones_like(temperatures), temperatures)
    probs = torch.softmax(lf / t.unsqueeze(1), dim=-1)
    if noise is None:
        noise = torch.empty_like(probs).exponential_(1, generator=generator)
    sampled = (probs / noise.clamp_min(1e-10)).argmax(dim=-1)
    return torch.where(greedy, lf.argmax(dim=-1), sampled)
