"""Tensor-level wrappers over the C ABI (include/nv_hip.h).

Each function checks shapes on the host (the kernels trust their arguments),
passes raw device pointers plus the current HIP stream, and raises on error.
Outputs are allocated with torch.empty unless an ``out`` buffer is given, so
every call is capturable in a hipGraph (torch.cuda.graph) once warmed up.
"""

import os

import torch

from byox_vllm_amd import _C
from byox_vllm_amd._C import check, dtype_code, lib, ptr, require_gpu, stream

_workspaces: dict[tuple, torch.Tensor] = {}
_retired: list[torch.Tensor] = []   # outgrown buffers stay alive: captured graphs may still point at them


def workspace(nbytes: int, device, tag: str = "main") -> torch.Tensor:
    """Grow-only scratch buffer per (device, tag).  Size it before graph capture."""
    # one buffer per stream: micro-batches captured on parallel streams must not share scratch space
    key = (device.index if isinstance(device, torch.device) else torch.device(device).index, tag, stream())
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        if ws is not None:
            _retired.append(ws)
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


class Partials:
    """Split-K output of nv_gemm left as fp32 slabs [n, M, N] in the GEMM workspace, plus the bias still to add.
    Valid until the next gemm() that writes slabs; consumed by add_rms_norm / rope_kv_store."""

    def __init__(self, slabs, n, rows, cols, bias, dtype):
        self.slabs, self.n, self.rows, self.cols, self.bias, self.dtype = slabs, n, rows, cols, bias, dtype


class PendingReduce:
    """Output of a row-parallel linear that still has to be summed over the tensor-parallel ranks: ``value`` is this
    rank's contribution (a tensor or the Partials of its split-K GEMM).  Consumed by RMSNorm, which runs the
    all-reduce fused with its residual add and normalisation (layers/comm.py)."""

    def __init__(self, value):
        self.value = value


class result_policy:
    """``with result_policy(1):`` — launches made inside write the results the next launch reads through to the memory
    side (include/nv_hip.h nv_set_result_policy; a captured launch keeps the policy it was captured under).  The
    setting is one word per process, so it is restored on the way out: runners with different policies can share a
    process."""

    def __init__(self, write_through: int):
        self.mask = int(write_through)

    def __enter__(self):
        self.previous = lib().nv_get_result_policy()
        check(lib().nv_set_result_policy(self.mask), "nv_set_result_policy")

    def __exit__(self, *exc):
        check(lib().nv_set_result_policy(self.previous), "nv_set_result_policy")


def _require_device_weight(w, what: str) -> None:
    """A LinearWeight built from host tensors would hand the kernel a host pointer (a GPU memory fault, not an error
    code): refuse it here."""
    if not w.data.is_cuda or (w.scales is not None and not w.scales.is_cuda):
        raise RuntimeError(f"{what}: the weight's data/scales are on {w.data.device}; they must be device memory")


def rms_norm(x, weight, eps, out=None):
    require_gpu(x, "rms_norm")
    hidden = x.size(-1)
    assert x.is_contiguous() and weight.numel() == hidden and weight.dtype == x.dtype
    out = torch.empty_like(x) if out is None else out
    rows = x.numel() // hidden
    check(lib().nv_rmsnorm(ptr(x), ptr(weight), ptr(out), rows, hidden, hidden, hidden, eps,
                           dtype_code(x.dtype), stream()), "nv_rmsnorm")
    return out


def add_rms_norm(x, residual, weight, eps, out=None, residual_out=None):
    if isinstance(x, Partials):
        assert x.bias is None and residual.is_contiguous() and residual.shape == (x.rows, x.cols)
        out = torch.empty_like(residual) if out is None else out
        residual_out = residual if residual_out is None else residual_out
        check(lib().nv_add_rmsnorm_splitk(ptr(x.slabs), x.n, ptr(residual), ptr(weight), ptr(out), ptr(residual_out),
                                          x.rows, x.cols, eps, dtype_code(residual.dtype), stream()),
              "nv_add_rmsnorm_splitk")
        return out, residual_out
    require_gpu(x, "add_rms_norm")
    hidden = x.size(-1)
    assert x.is_contiguous() and residual.is_contiguous() and x.shape == residual.shape
    assert weight.numel() == hidden and weight.dtype == x.dtype == residual.dtype
    out = torch.empty_like(x) if out is None else out
    residual_out = torch.empty_like(x) if residual_out is None else residual_out
    check(lib().nv_add_rmsnorm(ptr(x), ptr(residual), ptr(weight), ptr(out), ptr(residual_out),
                               x.numel() // hidden, hidden, eps, dtype_code(x.dtype), stream()), "nv_add_rmsnorm")
    return out, residual_out


def rope_kv_store(qkv, positions, cos_sin, num_heads, num_kv_heads, head_dim, slot_mapping=None,
                  k_cache=None, v_cache=None, q_norm_w=None, k_norm_w=None, eps=1e-6):
    """Rotate q and k inside ``qkv`` [T, (H+2Hkv)*D] in place; optionally store k/v into the paged cache.
    ``qkv`` may be the QKV GEMM's Partials: the summed, biased qkv row is then materialised first."""
    partials = None
    if isinstance(qkv, Partials):
        partials = qkv
        assert partials.cols == (num_heads + 2 * num_kv_heads) * head_dim
        qkv = torch.empty(partials.rows, partials.cols, dtype=partials.dtype, device=partials.slabs.device)
    require_gpu(qkv, "rope_kv_store")
    T = qkv.size(0)
    assert qkv.dim() == 2 and qkv.stride(1) == 1 and qkv.size(1) >= (num_heads + 2 * num_kv_heads) * head_dim
    assert positions.dtype == torch.int64 and positions.numel() == T and positions.is_contiguous()
    assert cos_sin.dtype == torch.float32 and cos_sin.is_contiguous() and cos_sin.size(-1) == head_dim
    if k_cache is not None:
        assert slot_mapping is not None and slot_mapping.dtype == torch.int32 and slot_mapping.numel() == T
        assert k_cache.is_contiguous() and v_cache.is_contiguous() and k_cache.dtype == qkv.dtype
        assert k_cache.size(-1) == head_dim and k_cache.size(-2) == num_kv_heads
    if partials is not None:
        check(lib().nv_rope_kv_store_splitk(ptr(partials.slabs), partials.n, ptr(partials.bias), ptr(qkv),
                                            qkv.stride(0), ptr(positions), ptr(cos_sin), ptr(slot_mapping),
                                            ptr(k_cache), ptr(v_cache), ptr(q_norm_w), ptr(k_norm_w), eps, T,
                                            num_heads, num_kv_heads, head_dim, dtype_code(qkv.dtype), stream()),
              "nv_rope_kv_store_splitk")
        return qkv
    check(lib().nv_rope_kv_store(ptr(qkv), qkv.stride(0), ptr(positions), ptr(cos_sin), ptr(slot_mapping),
                                 ptr(k_cache), ptr(v_cache), ptr(q_norm_w), ptr(k_norm_w), eps, T,
                                 num_heads, num_kv_heads, head_dim, dtype_code(qkv.dtype), stream()),
          "nv_rope_kv_store")
    return qkv


def gemm_rope_kv_store(x, w, bias, positions, cos_sin, num_heads, num_kv_heads, head_dim, slot_mapping=None,
                       k_cache=None, v_cache=None, q_norm_w=None, k_norm_w=None, eps=1e-6, out=None):
    """qkv projection + bias + [head norms] + RoPE + KV-cache store in one launch -> qkv [T, (H+2Hkv)*D]
    (q and k rotated, v as projected).  ``w``: LinearWeight of the packed q|k|v rows."""
    require_gpu(x, "gemm_rope_kv_store")
    _require_device_weight(w, "gemm_rope_kv_store")
    T, K = x.shape
    N = (num_heads + 2 * num_kv_heads) * head_dim
    assert K == w.in_features and N == w.out_features and x.stride(1) == 1 and x.dtype == w.dtype
    assert positions.dtype == torch.int64 and positions.numel() == T and positions.is_contiguous()
    assert cos_sin.dtype == torch.float32 and cos_sin.is_contiguous() and cos_sin.size(-1) == head_dim
    if bias is not None:
        assert bias.numel() == N and bias.dtype == x.dtype
    if k_cache is not None:
        assert slot_mapping is not None and slot_mapping.dtype == torch.int32 and slot_mapping.numel() == T
        assert k_cache.is_contiguous() and v_cache.is_contiguous() and k_cache.dtype == x.dtype
        assert k_cache.size(-1) == head_dim and k_cache.size(-2) == num_kv_heads
    if out is None:
        out = torch.empty(T, N, dtype=x.dtype, device=x.device)
    check(lib().nv_gemm_rope_kv_store(ptr(x), x.stride(0), ptr(w.data), ptr(w.scales), ptr(bias), ptr(out), out.stride(0),
                                      T, K, w.fmt, ptr(positions), ptr(cos_sin), ptr(slot_mapping), ptr(k_cache),
                                      ptr(v_cache), ptr(q_norm_w), ptr(k_norm_w), eps, num_heads, num_kv_heads,
                                      head_dim, dtype_code(x.dtype), stream()), "nv_gemm_rope_kv_store")
    return out


def kv_store(key, value, k_cache, v_cache, slot_mapping):
    """store_kvcache: key/value [T, Hkv, D] (may be strided views of a qkv buffer)."""
    require_gpu(key, "kv_store")
    T, hkv, d = key.shape
    row = hkv * d
    assert key.stride(-1) == 1 and value.stride(-1) == 1 and key.stride(1) == d and value.stride(1) == d
    assert k_cache.is_contiguous() and v_cache.is_contiguous() and k_cache.size(-1) * k_cache.size(-2) == row
    assert slot_mapping.dtype == torch.int32 and slot_mapping.numel() == T
    check(lib().nv_kv_store(ptr(key), key.stride(0), ptr(value), value.stride(0), ptr(k_cache), ptr(v_cache),
                            ptr(slot_mapping), T, row, dtype_code(key.dtype), stream()), "nv_kv_store")


def pick_decode_splits(batch, num_kv_heads, max_context):
    return lib().nv_paged_attn_decode_pick_splits(batch, num_kv_heads, max_context)


def paged_attn_decode(q, k_cache, v_cache, context_lens, block_tables, scale, out=None, num_splits=1):
    """q [B, H, D] (row-strided ok); caches [NB, bs, Hkv, D]; returns [B, H, D]."""
    require_gpu(q, "paged_attn_decode")
    B, H, D = q.shape
    nb, bs, hkv, d2 = k_cache.shape
    assert d2 == D and q.stride(2) == 1 and q.stride(1) == D
    assert k_cache.is_contiguous() and v_cache.is_contiguous() and v_cache.shape == k_cache.shape
    assert context_lens.dtype == torch.int32 and context_lens.numel() == B and context_lens.is_contiguous()
    assert block_tables.dtype == torch.int32 and block_tables.dim() == 2 and block_tables.size(0) >= B
    assert block_tables.stride(1) == 1
    out = torch.empty(B, H, D, dtype=q.dtype, device=q.device) if out is None else out
    assert out.stride(2) == 1 and out.stride(1) == D
    nbytes = lib().nv_paged_attn_decode_workspace_bytes(B, H, D, num_splits)
    ws = workspace(nbytes, q.device, "attn") if nbytes else None
    check(lib().nv_paged_attn_decode(ptr(q), q.stride(0), ptr(k_cache), ptr(v_cache), ptr(block_tables),
                                     block_tables.stride(0), ptr(context_lens), ptr(out), out.stride(0),
                                     ptr(ws), ws.numel() if ws is not None else 0, B, H, hkv, D, bs, scale,
                                     num_splits, dtype_code(q.dtype), stream()), "nv_paged_attn_decode")
    return out


def paged_attn_decode_static(q, k_cache, v_cache, context_lens, block_tables, scale, out, num_splits=1):
    """paged_attn_decode through the entry point that allocates nothing: ``out`` is required and the split partials
    live in the library's own per-GPU array (nv_paged_attn_decode_static)."""
    require_gpu(q, "paged_attn_decode_static")
    B, H, D = q.shape
    nb, bs, hkv, d2 = k_cache.shape
    assert d2 == D and q.stride(2) == 1 and q.stride(1) == D and out.shape == q.shape
    assert k_cache.is_contiguous() and v_cache.is_contiguous() and v_cache.shape == k_cache.shape
    assert context_lens.dtype == torch.int32 and context_lens.numel() == B and context_lens.is_contiguous()
    assert block_tables.dtype == torch.int32 and block_tables.dim() == 2 and block_tables.size(0) >= B
    assert block_tables.stride(1) == 1 and out.stride(2) == 1 and out.stride(1) == D
    check(lib().nv_paged_attn_decode_static(ptr(q), q.stride(0), ptr(k_cache), ptr(v_cache), ptr(block_tables),
                                            block_tables.stride(0), ptr(context_lens), ptr(out), out.stride(0),
                                            B, H, hkv, D, bs, scale, num_splits, dtype_code(q.dtype), stream()),
          "nv_paged_attn_decode_static")
    return out


def attn_prefill(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, scale, block_tables=None, out=None):
    """Causal varlen attention.  Packed k/v [Tk, Hkv, D] or, with block_tables, paged caches [NB, bs, Hkv, D]."""
    require_gpu(q, "attn_prefill")
    Tq, H, D = q.shape
    assert q.stride(2) == 1 and q.stride(1) == D
    assert cu_seqlens_q.dtype == torch.int32 and cu_seqlens_k.dtype == torch.int32
    n = cu_seqlens_q.numel() - 1
    if block_tables is None:
        hkv = k.size(1)
        assert k.stride(2) == 1 and k.stride(1) == D and v.stride(1) == D and k.stride(0) == v.stride(0)
        kv_stride, bs, bt_stride = k.stride(0), 0, 0
    else:
        assert k.is_contiguous() and v.is_contiguous() and k.dim() == 4
        hkv, bs = k.size(2), k.size(1)
        assert block_tables.dtype == torch.int32 and block_tables.size(0) >= n and block_tables.stride(1) == 1
        kv_stride, bt_stride = hkv * D, block_tables.stride(0)
    out = torch.empty(Tq, H, D, dtype=q.dtype, device=q.device) if out is None else out
    check(lib().nv_attn_prefill(ptr(q), q.stride(0), ptr(k), ptr(v), kv_stride, ptr(cu_seqlens_q), ptr(cu_seqlens_k),
                                ptr(block_tables), bt_stride, ptr(out), out.stride(0), n, max_seqlen_q, H, hkv, D, bs,
                                scale, dtype_code(q.dtype), stream()), "nv_attn_prefill")
    return out


def silu_mul(x, out=None):
    require_gpu(x, "silu_mul")
    assert x.is_contiguous() and x.dim() == 2 and x.size(1) % 2 == 0
    T, two_i = x.shape
    out = torch.empty(T, two_i // 2, dtype=x.dtype, device=x.device) if out is None else out
    check(lib().nv_silu_mul(ptr(x), ptr(out), T, two_i // 2, dtype_code(x.dtype), stream()), "nv_silu_mul")
    return out


def pick_split_k(M, N, K, wfmt=_C.W_INT4):
    """The library's own choice of split-K for this shape (see nv_gemm_pick_split_k)."""
    return lib().nv_gemm_pick_split_k(M, N, K, wfmt)


# Rows from which a product goes through the vendor library (nv_gemm_lt) instead of nv_gemm's decode-shaped tiles:
# prefill steps.  Measured on the layer shapes of Qwen3-0.6B, Qwen2-0.5B and Qwen2-7B, weight expansion included
# (scripts/r03_dense_prefill_gemm.py): at 1024 rows nv_gemm wins everywhere, at 2048 the two trade places by shape,
# from 4096 on the library wins or ties (1.3-1.7 times faster from 8192).  NV_GEMM_LT_ROWS=0 switches the path off.
MANY_ROWS = int(os.environ.get("NV_GEMM_LT_ROWS", "4096"))
LT_WORKSPACE_BYTES = 128 << 20


def expand_weight(w, gate_up_rows=False, out=None):
    """Row-major [N, K] copy, in the model dtype, of a LinearWeight in any format (nv_weight_expand)."""
    _require_device_weight(w, "expand_weight")
    N, K = w.out_features, w.in_features
    out = torch.empty(N, K, dtype=w.dtype, device=w.data.device) if out is None else out
    assert out.is_contiguous() and out.shape == (N, K) and out.dtype == w.dtype
    check(lib().nv_weight_expand(ptr(w.data), ptr(w.scales), w.fmt, N, K, int(gate_up_rows), dtype_code(w.dtype), ptr(out),
                                 stream()), "nv_weight_expand")
    return out


def gemm_many_rows(x, w, bias=None, out=None, silu_mul=False):
    """gemm() for prefill-sized x: the weight written out row-major into scratch, one vendor-library product, and for
    gate/up weights the SiluAndMul pass over its result (same roundings as the fused epilogue: the product rounded to
    the model dtype, silu rounded, the product with up rounded)."""
    M, K = x.shape
    N = w.out_features
    dev = x.device
    if w.fmt == _C.W_DENSE and not silu_mul:
        wr = w.data
    else:
        wr = workspace(N * K * x.element_size(), dev, "gemm_lt_weight")[:N * K * x.element_size()].view(x.dtype).view(N, K)
        expand_weight(w, gate_up_rows=silu_mul, out=wr)
    if silu_mul:
        y = workspace(M * N * x.element_size(), dev, "gemm_lt_gate_up")[:M * N * x.element_size()].view(x.dtype).view(M, N)
    else:
        y = torch.empty(M, N, dtype=x.dtype, device=dev) if out is None else out
        assert y.stride(1) == 1 and y.shape == (M, N)
    ws = workspace(LT_WORKSPACE_BYTES, dev, "gemm_lt")
    check(lib().nv_gemm_lt(ptr(x), x.stride(0), ptr(wr), ptr(bias), ptr(y), y.stride(0), M, N, K, dtype_code(x.dtype),
                           ptr(ws), LT_WORKSPACE_BYTES, stream()), "nv_gemm_lt")
    if not silu_mul:
        return y
    out = torch.empty(M, N // 2, dtype=x.dtype, device=dev) if out is None else out
    assert out.is_contiguous() and out.shape == (M, N // 2)
    check(lib().nv_silu_mul(ptr(y), ptr(out), M, N // 2, dtype_code(x.dtype), stream()), "nv_silu_mul")
    return out


_many_rows_blocked = 0


def block_many_rows(on: bool) -> None:
    """Keep products off the vendor-library route while ``on`` (counted: callers pair True with False).  For jobs whose
    ranks SHARE a GPU (rehearsals of the tensor-parallel runtime): the library's persistent stream-K workgroups and the
    peer-memory all-reduce's workgroups both wait for partners that must be resident at the same time, and with eight
    processes on one device the all-reduce's bounded wait has been seen to run out."""
    global _many_rows_blocked
    _many_rows_blocked = max(0, _many_rows_blocked + (1 if on else -1))


def use_many_rows_path(M, K, N) -> bool:
    return bool(MANY_ROWS) and M >= MANY_ROWS and not _many_rows_blocked and K % 128 == 0 and N % 16 == 0 \
        and lib().nv_gemm_lt_available() == 1 and not torch.cuda.is_current_stream_capturing()


def gemm(x, w, bias=None, out=None, split_k=None, silu_mul=False, partial_out=False):
    """x [M, K] @ w^T where ``w`` is a layers.quant.LinearWeight (dense, int8 or int4).

    silu_mul: ``w`` holds gate/up rows interleaved (LinearWeight.interleave_gate_up); returns silu(gate)*up [M, N/2].
    partial_out: return the split-K slabs as a Partials for a fused consumer instead of reducing them here."""
    require_gpu(x, "gemm")
    _require_device_weight(w, "gemm")
    M, K = x.shape
    assert K == w.in_features and x.stride(1) == 1 and x.dtype == w.dtype
    N = w.out_features
    if bias is not None:
        assert bias.numel() == N and bias.dtype == x.dtype
    if split_k is None:
        if use_many_rows_path(M, K, N) and (out is None or out.is_contiguous() or not silu_mul):
            return gemm_many_rows(x, w, bias, out, silu_mul)
        split_k = 1 if silu_mul else pick_split_k(M, N, K, w.fmt)
    split_k = max(1, min(split_k, K // 128))
    if split_k == 1:
        partial_out = False   # nothing to fuse: the GEMM epilogue writes the finished rows itself
    flags = (_C.GEMM_SILU_MUL if silu_mul else 0) | (_C.GEMM_PARTIAL_OUT if partial_out else 0)
    ws = None
    if split_k > 1 or partial_out:
        ws = workspace(lib().nv_gemm_workspace_bytes(M, N, split_k), x.device, "gemm")
    if partial_out:
        out_ptr, out_stride = None, N
    else:
        out = torch.empty(M, N // 2 if silu_mul else N, dtype=x.dtype, device=x.device) if out is None else out
        assert out.stride(1) == 1 and out.size(1) == (N // 2 if silu_mul else N)
        out_ptr, out_stride = ptr(out), out.stride(0)
    check(lib().nv_gemm(ptr(x), x.stride(0), ptr(w.data), ptr(w.scales), None if partial_out else ptr(bias), out_ptr,
                        out_stride, M, N, K, w.fmt, split_k, ptr(ws), ws.numel() if ws is not None else 0, flags,
                        dtype_code(x.dtype), stream()), "nv_gemm")
    if partial_out:
        return Partials(ws, split_k, M, N, bias, x.dtype)
    return out


def embed(ids, weight, vocab_start=0, out=None):
    require_gpu(weight, "embed")
    assert ids.dtype == torch.int64 and ids.is_contiguous() and weight.is_contiguous()
    T, hidden = ids.numel(), weight.size(1)
    out = torch.empty(T, hidden, dtype=weight.dtype, device=weight.device) if out is None else out
    check(lib().nv_embed(ptr(ids), ptr(weight), ptr(out), T, hidden, vocab_start, weight.size(0),
                         dtype_code(weight.dtype), stream()), "nv_embed")
    return out


def sample(logits, temperatures, noise=None, seed=0, out=None, seed_offset=None):
    require_gpu(logits, "sample")
    B, V = logits.shape
    assert logits.stride(1) == 1 and temperatures.dtype == torch.float32 and temperatures.numel() == B
    if noise is not None:
        assert noise.dtype == torch.float32 and noise.shape == logits.shape and noise.is_contiguous()
    out = torch.empty(B, dtype=torch.int64, device=logits.device) if out is None else out
    ws = workspace(lib().nv_sample_workspace_bytes(B, V), logits.device, "sample")
    check(lib().nv_sample(ptr(logits), logits.stride(0), ptr(temperatures), ptr(noise), ptr(out), ptr(ws), ws.numel(),
                          B, V, seed & 0xFFFFFFFFFFFFFFFF, ptr(seed_offset), dtype_code(logits.dtype), stream()),
          "nv_sample")
    return out


def lm_head_sample(x, w, temperatures, seed=0, out=None, seed_offset=None, logits=None):
    """LM head and sampler chained (embed_head.py:54-66 + sampler.py:11-15 of the reference): x [M, K] against the
    vocabulary matrix ``w`` (an unquantised LinearWeight) -> (tokens [M], logits [M, V]).  The GEMM's epilogue hands
    the sampler its statistics, so the logits are not read again; the tokens are those of ``sample(gemm(x, w), ...)``
    bit for bit, which is also what runs where the GEMM planner would split K (small vocabularies)."""
    require_gpu(x, "lm_head_sample")
    _require_device_weight(w, "lm_head_sample")
    M, K = x.shape
    V = w.out_features
    assert K == w.in_features and x.stride(1) == 1 and x.dtype == w.dtype and w.scales is None
    assert temperatures.dtype == torch.float32 and temperatures.numel() == M
    out = torch.empty(M, dtype=torch.int64, device=x.device) if out is None else out
    logits = torch.empty(M, V, dtype=x.dtype, device=x.device) if logits is None else logits
    assert logits.shape == (M, V) and logits.stride(1) == 1
    if V % 16 or pick_split_k(M, V, K, w.fmt) != 1:
        gemm(x, w, out=logits)
        return sample(logits, temperatures, seed=seed, out=out, seed_offset=seed_offset), logits
    ws = workspace(lib().nv_lm_head_sample_workspace_bytes(M, V), x.device, "sample")
    check(lib().nv_lm_head_sample(ptr(x), x.stride(0), ptr(w.data), w.fmt, ptr(logits), logits.stride(0),
                                  ptr(temperatures), ptr(out), ptr(ws), ws.numel(), M, V, K,
                                  seed & 0xFFFFFFFFFFFFFFFF, ptr(seed_offset), dtype_code(x.dtype), stream()),
          "nv_lm_head_sample")
    return out, logits


def stage_decode_inputs(ids, positions, slots, context_lens, block_tables, g):
    """Copy one step's metadata into the static buffers ``g`` of a captured graph (padding the tail)."""
    bs, padded = ids.numel(), g["padded_bs"]
    cols = block_tables.size(1)
    assert block_tables.is_contiguous() and cols <= g["block_tables"].size(1)
    check(lib().nv_stage_decode_inputs(ptr(ids), ptr(positions), ptr(slots), ptr(context_lens), ptr(block_tables),
                                       cols, ptr(g["input_ids"]), ptr(g["positions"]), ptr(g["slot_mapping"]),
                                       ptr(g["context_lens"]), ptr(g["block_tables"]), g["block_tables"].size(1),
                                       bs, padded, stream()), "nv_stage_decode_inputs")


STAGE_ADVANCE = 0x40000000   # header word 1 of the packed buffer: continue the previous step on the device
STAGE_STEP_BACK = 0x20000000  # header word 1: the previous launch was discarded, take its step count back first


def stage_decode_packed(packed, bs, bt_cols, g, prev_tokens=None, block_size=0):
    """Unpack one step's host buffer (already copied to the device) into the static graph buffers ``g``.
    ``prev_tokens`` (header form only): the launch also honours the STAGE_ADVANCE flag of the header."""
    if prev_tokens is not None:
        assert bs < 0 and prev_tokens.dtype == torch.int64 and block_size > 0
        check(lib().nv_stage_decode_packed_advance(
            ptr(packed), g["padded_bs"], ptr(g["input_ids"]), ptr(g["positions"]), ptr(g["slot_mapping"]),
            ptr(g["context_lens"]), ptr(g["block_tables"]), g["block_tables"].size(1), ptr(g.get("temperatures")),
            ptr(g.get("step")), ptr(prev_tokens), block_size, stream()), "nv_stage_decode_packed_advance")
        return
    check(lib().nv_stage_decode_packed(ptr(packed), bs, g["padded_bs"], bt_cols, ptr(g["input_ids"]),
                                       ptr(g["positions"]), ptr(g["slot_mapping"]), ptr(g["context_lens"]),
                                       ptr(g["block_tables"]), g["block_tables"].size(1), ptr(g.get("temperatures")),
                                       ptr(g.get("step")), stream()), "nv_stage_decode_packed")
