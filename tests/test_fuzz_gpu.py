"""Seeded random shapes through the attention kernels against the oracle: ragged lengths around the 16/32/128-token
tile edges, every head grouping the kernels special-case, both head widths, both dtypes, paged and packed keys."""

import numpy as np
import pytest
import torch

from oracle import attention as OA

pytestmark = pytest.mark.gpu
DT = {"bf16": torch.bfloat16, "fp16": torch.float16}
HEADS = [(14, 2), (28, 4), (16, 8), (4, 4), (8, 1), (6, 2), (16, 1)]   # groups of 7, 7, 2, 1, 8, 3, 16
EDGES = [1, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 511, 513]


def dev(t):
    return t.cuda()


def _len(rng):
    return int(rng.choice(EDGES)) if rng.rand() < 0.7 else int(rng.randint(1, 700))


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_prefill_attention(seed):
    from byox_vllm_amd import kernels
    rng = np.random.RandomState(100 + seed)
    name = "bf16" if seed % 3 else "fp16"
    dt = DT[name]
    H, Hkv = HEADS[seed % len(HEADS)]
    D = 64 if rng.rand() < 0.5 else 128
    n = int(rng.randint(1, 5))
    g = torch.Generator().manual_seed(seed)
    paged = seed % 2 == 1
    if not paged:
        lens = [_len(rng) for _ in range(n)]
        T = sum(lens)
        q = torch.randn(T, H, D, generator=g).to(dt)
        k = torch.randn(T, Hkv, D, generator=g).to(dt)
        v = torch.randn(T, Hkv, D, generator=g).to(dt)
        cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
        want = OA.attn_varlen(q, k, v, cu, cu, D ** -0.5)
        want_abs = OA.attn_varlen(q, k, v.abs(), cu, cu, D ** -0.5)
        got = kernels.attn_prefill(dev(q), dev(k), dev(v), dev(cu), dev(cu), max(lens), D ** -0.5)
    else:   # prefix-cache prefill: the last q_len tokens of each sequence attend to the whole paged sequence
        bs = int(rng.choice([16, 32, 256]))
        k_lens = [_len(rng) + 1 for _ in range(n)]
        q_lens = [int(rng.randint(1, kl + 1)) for kl in k_lens]
        nblk = sum(-(-kl // bs) for kl in k_lens) + 2
        kc = torch.randn(nblk, bs, Hkv, D, generator=g).to(dt)
        vc = torch.randn(nblk, bs, Hkv, D, generator=g).to(dt)
        kc[-1] = float("nan")                       # an unused block: nothing may read it into a result
        vc[-1] = float("nan")
        perm = iter(torch.randperm(nblk - 1, generator=g).tolist())
        width = max(-(-kl // bs) for kl in k_lens) + 1
        tables = torch.full((n, width), -1, dtype=torch.int32)
        for i, kl in enumerate(k_lens):
            for j in range(-(-kl // bs)):
                tables[i, j] = next(perm)
        q = torch.randn(sum(q_lens), H, D, generator=g).to(dt)
        cu_q = torch.tensor([0] + list(np.cumsum(q_lens)), dtype=torch.int32)
        cu_k = torch.tensor([0] + list(np.cumsum(k_lens)), dtype=torch.int32)
        vc_abs = vc.abs()
        vc_abs[-1] = float("nan")
        want = OA.attn_varlen(q, kc, vc, cu_q, cu_k, D ** -0.5, tables)
        want_abs = OA.attn_varlen(q, kc, vc_abs, cu_q, cu_k, D ** -0.5, tables)
        got = kernels.attn_prefill(dev(q), dev(kc), dev(vc), dev(cu_q), dev(cu_k), max(q_lens), D ** -0.5,
                                   block_tables=dev(tables))
    OA.check_close(got, want, want_abs, f"fuzz prefill seed {seed}")


def _fuzz_decode_case(seed, D=None, max_batch=9):
    rng = np.random.RandomState(500 + seed)
    name = "bf16" if seed % 2 else "fp16"
    dt = DT[name]
    H, Hkv = HEADS[seed % len(HEADS)]
    if D is None:
        D = 64 if seed % 4 < 2 else 128
    bs = int(rng.choice([16, 48, 256]))
    B = int(rng.randint(1, max_batch))
    lens = [0 if rng.rand() < 0.1 else _len(rng) for _ in range(B)]
    g = torch.Generator().manual_seed(seed)
    nblk = sum(-(-n // bs) for n in lens) + 2
    kc = torch.randn(nblk, bs, Hkv, D, generator=g).to(dt)
    vc = torch.randn(nblk, bs, Hkv, D, generator=g).to(dt)
    perm = iter(torch.randperm(nblk, generator=g).tolist())
    width = max(1, max(-(-n // bs) for n in lens)) + int(rng.randint(0, 3))
    tables = torch.full((B, width), -1, dtype=torch.int32)
    for i, n in enumerate(lens):
        for j in range(-(-n // bs)):
            tables[i, j] = next(perm)
        if n % bs:                                   # the tail of the last block holds junk from an earlier owner
            blk, off = int(tables[i, (n - 1) // bs]), n % bs
            kc[blk, off:] = float("nan")
            vc[blk, off:] = float("nan")
    q = torch.randn(B, H, D, generator=g).to(dt)
    ctx = torch.tensor(lens, dtype=torch.int32)
    want = OA.paged_attn_decode(q, kc, vc, ctx, tables, D ** -0.5)
    want_abs = OA.paged_attn_decode(q, kc, vc.abs(), ctx, tables, D ** -0.5)
    return rng, D, lens, q, kc, vc, ctx, tables, want, want_abs


@pytest.mark.parametrize("seed", range(10))
def test_fuzz_paged_decode_attention(seed):
    from byox_vllm_amd import kernels
    rng, D, lens, q, kc, vc, ctx, tables, want, want_abs = _fuzz_decode_case(seed)
    for splits in (1, int(rng.randint(2, 6))):
        got = kernels.paged_attn_decode(dev(q), dev(kc), dev(vc), dev(ctx), dev(tables), D ** -0.5, num_splits=splits)
        OA.check_close(got, want, want_abs, f"fuzz decode seed {seed} splits {splits} lens {lens}")


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_paged_decode_adaptive_wavefronts(seed, monkeypatch):
    """The adaptive form of the decode kernel (128-wide heads; a group keeps one or both wavefronts by its length
    against the launch's average) on ragged batches with empty rows and junk behind the contexts."""
    from byox_vllm_amd import kernels
    monkeypatch.setenv("NV_ATTN_WAVES", "0")
    rng, D, lens, q, kc, vc, ctx, tables, want, want_abs = _fuzz_decode_case(seed, D=128, max_batch=17)
    got = kernels.paged_attn_decode(dev(q), dev(kc), dev(vc), dev(ctx), dev(tables), D ** -0.5, num_splits=1)
    OA.check_close(got, want, want_abs, f"fuzz decode, adaptive, seed {seed} lens {lens}")


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_gemm(seed):
    """Random (M, N, K), weight format, split-K factor and epilogue against an fp64 F.linear on the dequantised weight."""
    from byox_vllm_amd import kernels
    from byox_vllm_amd.layers.quant import LinearWeight, interleave_gate_up
    from oracle import layers as L
    from oracle import quant as Q
    rng = np.random.RandomState(900 + seed)
    name = "bf16" if seed % 4 else "fp16"
    dt = DT[name]
    M = int(rng.choice([1, 2, 7, 16, 17, 33, 64, 100, 128, 129, 200, 515]))
    N = 16 * int(rng.randint(1, 40))
    K = 128 * int(rng.randint(1, 9))
    fmt = ["dense", "tiled", "w8a16", "w4a16"][seed % 4]
    silu = seed % 5 == 0 and N % 32 == 0
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, K, generator=g).to(dt)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dt)
    if fmt == "w4a16":
        w_ref = Q.dequantize_w4(*Q.quantize_w4(w))
    elif fmt == "w8a16":
        w_ref = Q.dequantize_w8(*Q.quantize_w8(w))
    else:
        w_ref = w
    wd = interleave_gate_up(dev(w)) if silu else dev(w)
    lw = LinearWeight.dense_tiled(wd) if fmt == "tiled" else LinearWeight.make(wd, None if fmt == "dense" else fmt)
    eps = 2.0 ** -8 if name == "bf16" else 2.0 ** -11
    if silu:
        want = L.silu_and_mul(L.linear(x, w_ref)).float()
        got = kernels.gemm(dev(x), lw, silu_mul=True).float().cpu()
        ulps = 2.0
    else:
        bias = (torch.randn(N, generator=g) * 0.1).to(dt)
        want = torch.nn.functional.linear(x.double(), w_ref.double(), bias.double()).to(dt).float()
        got = kernels.gemm(dev(x), lw, dev(bias), split_k=int(rng.randint(1, 6))).float().cpu()
        ulps = 1.0
    assert got.shape == want.shape
    bad = (got - want).abs() > ulps * eps * want.abs().clamp_min(1e-3) * 2   # same bar as the kernel tests' ulp_close
    assert not bad.any(), (M, N, K, fmt, silu, int(bad.sum()), (got - want).abs().max().item())


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_gemm_wide_int4_kernel(seed, monkeypatch):
    """The 32x32x16 int4 kernel forced on random shapes it admits (more than 64 rows, N a multiple of 32): ragged rows,
    ragged last column group, K splits down to one K-block per workgroup, every epilogue, both dtypes."""
    from byox_vllm_amd import kernels
    from byox_vllm_amd.layers.quant import LinearWeight, interleave_gate_up
    from oracle import layers as L
    from oracle import quant as Q
    monkeypatch.setenv("NV_GEMM_WIDE", "1")
    rng = np.random.RandomState(1700 + seed)
    name = "bf16" if seed % 3 else "fp16"
    dt = DT[name]
    M = int(rng.choice([65, 100, 128, 129, 200, 300, 515]))
    N = 32 * int(rng.randint(1, 24))
    K = 128 * int(rng.randint(1, 9))
    silu = seed % 4 == 0 and N % 64 == 0
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, K, generator=g).to(dt)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dt)
    w_ref = Q.dequantize_w4(*Q.quantize_w4(w))
    lw = LinearWeight.make(interleave_gate_up(dev(w)) if silu else dev(w), "w4a16")
    eps = 2.0 ** -8 if name == "bf16" else 2.0 ** -11
    if silu:
        want = L.silu_and_mul(L.linear(x, w_ref)).float()
        got = kernels.gemm(dev(x), lw, silu_mul=True).float().cpu()
        ulps = 2.0
    else:
        bias = (torch.randn(N, generator=g) * 0.1).to(dt)
        want = torch.nn.functional.linear(x.double(), w_ref.double(), bias.double()).to(dt).float()
        got = kernels.gemm(dev(x), lw, dev(bias), split_k=int(rng.randint(1, 9))).float().cpu()
        ulps = 1.0
    assert got.shape == want.shape
    bad = (got - want).abs() > ulps * eps * want.abs().clamp_min(1e-3) * 2
    assert not bad.any(), (M, N, K, silu, int(bad.sum()), (got - want).abs().max().item())
