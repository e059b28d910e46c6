"""HIP kernels through the C ABI against the CPU oracle and the reference-generated goldens."""

import os

import numpy as np
import pytest
import torch

from oracle import attention as OA
from oracle import layers as L
from oracle import quant as Q

pytestmark = pytest.mark.gpu
DT = {"bf16": torch.bfloat16, "fp16": torch.float16}
EPS = {"bf16": 2.0 ** -8, "fp16": 2.0 ** -11}   # one unit in the last place at magnitude 1


def dev(t):
    return t.to("cuda")


def t(a, dt):
    return torch.from_numpy(np.asarray(a)).to(dt)


def ulp_close(got, want, name, ulps=1.0):
    """|got - want| <= ulps * eps * max(|want|, tiny) elementwise (fp values that differ by rounding only)."""
    got, want = got.float().cpu(), want.float().cpu()
    tol = ulps * EPS[name] * want.abs().clamp_min(1e-3) * 2
    bad = (got - want).abs() > tol
    assert not bad.any(), f"{int(bad.sum())} of {bad.numel()} beyond {ulps} ulp; max abs {(got - want).abs().max()}"


@pytest.mark.parametrize("name", list(DT))
@pytest.mark.parametrize("h", [896, 128, 64])
def test_rmsnorm_vs_reference_golden(golden, name, h):
    from byox_vllm_amd import kernels
    g, dt = golden.npz("layers"), DT[name]
    k = f"rms_{name}_{h}_"
    x, r, w = t(g[k + "x"], dt), t(g[k + "r"], dt), t(g[k + "w"], dt)
    y = kernels.rms_norm(dev(x), dev(w), 1e-6)
    ulp_close(y, t(g[k + "y"], dt), name)
    y2, r2 = kernels.add_rms_norm(dev(x), dev(r), dev(w), 1e-6)
    ulp_close(y2, t(g[k + "y2"], dt), name)
    assert torch.equal(r2.cpu(), t(g[k + "r2"], dt)), "residual (a single fp32 add, then one rounding) must be bit-exact"
    # the sum of squares is the only place where summation order can differ: count exact matches too
    exact = (y.cpu() == t(g[k + "y"], dt)).float().mean().item()
    assert exact > 0.98


def test_rmsnorm_inplace_large_rows():
    from byox_vllm_amd import kernels
    torch.manual_seed(0)
    x = torch.randn(130, 3584).to(torch.bfloat16)
    r = torch.randn(130, 3584).to(torch.bfloat16)
    w = (1 + 0.1 * torch.randn(3584)).to(torch.bfloat16)
    want_y, want_r = L.add_rms_norm(x, r, w, 1e-6)
    xd, rd = dev(x), dev(r)
    kernels.add_rms_norm(xd, rd, dev(w), 1e-6, out=xd, residual_out=rd)
    assert torch.equal(rd.cpu(), want_r)
    ulp_close(xd, want_y, "bf16")


@pytest.mark.parametrize("name", list(DT))
@pytest.mark.parametrize("d", [64, 128])
def test_rope_bit_exact_vs_reference_golden(golden, name, d):
    from byox_vllm_amd.layers.rotary_embedding import RotaryEmbedding
    g, dt = golden.npz("layers"), DT[name]
    k = f"rope_{name}_{d}_"
    rope = RotaryEmbedding(d, d, 512, 1000000.0)
    assert torch.equal(rope.cos_sin_cache, torch.from_numpy(g[f"rope_table_{d}"]))
    rope = rope.to("cuda")
    q, kk = dev(t(g[k + "q"], dt)), dev(t(g[k + "k"], dt))
    rope(dev(torch.from_numpy(g[k + "pos"])), q, kk)
    assert torch.equal(q.cpu(), t(g[k + "q2"], dt))
    assert torch.equal(kk.cpu(), t(g[k + "k2"], dt))


@pytest.mark.parametrize("name", list(DT))
def test_silu_mul_vs_reference_golden(golden, name):
    from byox_vllm_amd import kernels
    g, dt = golden.npz("layers"), DT[name]
    y = kernels.silu_mul(dev(t(g[f"silu_{name}_x"], dt)))
    ulp_close(y, t(g[f"silu_{name}_y"], dt), name)   # device expf vs libm expf: at most the last bit


@pytest.mark.parametrize("qk_norm", [False, True])
@pytest.mark.parametrize("d", [64, 128])
def test_fused_rope_norm_kv_store(qk_norm, d):
    """q/k head-norm + RoPE in place + paged store == oracle rms_norm -> rope_qk -> store_kvcache."""
    from byox_vllm_amd import kernels
    torch.manual_seed(1)
    T, H, Hkv, bs, nb = 9, 4, 2, 16, 6
    dt = torch.bfloat16
    qkv = torch.randn(T, (H + 2 * Hkv) * d).to(dt)
    pos = torch.randint(0, 200, (T,))
    slots = torch.tensor([3, 17, 18, -1, 40, 95, 0, 64, 65], dtype=torch.int32)
    table = L.build_cos_sin_cache(d, 256, 1e6)
    qn = (1 + 0.1 * torch.randn(d)).to(dt)
    kn = (1 + 0.1 * torch.randn(d)).to(dt)
    q, k, v = qkv.split([H * d, Hkv * d, Hkv * d], dim=-1)
    q, k, v = q.reshape(T, H, d), k.reshape(T, Hkv, d), v.reshape(T, Hkv, d)
    if qk_norm:
        q, k = L.rms_norm(q, qn, 1e-6), L.rms_norm(k, kn, 1e-6)
    q2, k2 = L.rope_qk(pos, q, k, table)
    kc, vc = torch.zeros(nb, bs, Hkv, d, dtype=dt), torch.zeros(nb, bs, Hkv, d, dtype=dt)
    OA.store_kvcache(k2, v, kc, vc, slots)

    qkv_d, kc_d, vc_d = dev(qkv), dev(torch.zeros_like(kc)), dev(torch.zeros_like(vc))
    kernels.rope_kv_store(qkv_d, dev(pos), dev(table), H, Hkv, d, slot_mapping=dev(slots), k_cache=kc_d, v_cache=vc_d,
                          q_norm_w=dev(qn) if qk_norm else None, k_norm_w=dev(kn) if qk_norm else None)
    got_q = qkv_d[:, : H * d].reshape(T, H, d).cpu()
    got_k = qkv_d[:, H * d: (H + Hkv) * d].reshape(T, Hkv, d).cpu()
    if qk_norm:
        ulp_close(got_q, q2, "bf16", ulps=2)
        ulp_close(kc_d, kc, "bf16", ulps=2)
    else:
        assert torch.equal(got_q, q2) and torch.equal(got_k, k2)
        assert torch.equal(kc_d.cpu(), kc)
    assert torch.equal(vc_d.cpu(), vc), "the V copy is a byte copy"
    assert torch.equal(qkv_d[:, (H + Hkv) * d:].cpu(), qkv[:, (H + Hkv) * d:]), "v must not be modified"


def test_kv_store_strided_views_and_negative_slots():
    from byox_vllm_amd import kernels
    torch.manual_seed(2)
    T, Hkv, d, bs, nb = 7, 2, 64, 16, 4
    qkv = torch.randn(T, 6 * d).to(torch.float16)
    k = qkv[:, 2 * d: 4 * d].view(T, Hkv, d)
    v = qkv[:, 4 * d:].view(T, Hkv, d)
    slots = torch.tensor([5, -1, 63, 0, -12, 17, 33], dtype=torch.int32)
    kc, vc = torch.zeros(nb, bs, Hkv, d, dtype=torch.float16), torch.zeros(nb, bs, Hkv, d, dtype=torch.float16)
    OA.store_kvcache(k, v, kc, vc, slots)
    qd = dev(qkv)
    kd, vd = dev(torch.zeros_like(kc)), dev(torch.zeros_like(vc))
    kernels.kv_store(qd[:, 2 * d: 4 * d].view(T, Hkv, d), qd[:, 4 * d:].view(T, Hkv, d), kd, vd, dev(slots))
    assert torch.equal(kd.cpu(), kc) and torch.equal(vd.cpu(), vc)


def _paged_case(B, H, Hkv, D, bs, lens, dt, seed=0, extra_cols=2):
    g = torch.Generator().manual_seed(seed)
    nblk = sum(-(-n // bs) for n in lens) + 3
    q = torch.randn(B, H, D, generator=g).to(dt)
    kc = torch.randn(nblk, bs, Hkv, D, generator=g).to(dt)
    vc = torch.randn(nblk, bs, Hkv, D, generator=g).to(dt)
    perm = torch.randperm(nblk, generator=g).tolist()
    width = max(-(-n // bs) for n in lens) + extra_cols
    tables = torch.full((B, width), -1, dtype=torch.int32)
    it = iter(perm)
    for i, n in enumerate(lens):
        for j in range(-(-n // bs)):
            tables[i, j] = next(it)
    return q, kc, vc, torch.tensor(lens, dtype=torch.int32), tables


def _decode_oracle(q, kc, vc, ctx, tables, scale):
    """-> (oracle output, oracle output with |V|: the sum_j p_j |v_j| the rounding bound is built from)."""
    return (OA.paged_attn_decode(q, kc, vc, ctx, tables, scale),
            OA.paged_attn_decode(q, kc, vc.abs(), ctx, tables, scale))


PAGED_CASES = [
    dict(B=5, H=14, Hkv=2, D=64, bs=16, lens=[1, 16, 17, 100, 257]),        # Qwen2-0.5B heads, tiny blocks, ragged
    dict(B=3, H=28, Hkv=4, D=128, bs=256, lens=[255, 256, 700]),            # Qwen2-7B heads, production block size
    dict(B=4, H=16, Hkv=8, D=128, bs=32, lens=[31, 32, 33, 64]),            # Qwen3-0.6B heads
    dict(B=2, H=4, Hkv=4, D=64, bs=16, lens=[5, 48]),                       # MHA (group of 1)
    dict(B=3, H=14, Hkv=2, D=64, bs=16, lens=[0, 40, 0]),                   # graph padding rows (context 0)
    dict(B=7, H=14, Hkv=2, D=64, bs=256, lens=[1023, 1024, 1025, 1061, 2047, 3000, 33]),   # bench-sized contexts:
    dict(B=4, H=28, Hkv=4, D=128, bs=256, lens=[1024, 1057, 2049, 31]),     # every wavefront runs its chunk rotation
    dict(B=2, H=8, Hkv=2, D=128, bs=16, lens=[515, 258]),                   # zero-padded query slots (group 4 of 16)
]


@pytest.mark.parametrize("name", list(DT))
@pytest.mark.parametrize("cfg", PAGED_CASES)
@pytest.mark.parametrize("splits", [1, 2, 5])
def test_paged_decode_attention(name, cfg, splits):
    from byox_vllm_amd import kernels
    dt = DT[name]
    q, kc, vc, ctx, tables = _paged_case(dt=dt, **cfg)
    want, want_abs = _decode_oracle(q, kc, vc, ctx, tables, cfg["D"] ** -0.5)
    got = kernels.paged_attn_decode(dev(q), dev(kc), dev(vc), dev(ctx), dev(tables), cfg["D"] ** -0.5,
                                    num_splits=splits)
    OA.check_close(got, want, want_abs, f"paged decode splits={splits}")


@pytest.mark.parametrize("waves", ["1", "2", "4", "8", "16"])
@pytest.mark.parametrize("cfg", [PAGED_CASES[0], PAGED_CASES[2], PAGED_CASES[5], PAGED_CASES[6], PAGED_CASES[7]])
@pytest.mark.parametrize("splits", [1, 3])
@pytest.mark.parametrize("hint", ["0", "1"])
def test_paged_decode_forced_wavefront_counts(cfg, waves, splits, hint, monkeypatch):
    """Every compiled workgroup size of the decode kernel (1, 2, 4, 8, 16 wavefronts; 16 exists at D=64 only and
    falls back to 8 at D=128) is the same function, with default-policy and with non-temporal K/V loads (the form
    launches over large caches of 64-wide heads take; NV_ATTN_NT forces either).  Without the overrides the launch
    shape picks among 2, 4 and 8 wavefronts."""
    from byox_vllm_amd import kernels
    monkeypatch.setenv("NV_ATTN_WAVES", waves)
    monkeypatch.setenv("NV_ATTN_NT", hint)
    q, kc, vc, ctx, tables = _paged_case(dt=torch.bfloat16, **cfg)
    want, want_abs = _decode_oracle(q, kc, vc, ctx, tables, cfg["D"] ** -0.5)
    got = kernels.paged_attn_decode(dev(q), dev(kc), dev(vc), dev(ctx), dev(tables), cfg["D"] ** -0.5,
                                    num_splits=splits)
    OA.check_close(got, want, want_abs, f"paged decode waves={waves} splits={splits}")


@pytest.mark.parametrize("D,H,Hkv", [(64, 14, 2), (128, 28, 4)])
@pytest.mark.parametrize("n", [1024, 3000])
@pytest.mark.parametrize("splits", [1, 4])
def test_paged_decode_single_chunk_mutation(D, H, Hkv, n, splits):
    """Zero the V rows of ONE interior 32-key chunk: the kernel's output must move exactly as the oracle's does.
    A kernel that skipped that chunk (or any other: see the census test below) cannot pass."""
    from byox_vllm_amd import kernels
    q, kc, vc, ctx, tables = _paged_case(B=1, H=H, Hkv=Hkv, D=D, bs=256, lens=[n], dt=torch.bfloat16, seed=n + D)
    scale = D ** -0.5
    chunk = (n // 32) // 2 + 3                                  # an interior chunk, not aligned to a block start
    vc2 = vc.clone()
    for tok in range(32 * chunk, 32 * chunk + 32):
        vc2[tables[0, tok // 256].item(), tok % 256] = 0
    want0, abs0 = _decode_oracle(q, kc, vc, ctx, tables, scale)
    want1, abs1 = _decode_oracle(q, kc, vc2, ctx, tables, scale)
    run = lambda v: kernels.paged_attn_decode(dev(q), dev(kc), dev(v), dev(ctx), dev(tables), scale, num_splits=splits)  # noqa: E731
    got0, got1 = run(vc), run(vc2)
    OA.check_close(got0, want0, abs0, "before the mutation")
    OA.check_close(got1, want1, abs1, "after the mutation")
    moved_want = (want1.float() - want0.float())
    moved_got = (got1.float() - got0.float()).cpu()
    assert moved_want.abs().max().item() > 0
    # the oracle's own movement carries two bf16 output roundings; the kernel's must match it to that accuracy
    resid = (moved_got - moved_want).pow(2).mean().sqrt().item()
    assert resid < 0.35 * moved_want.pow(2).mean().sqrt().item(), (resid, moved_want.pow(2).mean().sqrt().item())


@pytest.mark.parametrize("D,H,Hkv", [(64, 14, 2), (128, 28, 4)])
@pytest.mark.parametrize("n", [1024, 1999, 3000])
@pytest.mark.parametrize("splits", [1, 2, 5])
@pytest.mark.parametrize("waves", ["1", "2", "4", "8", "16"])
def test_paged_decode_chunk_census(D, H, Hkv, n, splits, waves, monkeypatch):
    """V marks every 32-key chunk with its own output dimension (V[token, d] = 1 iff d == chunk % D), so O[d] is the
    softmax mass of the chunks mapped to d: a chunk that is skipped, read twice or read from the wrong rows shows up
    as a missing or doubled mass in its own dimension."""
    from byox_vllm_amd import kernels
    monkeypatch.setenv("NV_ATTN_WAVES", waves)
    q, kc, vc, ctx, tables = _paged_case(B=2, H=H, Hkv=Hkv, D=D, bs=256, lens=[n, n - 77], dt=torch.bfloat16, seed=n)
    q = (q * 0.25).to(torch.bfloat16)          # flatter softmax: every chunk keeps a mass far above the rounding bound
    vc.zero_()
    for b, nb in enumerate([n, n - 77]):
        tok = torch.arange(nb)
        for h in range(Hkv):
            vc[tables[b, tok // 256].long(), tok % 256, h, (tok // 32) % D] = 1.0
    want, want_abs = _decode_oracle(q, kc, vc, ctx, tables, D ** -0.5)
    got = kernels.paged_attn_decode(dev(q), dev(kc), dev(vc), dev(ctx), dev(tables), D ** -0.5, num_splits=splits)
    OA.check_close(got, want, want_abs, "chunk census")
    assert (want.float() > 0).any()
    assert torch.allclose(got.float().cpu().sum(-1), torch.ones(2, H), atol=2e-2)   # masses sum to one


ADAPTIVE_CASES = [
    dict(B=4, H=28, Hkv=4, D=128, bs=256, lens=[1024, 1057, 2049, 31]),     # one group keeps both wavefronts, three keep one
    dict(B=4, H=28, Hkv=4, D=128, bs=256, lens=[700, 700, 700, 700]),       # equal lengths: every group keeps one
    dict(B=5, H=16, Hkv=8, D=128, bs=32, lens=[0, 33, 0, 900, 64]),         # padding rows pull the average down
    dict(B=3, H=8, Hkv=2, D=128, bs=16, lens=[515, 258, 1]),                # zero-padded query slots
]


@pytest.mark.parametrize("name", list(DT))
@pytest.mark.parametrize("cfg", ADAPTIVE_CASES)
def test_paged_decode_adaptive_wavefronts(cfg, name, monkeypatch):
    """The form large launches of 128-wide heads take (NV_ATTN_WAVES=0 forces it on small ones): workgroups of two
    wavefronts in which a group keeps both only if it is at least 1.5 times as long as the launch's average group.
    Same function as every fixed workgroup size, whichever way a group decides."""
    from byox_vllm_amd import kernels
    monkeypatch.setenv("NV_ATTN_WAVES", "0")
    q, kc, vc, ctx, tables = _paged_case(dt=DT[name], **cfg)
    want, want_abs = _decode_oracle(q, kc, vc, ctx, tables, cfg["D"] ** -0.5)
    got = kernels.paged_attn_decode(dev(q), dev(kc), dev(vc), dev(ctx), dev(tables), cfg["D"] ** -0.5, num_splits=1)
    OA.check_close(got, want, want_abs, "paged decode, adaptive wavefront count")
    monkeypatch.setenv("NV_ATTN_WAVES", "2")
    fixed = kernels.paged_attn_decode(dev(q), dev(kc), dev(vc), dev(ctx), dev(tables), cfg["D"] ** -0.5, num_splits=1)
    mean = sum(cfg["lens"]) / len(cfg["lens"])
    for b, n in enumerate(cfg["lens"]):
        if 2 * n >= 3 * mean:        # kept both wavefronts: the very same arithmetic as the fixed two-wavefront launch
            assert torch.equal(got[b].cpu(), fixed[b].cpu()), f"row {b}"


@pytest.mark.parametrize("lens", [[3000, 500], [1999, 1999], [96, 3000]])
def test_paged_decode_adaptive_chunk_census(lens, monkeypatch):
    """The census of test_paged_decode_chunk_census for the adaptive form: a group that keeps one wavefront walks all
    of its chunks with it, a group that keeps two deals them round-robin; no chunk may be skipped or doubled."""
    from byox_vllm_amd import kernels
    monkeypatch.setenv("NV_ATTN_WAVES", "0")
    D, H, Hkv = 128, 28, 4
    q, kc, vc, ctx, tables = _paged_case(B=2, H=H, Hkv=Hkv, D=D, bs=256, lens=lens, dt=torch.bfloat16, seed=lens[0])
    q = (q * 0.25).to(torch.bfloat16)
    vc.zero_()
    for b, nb in enumerate(lens):
        tok = torch.arange(nb)
        for h in range(Hkv):
            vc[tables[b, tok // 256].long(), tok % 256, h, (tok // 32) % D] = 1.0
    want, want_abs = _decode_oracle(q, kc, vc, ctx, tables, D ** -0.5)
    got = kernels.paged_attn_decode(dev(q), dev(kc), dev(vc), dev(ctx), dev(tables), D ** -0.5, num_splits=1)
    OA.check_close(got, want, want_abs, "chunk census, adaptive")
    assert torch.allclose(got.float().cpu().sum(-1), torch.ones(2, H), atol=2e-2)


def test_paged_decode_ignores_garbage_past_context():
    """Slots past the context length may hold NaN/Inf bit patterns (recycled blocks): they must not leak."""
    from byox_vllm_amd import kernels
    q, kc, vc, ctx, tables = _paged_case(B=2, H=14, Hkv=2, D=64, bs=16, lens=[19, 35], dt=torch.bfloat16)
    want, want_abs = _decode_oracle(q, kc, vc, ctx, tables, 0.125)
    for b, n in enumerate([19, 35]):
        blk, off = tables[b, n // 16].item(), n % 16
        kc[blk, off:] = float("nan")
        vc[blk, off:] = float("inf")
    got = kernels.paged_attn_decode(dev(q), dev(kc), dev(vc), dev(ctx), dev(tables), 0.125)
    OA.check_close(got, want, want_abs, "garbage past the context")


def test_paged_decode_softmax_rescale_branch():
    """A key far larger than the running max late in the sequence forces the online-softmax rescale."""
    from byox_vllm_amd import kernels
    q, kc, vc, ctx, tables = _paged_case(B=1, H=14, Hkv=2, D=64, bs=16, lens=[200], dt=torch.bfloat16, seed=3)
    blk = tables[0, 150 // 16].item()
    kc[blk, 150 % 16] = (q[0, 0] * 6).to(torch.bfloat16)      # spike for head 0 at token 150
    want, want_abs = _decode_oracle(q, kc, vc, ctx, tables, 0.125)
    got = kernels.paged_attn_decode(dev(q), dev(kc), dev(vc), dev(ctx), dev(tables), 0.125)
    OA.check_close(got, want, want_abs, "rescale branch")


@pytest.mark.parametrize("name", list(DT))
@pytest.mark.parametrize("cfg", [
    dict(H=4, Hkv=2, D=64, lens=[5, 19, 64, 1]),
    dict(H=14, Hkv=2, D=64, lens=[130, 33]),
    dict(H=4, Hkv=1, D=128, lens=[70, 16, 17]),
])
def test_prefill_attention_packed(name, cfg):
    from byox_vllm_amd import kernels
    dt = DT[name]
    g = torch.Generator().manual_seed(5)
    T = sum(cfg["lens"])
    H, Hkv, D = cfg["H"], cfg["Hkv"], cfg["D"]
    qkv = torch.randn(T, (H + 2 * Hkv) * D, generator=g).to(dt)   # strided views, as the model passes them
    q = qkv[:, : H * D].view(T, H, D)
    k = qkv[:, H * D: (H + Hkv) * D].view(T, Hkv, D)
    v = qkv[:, (H + Hkv) * D:].view(T, Hkv, D)
    cu = torch.tensor([0] + list(np.cumsum(cfg["lens"])), dtype=torch.int32)
    want = OA.attn_varlen(q, k, v, cu, cu, D ** -0.5)
    want_abs = OA.attn_varlen(q, k, v.abs(), cu, cu, D ** -0.5)
    qd = dev(qkv)
    got = kernels.attn_prefill(qd[:, : H * D].view(T, H, D), qd[:, H * D: (H + Hkv) * D].view(T, Hkv, D),
                               qd[:, (H + Hkv) * D:].view(T, Hkv, D), dev(cu), dev(cu), max(cfg["lens"]), D ** -0.5)
    OA.check_close(got, want, want_abs, "packed prefill")


@pytest.mark.parametrize("qt", ["1", "2"])
@pytest.mark.parametrize("cfg", [
    dict(H=14, Hkv=2, D=64, lens=[300, 97, 257, 128]),     # several query tiles per wavefront, ragged ends
    dict(H=8, Hkv=4, D=128, lens=[200, 65]),
])
def test_prefill_attention_query_tiles_per_wavefront(cfg, qt, monkeypatch):
    """The prefill kernel's tiling choices (one or two 16-query tiles per wavefront, prefetch on or off) are all the
    same function: forced either way, long ragged sequences match the oracle."""
    from byox_vllm_amd import kernels
    monkeypatch.setenv("NV_PREFILL_QT", qt)
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(21)
    T = sum(cfg["lens"])
    H, Hkv, D = cfg["H"], cfg["Hkv"], cfg["D"]
    q = torch.randn(T, H, D, generator=g).to(dt)
    k = torch.randn(T, Hkv, D, generator=g).to(dt)
    v = torch.randn(T, Hkv, D, generator=g).to(dt)
    cu = torch.tensor([0] + list(np.cumsum(cfg["lens"])), dtype=torch.int32)
    want = OA.attn_varlen(q, k, v, cu, cu, D ** -0.5)
    want_abs = OA.attn_varlen(q, k, v.abs(), cu, cu, D ** -0.5)
    got = kernels.attn_prefill(dev(q), dev(k), dev(v), dev(cu), dev(cu), max(cfg["lens"]), D ** -0.5)
    OA.check_close(got, want, want_abs, f"prefill qt={qt}")


def test_prefill_attention_prefix_cache_paged():
    """seqlen_q < seqlen_k: keys come from the paged cache, mask aligned to the bottom-right corner."""
    from byox_vllm_amd import kernels
    dt, H, Hkv, D, bs = torch.bfloat16, 4, 2, 64, 16
    k_lens, cached = [22, 40, 16 + 3], [16, 32, 16]
    q_lens = [n - c for n, c in zip(k_lens, cached)]
    _, kc, vc, _, tables = _paged_case(B=3, H=H, Hkv=Hkv, D=D, bs=bs, lens=k_lens, dt=dt, seed=8)
    g = torch.Generator().manual_seed(9)
    q = torch.randn(sum(q_lens), H, D, generator=g).to(dt)
    cu_q = torch.tensor([0] + list(np.cumsum(q_lens)), dtype=torch.int32)
    cu_k = torch.tensor([0] + list(np.cumsum(k_lens)), dtype=torch.int32)
    want = OA.attn_varlen(q, kc, vc, cu_q, cu_k, D ** -0.5, tables)
    want_abs = OA.attn_varlen(q, kc, vc.abs(), cu_q, cu_k, D ** -0.5, tables)
    got = kernels.attn_prefill(dev(q), dev(kc), dev(vc), dev(cu_q), dev(cu_k), max(q_lens), D ** -0.5,
                               block_tables=dev(tables))
    OA.check_close(got, want, want_abs, "prefix-cache prefill")


GEMM_SHAPES = [(1, 128, 128), (7, 192, 256), (32, 1152, 896), (128, 896, 896), (130, 256, 384), (300, 64, 128)]


@pytest.mark.parametrize("name", list(DT))
@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("fmt", ["dense", "tiled", "w8a16", "w4a16"])
@pytest.mark.parametrize("split_k", [1, 3])
def test_gemm_dense_and_dequant(name, M, N, K, fmt, split_k):
    """nv_gemm == F.linear on the (dequantised) weight; fp32 accumulation, one rounding at the end."""
    from byox_vllm_amd import kernels
    from byox_vllm_amd.layers.quant import LinearWeight
    dt = DT[name]
    g = torch.Generator().manual_seed(M * 7 + N)
    x = torch.randn(M, K, generator=g).to(dt)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dt)
    bias = (torch.randn(N, generator=g) * 0.1).to(dt)
    if fmt == "tiled":      # unquantised weights in the streaming layout: same values, other order
        lw = LinearWeight.dense_tiled(dev(w))
        assert torch.equal(lw.logical_dense().cpu(), w)
    else:
        lw = LinearWeight.make(dev(w), None if fmt == "dense" else fmt)
    if fmt == "w4a16":
        packed, scales = lw.logical_int4()
        assert torch.equal(packed.cpu(), Q.quantize_w4(w)[0]) and torch.equal(scales.cpu(), Q.quantize_w4(w)[1])
        w_ref = Q.dequantize_w4(packed.cpu(), scales.cpu())
    elif fmt == "w8a16":
        q8, scales = lw.logical_int8()
        assert torch.equal(q8.cpu(), Q.quantize_w8(w)[0]) and torch.equal(scales.cpu(), Q.quantize_w8(w)[1])
        w_ref = Q.dequantize_w8(q8.cpu(), scales.cpu())
    else:
        w_ref = w
    want = torch.nn.functional.linear(x.double(), w_ref.double(), bias.double())   # exact up to fp64
    got = kernels.gemm(dev(x), lw, dev(bias), split_k=split_k)
    # fp32 accumulation error is far below one output rounding; allow 1 ulp of the output dtype
    ulp_close(got, want.to(dt), name, ulps=1.0)


@pytest.mark.parametrize("name", list(DT))
@pytest.mark.parametrize("M,N,K", [(512, 128, 128), (777, 256, 384), (1300, 1152, 896)])
@pytest.mark.parametrize("fmt", ["tiled", "w8a16", "w4a16"])
def test_gemm_many_rows(name, M, N, K, fmt):
    """Prefill-sized M (many 128-row tiles, a ragged last one): same contract as at decode sizes."""
    from byox_vllm_amd import kernels
    from byox_vllm_amd.layers.quant import LinearWeight
    dt = DT[name]
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g).to(dt)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dt)
    bias = (torch.randn(N, generator=g) * 0.1).to(dt)
    if fmt == "tiled":
        lw, w_ref = LinearWeight.dense_tiled(dev(w)), w
    elif fmt == "w4a16":
        lw, w_ref = LinearWeight.make(dev(w), fmt), Q.dequantize_w4(*Q.quantize_w4(w))
    else:
        lw, w_ref = LinearWeight.make(dev(w), fmt), Q.dequantize_w8(*Q.quantize_w8(w))
    want = torch.nn.functional.linear(x.double(), w_ref.double(), bias.double())
    got = kernels.gemm(dev(x), lw, dev(bias))
    ulp_close(got, want.to(dt), name, ulps=1.0)


def _weights_for(fmt, w):
    """(kernel-side LinearWeight on the GPU, the oracle's dequantised reference weight on the CPU)."""
    from byox_vllm_amd.layers.quant import LinearWeight
    if fmt == "tiled":
        return LinearWeight.dense_tiled(dev(w)), w
    if fmt == "dense":
        return LinearWeight.dense(dev(w)), w
    if fmt == "w4a16":
        return LinearWeight.make(dev(w), fmt), Q.dequantize_w4(*Q.quantize_w4(w))
    return LinearWeight.make(dev(w), fmt), Q.dequantize_w8(*Q.quantize_w8(w))


FORCED = [("w4a16", n) for n in (1, 2, 3)] + [("w8a16", n) for n in (1, 2, 3)] + \
         [("tiled", n) for n in (1, 2)] + [("dense", n) for n in (1, 2)]


@pytest.mark.parametrize("fmt,ntw", FORCED)
@pytest.mark.parametrize("bm", ["16", "32", "64", "128"])
@pytest.mark.parametrize("split_k", [1, 3, 9])
def test_gemm_forced_tile_variants(fmt, ntw, bm, split_k, monkeypatch):
    """Every compiled (weight format, tiles per wavefront, rows per workgroup) instantiation, with and without
    split-K, is the same function: the planner normally picks among them by shape, here each one is forced on a
    shape with ragged rows (130 = 8 x 16 + 2) and a ragged last column block (29 tiles of 16 columns)."""
    from byox_vllm_amd import kernels
    monkeypatch.setenv("NV_GEMM_NTW", str(ntw))
    monkeypatch.setenv("NV_GEMM_BM", bm)
    M, N, K = 130, 464, 1152
    g = torch.Generator().manual_seed(ntw * 100 + int(bm) + split_k)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16)
    bias = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16)
    lw, w_ref = _weights_for(fmt, w)
    want = torch.nn.functional.linear(x.double(), w_ref.double(), bias.double())
    got = kernels.gemm(dev(x), lw, dev(bias), split_k=split_k)
    ulp_close(got, want.to(torch.bfloat16), "bf16", ulps=1.0)


@pytest.mark.parametrize("fmt,ntw", [("w4a16", 1), ("w4a16", 2), ("w4a16", 3), ("w8a16", 3), ("tiled", 2)])
@pytest.mark.parametrize("bm", ["16", "64", "128"])
def test_gemm_forced_tile_variants_silu_epilogue(fmt, ntw, bm, monkeypatch):
    from byox_vllm_amd import kernels
    from byox_vllm_amd.layers.quant import LinearWeight, interleave_gate_up
    monkeypatch.setenv("NV_GEMM_NTW", str(ntw))
    monkeypatch.setenv("NV_GEMM_BM", bm)
    M, K, inter = 130, 512, 232
    g = torch.Generator().manual_seed(ntw + int(bm))
    x = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(2 * inter, K, generator=g) * 0.05).to(torch.bfloat16)
    _, w_ref = _weights_for(fmt, w)
    wi = interleave_gate_up(dev(w))
    lw = LinearWeight.dense_tiled(wi) if fmt == "tiled" else LinearWeight.make(wi, fmt)
    want = L.silu_and_mul(L.linear(x, w_ref))
    got = kernels.gemm(dev(x), lw, silu_mul=True)
    assert got.shape == (M, inter)
    ulp_close(got, want, "bf16", ulps=2.0)


@pytest.mark.parametrize("name", list(DT))
@pytest.mark.parametrize("M,N,K", [(65, 32, 128), (128, 224, 256), (130, 416, 384), (200, 1152, 896), (128, 4608, 3584)])
@pytest.mark.parametrize("mode", ["bias", "plain-split1", "split3", "split7", "silu"])
def test_gemm_wide_kernel(name, M, N, K, mode, monkeypatch):
    """The 32x32x16-MFMA int4 kernel (12 wavefronts, 192 columns per workgroup, x by LDS-DMA), forced on shapes with
    ragged rows, a ragged last column group (N / 32 not a multiple of 6), a single K-block, K splits that leave one
    K-block per workgroup, every epilogue, both dtypes: the same function as the 16x16x32 kernel, against the fp64
    product with the oracle's dequantised weights."""
    from byox_vllm_amd import kernels
    from byox_vllm_amd.layers.quant import LinearWeight, interleave_gate_up
    if name == "fp16" and K > 1024:
        pytest.skip("an fp32 sum over 3584 terms is off by more than one fp16 ulp of an output that cancels to almost "
                    "zero (both kernels return the same bits there): the long-K shape is checked in bf16")
    monkeypatch.setenv("NV_GEMM_WIDE", "1")
    dt = DT[name]
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(dt)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dt)
    w_ref = Q.dequantize_w4(*Q.quantize_w4(w))
    if mode == "silu":
        if N % 64:
            pytest.skip("gate and up halves need N / 2 to be a multiple of 32 here")
        want = L.silu_and_mul(L.linear(x, w_ref))
        got = kernels.gemm(dev(x), LinearWeight.make(interleave_gate_up(dev(w)), "w4a16"), silu_mul=True)
        ulp_close(got, want, name, ulps=2.0)
        return
    bias = (torch.randn(N, generator=g) * 0.1).to(dt) if mode == "bias" else None
    split = {"bias": 1, "plain-split1": 1, "split3": 3, "split7": 7}[mode]
    want = torch.nn.functional.linear(x.double(), w_ref.double(), None if bias is None else bias.double())
    got = kernels.gemm(dev(x), LinearWeight.make(dev(w), "w4a16"), None if bias is None else dev(bias), split_k=split)
    ulp_close(got, want.to(dt), name, ulps=1.0)
    monkeypatch.setenv("NV_GEMM_WIDE", "0")      # and bit for bit what the 16x16x32 kernel returns for one K split
    if split == 1:                               # (with several splits the two kernels' slabs cover the same K ranges too)
        other = kernels.gemm(dev(x), LinearWeight.make(dev(w), "w4a16"), None if bias is None else dev(bias), split_k=split)
        assert (got.float() - other.float()).abs().max().item() <= 2.0 ** -7 * want.abs().max().item()


@pytest.mark.parametrize("name", list(DT))
@pytest.mark.parametrize("fmt", ["w4a16", "w8a16"])
@pytest.mark.parametrize("M,N,K", [(65, 160, 128), (128, 224, 256), (130, 480, 384), (200, 1184, 896), (128, 4608, 3584)])
@pytest.mark.parametrize("mode", ["bias", "plain-split1", "split3", "silu"])
def test_gemm_mixed_width_kernel(name, fmt, M, N, K, mode, monkeypatch):
    """The 160-column workgroup (column groups of 3, 3, 2, 2 tiles, the second k-step group mirrored over the SIMDs),
    forced on shapes with ragged rows, a last workgroup that ends inside any of its column groups, one K-block, K
    splits, every epilogue, both quantised formats and dtypes: the same function as the uniform kernels, bit for bit
    (the same k-steps are summed in the same order), and within 1 ulp of the fp64 product."""
    from byox_vllm_amd import kernels
    from byox_vllm_amd.layers.quant import LinearWeight, interleave_gate_up
    if name == "fp16" and K > 1024:
        pytest.skip("long-K shapes are checked in bf16 (an fp32 sum over 3584 terms against one fp16 ulp near zero)")
    dt = DT[name]
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(dt)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dt)
    w_ref = Q.dequantize_w4(*Q.quantize_w4(w)) if fmt == "w4a16" else Q.dequantize_w8(*Q.quantize_w8(w))

    def both(fn):
        monkeypatch.setenv("NV_GEMM_MIXED", "1")
        got = fn()
        monkeypatch.setenv("NV_GEMM_MIXED", "0")
        monkeypatch.setenv("NV_GEMM_WIDE", "0")
        return got, fn()

    if mode == "silu":
        if N % 32:
            pytest.skip("gate and up halves need N / 2 to be a multiple of 16")
        lw = LinearWeight.make(interleave_gate_up(dev(w)), fmt)
        got, other = both(lambda: kernels.gemm(dev(x), lw, silu_mul=True))
        assert torch.equal(got, other)
        # against the oracle's chain of three roundings (gate and up to the model dtype, silu, product): a gate that
        # the fp32 accumulation rounds to the neighbouring value moves silu by one ulp and the product by up to three
        ulp_close(got, L.silu_and_mul(L.linear(x, w_ref)), name, ulps=3.0)
        return
    bias = (torch.randn(N, generator=g) * 0.1).to(dt) if mode == "bias" else None
    split = {"bias": 1, "plain-split1": 1, "split3": 3}[mode]
    if split > K // 128:
        pytest.skip("fewer K-blocks than splits")
    want = torch.nn.functional.linear(x.double(), w_ref.double(), None if bias is None else bias.double())
    lw = LinearWeight.make(dev(w), fmt)
    got, other = both(lambda: kernels.gemm(dev(x), lw, None if bias is None else dev(bias), split_k=split))
    ulp_close(got, want.to(dt), name, ulps=1.0)
    assert torch.equal(got, other)


def _linear_fp64_rounded(x, w_ref, bias=None, rows=16384):
    """F.linear in fp64 on the CPU, rounded once to the model dtype; W is walked in row slices to bound memory."""
    xd = x.double()
    outs = []
    for lo in range(0, w_ref.size(0), rows):
        b = None if bias is None else bias[lo: lo + rows].double()
        outs.append(torch.nn.functional.linear(xd, w_ref[lo: lo + rows].double(), b).to(x.dtype))
    return torch.cat(outs, dim=1)


# the layer linears of Qwen2-7B (h 3584, q/k/v 3584 + 2 x 512, intermediate 18944) and its LM head, at the two batch
# sizes the benchmark runs them with: the planner's own choices for these shapes (3 / 2 tiles per wavefront, split-K 9
# for down_proj, 128-row tiles) are the kernels a 7B decode step spends its time in
QWEN2_7B_LINEARS = [("qkv", 4608, 3584, "w4a16"), ("o", 3584, 3584, "w4a16"), ("down", 3584, 18944, "w4a16"),
                    ("qkv", 4608, 3584, "w8a16"), ("down", 3584, 18944, "w8a16"), ("lm_head", 152064, 3584, "tiled")]


@pytest.mark.parametrize("M", [128, 512])
@pytest.mark.parametrize("which,N,K,fmt", QWEN2_7B_LINEARS)
def test_gemm_qwen2_7b_shapes(which, N, K, fmt, M):
    from byox_vllm_amd import kernels
    g = torch.Generator().manual_seed(N + M)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16)
    bias = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16) if which == "qkv" else None
    lw, w_ref = _weights_for(fmt, w)
    want = _linear_fp64_rounded(x, w_ref, bias)
    got = kernels.gemm(dev(x), lw, None if bias is None else dev(bias))          # split-K and tiles: the planner's
    ulp_close(got, want, "bf16", ulps=1.0)
    if which in ("o", "down"):      # as the decode step runs them: slabs left for the next add-RMSNorm to sum
        res = torch.randn(M, N, generator=g).to(torch.bfloat16)
        nw = (1 + 0.1 * torch.randn(N, generator=g)).to(torch.bfloat16)
        parts = kernels.gemm(dev(x), lw, partial_out=True)
        if isinstance(parts, kernels.Partials):
            out, new_res = kernels.add_rms_norm(parts, dev(res), dev(nw), 1e-6)
            # the fused consumer rounds the slab sum to bf16 first, exactly like the stand-alone GEMM did
            want_out, want_res = L.add_rms_norm(got.cpu(), res, nw, 1e-6)
            assert torch.equal(new_res.cpu(), want_res)
            ulp_close(out, want_out, "bf16", ulps=1.0)


# One rank's linears of Qwen2-7B sharded over 8 ranks (28/4 heads of 128: a KV head per rank pair, 4 query slots per
# rank; 148 MLP K-blocks dealt 19/18, padded to 19), int4, at the 512 rows the 8-GPU configuration decodes with:
# q|k|v (4 + 1 + 1) x 128 rows; o_proj K = 4 x 128; gate|up 2 x 19 x 128 rows with the SiLU epilogue; down_proj
# K = 19 x 128; the vocabulary shard 152064 / 8.
QWEN2_7B_TP8_SHARDS = [("qkv", 768, 3584, "w4a16"), ("o", 3584, 512, "w4a16"), ("gate_up", 4864, 3584, "w4a16"),
                       ("down", 3584, 2432, "w4a16"), ("lm_head", 19008, 3584, "tiled")]


@pytest.mark.parametrize("M", [128, 512])
@pytest.mark.parametrize("which,N,K,fmt", QWEN2_7B_TP8_SHARDS)
def test_gemm_qwen2_7b_tp8_shard_shapes(which, N, K, fmt, M):
    """Every GEMM a rank of the 8-way tensor-parallel Qwen2-7B runs, with the planner's own tiles and split-K for the
    shard shape, against the fp64 product of the dequantised weight rounded once; o/down also as slabs summed by the
    following add-RMSNorm (what the captured step does before the all-reduce)."""
    from byox_vllm_amd import kernels
    from byox_vllm_amd.layers.quant import LinearWeight, interleave_gate_up
    g = torch.Generator().manual_seed(N + M + K)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16)
    if which == "gate_up":
        w_ref = Q.dequantize_w4(*Q.quantize_w4(w))
        want = L.silu_and_mul(_linear_fp64_rounded(x, w_ref)).float()
        got = kernels.gemm(dev(x), LinearWeight.make(interleave_gate_up(dev(w)), fmt), silu_mul=True).float().cpu()
        err = (got - want).abs()
        unit = EPS["bf16"] * want.abs().clamp_min(1e-3) * 2
        assert (err <= unit).float().mean().item() > 0.995 and (err <= 4 * unit).all()
        return
    bias = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16) if which == "qkv" else None
    lw, w_ref = _weights_for(fmt, w)
    want = _linear_fp64_rounded(x, w_ref, bias)
    got = kernels.gemm(dev(x), lw, None if bias is None else dev(bias))
    ulp_close(got, want, "bf16", ulps=1.0)
    if which in ("o", "down"):
        res = torch.randn(M, N, generator=g).to(torch.bfloat16)
        nw = (1 + 0.1 * torch.randn(N, generator=g)).to(torch.bfloat16)
        parts = kernels.gemm(dev(x), lw, partial_out=True)
        if isinstance(parts, kernels.Partials):
            out, new_res = kernels.add_rms_norm(parts, dev(res), dev(nw), 1e-6)
            want_out, want_res = L.add_rms_norm(got.cpu(), res, nw, 1e-6)
            assert torch.equal(new_res.cpu(), want_res)
            ulp_close(out, want_out, "bf16", ulps=1.0)


@pytest.mark.parametrize("wide", ["0", "1"])
@pytest.mark.parametrize("M", [128, 512])
@pytest.mark.parametrize("fmt", ["w4a16", "w8a16"])
def test_gemm_qwen2_7b_gate_up_silu(fmt, M, wide, monkeypatch):
    """gate_up of Qwen2-7B (37888 x 3584) with the SiLU epilogue, for int4 through the 16x16x32 kernel and through the
    32x32x16 kernel (the planner's choice for this shape).  The oracle rounds gate and up to bf16 before the
    activation; a gate that sits on a rounding boundary may land one ulp apart in two correct implementations, which
    the activation and the product carry to the output: nearly all elements agree to 1 ulp, none is off by more than 4."""
    if fmt != "w4a16" and wide == "1":
        pytest.skip("the 32x32x16 kernel exists for int4 weights")
    monkeypatch.setenv("NV_GEMM_WIDE", wide)
    from byox_vllm_amd import kernels
    from byox_vllm_amd.layers.quant import LinearWeight, interleave_gate_up
    K, inter = 3584, 18944
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(2 * inter, K, generator=g) * 0.02).to(torch.bfloat16)
    w_ref = Q.dequantize_w4(*Q.quantize_w4(w)) if fmt == "w4a16" else Q.dequantize_w8(*Q.quantize_w8(w))
    want = L.silu_and_mul(_linear_fp64_rounded(x, w_ref)).float()
    got = kernels.gemm(dev(x), LinearWeight.make(interleave_gate_up(dev(w)), fmt), silu_mul=True).float().cpu()
    assert got.shape == (M, inter)
    err = (got - want).abs()
    unit = EPS["bf16"] * want.abs().clamp_min(1e-3) * 2
    assert (err > 4 * unit).sum().item() == 0, err.max().item()
    assert (err > unit).float().mean().item() < 2e-3


@pytest.mark.parametrize("name", list(DT))
@pytest.mark.parametrize("fmt", ["dense", "tiled", "w8a16", "w4a16"])
@pytest.mark.parametrize("gate_up", [False, True])
def test_weight_expand_is_the_oracles_dequantisation(name, fmt, gate_up):
    """nv_weight_expand: any nv_gemm weight format back to row-major in the model dtype, bit for bit the weight the
    oracle multiplies by (dense: the weight itself; int4 / int8: oracle/quant.py's dequantisation); rows stored for
    the silu(gate)*up epilogue come back as [gate rows; up rows]."""
    from byox_vllm_amd import kernels
    from byox_vllm_amd.layers.quant import interleave_gate_up
    dt = DT[name]
    g = torch.Generator().manual_seed(5)
    N, K = 352, 384
    w = (torch.randn(N, K, generator=g) * 0.05).to(dt)
    stored = interleave_gate_up(w) if gate_up else w
    lw, _ = _weights_for(fmt, stored)
    want = w if fmt in ("dense", "tiled") else (Q.dequantize_w4(*Q.quantize_w4(w)) if fmt == "w4a16" else
                                                 Q.dequantize_w8(*Q.quantize_w8(w)))
    got = kernels.expand_weight(lw, gate_up_rows=gate_up)
    assert got.dtype == dt and torch.equal(got.cpu().view(torch.int16), want.view(torch.int16))


@pytest.mark.parametrize("name", list(DT))
@pytest.mark.parametrize("M,N,K", [(2048, 128, 128), (2500, 1152, 896), (4100, 896, 4864)])
@pytest.mark.parametrize("fmt", ["dense", "tiled", "w8a16", "w4a16"])
def test_gemm_many_rows_through_the_vendor_library(name, M, N, K, fmt, monkeypatch):
    """From kernels.MANY_ROWS rows on, gemm() writes the weight out row-major and multiplies through nv_gemm_lt
    (hipBLASLt): same contract as nv_gemm (F.linear on the dequantised weight, fp32 accumulation, one rounding; with
    and without bias, into a strided output), and within one rounding of what nv_gemm itself returns."""
    from byox_vllm_amd import kernels
    dt = DT[name]
    assert kernels.lib().nv_gemm_lt_available() == 1 and kernels.MANY_ROWS == 4096
    monkeypatch.setattr(kernels, "MANY_ROWS", 2048)
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g).to(dt)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dt)
    bias = (torch.randn(N, generator=g) * 0.1).to(dt)
    lw, w_ref = _weights_for(fmt, w)
    calls = []
    real = kernels.gemm_many_rows
    monkeypatch.setattr(kernels, "gemm_many_rows", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    # One rounding of the exact product, plus what fp32 accumulation of K terms leaves behind where the products cancel
    # (a random walk of K roundings at 2^-24 of partial sums of a few units: 3e-5 at K = 4864, visible only in fp16
    # outputs close to zero; nv_gemm shows the same, scripts/r03_lt_accuracy.py).
    floor = 8.0 * K ** 0.5 * 2.0 ** -24

    def close(got, want, ulps):
        got, want = got.double().cpu(), want.double().cpu()
        bad = (got - want).abs() > ulps * EPS[name] * 2 * want.abs() + floor
        assert not bad.any(), f"{int(bad.sum())} of {bad.numel()} beyond {ulps} ulp; max abs {(got - want).abs().max()}"

    got = kernels.gemm(dev(x), lw, dev(bias))
    close(got, torch.nn.functional.linear(x.double(), w_ref.double(), bias.double()), 1.0)
    wide = dev(torch.zeros(M, N + 64, dtype=dt))
    got2 = kernels.gemm(dev(x), lw, out=wide[:, 8:8 + N])
    close(got2, torch.nn.functional.linear(x.double(), w_ref.double()), 1.0)
    assert wide[:, :8].eq(0).all() and wide[:, 8 + N:].eq(0).all() and len(calls) == 2
    monkeypatch.setattr(kernels, "MANY_ROWS", 0)
    own = kernels.gemm(dev(x), lw, dev(bias))
    assert len(calls) == 2
    close(got, own, 2.0)      # each within one rounding of the exact product


@pytest.mark.parametrize("M", [640, 2100])
@pytest.mark.parametrize("vendor", [True, False])
def test_gemm_many_rows_silu_mul(M, vendor, monkeypatch):
    from byox_vllm_amd import kernels
    monkeypatch.setattr(kernels, "MANY_ROWS", 2048 if vendor else 0)
    from byox_vllm_amd.layers.quant import LinearWeight, interleave_gate_up
    g = torch.Generator().manual_seed(23)
    K, inter = 256, 1536
    x = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(2 * inter, K, generator=g) * 0.05).to(torch.bfloat16)
    want = L.silu_and_mul(L.linear(x, Q.dequantize_w4(*Q.quantize_w4(w))))
    got = kernels.gemm(dev(x), LinearWeight.make(interleave_gate_up(dev(w)), "w4a16"), silu_mul=True)
    assert got.shape == (M, inter)
    ulp_close(got, want, "bf16", ulps=2.0)


@pytest.mark.parametrize("fmt", ["dense", "w4a16"])
@pytest.mark.parametrize("M", [5, 128])
def test_gemm_silu_mul_epilogue(fmt, M):
    """gate_up GEMM with SiluAndMul in the epilogue == oracle linear -> silu_and_mul (eager roundings)."""
    from byox_vllm_amd import kernels
    from byox_vllm_amd.layers.quant import LinearWeight, interleave_gate_up
    g = torch.Generator().manual_seed(21)
    K, inter = 256, 192
    x = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(2 * inter, K, generator=g) * 0.05).to(torch.bfloat16)
    w_ref = w if fmt == "dense" else Q.dequantize_w4(*Q.quantize_w4(w))
    want = L.silu_and_mul(L.linear(x, w_ref))
    lw = LinearWeight.make(interleave_gate_up(dev(w)), None if fmt == "dense" else fmt)
    got = kernels.gemm(dev(x), lw, silu_mul=True)
    assert got.shape == (M, inter)
    ulp_close(got, want, "bf16", ulps=2.0)


def test_splitk_partials_feed_add_rmsnorm_and_rope():
    """Slabs left by the GEMM, summed inside the consumer kernels, give what GEMM -> consumer gives."""
    from byox_vllm_amd import kernels
    from byox_vllm_amd.layers.quant import LinearWeight
    g = torch.Generator().manual_seed(22)
    M, K, hid = 37, 512, 128
    x = dev(torch.randn(M, K, generator=g).to(torch.bfloat16))
    w = dev((torch.randn(hid, K, generator=g) * 0.05).to(torch.bfloat16))
    res = dev(torch.randn(M, hid, generator=g).to(torch.bfloat16))
    nw = dev((1 + 0.1 * torch.randn(hid, generator=g)).to(torch.bfloat16))
    lw = LinearWeight.dense(w)
    y = kernels.gemm(x, lw, split_k=4)
    want_out, want_res = kernels.add_rms_norm(y, res.clone(), nw, 1e-6)
    parts = kernels.gemm(x, lw, split_k=4, partial_out=True)
    assert isinstance(parts, kernels.Partials) and parts.n == 4
    got_res = res.clone()
    got_out, _ = kernels.add_rms_norm(parts, got_res, nw, 1e-6)
    assert torch.equal(got_out, want_out) and torch.equal(got_res, want_res)
    # qkv slabs + bias -> rope + cache store
    H, Hkv, d, T = 2, 1, 64, 9
    wq = dev((torch.randn((H + 2 * Hkv) * d, K, generator=g) * 0.05).to(torch.bfloat16))
    bq = dev((torch.randn((H + 2 * Hkv) * d, generator=g) * 0.1).to(torch.bfloat16))
    xq = dev(torch.randn(T, K, generator=g).to(torch.bfloat16))
    pos = dev(torch.arange(3, 3 + T))
    slots = dev(torch.tensor([0, 1, 2, -1, 20, 21, 22, 23, 40], dtype=torch.int32))
    table = dev(L.build_cos_sin_cache(d, 64, 1e6))
    caches = [dev(torch.zeros(4, 16, Hkv, d, dtype=torch.bfloat16)) for _ in range(4)]
    lq = LinearWeight.dense(wq)
    want = kernels.rope_kv_store(kernels.gemm(xq, lq, bq, split_k=2), pos, table, H, Hkv, d, slot_mapping=slots,
                                 k_cache=caches[0], v_cache=caches[1])
    got = kernels.rope_kv_store(kernels.gemm(xq, lq, bq, split_k=2, partial_out=True), pos, table, H, Hkv, d,
                                slot_mapping=slots, k_cache=caches[2], v_cache=caches[3])
    assert torch.equal(got, want) and torch.equal(caches[0], caches[2]) and torch.equal(caches[1], caches[3])


def test_gemm_matches_cpu_flinear_closely():
    """Against the oracle's own F.linear in bf16 (different summation order): same values up to the last bit."""
    from byox_vllm_amd import kernels
    from byox_vllm_amd.layers.quant import LinearWeight
    g = torch.Generator().manual_seed(11)
    x = torch.randn(64, 896, generator=g).to(torch.bfloat16)
    w = (torch.randn(1152, 896, generator=g) * 0.02).to(torch.bfloat16)
    want = L.linear(x, w)
    got = kernels.gemm(dev(x), LinearWeight.dense(dev(w)))
    ulp_close(got, want, "bf16", ulps=1.0)


def test_embed_gather_with_vocab_shard():
    from byox_vllm_amd import kernels
    g = torch.Generator().manual_seed(12)
    w = torch.randn(64, 128, generator=g).to(torch.bfloat16)
    ids = torch.tensor([0, 63, 64, 127, 70, 5])
    for rank in (0, 1):
        want = L.vocab_embedding(ids, w, rank, 2)
        got = kernels.embed(dev(ids), dev(w), vocab_start=64 * rank)
        assert torch.equal(got.cpu(), want)


def test_sampler_replays_reference_draws(golden):
    from byox_vllm_amd import kernels
    g = golden.npz("sampler")
    logits = dev(t(g["logits"], torch.bfloat16))
    toks = kernels.sample(logits, dev(torch.from_numpy(g["temps"])), noise=dev(torch.from_numpy(g["noise"])))
    assert toks.cpu().tolist() == g["tokens"].tolist()
    greedy = kernels.sample(logits, dev(torch.zeros(6)))
    assert greedy.cpu().tolist() == logits.float().argmax(-1).cpu().tolist()


def test_sampler_large_vocab_greedy_and_distribution():
    from byox_vllm_amd import kernels
    g = torch.Generator().manual_seed(13)
    logits = (torch.randn(8, 151936, generator=g) * 2).to(torch.bfloat16)
    want = L.sample(logits, torch.zeros(8))
    got = kernels.sample(dev(logits), dev(torch.zeros(8)))
    assert got.cpu().tolist() == want.tolist()
    # in-kernel Exp(1) noise: a two-token distribution with p = (0.75, 0.25) is sampled at about that rate
    two = torch.full((4000, 64), -30.0)
    two[:, 3], two[:, 9] = np.log(3.0), 0.0
    toks = kernels.sample(dev(two.to(torch.bfloat16)), dev(torch.ones(4000)), seed=77).cpu()
    assert set(toks.tolist()) <= {3, 9}
    assert abs((toks == 3).float().mean().item() - 0.75) < 0.04


@pytest.mark.parametrize("V", [20000, 20003])
@pytest.mark.parametrize("temp", [1.0, 0.5])
def test_sampler_in_kernel_randomness_follows_softmax(V, temp):
    """Without injected noise the sampler draws from softmax(logits / T) (inverse CDF over slice sums): candidates
    spread over all slices of the row come out at their softmax rates; V % 8 != 0 takes the unaligned path."""
    from byox_vllm_amd import kernels
    B = 8000
    idx = [3, 2500, 7000, 9999, 12001, 15000, 17500, V - 1]
    p = np.array([0.30, 0.20, 0.15, 0.10, 0.10, 0.07, 0.05, 0.03])
    row = torch.full((V,), -40.0)
    row[idx] = torch.from_numpy(np.log(p)).float()
    logits = row.to(torch.bfloat16).unsqueeze(0).expand(B, V).contiguous()
    want = torch.softmax(logits[0].float() / temp, dim=-1)[idx].numpy()
    toks = kernels.sample(dev(logits), dev(torch.full((B,), temp)), seed=123).cpu().numpy()
    assert set(toks.tolist()) <= set(idx)
    freq = np.array([(toks == i).mean() for i in idx])
    assert np.abs(freq - want).max() < 0.02, (freq, want)
    # another seed gives other draws; greedy rows in the same launch are exact
    temps = torch.full((B,), temp)
    temps[::2] = 0.0
    toks2 = kernels.sample(dev(logits), dev(temps), seed=124).cpu().numpy()
    assert (toks2[::2] == 3).all() and (toks2[1::2] != toks[1::2]).any()


def test_sampler_masked_logits_and_limits():
    """-inf logits (a masked vocabulary) carry no mass and are never drawn, whole parts of them included; a row with
    a single finite logit returns it at any temperature; a vocabulary beyond the draw's LDS budget is refused."""
    from byox_vllm_amd import kernels
    V, B = 5000, 64
    g = torch.Generator().manual_seed(3)
    logits = torch.full((B, V), float("-inf"))
    allowed = [7, 63, 64, 1000, 1001, 4095, 4999]
    logits[:, allowed] = torch.randn(B, len(allowed), generator=g)
    logits[0] = float("-inf")
    logits[0, 4999] = -3.0
    temps = torch.tensor([0.0, 0.7, 1.3, 5.0])[torch.arange(B) % 4].float()
    lg = dev(logits.to(torch.bfloat16))
    seen = set()
    for seed in range(20):
        toks = kernels.sample(lg, dev(temps), seed=seed).cpu()
        assert set(toks.tolist()) <= set(allowed)
        assert toks[0].item() == 4999
        greedy = temps == 0
        assert toks[greedy].tolist() == lg.float().argmax(-1).cpu()[greedy].tolist()
        seen |= set(toks[~greedy].tolist())
    assert len(seen) >= 5
    # the largest row the draw's LDS image admits (part statistics in dynamic LDS next to the kernel's statics), and
    # one part more
    vmax = 64 * 8128
    big = torch.full((2, vmax), -4.0, dtype=torch.bfloat16)
    big[0, vmax - 1], big[1, 77] = 3.0, 3.0
    assert kernels.sample(dev(big), dev(torch.zeros(2))).tolist() == [vmax - 1, 77]
    assert set(kernels.sample(dev(big), dev(torch.full((2,), 0.05)), seed=1).tolist()) == {vmax - 1, 77}
    with pytest.raises(RuntimeError, match="bad shape"):
        kernels.sample(dev(torch.zeros(1, vmax + 1, dtype=torch.bfloat16)), dev(torch.ones(1)))


@pytest.mark.parametrize("name", list(DT))
@pytest.mark.parametrize("M,V,K,ntw", [(1, 4096, 256, "1"), (5, 4112, 128, "2"), (16, 20000, 256, "1"), (33, 8208, 384, "2"),
                                       (128, 151936, 128, "2"), (200, 30000 - 30000 % 16, 256, "1")])
def test_lm_head_sample_equals_gemm_then_sample(name, M, V, K, ntw, monkeypatch):
    """The chained LM head + sampler (statistics from the GEMM's epilogue) stores the logits nv_gemm stores and draws
    the tokens nv_sample draws from them, bit for bit: greedy rows, sampled rows at several temperatures, vocabularies
    that end inside a 64-column part, both tile widths, tiled and row-major weights."""
    from byox_vllm_amd import kernels
    from byox_vllm_amd.layers.quant import LinearWeight
    monkeypatch.setenv("NV_GEMM_NTW", ntw)
    dt = DT[name]
    g = torch.Generator().manual_seed(M * 7 + V)
    x = dev(torch.randn(M, K, generator=g).to(dt))
    w = dev((torch.randn(V, K, generator=g) * 0.3).to(dt))
    temps = torch.tensor([0.0, 0.6, 1.0, 1.7])[torch.arange(M) % 4].float()
    for packed in (LinearWeight.dense_tiled(w) if K % 128 == 0 else LinearWeight.dense(w), LinearWeight.dense(w)):
        want_logits = kernels.gemm(x, packed, split_k=1)
        want = kernels.sample(want_logits, dev(temps), seed=99)
        lib_call = kernels.lib().nv_lm_head_sample
        ws = kernels.workspace(kernels.lib().nv_lm_head_sample_workspace_bytes(M, V), x.device, "sample")
        ws.zero_()
        tokens = torch.empty(M, dtype=torch.int64, device=x.device)
        logits = torch.empty(M, V, dtype=dt, device=x.device)
        kernels.check(lib_call(kernels.ptr(x), x.stride(0), kernels.ptr(packed.data), packed.fmt, kernels.ptr(logits),
                               logits.stride(0), kernels.ptr(dev(temps)), kernels.ptr(tokens), kernels.ptr(ws),
                               ws.numel(), M, V, K, 99, None, kernels.dtype_code(dt), kernels.stream()),
                      "nv_lm_head_sample")
        assert torch.equal(logits, want_logits)
        assert tokens.cpu().tolist() == want.cpu().tolist()
        greedy = temps == 0
        assert tokens.cpu()[greedy].tolist() == want_logits.float().argmax(-1).cpu()[greedy].tolist()
        assert len(set(tokens.cpu()[~greedy].tolist())) > 1 or M < 3
        # the python wrapper takes the same route (or gemm + sample where the planner splits K)
        got, got_logits = kernels.lm_head_sample(x, packed, dev(temps), seed=99)
        assert torch.equal(got_logits, kernels.gemm(x, packed)) or kernels.pick_split_k(M, V, K, packed.fmt) == 1
        if kernels.pick_split_k(M, V, K, packed.fmt) == 1:
            assert got.cpu().tolist() == want.cpu().tolist()


def test_lm_head_sample_follows_softmax():
    """Rows of one hidden state against a vocabulary matrix with a few strong columns: the fused draw comes out at
    the softmax rates of the stored logits."""
    from byox_vllm_amd import kernels
    from byox_vllm_amd.layers.quant import LinearWeight
    V, K, B = 20000, 128, 512
    idx = [5, 2600, 9000, 15555, V - 1]
    w = torch.zeros(V, K)
    w[:, 0] = -30.0
    w[idx, 0] = torch.tensor([1.0, 0.5, 0.0, -0.5, 0.8])
    x = torch.zeros(B, K)
    x[:, 0] = 1.0
    packed = LinearWeight.dense_tiled(dev(w.to(torch.bfloat16)))
    counts = np.zeros(len(idx))
    for seed in range(8):
        toks, logits = kernels.lm_head_sample(dev(x.to(torch.bfloat16)), packed, dev(torch.full((B,), 0.8)), seed=seed)
        toks = toks.cpu().numpy()
        assert set(toks.tolist()) <= set(idx)
        counts += np.array([(toks == i).sum() for i in idx])
    want = torch.softmax(logits[0].float().cpu() / 0.8, dim=-1)[idx].numpy()
    assert np.abs(counts / counts.sum() - want).max() < 0.03, (counts / counts.sum(), want)


def test_stage_decode_inputs_pads_like_reference():
    from byox_vllm_amd import kernels
    bs, padded, cols, gcols = 3, 8, 2, 5
    g = dict(input_ids=dev(torch.full((8,), 7, dtype=torch.int64)), positions=dev(torch.full((8,), 7, dtype=torch.int64)),
             slot_mapping=dev(torch.full((8,), 7, dtype=torch.int32)), context_lens=dev(torch.full((8,), 7, dtype=torch.int32)),
             block_tables=dev(torch.full((8, gcols), 9, dtype=torch.int32)), padded_bs=padded)
    ids, pos = torch.tensor([11, 12, 13]), torch.tensor([4, 5, 6])
    slots, ctx = torch.tensor([40, 50, 60], dtype=torch.int32), torch.tensor([5, 6, 7], dtype=torch.int32)
    bt = torch.tensor([[1, 2], [3, -1], [4, 5]], dtype=torch.int32)
    kernels.stage_decode_inputs(dev(ids), dev(pos), dev(slots), dev(ctx), dev(bt), g)
    assert g["input_ids"].cpu().tolist() == [11, 12, 13, 0, 0, 0, 0, 0]
    assert g["slot_mapping"].cpu().tolist() == [40, 50, 60, -1, -1, -1, -1, -1]     # model_runner.py:248-249
    assert g["context_lens"].cpu().tolist() == [5, 6, 7, 0, 0, 0, 0, 0]             # model_runner.py:250-251
    assert g["block_tables"][:3, :2].cpu().tolist() == bt.tolist()
    assert g["block_tables"][:, 2:].cpu().eq(9).all()                              # stale columns stay, as in the reference


@pytest.mark.parametrize("fmt", ["tiled", "w8a16", "w4a16"])
@pytest.mark.parametrize("d,qk_norm,with_bias", [(64, False, True), (128, True, False), (64, True, True)])
@pytest.mark.parametrize("T", [1, 9, 130])
def test_gemm_rope_kv_store_equals_gemm_then_rope(fmt, d, qk_norm, with_bias, T):
    """The one-launch q/k/v projection (+bias, head norms, RoPE, cache store) writes the same bits as nv_gemm
    followed by nv_rope_kv_store, for every row-tile size the launch picks."""
    from byox_vllm_amd import kernels
    from byox_vllm_amd.layers.quant import LinearWeight
    g = torch.Generator().manual_seed(T * 13 + d)
    H, Hkv, K, bs, nb = 4, 2, 256, 16, 12
    dt = torch.bfloat16
    N = (H + 2 * Hkv) * d
    x = torch.randn(T, K, generator=g).to(dt)
    w = (torch.randn(N, K, generator=g) * 0.06).to(dt)
    bias = (torch.randn(N, generator=g) * 0.1).to(dt) if with_bias else None
    qn, kn = ((1 + 0.1 * torch.randn(d, generator=g)).to(dt) for _ in range(2))
    pos = torch.randint(0, 200, (T,), generator=g)
    slots = torch.randperm(nb * bs, generator=g)[:T].to(torch.int32)
    slots[T // 2] = -1
    table = L.build_cos_sin_cache(d, 256, 1e6)
    lw = LinearWeight.dense_tiled(dev(w)) if fmt == "tiled" else LinearWeight.make(dev(w), fmt)
    norms = dict(q_norm_w=dev(qn), k_norm_w=dev(kn)) if qk_norm else {}

    kc1, vc1 = (dev(torch.zeros(nb, bs, Hkv, d, dtype=dt)) for _ in range(2))
    qkv1 = kernels.gemm(dev(x), lw, None if bias is None else dev(bias), split_k=1)
    kernels.rope_kv_store(qkv1, dev(pos), dev(table), H, Hkv, d, slot_mapping=dev(slots), k_cache=kc1, v_cache=vc1, **norms)

    for bm in ("16", "32", "64", "128", ""):
        if bm:
            os.environ["NV_ROPE_GEMM_BM"] = bm
        else:
            os.environ.pop("NV_ROPE_GEMM_BM", None)
        kc2, vc2 = (dev(torch.zeros(nb, bs, Hkv, d, dtype=dt)) for _ in range(2))
        qkv2 = kernels.gemm_rope_kv_store(dev(x), lw, None if bias is None else dev(bias), dev(pos), dev(table), H, Hkv, d,
                                          slot_mapping=dev(slots), k_cache=kc2, v_cache=vc2, **norms)
        assert torch.equal(qkv2.cpu(), qkv1.cpu()), f"qkv differs at BM={bm or 'auto'}"
        assert torch.equal(kc2.cpu(), kc1.cpu()) and torch.equal(vc2.cpu(), vc1.cpu())
    os.environ.pop("NV_ROPE_GEMM_BM", None)


def test_stage_decode_packed_plain_and_header_form():
    """One packed int32 buffer per step; in the header form batch size and table width are read on the device."""
    from byox_vllm_amd import kernels
    padded, gcols = 8, 5

    def fresh():
        return dict(input_ids=dev(torch.full((8,), 7, dtype=torch.int64)), positions=dev(torch.full((8,), 7, dtype=torch.int64)),
                    slot_mapping=dev(torch.full((8,), 7, dtype=torch.int32)), context_lens=dev(torch.full((8,), 7, dtype=torch.int32)),
                    block_tables=dev(torch.full((8, gcols), 9, dtype=torch.int32)), temperatures=dev(torch.zeros(8)),
                    step=dev(torch.zeros(1, dtype=torch.int64)), padded_bs=padded)

    temps = np.array([0.5, 1.0, 0.0], dtype=np.float32)
    payload = np.concatenate([[11, 12, 13], [4, 5, 6], [40, 50, 60], [5, 6, 7], temps.view(np.int32), [1, 2, 3, -1, 4, 5]]).astype(np.int32)

    def check(g):
        assert g["input_ids"].cpu().tolist() == [11, 12, 13, 0, 0, 0, 0, 0]
        assert g["positions"].cpu().tolist() == [4, 5, 6, 0, 0, 0, 0, 0]
        assert g["slot_mapping"].cpu().tolist() == [40, 50, 60, -1, -1, -1, -1, -1]
        assert g["context_lens"].cpu().tolist() == [5, 6, 7, 0, 0, 0, 0, 0]
        assert g["temperatures"].cpu().tolist() == [0.5, 1.0, 0.0, 1.0, 1.0, 1.0, 1.0, 1.0]
        assert g["block_tables"][:3, :2].cpu().tolist() == [[1, 2], [3, -1], [4, 5]]
        assert g["block_tables"][:, 2:].cpu().eq(9).all() and g["block_tables"][3:].cpu().eq(9).all()
        assert g["step"].item() == 1

    g = fresh()
    kernels.stage_decode_packed(dev(torch.from_numpy(payload)), 3, 2, g)
    check(g)
    g = fresh()
    with_header = np.concatenate([[3, 2], payload, np.full(40, 12345)]).astype(np.int32)   # junk behind the step's words
    kernels.stage_decode_packed(dev(torch.from_numpy(with_header)), -1, 0, g)
    check(g)
    # a header that exceeds the static buffers is clamped on the device, never written out of bounds
    g = fresh()
    bad = np.concatenate([[100, 100], np.zeros(8 * (5 + gcols), dtype=np.int32)]).astype(np.int32)
    kernels.stage_decode_packed(dev(torch.from_numpy(bad)), -1, 0, g)
    assert g["context_lens"].cpu().tolist() == [0] * 8 and g["block_tables"].cpu().eq(0).all()


def test_stage_decode_advance_continues_the_previous_step_on_the_device():
    """Header flag NV_STAGE_ADVANCE: ids <- the previous step's tokens, positions and context lengths + 1, slots from
    the block tables already on the device (what ModelRunner.prepare_decode computes per sequence,
    model_runner.py:206-226), rows past bs and everything else untouched; without the flag the same launch stages
    the packed buffer."""
    from byox_vllm_amd import kernels
    from byox_vllm_amd.engine.model_runner import ModelRunner
    from byox_vllm_amd.engine.sequence import Sequence
    padded, gcols, block = 8, 5, 16
    g = dict(input_ids=dev(torch.zeros(8, dtype=torch.int64)), positions=dev(torch.zeros(8, dtype=torch.int64)),
             slot_mapping=dev(torch.zeros(8, dtype=torch.int32)), context_lens=dev(torch.zeros(8, dtype=torch.int32)),
             block_tables=dev(torch.zeros((8, gcols), dtype=torch.int32)), temperatures=dev(torch.zeros(8)),
             step=dev(torch.zeros(1, dtype=torch.int64)), padded_bs=padded)
    prev = dev(torch.tensor([101, 102, 103, 104, 0, 0, 0, 0], dtype=torch.int64))
    # four sequences: mid-block, last position of a block, first position of the next (allocated) block, and one
    # whose next position falls into a block the table does not hold yet (-1 padding)
    lens = [5, 31, 32, 48]
    tables = [[3, -1, -1], [4, 5, -1], [6, 7, 8], [9, 10, 11]]
    temps = np.array([0.5, 1.0, 0.0, 0.7], dtype=np.float32)
    payload = np.concatenate([[3, gcols - 2], [11, 12, 13, 14], [n - 1 for n in lens],
                              [tables[i][(n - 1) // block] * block + (n - 1) % block for i, n in enumerate(lens)], lens,
                              temps.view(np.int32), np.array(tables).reshape(-1)]).astype(np.int32)
    payload[0] = 4
    buf = dev(torch.from_numpy(payload.copy()))
    kernels.stage_decode_packed(buf, -1, 0, g, prev_tokens=prev, block_size=block)
    assert g["input_ids"].cpu().tolist()[:4] == [11, 12, 13, 14] and g["step"].item() == 1
    before_tables = g["block_tables"].cpu().clone()
    header_only = payload.copy()
    header_only[1] |= kernels.STAGE_ADVANCE
    header_only[2:] = 777                    # nothing behind the header may be read
    buf.copy_(torch.from_numpy(header_only))
    kernels.stage_decode_packed(buf, -1, 0, g, prev_tokens=prev, block_size=block)
    want_pos = lens                           # the new token's position = the old length
    assert g["input_ids"].cpu().tolist() == [101, 102, 103, 104, 0, 0, 0, 0]
    assert g["positions"].cpu().tolist() == want_pos + [0] * 4
    assert g["context_lens"].cpu().tolist() == [n + 1 for n in lens] + [0] * 4
    assert g["slot_mapping"].cpu().tolist() == [3 * 16 + 5, 5 * 16 + 15, 8 * 16 + 0, -1, -1, -1, -1, -1]
    assert torch.equal(g["block_tables"].cpu(), before_tables) and g["temperatures"].cpu().tolist()[:4] == temps.tolist()
    assert g["step"].item() == 2
    # NV_STAGE_STEP_BACK: the step counter first gives back the previous launch's count, in either form; the flag is
    # not taken for part of the table width
    header_only[1] |= kernels.STAGE_STEP_BACK
    buf.copy_(torch.from_numpy(header_only))
    kernels.stage_decode_packed(buf, -1, 0, g, prev_tokens=prev, block_size=block)
    assert g["step"].item() == 2 and g["positions"].cpu().tolist()[:4] == [n + 1 for n in lens]
    assert g["slot_mapping"].cpu().tolist()[:4] == [3 * 16 + 6, -1, 8 * 16 + 1, -1]
    full_again = payload.copy()
    full_again[1] |= kernels.STAGE_STEP_BACK
    buf.copy_(torch.from_numpy(full_again))
    kernels.stage_decode_packed(buf, -1, 0, g, prev_tokens=prev, block_size=block)
    assert g["step"].item() == 2 and g["input_ids"].cpu().tolist()[:4] == [11, 12, 13, 14]
    assert g["positions"].cpu().tolist()[:4] == [n - 1 for n in lens]
    assert torch.equal(g["block_tables"].cpu(), before_tables)


def test_bad_arguments_are_rejected_not_launched():
    from byox_vllm_amd import kernels
    from byox_vllm_amd.layers.quant import LinearWeight
    x = dev(torch.zeros(4, 100, dtype=torch.bfloat16))
    with pytest.raises(RuntimeError, match="nv_rmsnorm"):
        kernels.rms_norm(x, dev(torch.ones(100, dtype=torch.bfloat16)), 1e-6)       # hidden % 8 != 0
    with pytest.raises(RuntimeError, match="nv_gemm"):
        kernels.gemm(dev(torch.zeros(4, 96, dtype=torch.bfloat16)), LinearWeight.dense(dev(torch.zeros(64, 96, dtype=torch.bfloat16))))
    with pytest.raises(TypeError):
        kernels.rms_norm(dev(torch.zeros(4, 128)), dev(torch.ones(128)), 1e-6)      # fp32 is not a HIP-path dtype
    with pytest.raises(RuntimeError, match="no CPU compute path"):
        kernels.rms_norm(torch.zeros(4, 128, dtype=torch.bfloat16), torch.ones(128, dtype=torch.bfloat16), 1e-6)


@pytest.mark.parametrize("fmt", ["w4a16", "w8a16", "tiled"])
def test_result_store_policy_changes_no_bit(fmt):
    """nv_set_result_policy only picks the cache policy of the stores: normalised rows, residual rows, SiLU(gate)*up
    rows, split-K slabs (through the norm that sums them) and the q/k/v + cache rows of both RoPE forms are the same
    bits with it on; the setting is restored by the context manager; values other than 0 and 1 are refused."""
    from byox_vllm_amd import kernels
    from byox_vllm_amd._C import lib
    from byox_vllm_amd.layers.quant import LinearWeight, interleave_gate_up
    g = torch.Generator().manual_seed(len(fmt))
    dt = torch.bfloat16
    M, K, I, H, Hkv, d, bs, nb, mask = 100, 1024, 384, 4, 2, 64, 16, 12, 1
    x = dev(torch.randn(M, K, generator=g).to(dt))
    res = dev(torch.randn(M, K, generator=g).to(dt))
    nw = dev((1 + 0.1 * torch.randn(K, generator=g)).to(dt))
    make = (lambda w: LinearWeight.dense_tiled(w)) if fmt == "tiled" else (lambda w: LinearWeight.make(w, fmt))
    w_gu = make(dev(interleave_gate_up((torch.randn(2 * I, K, generator=g) * 0.05).to(dt))))
    w_o = make(dev((torch.randn(K, K, generator=g) * 0.05).to(dt)))
    w_qkv = make(dev((torch.randn((H + 2 * Hkv) * d, K, generator=g) * 0.05).to(dt)))
    pos = dev(torch.randint(0, 200, (M,), generator=g))
    slots = dev(torch.randperm(nb * bs, generator=g)[:M].to(torch.int32))
    table = dev(L.build_cos_sin_cache(d, 256, 1e6))

    def everything():
        out = {}
        out["norm"] = kernels.rms_norm(x, nw, 1e-6)
        out["add_norm"], out["residual"] = kernels.add_rms_norm(x, res.clone(), nw, 1e-6)
        out["act"] = kernels.gemm(x, w_gu, silu_mul=True)
        slabs = kernels.gemm(x, w_o, split_k=2, partial_out=True)
        out["slab_norm"], out["slab_residual"] = kernels.add_rms_norm(slabs, res.clone(), nw, 1e-6)
        kc, vc = (dev(torch.zeros(nb, bs, Hkv, d, dtype=dt)) for _ in range(2))
        out["qkv_fused"] = kernels.gemm_rope_kv_store(x, w_qkv, None, pos, table, H, Hkv, d, slot_mapping=slots,
                                                      k_cache=kc, v_cache=vc)
        out["k_fused"], out["v_fused"] = kc, vc
        kc2, vc2 = (dev(torch.zeros(nb, bs, Hkv, d, dtype=dt)) for _ in range(2))
        qkv = kernels.gemm(x, w_qkv, split_k=1)
        kernels.rope_kv_store(qkv, pos, table, H, Hkv, d, slot_mapping=slots, k_cache=kc2, v_cache=vc2)
        out["qkv_two"], out["k_two"], out["v_two"] = qkv, kc2, vc2
        torch.cuda.synchronize()
        return {k: v.cpu() for k, v in out.items()}

    assert lib().nv_get_result_policy() == 0
    plain = everything()
    with kernels.result_policy(mask):
        assert lib().nv_get_result_policy() == mask
        hinted = everything()
    assert lib().nv_get_result_policy() == 0
    for k in plain:
        assert torch.equal(plain[k], hinted[k]), k
    assert lib().nv_set_result_policy(2) != 0 and lib().nv_get_result_policy() == 0


def test_vendor_library_plan_table_is_bounded():
    """A serving process sees many prefill row counts: nv_gemm_lt keeps one plan per (rows, shape) and empties the table
    when it holds 512.  Six hundred row counts in a row, every product still the right one."""
    from byox_vllm_amd import kernels
    from byox_vllm_amd.layers.quant import LinearWeight
    g = torch.Generator().manual_seed(3)
    w = (torch.randn(128, 128, generator=g) * 0.05).to(torch.bfloat16)
    x = torch.randn(1000, 128, generator=g).to(torch.bfloat16)
    lw, xd = LinearWeight.dense(dev(w)), dev(x)
    want = torch.nn.functional.linear(x.double(), w.double())
    for rows in range(400, 1000):
        got = kernels.gemm_many_rows(xd[:rows], lw)
        if rows % 97 == 0 or rows == 999:
            ulp_close(got, want[:rows].to(torch.bfloat16), "bf16", ulps=1.0)
