"""nv_paged_attn_decode_static (the entry point that allocates nothing: split partials in an array built into the
library) against nv_paged_attn_decode (caller's workspace) through the C ABI: the same bits, on two calls in a row,
the second with a smaller input, and against the oracle's rounding bound."""

import pytest
import torch

from oracle import attention as OA

pytestmark = pytest.mark.gpu


def _case(B, H, Hkv, D, bs, lens, seed):
    g = torch.Generator().manual_seed(seed)
    nblk = sum(-(-n // bs) for n in lens) + 2
    q = torch.randn(B, H, D, generator=g).to(torch.bfloat16)
    kc = torch.randn(nblk, bs, Hkv, D, generator=g).to(torch.bfloat16)
    vc = torch.randn(nblk, bs, Hkv, D, generator=g).to(torch.bfloat16)
    perm = iter(torch.randperm(nblk, generator=g).tolist())
    tables = torch.full((B, max(-(-n // bs) for n in lens)), -1, dtype=torch.int32)
    for i, n in enumerate(lens):
        for j in range(-(-n // bs)):
            tables[i, j] = next(perm)
    return q, kc, vc, torch.tensor(lens, dtype=torch.int32), tables


@pytest.mark.parametrize("D,H,Hkv", [(64, 14, 2), (128, 28, 4)])
def test_static_scratch_entry_point_matches_workspace_entry_point(D, H, Hkv):
    from byox_vllm_amd import _C, kernels
    scale = D ** -0.5
    # first a large call that fills much of the array, then a smaller one that must not see its leftovers
    calls = [dict(B=24, lens=[1500 + 37 * i for i in range(24)], splits=6, seed=1),
             dict(B=3, lens=[700, 33, 1025], splits=4, seed=2),
             dict(B=2, lens=[90, 300], splits=1, seed=3)]
    for c in calls:
        q, kc, vc, ctx, tables = _case(c["B"], H, Hkv, D, 256, c["lens"], c["seed"])
        qd, kd, vd, cd, td = (t.cuda() for t in (q, kc, vc, ctx, tables))
        want_ws = kernels.paged_attn_decode(qd, kd, vd, cd, td, scale, num_splits=c["splits"])
        out = torch.full_like(qd, float("nan"))
        torch.cuda.synchronize()
        before = torch.cuda.memory_stats()["allocation.all.allocated"]
        free_before, _ = torch.cuda.mem_get_info()             # hipMemGetInfo: the DEVICE's free memory
        kernels.paged_attn_decode_static(qd, kd, vd, cd, td, scale, out=out, num_splits=c["splits"])
        assert torch.cuda.memory_stats()["allocation.all.allocated"] == before      # torch's allocator was not asked
        torch.cuda.synchronize()
        # and neither was the driver: a hipMalloc inside the library (of the 0.2-4.5 MB of partials these calls need,
        # rounded up to the driver's 2 MiB granules) would show as less free device memory.  The very first call's
        # hipGetSymbolAddress maps nothing: the array is part of the code object loaded with the library.
        assert torch.cuda.mem_get_info()[0] >= free_before, "device memory was allocated during the call"
        assert torch.equal(out, want_ws), f"call {c} differs from the workspace entry point"
        want, want_abs = (OA.paged_attn_decode(q, kc, v, ctx, tables, scale) for v in (vc, vc.abs()))
        OA.check_close(out, want, want_abs, "static-scratch decode")
    # beyond the built-in array: a clear error, nothing launched
    lib = _C.lib()
    assert lib.nv_paged_attn_decode_workspace_bytes(4096, 28, 128, 32) > (16 << 20)
    q, kc, vc, ctx, tables = (t.cuda() for t in _case(2, H, Hkv, D, 256, [40, 50], 4))
    code = lib.nv_paged_attn_decode_static(q.data_ptr(), q.stride(0), kc.data_ptr(), vc.data_ptr(), tables.data_ptr(),
                                           tables.stride(0), ctx.data_ptr(), q.data_ptr(), q.stride(0), 4096, 28, 4,
                                           128, 256, scale, 32, 0, None)
    assert code == -1 and b"built-in array" in lib.nv_last_error()
