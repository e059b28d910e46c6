"""Rounding of the vendor-library product against the exact one: elements beyond 1 ulp of the output dtype, with and
without a library workspace (a split-K algorithm needs one for its partial sums)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from byox_vllm_amd import kernels
from byox_vllm_amd._C import check, dtype_code, lib, ptr, stream

def one(M, N, K, dt, ws_bytes):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(M, K, generator=g).to(dt).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).to(dt).cuda()
    out = torch.empty(M, N, dtype=dt, device="cuda")
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device="cuda")
    def call():
        check(lib().nv_gemm_lt(ptr(x), K, ptr(w), None, ptr(out), N, M, N, K, dtype_code(dt), ptr(ws), ws_bytes, stream()), "lt")
    call(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): call()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 100
    want = (x.double() @ w.double().t())
    eps = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    diff = (out.double() - want).abs()
    err = diff / (want.abs().clamp_min(1e-3) * eps * 2)
    bad = err > 1
    print(f"M={M} N={N} K={K} {str(dt)[6:]:8s} workspace {ws_bytes >> 20:3d} MiB: {us:8.1f} us  max {err.max().item():.2f} ulp, "
          f"{int(bad.sum())} of {err.numel()} beyond 1; among those max |want| {want.abs()[bad].max().item() if bad.any() else 0:.4f}, "
          f"max abs diff {diff[bad].max().item() if bad.any() else 0:.2e}", flush=True)
    lw = __import__("byox_vllm_amd.layers.quant", fromlist=["LinearWeight"]).LinearWeight.dense_tiled(w)
    kernels.MANY_ROWS = 0
    own = kernels.gemm(x, lw)
    diff = (own.double() - want).abs()
    err = diff / (want.abs().clamp_min(1e-3) * eps * 2)
    bad = err > 1
    print(f"   nv_gemm (tiled): max {err.max().item():.2f} ulp, {int(bad.sum())} beyond 1; among those max |want| "
          f"{want.abs()[bad].max().item() if bad.any() else 0:.4f}, max abs diff {diff[bad].max().item() if bad.any() else 0:.2e}", flush=True)

with torch.inference_mode():
    for dt in (torch.float16, torch.bfloat16):
        for M, N, K in ((4100, 896, 4864), (2500, 1152, 896), (14592, 1024, 3072), (14592, 3584, 18944), (2048, 896, 896)):
            for wsb in (128 << 20,):
                one(M, N, K, dt, wsb)
