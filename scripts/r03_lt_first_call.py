"""Host time of the FIRST nv_gemm_lt call of a shape in a fresh process (algorithm search, lazy loading of the library's
kernels) against a repeated call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from byox_vllm_amd._C import check, dtype_code, lib, ptr, stream
dt = torch.bfloat16
ws = torch.empty(128 << 20, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for M, N, K in ((16384, 4736, 3584), (12300, 4736, 3584), (12300, 3584, 2432), (12300, 768, 3584), (12300, 3584, 512),
                (5000, 4736, 3584), (7777, 9728, 896), (4096, 1152, 896)):
    x = torch.randn(M, K, device="cuda").to(dt); w = torch.randn(N, K, device="cuda").to(dt)
    out = torch.empty(M, N, device="cuda", dtype=dt)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        check(lib().nv_gemm_lt(ptr(x), K, ptr(w), None, ptr(out), N, M, N, K, dtype_code(dt), ptr(ws), 128 << 20, stream()), "lt")
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"M={M} N={N} K={K}: first call {ts[0]:9.2f} ms, then {ts[1]:7.2f}, {ts[2]:7.2f}", flush=True)
