"""Prefill-sized products: nv_gemm (tiled dense and int4 weights) against the vendor-library path of kernels.gemm
(weight written out row-major into scratch + hipBLASLt product, both timed), Qwen3-0.6B / Qwen2-0.5B / Qwen2-7B
layer shapes, launched eagerly as a prefill step launches them.  Usage: python scripts/r03_dense_prefill_gemm.py [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from byox_vllm_amd import kernels
from byox_vllm_amd.layers.quant import LinearWeight

def timeit(fn, reps=20):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps

def run(name, M, N, K, dt=torch.bfloat16):
    w = (torch.randn(N, K, device="cuda") * 0.02).to(dt)
    x = torch.randn(M, K, device="cuda").to(dt)
    out = torch.empty(M, N, device="cuda", dtype=dt)
    flops = 2.0 * M * N * K
    res = {}
    for fmt in ("tiled", "w4a16"):
        lw = LinearWeight.dense_tiled(w) if fmt == "tiled" else LinearWeight.make(w, fmt)
        kernels.MANY_ROWS = 0
        res[f"nv_gemm {fmt}"] = timeit(lambda: kernels.gemm(x, lw, out=out))
        kernels.MANY_ROWS = 1
        res[f"vendor path {fmt}"] = timeit(lambda: kernels.gemm(x, lw, out=out))
    print(f"{name:13s} M={M} N={N} K={K}: " + "  ".join(f"{k} {v:7.1f} us {flops / v / 1e6:6.1f} TF" for k, v in res.items()), flush=True)

if __name__ == "__main__":
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 14592
    with torch.inference_mode():
        for name, N, K in (("0.6b qkv", 4096, 1024), ("0.6b o", 1024, 2048), ("0.6b gate_up", 6144, 1024), ("0.6b down", 1024, 3072),
                           ("0.5b qkv", 1152, 896), ("0.5b gate_up", 9728, 896), ("0.5b down", 896, 4864),
                           ("7b qkv", 4608, 3584), ("7b gate_up", 37888, 3584), ("7b down", 3584, 18944)):
            run(name, M, N, K)
