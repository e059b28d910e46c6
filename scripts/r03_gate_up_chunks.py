"""gate/up of a prefill step through the vendor library: one product + one SiluAndMul pass over [M, 2I], against the
same work cut into column chunks small enough for the chunk's product to be consumed from the memory-side cache."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from byox_vllm_amd import kernels
from byox_vllm_amd._C import check, dtype_code, lib, ptr, stream

def timeit(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps

def run(M, I, K, chunks_list, dt=torch.bfloat16):
    x = torch.randn(M, K, device="cuda").to(dt)
    gate = (torch.randn(I, K, device="cuda") * 0.02).to(dt)
    up = (torch.randn(I, K, device="cuda") * 0.02).to(dt)
    ws = torch.empty(128 << 20, dtype=torch.uint8, device="cuda")
    for chunks in chunks_list:
        C = I // chunks
        assert C * chunks == I and C % 8 == 0
        wr = [torch.cat((gate[c * C:(c + 1) * C], up[c * C:(c + 1) * C])).contiguous() for c in range(chunks)]
        y = torch.empty(M, 2 * C, device="cuda", dtype=dt)
        outs = [torch.empty(M, C, device="cuda", dtype=dt) for _ in range(chunks)]
        def step():
            for c in range(chunks):
                check(lib().nv_gemm_lt(ptr(x), K, ptr(wr[c]), None, ptr(y), 2 * C, M, 2 * C, K, dtype_code(dt), ptr(ws), 128 << 20, stream()), "lt")
                check(lib().nv_silu_mul(ptr(y), ptr(outs[c]), M, C, dtype_code(dt), stream()), "silu")
        us = timeit(step)
        print(f"M={M} I={I} K={K} chunks {chunks:2d} (product of {M * 2 * C * 2 >> 20} MiB each): {us:8.1f} us", flush=True)

with torch.inference_mode():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    run(M, 4864, 896, (1, 2, 4, 8, 19))
    run(M, 3072, 1024, (1, 2, 4, 8))
    run(M // 2, 18944, 3584, (1, 2, 4, 8, 16))
