"""LM head of a decode step (128 rows x 151936 columns, K = 896 / 3584): the vendor library's product on the row-major
weight (+ the sampler's scan over the stored logits) against nv_lm_head_sample, both replayed from a captured graph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from byox_vllm_amd import kernels
from byox_vllm_amd._C import check, dtype_code, lib, ptr, stream
from byox_vllm_amd.layers.quant import LinearWeight

def graph_time(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps

def run(M, V, K, copies, dt=torch.bfloat16):
    x = torch.randn(M, K, device="cuda").to(dt)
    ws_ = [(torch.randn(V, K, device="cuda") * 0.02).to(dt) for _ in range(copies)]   # cycled: no cache reuse between launches
    tiled = [LinearWeight.dense_tiled(w) for w in ws_]
    logits = torch.empty(M, V, device="cuda", dtype=dt)
    temps = torch.full((M,), 0.6, device="cuda")
    out = torch.empty(M, dtype=torch.int64, device="cuda")
    ws = torch.empty(128 << 20, dtype=torch.uint8, device="cuda")
    def vendor():
        for w in ws_:
            check(lib().nv_gemm_lt(ptr(x), K, ptr(w), None, ptr(logits), V, M, V, K, dtype_code(dt), ptr(ws), 128 << 20, stream()), "lt")
    def vendor_sample():
        for w in ws_:
            check(lib().nv_gemm_lt(ptr(x), K, ptr(w), None, ptr(logits), V, M, V, K, dtype_code(dt), ptr(ws), 128 << 20, stream()), "lt")
            kernels.sample(logits, temps, seed=1, out=out)
    def own():
        for w in tiled:
            kernels.lm_head_sample(x, w, temps, seed=1, out=out, logits=logits)
    a, b, c = graph_time(vendor) / copies, graph_time(vendor_sample) / copies, graph_time(own) / copies
    print(f"M={M} V={V} K={K}: vendor product {a:7.1f} us, vendor product + nv_sample {b:7.1f} us, nv_lm_head_sample {c:7.1f} us", flush=True)

with torch.inference_mode():
    run(128, 151936, 896, 2)
    run(128, 152064, 3584, 2)
    run(32, 151936, 896, 2)
