"""LM head + sampler on the decode shape: nv_gemm then nv_sample against the chained nv_lm_head_sample."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from byox_vllm_amd import kernels
from byox_vllm_amd.layers.quant import LinearWeight

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
V, K = 151936, int(sys.argv[2]) if len(sys.argv) > 2 else 896
w = LinearWeight.dense_tiled((torch.randn(V, K, device="cuda") * 0.05).to(torch.bfloat16))
xs = [torch.randn(B, K, device="cuda").to(torch.bfloat16) for _ in range(4)]
logits = torch.empty(B, V, dtype=torch.bfloat16, device="cuda")
out = torch.empty(B, dtype=torch.int64, device="cuda")


def timed(name, fn):
    for x in xs:
        fn(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for x in xs:
            fn(x)
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        g.replay()
    b.record(); torch.cuda.synchronize()
    print(f"{name}: {a.elapsed_time(b) * 1e3 / 80:.2f} us", flush=True)


for temp in (0.0, 0.6):
    temps = torch.full((B,), temp, device="cuda")
    timed(f"T={temp} gemm only            ", lambda x: kernels.gemm(x, w, out=logits))
    timed(f"T={temp} sample only          ", lambda x: kernels.sample(logits, temps, seed=1, out=out))
    timed(f"T={temp} gemm + sample        ", lambda x: (kernels.gemm(x, w, out=logits), kernels.sample(logits, temps, seed=1, out=out)))
    timed(f"T={temp} chained head + sample", lambda x: kernels.lm_head_sample(x, w, temps, seed=1, out=out, logits=logits))
