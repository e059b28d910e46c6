"""Micro-benchmark of nv_sample on the LM-head output shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from byox_vllm_amd import kernels
B, V = 128, 151936
logits = [torch.randn(B, V, device="cuda").to(torch.bfloat16) for _ in range(4)]
for temp in (0.0, 0.6):
    temps = torch.full((B,), temp, device="cuda")
    out = torch.empty(B, dtype=torch.int64, device="cuda")
    for l in logits:
        kernels.sample(l, temps, seed=1, out=out)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for l in logits:
            kernels.sample(l, temps, seed=1, out=out)
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        g.replay()
    b.record(); torch.cuda.synchronize()
    print(f"temperature {temp}: {a.elapsed_time(b) * 1e3 / 80:.2f} us per sample call ({B * V * 2 / 1e6:.0f} MB of logits)")
