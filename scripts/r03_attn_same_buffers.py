"""Wavefront-count settings of the paged decode kernel timed on the SAME buffers, interleaved, several rounds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from byox_vllm_amd import kernels

B, H, Hkv, D, S, L, bs = (int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "128,28,4,128,1024,8,256".split(",")))
settings = sys.argv[2:] or ["0", "1", "2"]
dev, dt = "cuda", torch.bfloat16
nblk = B * (-(-S // bs))
kv = torch.randn(2, L, nblk, bs, Hkv, D, device=dev, dtype=dt)
q = torch.randn(B, H, D, device=dev, dtype=dt)
ctx = torch.full((B,), S, dtype=torch.int32, device=dev)
tables = torch.arange(nblk, dtype=torch.int32, device=dev).view(B, -1)
out = torch.empty_like(q)
graphs = {}
for w in settings:
    os.environ["NV_ATTN_WAVES"] = w
    for l in range(L):
        kernels.paged_attn_decode(q, kv[0, l], kv[1, l], ctx, tables, D ** -0.5, out=out)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for l in range(L):
            kernels.paged_attn_decode(q, kv[0, l], kv[1, l], ctx, tables, D ** -0.5, out=out)
    graphs[w] = g
res = {w: [] for w in settings}
for rnd in range(7):
    for w in settings:
        g = graphs[w]
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            g.replay()
        b.record(); torch.cuda.synchronize()
        res[w].append(a.elapsed_time(b) * 1e3 / (20 * L))
for w in settings:
    xs = sorted(res[w])
    print(f"B={B} Hkv={Hkv} D={D} S={S} waves={w}: median {xs[len(xs) // 2]:.2f} us  min {xs[0]:.2f}  max {xs[-1]:.2f}")
