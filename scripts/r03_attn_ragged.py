"""One long sequence among short ones: what the wavefront count per (sequence, KV head) group costs a ragged batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from byox_vllm_amd import kernels

def run(B, H, Hkv, D, short, long_, n_long, bs=256, L=8, dt=torch.bfloat16):
    dev = "cuda"
    per = -(-long_ // bs)
    nblk = B * per
    kv = torch.randn(2, L, nblk, bs, Hkv, D, device=dev, dtype=dt)
    q = torch.randn(B, H, D, device=dev, dtype=dt)
    lens = [long_] * n_long + [short] * (B - n_long)
    ctx = torch.tensor(lens, dtype=torch.int32, device=dev)
    tables = torch.arange(nblk, dtype=torch.int32, device=dev).view(B, per)
    out = torch.empty_like(q)
    g = torch.cuda.CUDAGraph()
    for l in range(L):
        kernels.paged_attn_decode(q, kv[0, l], kv[1, l], ctx, tables, D ** -0.5, out=out)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for l in range(L):
            kernels.paged_attn_decode(q, kv[0, l], kv[1, l], ctx, tables, D ** -0.5, out=out)
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        g.replay()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / (20 * L)
    mb = sum(lens) * 2 * Hkv * D * 2 / 1e6
    print(f"B={B} Hkv={Hkv} D={D}: {n_long} x {long_} + {B - n_long} x {short} tokens ({mb:.0f} MB) waves={os.environ.get('NV_ATTN_WAVES', 'auto')}: {us:7.2f} us", flush=True)

for w in ("0", "1", "2", "4"):
    os.environ["NV_ATTN_WAVES"] = w
    shapes = [(256, 2048, 1), (1024, 2048, 4), (256, 4096, 1)] if len(sys.argv) < 2 else [
        tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
    for short, long_, n_long in shapes:
        run(128, 28, 4, 128, short, long_, n_long)
