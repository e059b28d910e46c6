"""Micro-benchmark of nv_gemm on the decode shapes: L distinct weight copies visited in turn inside one hipGraph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from byox_vllm_amd import kernels
from byox_vllm_amd.layers.quant import LinearWeight

def run(name, M, N, K, fmt, L, split_k=None, silu=False, partial=False, dt=torch.bfloat16):
    dev = "cuda"
    ws = []
    wsrc = (torch.randn(N, K, device=dev) * 0.02).to(dt)
    base = LinearWeight.dense_tiled(wsrc) if fmt == "tiled" else LinearWeight.make(wsrc, fmt)
    for _ in range(L):
        w = LinearWeight(base.fmt, base.data.clone(), None if base.scales is None else base.scales.clone(), N, K, dt)
        ws.append(w)
    pad = int(os.environ.get("NV_BENCH_X_PAD", "0"))      # extra elements per row of x: a row stride that is not a
    x = torch.randn(M, K + pad, device=dev).to(dt)[:, :K]  # multiple of 4 KB spreads the tile over the L2 channels
    out = torch.empty(M, N // 2 if silu else N, device=dev, dtype=dt)
    def call(w):
        return kernels.gemm(x, w, out=None if partial else out, split_k=split_k, silu_mul=silu, partial_out=partial)
    for w in ws:
        call(w)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for w in ws:
            call(w)
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    a.record()
    for _ in range(reps):
        g.replay()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / (reps * L)
    flops = 2.0 * M * N * K
    print(f"{name:10s} M={M} N={N} K={K} {fmt or 'dense'} split={split_k} silu={silu} partial={partial}: "
          f"{us:8.2f} us  {flops / us / 1e6:7.1f} TFLOP/s  {base.nbytes / us / 1e6:6.2f} TB/s(w)", flush=True)

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "0.5b"
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    if which == "0.5b":
        run("qkv", M, 1152, 896, "w4a16", 24, partial=True)
        run("o", M, 896, 896, "w4a16", 24, partial=True)
        run("gate_up", M, 9728, 896, "w4a16", 24, silu=True)
        run("down", M, 896, 4864, "w4a16", 24, partial=True)
        run("lm_head", M, 151936, 896, None, 2)
    elif which == "7b":
        run("qkv", M, 4608, 3584, "w4a16", 8, partial=True)
        run("o", M, 3584, 3584, "w4a16", 8, partial=True)
        run("gate_up", M, 37888, 3584, "w4a16", 4, silu=True)
        run("down", M, 3584, 18944, "w4a16", 4, partial=True)
        run("lm_head", M, 152064, 3584, None, 2)
    elif which == "7bcold":   # enough copies that no weight is still in the 256 MB Infinity Cache when it is read again
        run("qkv", M, 4608, 3584, "w4a16", 80, partial=True)
        run("o", M, 3584, 3584, "w4a16", 100, partial=True)
        run("gate_up", M, 37888, 3584, "w4a16", 10, silu=True)
        run("down", M, 3584, 18944, "w4a16", 20, partial=True)
    elif which == "sweep7b":
        for sk in (2, 3, 4, 6, 9):
            run("down", M, 3584, 18944, "w4a16", 4, split_k=sk, partial=True)
        for sk in (1, 2, 3, 4, 6):
            run("qkv", M, 4608, 3584, "w4a16", 8, split_k=sk, partial=True)
        for sk in (2, 3, 4, 7):
            run("o", M, 3584, 3584, "w4a16", 8, split_k=sk, partial=True)
    elif which == "sweep05b":
        for sk in (2, 4, 7):
            run("qkv", M, 1152, 896, "w4a16", 24, split_k=sk, partial=True)
        for sk in (2, 4, 7):
            run("o", M, 896, 896, "w4a16", 24, split_k=sk, partial=True)
        for sk in (4, 9, 18, 38):
            run("down", M, 896, 4864, "w4a16", 24, split_k=sk, partial=True)
        run("gate_up", M, 9728, 896, "w4a16", 24, silu=True)
    elif which == "dense05b":
        for fmt in (None, "tiled"):
            run("qkv", M, 1152, 896, fmt, 24, partial=True)
            run("o", M, 896, 896, fmt, 24, partial=True)
            run("gate_up", M, 9728, 896, fmt, 24, silu=True)
            run("down", M, 896, 4864, fmt, 24, partial=True)
    elif which == "lmhead":
        for n in (32768, 151936):
            run("lm_head", M, n, 896, None, 4)
            run("lm_head", M, n, 896, "tiled", 4)
        run("lm_head7b", M, 152064, 3584, None, 2)
        run("lm_head7b", M, 152064, 3584, "tiled", 2)
    elif which == "gateup7b":
        run("gate_up", M, 37888, 3584, "w4a16", 4, silu=True)
    elif which == "gateup05b":
        run("gate_up", M, 9728, 896, "w4a16", 24, silu=True)
