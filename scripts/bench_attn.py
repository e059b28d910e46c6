"""Micro-benchmark of nv_paged_attn_decode: L layers' caches visited in turn inside one hipGraph (no L3 reuse)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from byox_vllm_amd import kernels

def run(B, H, Hkv, D, S, L, splits_list, bs=256, dt=torch.bfloat16):
    dev = "cuda"
    nblk = B * (-(-S // bs))
    kv = torch.randn(2, L, nblk, bs, Hkv, D, device=dev, dtype=dt)
    q = torch.randn(B, H, D, device=dev, dtype=dt)
    ctx = torch.full((B,), S, dtype=torch.int32, device=dev)
    tables = torch.arange(nblk, dtype=torch.int32, device=dev).view(B, -1)
    out = torch.empty_like(q)
    bytes_per = B * S * 2 * Hkv * D * 2
    for splits in splits_list:
        for l in range(L):
            kernels.paged_attn_decode(q, kv[0, l], kv[1, l], ctx, tables, D ** -0.5, out=out, num_splits=splits)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for l in range(L):
                kernels.paged_attn_decode(q, kv[0, l], kv[1, l], ctx, tables, D ** -0.5, out=out, num_splits=splits)
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        a.record()
        for _ in range(reps):
            g.replay()
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / (reps * L)
        print(f"B={B} H={H}/{Hkv} D={D} S={S} waves={os.environ.get('NV_ATTN_WAVES','auto')} splits={splits}: "
              f"{us:7.2f} us/launch  {bytes_per / us / 1e6:6.2f} TB/s", flush=True)

def run_prefill(n, S, H, Hkv, D, dt=torch.bfloat16):
    """Causal varlen prefill attention on n packed sequences of S tokens."""
    dev = "cuda"
    T = n * S
    q = torch.randn(T, H, D, device=dev, dtype=dt)
    k = torch.randn(T, Hkv, D, device=dev, dtype=dt)
    v = torch.randn(T, Hkv, D, device=dev, dtype=dt)
    cu = torch.arange(0, T + 1, S, dtype=torch.int32, device=dev)
    out = torch.empty_like(q)
    call = lambda: kernels.attn_prefill(q, k, v, cu, cu, S, D ** -0.5, out=out)
    call(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    a.record()
    for _ in range(reps):
        call()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / reps
    flops = 4.0 * n * (S * (S + 1) / 2) * H * D
    print(f"prefill n={n} S={S} H={H}/{Hkv} D={D} qt={os.environ.get('NV_PREFILL_QT', 'auto')}: {us:8.1f} us  "
          f"{flops / us / 1e6:6.1f} TFLOP/s (causal flops)", flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "0.5b"
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    if which == "0.5b":
        run(128, 14, 2, 64, S, 24, [1, 2, 4])
    elif which == "pmc":      # one configuration only, for counter passes: every paged_decode launch is the same
        run(128, 14, 2, 64, S, 24, [1])
    elif which == "pmc7b":
        run(128, 28, 4, 128, S, 8, [1])
    elif which == "7b":
        run(128, 28, 4, 128, S, 8, [1, 2])
    elif which == "qwen3":    # Qwen3-0.6B geometry at the short contexts of the reference's own bench script
        run(128, 16, 8, 128, S, 28, [1])
    elif which == "batches":  # split choice at the smaller graph batch sizes (combine launch included)
        for B in (8, 16, 32, 64):
            run(B, 14, 2, 64, S, 24, [1, 2, 4, 8, 16])
        for B in (8, 32):
            run(B, 28, 4, 128, S, 8, [1, 2, 4, 8])
    elif which == "split2":   # the split kernel alone (read its duration from a rocprofv3 --stats run)
        run(128, 14, 2, 64, S, 24, [2])
    elif which == "7b512":
        run(512, 28, 4, 128, S, 4, [1])
    elif which == "prefill":
        run_prefill(8, 1024, 14, 2, 64)
        run_prefill(2, 4096, 14, 2, 64)
        run_prefill(8, 1024, 28, 4, 128)
        run_prefill(2, 4096, 28, 4, 128)
        run_prefill(64, 128, 14, 2, 64)
    elif which == "mall":     # working sets of 1, 2, 3 and 24 layers: does a cache-resident KV (256 MB Infinity Cache) stream faster?
        for L in (1, 2, 3, 24):
            print("layers cycled:", L)
            run(128, 14, 2, 64, S, L, [1])
    elif which == "small":
        run(32, 14, 2, 64, 1024, 24, [1, 2, 4, 8])
        run(1, 14, 2, 64, 1024, 24, [1, 4, 8, 16])
    elif which == "layout":
        run(128, 14, 2, 64, 1024, 24, [1])      # rows of the two kv heads interleaved (reference layout)
        run(256, 7, 1, 64, 1024, 24, [1])       # same bytes, every stream contiguous
        run(128, 14, 2, 64, 4096, 6, [1])       # longer context: prologue/epilogue amortised
        run(256, 7, 1, 64, 4096, 6, [1])
    elif which == "rows256":  # can 128 workgroups that read contiguous 256-byte rows keep up with 256 workgroups on 128-byte rows?
        run(128, 7, 1, 128, 1100, 24, [1])      # one kv head of 128: contiguous 256-byte rows, 128 workgroups
        run(128, 7, 1, 128, 4096, 6, [1])
        run(128, 14, 2, 64, 1100, 24, [1])      # two kv heads of 64: the same bytes as interleaved 128-byte rows, 256 workgroups
        run(128, 14, 2, 64, 4096, 6, [1])
        run(256, 7, 1, 128, 1100, 12, [1])
