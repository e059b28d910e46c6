"""Do two independent kernel chains overlap when captured as two branches of one hipGraph?
Chain = n dependent small GEMMs (0.5B o-projection shape at 64 rows).  (a) both chains one after the other on one
stream, (b) forked onto two streams inside one capture, (c) a single chain at 128 rows (the same total work), (d) two graphs replayed on two streams."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from byox_vllm_amd import kernels
from byox_vllm_amd.layers.quant import LinearWeight

def replay_us(g, reps=20):
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps

def main(n=48, K=896, N=896):
    dev, dt = "cuda", torch.bfloat16
    ws = [LinearWeight.make((torch.randn(N, K, device=dev) * 0.03).to(dt), "w4a16") for _ in range(n)]
    def chain(x):
        for w in ws:
            x = kernels.gemm(x, w)
        return x
    xa, xb = (torch.randn(64, K, device=dev).to(dt) for _ in range(2))
    xfull = torch.randn(128, K, device=dev).to(dt)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(s1):
        chain(xa); chain(xfull)
    with torch.cuda.stream(s2):
        chain(xb)
    torch.cuda.synchronize()
    g_seq = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_seq, stream=s1):
        chain(xa); chain(xb)
    g_full = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_full, stream=s1):
        chain(xfull)
    g_fork = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_fork, stream=s1):
        s2.wait_stream(s1)
        chain(xa)
        with torch.cuda.stream(s2):
            chain(xb)
        s1.wait_stream(s2)
    # (d) two graphs, one per chain, each replayed on its own stream
    g_a, g_b = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_a, stream=s1):
        chain(xa)
    with torch.cuda.graph(g_b, stream=s2):
        chain(xb)
    def both():
        with torch.cuda.stream(s1):
            g_a.replay()
        with torch.cuda.stream(s2):
            g_b.replay()
    import time
    both(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        both()
    torch.cuda.synchronize()
    two_graphs = (time.perf_counter() - t0) * 1e6 / 20
    with torch.cuda.stream(s1):
        one_graph = replay_us(g_a)
    print(f"{n} GEMMs per chain: one 64-row chain as a graph {one_graph:8.1f} us | two graphs on two streams {two_graphs:8.1f} us", flush=True)
    print(f"{n} GEMMs per chain: two 64-row chains in sequence {replay_us(g_seq):8.1f} us | forked {replay_us(g_fork):8.1f} us | "
          f"one 128-row chain {replay_us(g_full):8.1f} us", flush=True)

if __name__ == "__main__":
    main()
