"""Time of one GEMM launch against K at fixed M, N: the slope is the K loop, the intercept launch + prologue + epilogue."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.bench_gemm import run
N = int(sys.argv[1]) if len(sys.argv) > 1 else 37888
silu = (sys.argv[2] == "silu") if len(sys.argv) > 2 else True
for kb in (1, 2, 4, 7, 14, 28, 56):
    K = 128 * kb
    L = max(2, min(24, int(600e6 // (N * K // 2))))
    run(f"kb={kb}", 128, N, K, "w4a16", L, split_k=None if silu else 1, silu=silu, partial=not silu)
