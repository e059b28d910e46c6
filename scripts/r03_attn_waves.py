"""Wavefronts per workgroup of the paged decode kernel over launch shapes (NV_ATTN_WAVES forced per run)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_attn
shapes = [  # B, H, Hkv, D, S, layers cycled
    (128, 28, 4, 128, 1024, 8), (128, 28, 4, 128, 4096, 2), (256, 28, 4, 128, 1024, 4), (64, 28, 4, 128, 1024, 8),
    (512, 14, 2, 64, 1024, 6), (256, 14, 2, 64, 1024, 12), (256, 14, 2, 64, 4096, 3), (128, 16, 8, 128, 180, 28),
]
for shape in shapes:
    for w in sys.argv[1:]:
        os.environ["NV_ATTN_WAVES"] = w
        bench_attn.run(*shape, [1])
