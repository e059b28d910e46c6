"""Prefill throughput of the whole model (ModelRunner.run with is_prefill=True): tokens/s for a few prompt shapes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from byox_vllm_amd.config import Config, qwen2_config
from byox_vllm_amd.engine.model_runner import ModelRunner
from byox_vllm_amd.engine.sequence import Sequence

model = sys.argv[1] if len(sys.argv) > 1 else "qwen2-0.5b"
quant = sys.argv[2] if len(sys.argv) > 2 else "w4a16"
quant = None if quant == "none" else quant
hf = qwen2_config(model, dtype=torch.bfloat16)
cfg = Config("synthetic", hf_config=hf, quantization=quant, kvcache_block_size=256, num_kvcache_blocks=160,
             max_model_len=8192, max_num_seqs=64, max_num_batched_tokens=32768, enforce_eager=True)
r = ModelRunner(cfg, 0, None)
shapes = ((1, 512), (1, 2048), (1, 8192), (8, 1024), (16, 2048))
if len(sys.argv) > 3:    # e.g. "1x256,1x1024,4x1024"
    shapes = tuple(tuple(int(v) for v in item.split("x")) for item in sys.argv[3].split(","))
for nseq, length in shapes:
    seqs, nb = [], 0
    for _ in range(nseq):
        s = Sequence(list(range(1, length + 1)), 256)
        blocks = -(-length // 256)
        s.block_table = list(range(nb, nb + blocks)); nb += blocks
        seqs.append(s)
    r.run(seqs, True); torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        r.run(seqs, True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    h = hf
    flops = 2 * nseq * length * (sum(p.numel() for n, p in r.model.named_parameters() if "embed" not in n) if quant is None else 0)
    print(f"{model} {quant or 'bf16'} prefill {nseq} x {length}: {dt * 1e3:8.2f} ms  {nseq * length / dt / 1e3:8.1f} k tok/s", flush=True)
