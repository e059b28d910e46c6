"""Where the host time of an eager prefill step goes: cProfile over ModelRunner.run(prefill) on a short prompt."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from byox_vllm_amd.config import Config, qwen2_config
from byox_vllm_amd.engine.model_runner import ModelRunner
from byox_vllm_amd.engine.sequence import Sequence

hf = qwen2_config("qwen2-0.5b", dtype=torch.bfloat16)
cfg = Config("synthetic", hf_config=hf, quantization="w4a16", kvcache_block_size=256, num_kvcache_blocks=64,
             max_model_len=4096, max_num_seqs=16, max_num_batched_tokens=8192, enforce_eager=True)
r = ModelRunner(cfg, 0, None)
s = Sequence(list(range(1, 513)), 256)
s.block_table = [0, 1]
for _ in range(3):
    r.run([s], True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    r.run([s], True)
torch.cuda.synchronize()
print(f"prefill 1 x 512: {(time.perf_counter() - t0) * 100:.2f} ms per step")
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    r.run([s], True)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
