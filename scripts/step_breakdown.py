"""Where one decode step spends its time: host prepare, graph replay, LM head, sampler (debug aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from byox_vllm_amd.config import Config, qwen2_config
from byox_vllm_amd.engine.model_runner import ModelRunner
from byox_vllm_amd.utils.context import get_context, reset_context

wl = sys.argv[1] if len(sys.argv) > 1 else bench.DEFAULT_WORKLOAD
model_name, quant, dt, batch, context = bench.WORKLOADS[wl]
hf = qwen2_config(model_name, dtype=bench.DTYPES[dt])
bps = (context + 200) // 256 + 1
cfg = Config("synthetic", hf_config=hf, quantization=quant, kvcache_block_size=256, num_kvcache_blocks=batch * bps + 1,
             max_model_len=bps * 256, max_num_seqs=batch, max_num_batched_tokens=8192, graph_batch_sizes=[batch])
r = ModelRunner(cfg, 0, None)
r.kv_cache.normal_()
seqs, nb = bench.make_sequences(batch, context, 256)
drv = bench.StepDriver(r, seqs, nb)
for _ in range(5):
    drv.step()
torch.cuda.synchronize()
# wall time of a whole step versus the device time of the graph replay inside it
replay = r._replay
spans = []
def timed_replay(bs):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); replay(bs); b.record()
    spans.append((a, b))
r._replay = timed_replay
N = 30
t0 = time.perf_counter()
for _ in range(N):
    drv.step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / N * 1e3
gpu = sum(a.elapsed_time(b) for a, b in spans) / len(spans)
print(f"wall {wall:.3f} ms/step, graph replay on device {gpu:.3f} ms, host + copies + sync {wall - gpu:.3f} ms")

# host-side split of one step (wall clock of each Python-level piece; the device runs asynchronously under them)
import numpy as np
from byox_vllm_amd import kernels as K
r._replay = replay
acc = {}
def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter(); out = fn(*a, **k); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
        return out
    return w
r._pack_decode = timed("pack", r._pack_decode)
r._replay = timed("graph launch", r._replay)
K.stage_decode_packed = timed("stage launch", K.stage_decode_packed)
run = r.run
r.run = timed("run total", run)
t0 = time.perf_counter()
for _ in range(N):
    drv.step()
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / N * 1e6
print({k: round(v / N * 1e6, 1) for k, v in acc.items()}, f"us per step; whole step {tot:.1f} us (driver bookkeeping = whole - run)")
