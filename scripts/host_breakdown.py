"""Where the host's share of a decode step goes (0.5B int4, batch 128): pack, graph launch, wait, read-back, driver."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from byox_vllm_amd.config import Config, qwen2_config
from byox_vllm_amd.engine.model_runner import ModelRunner

hf = qwen2_config("qwen2-0.5b", dtype=torch.bfloat16)
batch, context, bs = 128, 1024, 256
per = context // bs + 2
cfg = Config("synthetic", hf_config=hf, quantization="w4a16", kvcache_block_size=bs, num_kvcache_blocks=batch * per + 1,
             max_model_len=per * bs, max_num_seqs=batch, max_num_batched_tokens=8192, graph_batch_sizes=[batch], seed=0)
r = ModelRunner(cfg, 0, None)
r.kv_cache.normal_()
seqs, nb = bench.make_sequences(batch, context, bs)
drv = bench.StepDriver(r, seqs, nb)
T = dict(pack=0.0, launch=0.0, wait=0.0, read=0.0, driver=0.0)
pc = time.perf_counter
def run_graph(seqs):
    g, host = r.graph_vars, r._host_stage_np
    t0 = pc()
    n, cols = r._pack_decode(seqs, host[2:])
    host[0], host[1] = len(seqs), cols
    padded = g["padded_bs"] = next(x for x in r.graph_bs if x >= len(seqs))
    t1 = pc()
    r._replay(padded)
    t2 = pc()
    torch.cuda.current_stream().synchronize()
    t3 = pc()
    out = r._host_tokens_np[:len(seqs)].tolist()
    t4 = pc()
    T["pack"] += t1 - t0; T["launch"] += t2 - t1; T["wait"] += t3 - t2; T["read"] += t4 - t3
    return out
r._run_decode_graph = run_graph
for _ in range(8):
    drv.step()
for k in T: T[k] = 0.0
torch.cuda.synchronize()
N = 200
t0 = pc()
for _ in range(N):
    drv.step()
torch.cuda.synchronize()
total = (pc() - t0) / N
T["driver"] = total - sum(T.values()) / N
print(f"step {total * 1e6:.1f} us: " + " | ".join(f"{k} {v / N * 1e6 if k != 'driver' else v * 1e6:.1f}" for k, v in T.items()))
# device time of one replay
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    r._replay(batch)
b.record(); torch.cuda.synchronize()
print(f"graph replay back to back: {a.elapsed_time(b) * 1e3 / 20:.1f} us")
r.exit()
