"""Host time of a decode step on the device-advance path (0.5B int4, batch 128): the phases of _run_decode_graph timed
in place, plus the driver's token appends."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from byox_vllm_amd import kernels
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
T = dict(check=0.0, header=0.0, launch=0.0, wait=0.0, read=0.0, cache=0.0)
pc = time.perf_counter
full = [0]

def run_graph(seqs):
    t0 = pc()
    bs_ = len(seqs)
    g, host = r.graph_vars, r._host_stage_np
    padded = next(x for x in r.graph_bs if x >= bs_)
    ntok = r._continues_previous_step(seqs, padded)
    t1 = pc()
    cache = getattr(r, "_pack_cache", None)
    if ntok is not None:
        host[0], host[1] = bs_, cache["key"][1] | kernels.STAGE_ADVANCE
    else:
        full[0] += 1
        n, cols = r._pack_decode(seqs, host[2:])
        host[0], host[1] = bs_, cols
        cache = r._pack_cache
        ntok = cache["ntok"].tolist()
    g["padded_bs"] = padded
    t2 = pc()
    r._replay(padded)
    t3 = pc()
    r._wait_for_step()
    t4 = pc()
    tokens = r._host_tokens_np[:bs_].tolist()
    t5 = pc()
    cache["padded"], cache["tokens"], cache["next_ntok"] = padded, tokens, [x + 1 for x in ntok]
    t6 = pc()
    for k, v in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)):
        T[k] += v
    return tokens

r._run_decode_graph = run_graph
for _ in range(8):
    drv.step()
for k in T: T[k] = 0.0
full[0] = 0
torch.cuda.synchronize()
N = 200
t0 = pc()
for _ in range(N):
    drv.step()
torch.cuda.synchronize()
total = (pc() - t0) / N
inside = sum(T.values()) / N
print(f"step {total * 1e6:.1f} us ({full[0]} of {N} steps packed in full): " +
      " | ".join(f"{k} {v / N * 1e6:.1f}" for k, v in T.items()) + f" | outside the runner {(total - inside) * 1e6:.1f}")
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    r._replay(batch)
b.record(); torch.cuda.synchronize()
print(f"graph replay back to back: {a.elapsed_time(b) * 1e3 / 20:.1f} us")
r.exit()
