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
acc = dict(prepare=0.0, stage_replay=0.0, head=0.0, sample=0.0, tolist=0.0, total=0.0)
N = 30
for _ in range(N):
    t0 = time.perf_counter()
    ids, pos = r._prepare_decode(seqs)
    temps = r._prepare_sample(seqs)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    ctx = get_context()
    g = r.graph_vars
    g["padded_bs"] = batch
    from byox_vllm_amd import kernels
    with torch.inference_mode():
        kernels.stage_decode_inputs(ids, pos, ctx.slot_mapping, ctx.context_lens, ctx.block_tables.contiguous(), g)
        r.graphs[batch].replay()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        logits = r.model.compute_logits(g["outputs"][:batch])
        torch.cuda.synchronize(); t3 = time.perf_counter()
        toks = r.sampler(logits, temps)
        torch.cuda.synchronize(); t4 = time.perf_counter()
    tl = toks.tolist(); t5 = time.perf_counter()
    reset_context()
    for s, t in zip(seqs, tl):
        s.append_token(t)
        if len(s) % 256 == 1:
            s.block_table.append(drv.next_block); drv.next_block += 1
    acc["prepare"] += t1 - t0; acc["stage_replay"] += t2 - t1; acc["head"] += t3 - t2
    acc["sample"] += t4 - t3; acc["tolist"] += t5 - t4; acc["total"] += t5 - t0
print({k: round(v / N * 1e3, 3) for k, v in acc.items()}, "ms per step")
