"""Where rank 0 of a tensor-parallel job spends the part of its step that lies outside the captured graph (collectives
replaced by nothing, see r03_tp_one_rank.py): every piece bracketed by a device synchronize, so the pieces add up to more
than the unsynchronised step.  Usage: python scripts/r03_tp_one_rank_breakdown.py WORKLOAD TP"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import bench
import byox_vllm_amd.layers.linear as LIN
import byox_vllm_amd.layers.embed_head as EH
from byox_vllm_amd.config import Config, qwen2_config
from byox_vllm_amd.engine.model_runner import ModelRunner

workload, tp = sys.argv[1], int(sys.argv[2])
steps = 40
model_name, quantization, dtype_name, batch, context = bench.WORKLOADS[workload]
hf = qwen2_config(model_name, dtype=bench.DTYPES[dtype_name])
dist.init_process_group("gloo", "tcp://127.0.0.1:29618", world_size=1, rank=0)
LIN.tp_all_reduce = EH.tp_all_reduce = lambda t: None
EH.tp_gather = lambda t, rank, size: [t]
block_size = 256
blocks_per_seq = (context + steps + 80) // block_size + 1
cfg = Config("synthetic", hf_config=hf, quantization=quantization, tensor_parallel_size=tp, kvcache_block_size=block_size,
             num_kvcache_blocks=batch * blocks_per_seq + 1, max_model_len=blocks_per_seq * block_size, max_num_seqs=batch,
             max_num_batched_tokens=max(8192, blocks_per_seq * block_size), enforce_eager=False, graph_batch_sizes=[batch])
cfg.shm_name = f"nv_one_rank_{os.getpid()}"
runner = ModelRunner(cfg, 0, [])
runner.kv_cache.normal_()
seqs, next_block = bench.make_sequences(batch, context, block_size)
driver = bench.StepDriver(runner, seqs, next_block, 1)
spent = collections.defaultdict(float)

def timed(obj, name, label=None):
    real = getattr(obj, name)
    def wrapper(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = real(*a, **k)
        torch.cuda.synchronize(); spent[label or name] += time.perf_counter() - t0
        return out
    setattr(obj, name, wrapper)

for _ in range(8):
    driver.step()
timed(runner, "_prepare_step"); timed(runner, "_pack_decode"); timed(runner, "_write_shm"); timed(runner, "_replay", "graph replay")
timed(runner.model, "compute_logits", "LM-head shard + gather"); timed(runner, "sampler", "sampler launch")
import byox_vllm_amd.kernels as K
timed(K, "stage_decode_inputs", "stage kernel (seven arrays)")
timed(K, "stage_decode_packed", "stage kernel (packed)")
timed(runner, "run_step", "run_step (whole)")
timed(runner, "run_packed", "run_packed (whole)")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    driver.step()
torch.cuda.synchronize(); total = (time.perf_counter() - t0) / steps
print(f"{workload} rank 0 of tp={tp}: synchronised step {total * 1e3:.3f} ms")
for k, v in spent.items():
    print(f"  {k:50s} {v / steps * 1e6:8.1f} us")
runner.exit()
