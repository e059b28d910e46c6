"""What ONE rank of a tensor-parallel job computes per decode step, measured on one GPU with no peers: the runner and
its model are built exactly as rank 0 of `tp` ranks builds them (sharded weights, KV heads of this rank only, graph
without head, LM head shard + sampler outside), and every collective is replaced by nothing (a sum over one
contribution, a gather of one part).  The tokens are meaningless; the time is the rank's kernels and host work, i.e.
the step time of a tensor-parallel job whose collectives cost nothing: an upper bound on its scaling.
Usage: python scripts/r03_tp_one_rank.py WORKLOAD TP [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import bench
import byox_vllm_amd.layers.linear as LIN
import byox_vllm_amd.layers.embed_head as EH
from byox_vllm_amd.config import Config, qwen2_config
from byox_vllm_amd.engine.model_runner import ModelRunner

workload, tp = sys.argv[1], int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 64
model_name, quantization, dtype_name, batch, context = bench.WORKLOADS[workload]
hf = qwen2_config(model_name, dtype=bench.DTYPES[dtype_name])
dist.init_process_group("gloo", "tcp://127.0.0.1:29617", world_size=1, rank=0)   # the runner's barriers need a group
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
for _ in range(8):
    driver.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    driver.step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
g = runner.graphs[batch]
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    g.replay()
torch.cuda.synchronize()
gt = (time.perf_counter() - t0) / steps
print(f"{workload} rank 0 of tp={tp}: step {dt * 1e3:.3f} ms (graph of the layers alone, replayed back to back: {gt * 1e3:.3f} ms); "
      f"KV heads on this rank {runner.kv_cache.shape[-2]}", flush=True)
runner.exit()
