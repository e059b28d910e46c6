"""Decode step time with the result-store policy off / on over batch sizes (Qwen2-0.5B int4, context 1024), to place the
lower bound of rows from which the write-through stores pay."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from byox_vllm_amd.config import Config, qwen2_config
from byox_vllm_amd.engine.model_runner import ModelRunner
from byox_vllm_amd.layers.rotary_embedding import get_rope

quant = sys.argv[1] if len(sys.argv) > 1 else "w4a16"
hf = qwen2_config("qwen2-0.5b", dtype=torch.bfloat16)
context, bs = 1024, 256
per = context // bs + 3
for batch in (1, 2, 4, 8, 16, 32, 64):
    res = {}
    for rnd in range(2):
        for pol in ("0", "1"):
            os.environ["NV_RESULT_POLICY"] = pol
            get_rope.cache_clear()
            cfg = Config("synthetic", hf_config=hf, quantization=quant or None, kvcache_block_size=bs,
                         num_kvcache_blocks=batch * per + 1, max_model_len=per * bs, max_num_seqs=batch,
                         max_num_batched_tokens=8192, graph_batch_sizes=[batch], seed=0)
            r = ModelRunner(cfg, 0, None)
            r.kv_cache.normal_()
            seqs, nb = bench.make_sequences(batch, context, bs)
            drv = bench.StepDriver(r, seqs, nb)
            for _ in range(8):
                drv.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                drv.step()
            torch.cuda.synchronize()
            res.setdefault(pol, []).append((time.perf_counter() - t0) / 200 * 1e3)
            r.exit()
            torch.cuda.empty_cache()
    print(f"batch {batch:3d} {quant or 'bf16'}: policy off {min(res['0']):.4f} ms  on {min(res['1']):.4f} ms  ({(min(res['1']) / min(res['0']) - 1) * 100:+.2f} %)", flush=True)
