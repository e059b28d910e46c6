"""Would splitting the decode batch into micro-batches replayed concurrently on separate streams pay?
Measurement only: k independent ModelRunner instances (own weights, own KV cache, own graph) of batch B/k each are
replayed together, every graph on a stream of its own, against one runner of batch B.  Steps replay the same staged
step again and again (the sequences do not advance): only the time is of interest.
    python scripts/bench_microbatch.py [workload] [k ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from byox_vllm_amd import kernels
from byox_vllm_amd.config import Config, qwen2_config
from byox_vllm_amd.engine.model_runner import ModelRunner
from byox_vllm_amd.layers.rotary_embedding import get_rope


def build(hf, quantization, batch, context, block_size=256):
    blocks_per_seq = (context + 64) // block_size + 1
    config = Config("synthetic", hf_config=hf, quantization=quantization, kvcache_block_size=block_size,
                    num_kvcache_blocks=batch * blocks_per_seq + 1, max_model_len=blocks_per_seq * block_size,
                    max_num_seqs=batch, max_num_batched_tokens=max(8192, blocks_per_seq * block_size),
                    graph_batch_sizes=[batch], seed=0)
    r = ModelRunner(config, 0, None)
    r.kv_cache.normal_()
    seqs, nb = bench.make_sequences(batch, context, block_size)
    d = bench.StepDriver(r, seqs, nb)
    for _ in range(3):
        d.step()          # leaves a real step staged in the runner's pinned buffer
    # every runner gets scratch buffers of its own: the captured graphs of different runners run concurrently
    kernels._retired.extend(kernels._workspaces.values())
    kernels._workspaces.clear()
    return r


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else bench.DEFAULT_WORKLOAD
    ks = [int(a) for a in sys.argv[2:]] or [1, 2, 4]
    model_name, quantization, dtype_name, batch, context = bench.WORKLOADS[workload]
    hf = qwen2_config(model_name, dtype=bench.DTYPES[dtype_name])
    for k in ks:
        get_rope.cache_clear()
        runners = [build(hf, quantization, batch // k, context) for _ in range(k)]
        pool = [torch.cuda.Stream() for _ in range(8)]
        # HIP maps streams onto a few hardware queues (4 per process by default): two streams on one queue do not
        # overlap.  NV_MB_STREAMS=i,j,... picks the streams of the pool to use.
        pick = [int(x) for x in os.environ.get("NV_MB_STREAMS", ",".join(map(str, range(k)))).split(",")][:k]
        streams = [pool[i] for i in pick]
        torch.cuda.synchronize()

        def step():
            for r, s in zip(runners, streams):
                with torch.cuda.stream(s):
                    r._replay(batch // k)
            for s in streams:
                s.synchronize()
        for _ in range(5):
            step()
        t0 = time.perf_counter()
        n = 50
        for _ in range(n):
            step()
        dt = (time.perf_counter() - t0) / n
        # the same graphs one after the other on ONE stream: what the split costs without any overlap
        def serial():
            for r in runners:
                r._replay(batch // k)
            torch.cuda.current_stream().synchronize()
        for _ in range(3):
            serial()
        t0 = time.perf_counter()
        for _ in range(n):
            serial()
        ds = (time.perf_counter() - t0) / n
        print(f"{workload}: streams {pick} {k} micro-batch(es) of {batch // k}: concurrent {dt * 1e6:8.1f} us/step "
              f"({batch / dt:9.0f} tok/s) | one stream {ds * 1e6:8.1f} us/step", flush=True)
        for r in runners:
            r.exit()
        del runners
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
