"""Decode-step benchmark of the paged-KV hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload NAME] [--parallelism dp|tp]

A step = ModelRunner.run(seqs, is_prefill=False): host metadata -> hipGraph
replay of the decode forward -> LM head -> sampler -> token ids on the host,
for one batch of synthetic sequences whose KV cache (random values) is already
resident in HBM.  Rank 0 prints ONE JSON line; `value` is whole-job decode
tokens/s, `roofline` prices the paged-attention kernel against HBM peak and
`cpu_baseline` times the CPU oracle on a bounded sample of the same workload.

For N > 1 launch under torch.distributed.run (one rank per GPU).  The default
there is `--parallelism tp`: the reference's own split (column/row-sharded
linears, RCCL all-reduce over xGMI captured inside the decode graph, logits
gather), the SAME batch spread over N GPUs ("scaling": "strong").  After the
timed tensor-parallel run the same job also times N independent replicas of the
single-GPU step (no data-path collective) and reports that as the extra key
`replicas_tok_s`; `--parallelism dp` makes the replicas the headline instead.
"""

import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak

WORKLOADS = {
    # name: (model, quantization, dtype, batch, context)
    "qwen2-0.5b-w4a16-b128-s1024": ("qwen2-0.5b", "w4a16", "bf16", 128, 1024),
    "qwen2-0.5b-bf16-b32-s1024": ("qwen2-0.5b", None, "bf16", 32, 1024),
    "qwen2-0.5b-w8a16-b64-s1024": ("qwen2-0.5b", "w8a16", "bf16", 64, 1024),
    "qwen2-7b-w4a16-b128-s1024": ("qwen2-7b", "w4a16", "bf16", 128, 1024),
    "qwen2-7b-w4a16-b512-s1024": ("qwen2-7b", "w4a16", "bf16", 512, 1024),
    "qwen2-0.5b-fp16-b1-s128": ("qwen2-0.5b", None, "fp16", 1, 128),
}
# The reference's own benchmark script (bench.py:17-42 of nanovllm): 128 requests, prompt and output lengths drawn
# from U[100, 128], temperature 0.6, ignore_eos, through LLM.generate (prefill and decode mixed by the scheduler).
# name: (model, quantization, dtype)
GENERATE_WORKLOADS = {
    "ref-bench": ("qwen3-0.6b", None, "bf16"),                 # the reference's model (Qwen3-0.6B, bf16)
    "ref-bench-qwen2-0.5b-w4a16": ("qwen2-0.5b", "w4a16", "bf16"),
}
DEFAULT_WORKLOAD = "qwen2-0.5b-w4a16-b128-s1024"
DTYPES = {"bf16": torch.bfloat16, "fp16": torch.float16}


def make_sequences(batch, context, block_size, vocab_cap=10000, temperature=0.6):
    from byox_vllm_amd.engine.sequence import Sequence
    from byox_vllm_amd.sampling_params import SamplingParams
    random.seed(0)
    seqs = []
    next_block = 0
    for _ in range(batch):
        ids = [random.randrange(vocab_cap) for _ in range(context)]
        seq = Sequence(ids, block_size, SamplingParams(temperature=temperature, max_tokens=1 << 30, ignore_eos=True))
        n = seq.num_blocks
        seq.block_table = list(range(next_block, next_block + n))
        next_block += n
        seqs.append(seq)
    return seqs, next_block


class StepDriver:
    """Advances every sequence by one token per step, handing out blocks as the scheduler would."""

    def __init__(self, runner, seqs, next_block, tp=1):
        self.runner, self.seqs, self.next_block, self.tp = runner, seqs, next_block, tp

    def step(self):
        tokens = self.runner.call("run", self.seqs, False)
        if self.tp > 1:   # only TP rank 0 samples: every rank must feed the same ids into the next step
            dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
            t = torch.tensor(tokens if tokens is not None else [0] * len(self.seqs), device=dev)
            dist.broadcast(t, src=0)
            tokens = t.tolist()
        for seq, tok in zip(self.seqs, tokens):
            seq.append_token(tok)
            if seq.num_tokens % seq.block_size == 1:
                seq.block_table.append(self.next_block)
                self.next_block += 1


def kv_bytes_per_launch(seqs, hf, tp):
    """Algorithmic bytes one paged-attention launch (one layer) must read: K and V of every cached token."""
    hkv = max(1, hf.num_key_value_heads // tp)
    return sum(len(s) for s in seqs) * 2 * hkv * hf.head_dim * hf.dtype.itemsize


def step_bytes(runner, seqs, hf, tp):
    """Algorithmic HBM bytes of one whole decode step on this GPU (weights + scales + LM head + KV read/write)."""
    from byox_vllm_amd.layers.linear import LinearBase
    w = sum(m.packed.nbytes for m in runner.model.modules() if isinstance(m, LinearBase))
    head = runner.model.lm_head._streaming_weight().nbytes   # an untied head keeps only its streaming copy
    kv_read = kv_bytes_per_launch(seqs, hf, tp) * hf.num_hidden_layers
    kv_write = len(seqs) * hf.num_hidden_layers * 2 * max(1, hf.num_key_value_heads // tp) * hf.head_dim * hf.dtype.itemsize
    return w + head + kv_read + kv_write


def time_attention_kernel(runner, seqs, reps=10):
    """Average duration of one paged-attention launch, measured with HIP events on the launch stream.

    The launches of one decode step (one per layer, each on its own layer's KV cache, so nothing is
    re-read from the Infinity Cache) are captured into a hipGraph with the step's real block tables and
    context lengths; events bracket `reps` back-to-back replays.  Host launch gaps cannot leak into the
    figure this way, which an event pair around an eagerly launched kernel does not guarantee."""
    from byox_vllm_amd.utils.context import reset_context
    hf = runner.config.hf_config
    runner._prepare_decode(seqs)                     # the step's metadata, on the device
    from byox_vllm_amd.utils.context import get_context
    ctx = get_context()
    layers = [l.self_attn.attn for l in runner.model.model.layers]
    h = layers[0].num_heads
    q = torch.randn(len(seqs), h, hf.head_dim, device=runner.device, dtype=hf.dtype)
    out = torch.empty_like(q)
    runner._set_decode_splits(len(seqs))

    def one_step():
        for attn in layers:
            attn.forward_fused(q, None, None, out=out)

    with torch.inference_mode():
        one_step()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            one_step()
        graph.replay()
        torch.cuda.synchronize()
        rounds = []
        for _ in range(5):   # median of 5 rounds: one slow round (clock ramp, another tenant) must not set the figure
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                graph.replay()
            b.record()
            torch.cuda.synchronize()
            rounds.append(a.elapsed_time(b))
    reset_context()
    launches = reps * len(layers)
    return sorted(rounds)[len(rounds) // 2] * 1e-3 / launches, launches * 5


def cpu_baseline(model_name, quantization, dtype, batch, context, budget_s=20.0):
    """The CPU oracle (a port of the reference's PyTorch path) on a bounded sample: the workload's own batch and
    context, a few steps (a reduced batch would re-read the weights per 8 rows instead of per 128 and understate the
    CPU); only the 7B geometry, whose setup alone takes minutes, is sampled at batch 8."""
    from byox_vllm_amd.config import qwen2_config
    from oracle import model as OM
    hf = qwen2_config(model_name, dtype=dtype)
    sample_batch = batch if model_name != "qwen2-7b" else min(batch, 8)
    bs = 256
    t0 = time.perf_counter()
    weights = OM.random_weights(hf, seed=0)
    if quantization:
        weights = OM.fake_quantize(weights, 4 if quantization == "w4a16" else 8)
    blocks_per_seq = (context + 8) // bs + 1
    om = OM.OracleQwen(hf, weights, bs, sample_batch * blocks_per_seq)
    for kv in om.kv:          # random history: one block drawn, the others copies (the timing does not see values)
        kv[:, :, :1].normal_()
        kv[:, :, 1:] = kv[:, :, :1]
    tables = [list(range(i * blocks_per_seq, (i + 1) * blocks_per_seq)) for i in range(sample_batch)]
    lens = [context] * sample_batch
    ids = torch.randint(0, 10000, (sample_batch,))
    setup = time.perf_counter() - t0
    steps, spent = 0, 0.0
    while steps < 1 or (spent < budget_s / 2 and steps < 4):
        lens = [n + 1 for n in lens]
        ctx, pos = OM.decode_ctx(lens, tables, bs)
        t1 = time.perf_counter()
        logits = om.compute_logits(om.forward(ids, pos, ctx), ctx)
        ids = logits.float().argmax(-1)
        spent += time.perf_counter() - t1
        steps += 1
    return {"value": round(sample_batch * steps / spent, 2), "unit": "tokens/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{steps} decode steps, batch {sample_batch} of {batch}, context {context}, same model/format, "
                      f"eager PyTorch CPU oracle (setup {setup:.0f}s not timed)"}


def time_prefill_attention(hf, lens, reps=20):
    """One causal varlen prefill-attention launch on packed sequences of the given lengths, HIP-event timed inside a
    hipGraph -> (seconds per launch, causal flops per launch = sum_i 2 * len_i^2 * H * D, QK^T and PV halves)."""
    from byox_vllm_amd import kernels
    H, Hkv, D = hf.num_attention_heads, hf.num_key_value_heads, hf.head_dim
    T = sum(lens)
    q = torch.randn(T, H, D, device="cuda", dtype=hf.dtype)
    k = torch.randn(T, Hkv, D, device="cuda", dtype=hf.dtype)
    v = torch.randn(T, Hkv, D, device="cuda", dtype=hf.dtype)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device="cuda")
    out = torch.empty_like(q)
    call = lambda: kernels.attn_prefill(q, k, v, cu, cu, max(lens), D ** -0.5, out=out)  # noqa: E731
    call()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(reps):
            call()
    graph.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    graph.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / reps, float(sum(2 * n * n * H * D for n in lens))


def run_generate_workload(args):
    """`--workload ref-bench*`: whole requests through LLM.generate, as the reference's bench.py times them.
    `value` = requested output tokens / wall seconds of generate() (prefill included), mean over --steps runs."""
    from byox_vllm_amd import LLM, SamplingParams
    from byox_vllm_amd.config import qwen2_config
    assert args.gpus == 1, "the generate workloads run on one GPU"
    model_name, quantization, dtype_name = GENERATE_WORKLOADS[args.workload]
    hf = qwen2_config(model_name, dtype=DTYPES[dtype_name])
    random.seed(0)
    num_seqs, lo, hi = 128, 100, 128
    prompts = [[random.randint(0, 10000) for _ in range(random.randint(lo, hi))] for _ in range(num_seqs)]
    params = [SamplingParams(temperature=0.6, ignore_eos=True, max_tokens=random.randint(lo, hi)) for _ in range(num_seqs)]
    llm = LLM("synthetic", hf_config=hf, quantization=quantization, enforce_eager=args.eager, max_model_len=256,
              max_num_seqs=num_seqs, max_num_batched_tokens=16384, kvcache_block_size=256, num_kvcache_blocks=2 * num_seqs + 8)
    llm.generate([[1, 2, 3]], SamplingParams(max_tokens=4), use_tqdm=False)          # the reference's warm-up request
    runs = []
    for i in range(args.warmup + args.steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        llm.generate(prompts, params, use_tqdm=False)
        torch.cuda.synchronize()
        if i >= args.warmup:
            runs.append((time.perf_counter() - t0, dict(llm.last_stats)))
    llm._exit()
    out_tokens = sum(sp.max_tokens for sp in params)
    wall = sum(r[0] for r in runs) / len(runs)
    agg = {k: sum(r[1][k] for r in runs) for k in runs[0][1]}
    kern_s, flops = time_prefill_attention(hf, [len(p) for p in prompts])
    achieved = flops / kern_s / 1e12
    line = {
        "metric": "generate output tokens/s (prefill + decode, LLM.generate)", "value": round(out_tokens / wall, 1),
        "unit": "tokens/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(wall * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtype_name, "data": "synthetic (random weights, random prompts)",
        "config": {"workload": args.workload, "geometry": model_name, "weight_format": quantization or dtype_name,
                   "requests": num_seqs, "prompt_len": f"U[{lo},{hi}]", "output_len": f"U[{lo},{hi}]", "temperature": 0.6,
                   "step": "one LLM.generate over all requests (reference bench.py:17-42)", "graph": not args.eager},
        "prefill_tok_s": round(agg["prefill_tokens"] / agg["prefill_s"], 1),
        "decode_tok_s": round(agg["decode_tokens"] / agg["decode_s"], 1),
        "prefill_steps": agg["prefill_steps"] // len(runs), "decode_steps": agg["decode_steps"] // len(runs),
        "roofline": {"bound": "mfma", "kernel": "prefill_lds_kernel (causal varlen attention of the prefill step)",
                     "achieved": round(achieved, 1), "peak": 2500.0, "unit": "TFLOP/s",
                     "frac": round(achieved / 2500.0, 4), "traffic": None, "flops_per_launch": flops,
                     "avg_launch_us": round(kern_s * 1e6, 2)},
    }
    print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS) + sorted(GENERATE_WORKLOADS))
    ap.add_argument("--parallelism", default=None, choices=["dp", "tp"],
                    help="N > 1: tp (default) = one model sharded over the ranks; dp = one replica per rank")
    ap.add_argument("--no-replicas", action="store_true", help="tp: skip the extra replicas_tok_s leg")
    ap.add_argument("--allreduce", default="rccl", choices=["rccl", "xgmi"],
                    help="tp: torch.distributed all-reduce (RCCL) or the library's all-reduce fused with add-RMSNorm")
    ap.add_argument("--eager", action="store_true", help="no hipGraph (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-7b", action="store_true", help="also time the CPU oracle for 7B (minutes of setup)")
    args = ap.parse_args()

    if args.workload in GENERATE_WORKLOADS:
        args.steps, args.warmup = min(args.steps, 5), min(args.warmup, 1)
        torch.cuda.set_device(0)
        return run_generate_workload(args)
    if os.environ.get("NV_BENCH_FAULT_TIMEOUT"):   # debugging aid: dump every thread's stack and exit if the run stalls
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["NV_BENCH_FAULT_TIMEOUT"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    # NV_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (ranks share devices)
    backend = os.environ.get("NV_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from byox_vllm_amd.config import Config, qwen2_config
    from byox_vllm_amd.engine.model_runner import ModelRunner

    model_name, quantization, dtype_name, batch, context = WORKLOADS[args.workload]
    hf = qwen2_config(model_name, dtype=DTYPES[dtype_name])
    if args.parallelism is None:
        args.parallelism = "tp" if world > 1 else "dp"
    tp = world if (args.parallelism == "tp" and world > 1) else 1
    if tp > 1:
        assert hf.num_key_value_heads % tp == 0 or tp % hf.num_key_value_heads == 0, (
            f"{model_name}: {hf.num_key_value_heads} KV heads neither divide nor are divided by {tp} ranks; "
            "use --parallelism dp")
    block_size = 256
    total_steps = args.warmup + args.steps + 2
    blocks_per_seq = (context + total_steps) // block_size + 1

    def build(tp_size):
        config = Config("synthetic", hf_config=hf, quantization=quantization, tensor_parallel_size=tp_size,
                        kvcache_block_size=block_size, num_kvcache_blocks=batch * blocks_per_seq + 1,
                        max_model_len=blocks_per_seq * block_size, max_num_seqs=batch,
                        max_num_batched_tokens=max(8192, blocks_per_seq * block_size),
                        enforce_eager=args.eager, graph_batch_sizes=[batch], seed=rank if tp_size == 1 else 0,
                        xgmi_allreduce=tp_size > 1 and args.allreduce == "xgmi")
        try:
            r = ModelRunner(config, rank if tp_size > 1 else 0, None)
        except Exception as e:   # a collective that cannot be captured on this stack must not cost the whole measurement
            if tp_size == 1 or config.enforce_eager:
                raise
            print(f"[bench] rank {rank}: capturing the tensor-parallel decode graph failed ({e!r}); running the step "
                  "eagerly (graph_captured: false)", file=sys.stderr, flush=True)
            config.enforce_eager = True
            r = ModelRunner(config, rank, None)
        r.kv_cache.normal_()   # synthetic KV history: throughput does not depend on the values
        seqs_, next_block = make_sequences(batch, context, block_size)
        return r, seqs_, StepDriver(r, seqs_, next_block, tp_size)

    def timed(driver):
        """W untimed steps, then K steps between barrier + synchronize on both sides; the MAX over ranks."""
        for _ in range(args.warmup):
            driver.step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            driver.step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        return dt

    tp_error = None
    try:
        if tp > 1 and os.environ.get("NV_BENCH_FAIL_TP"):      # test hook: exercises the fallback below
            raise RuntimeError("tensor-parallel leg failed on request (NV_BENCH_FAIL_TP)")
        runner, seqs, driver = build(tp)
        elapsed = timed(driver)
    except Exception as e:
        # A tensor-parallel step that fails on this stack (on every rank alike: a rank that fails alone leaves its
        # peers inside a collective) must not cost the whole measurement: fall back to one replica per rank, say so.
        if tp == 1:
            raise
        tp_error = repr(e)
        print(f"[bench] rank {rank}: the tensor-parallel leg failed ({tp_error}); measuring {world} replicas instead",
              file=sys.stderr, flush=True)
        runner = driver = None
        torch.cuda.empty_cache()
        from byox_vllm_amd.layers.rotary_embedding import get_rope
        get_rope.cache_clear()
        tp = 1
        runner, seqs, driver = build(1)
        elapsed = timed(driver)

    replicas = world // tp
    tokens = batch * args.steps * replicas
    bytes_step = step_bytes(runner, seqs, hf, tp)
    kern_s, launches = time_attention_kernel(runner, seqs)
    kv_bytes = kv_bytes_per_launch(seqs, hf, tp)
    achieved = kv_bytes / kern_s / 1e9
    # HBM bytes per launch from the rocprofv3 PMC passes (FETCH_SIZE x 2 on gfx950, + WRITE_SIZE) kept under
    # profiles/: counters cannot be read from inside this process, so the measured traffic/algorithmic ratio
    # of the same kernel and head geometry is applied to this run's byte count.
    traffic = None
    pmc_file = os.path.join(ROOT, "profiles", "r02_attn_decode_hbm_traffic.json")
    if os.path.exists(pmc_file):
        pmc = json.load(open(pmc_file))
        for entry in [pmc] + pmc.get("others", []):
            if entry["shape"]["D"] == hf.head_dim and entry["shape"]["Hkv"] == max(1, hf.num_key_value_heads // tp):
                traffic = int(kv_bytes * entry["traffic_over_algorithmic"])

    if rank == 0:
        line = {
            "metric": "decode tokens/s",
            "value": round(tokens / elapsed, 1),
            "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak" if tp == 1 else "strong", "vs_baseline": None,
            "dtype": dtype_name, "data": "synthetic (random weights, random KV history, random prompts)",
            "config": {"workload": args.workload, "geometry": model_name, "weight_format": quantization or dtype_name,
                       "batch_per_replica": batch, "context": context, "kv_block_size": block_size,
                       "parallelism": f"{'tp' if tp > 1 else 'dp'}{world}", "graph": not args.eager,
                       "step": "ModelRunner.run decode: host prepare + hipGraph forward + LM head + sampler + D2H"},
            "rccl_ranks": dist.get_world_size() if world > 1 else 1,
            "backend": (("rccl" if backend == "nccl" else backend) if world > 1 else None),
            "graph_captured": not runner.enforce_eager,
            "allreduce": (("nv_allreduce_add_rmsnorm (peer-mapped memory)" if runner.comm is not None else
                           "torch.distributed all_reduce") if tp > 1 else None),
            "tp_error": tp_error,
            "step_hbm_fraction": round(bytes_step / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
            "step_algorithmic_bytes": bytes_step,
            "roofline": {"bound": "hbm", "kernel": "paged_decode_kernel (+ decode_combine_kernel)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "bytes_per_launch": kv_bytes, "avg_launch_us": round(kern_s * 1e6, 2), "launches_timed": launches},
        }
        if world == 1 and not args.no_cpu_baseline and (model_name != "qwen2-7b" or args.cpu_baseline_7b):
            line["cpu_baseline"] = cpu_baseline(model_name, quantization, DTYPES[dtype_name], batch, context)
    runner.exit()
    if tp > 1 and not args.no_replicas:
        # the same job as N independent replicas of the single-GPU step: what the hardware does with this workload
        # when no collective sits on the critical path (every rank takes part: the timing has barriers)
        del runner, driver
        torch.cuda.empty_cache()
        from byox_vllm_amd.layers.rotary_embedding import get_rope
        get_rope.cache_clear()
        runner, seqs, driver = build(1)
        rep_elapsed = timed(driver)
        runner.exit()
        if rank == 0:
            line["replicas_tok_s"] = round(batch * args.steps * world / rep_elapsed, 1)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
