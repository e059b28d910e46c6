"""Decode-step benchmark of the paged-KV hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload NAME] [--parallelism dp|tp]

A step = ModelRunner.run(seqs, is_prefill=False): host metadata -> hipGraph
replay of the decode forward -> LM head -> sampler -> token ids on the host,
for one batch of synthetic sequences whose KV cache (random values) is already
resident in HBM.  Rank 0 prints ONE JSON line; `value` is whole-job decode
tokens/s, `roofline` prices the paged-attention kernel against HBM peak and
`cpu_baseline` times the CPU oracle on a bounded sample of the same workload.

For N > 1 the ranks run one per GPU under torch.distributed.run; started plainly
(`python bench.py --gpus N`, no WORLD_SIZE in the environment) this script
starts that launcher itself as a child process, before it touches a GPU, and
passes the child's JSON line and exit code on (the reference spawns its own
ranks too, nanovllm/engine/llm_engine.py:24-32).  The default there is
`--parallelism tp`: the reference's own split (column/row-sharded linears,
RCCL all-reduce over xGMI captured inside the decode graph, logits gather), the
SAME batch spread over N GPUs ("scaling": "strong").  After the timed
tensor-parallel run the same job also times the other all-reduce
implementation (`--allreduce rccl` is the headline, the peer-memory all-reduce
fused with add+RMSNorm is reported as `xgmi_allreduce_tok_s`, or the other way
round) and N independent replicas of the single-GPU step (no data-path
collective, `replicas_tok_s`); `--parallelism dp` makes the replicas the
headline instead.  The replicas are timed FIRST and their line is handed to a
small keeper process: should a tensor-parallel leg then take the ranks down
(a GPU fault aborts the process, no exception to catch) or stall past
NV_BENCH_LEG_DEADLINE seconds, the keeper prints that line, marked as such in
`tp_error`, so a lease of a multi-GPU node always yields a measurement.
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


def cpu_baseline(model_name, quantization, dtype, batch, context, budget_s=20.0, **geometry_overrides):
    """The CPU oracle (a port of the reference's PyTorch path) on a bounded sample of the workload: its own batch and
    context, a few decode steps.  Qwen2-0.5B runs whole.  Qwen2-7B would take minutes just to draw its weights, so a
    TWO-layer model of the same geometry (full-size embedding and LM head) is timed and the step is extrapolated:
    28 x (time of the layer stack / 2) + time of the LM head; every layer does the same work, and the sample says so."""
    from byox_vllm_amd.config import qwen2_config
    from oracle import model as OM
    full = qwen2_config(model_name, dtype=dtype, **geometry_overrides)      # overrides: tests shrink the vocabulary
    sampled_layers = full.num_hidden_layers if model_name != "qwen2-7b" else 2
    hf = qwen2_config(model_name, dtype=dtype, **dict(geometry_overrides, num_hidden_layers=sampled_layers))
    bs = 256
    t0 = time.perf_counter()
    weights = OM.random_weights(hf, seed=0)
    if quantization:
        weights = OM.fake_quantize(weights, 4 if quantization == "w4a16" else 8)
    blocks_per_seq = (context + 8) // bs + 1
    om = OM.OracleQwen(hf, weights, bs, batch * blocks_per_seq)
    for kv in om.kv:          # random history: one block drawn, the others copies (the timing does not see values)
        kv[:, :, :1].normal_()
        kv[:, :, 1:] = kv[:, :, :1]
    tables = [list(range(i * blocks_per_seq, (i + 1) * blocks_per_seq)) for i in range(batch)]
    lens = [context] * batch
    ids = torch.randint(0, min(10000, hf.vocab_size), (batch,))
    setup = time.perf_counter() - t0
    steps, t_layers, t_head = 0, 0.0, 0.0
    while steps < 1 or (t_layers + t_head < budget_s / 2 and steps < 4):
        lens = [n + 1 for n in lens]
        ctx, pos = OM.decode_ctx(lens, tables, bs)
        t1 = time.perf_counter()
        hidden = om.forward(ids, pos, ctx)
        t2 = time.perf_counter()
        logits = om.compute_logits(hidden, ctx)
        ids = logits.float().argmax(-1)
        t_layers += t2 - t1
        t_head += time.perf_counter() - t2
        steps += 1
    spent = t_layers * full.num_hidden_layers / sampled_layers + t_head
    how = "whole model" if sampled_layers == full.num_hidden_layers else (
        f"{sampled_layers} of {full.num_hidden_layers} decoder layers timed with the full embedding and LM head, layer "
        f"time scaled by {full.num_hidden_layers}/{sampled_layers}")
    return {"value": round(batch * steps / spent, 2), "unit": "tokens/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{steps} decode steps, batch {batch}, context {context}, same geometry/format, {how}; "
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


KEEPER = r"""
import os, select, signal, sys
# a launcher that ends a failed job signals the ranks' whole process groups, this process included, possibly while
# the rank that died is still writing its core file and has not closed the pipe yet: stay until the pipe says so
signal.signal(signal.SIGTERM, signal.SIG_IGN); signal.signal(signal.SIGINT, signal.SIG_IGN)
deadline = float(sys.argv[1]); parent = int(sys.argv[2])
line, buf = None, b""
while True:
    ready, _, _ = select.select([0], [], [], deadline if line is not None else None)
    if not ready:                                   # a leg stalled after the provisional line arrived
        try:
            os.kill(parent, signal.SIGKILL)
        except OSError:
            pass
        break
    chunk = os.read(0, 1 << 16)
    if not chunk:                                   # the writer is gone
        break
    buf += chunk
    while b"\n" in buf:
        ln, buf = buf.split(b"\n", 1)
        if ln == b"DONE":
            sys.exit(0)
        line = ln
if line is not None:
    sys.stdout.buffer.write(line + b"\n"); sys.stdout.flush()
"""


class LineKeeper:
    """Rank 0 of a multi-rank job: a child process (started before this process touches the GPU) that holds the last
    complete JSON line it was given and prints it if this process ends without saying DONE, or goes quiet for longer
    than NV_BENCH_LEG_DEADLINE seconds (default 900; the keeper then also ends this process, so that only one line
    is ever printed).  The tensor-parallel legs have never run on more than one device; a fault there aborts the
    process from a runtime thread, where no Python handler gets a turn."""

    def __init__(self):
        import subprocess
        deadline = os.environ.get("NV_BENCH_LEG_DEADLINE", "900")
        self.proc = subprocess.Popen([sys.executable, "-c", KEEPER, deadline, str(os.getpid())], stdin=subprocess.PIPE)

    def hold(self, line: dict):
        self.proc.stdin.write((json.dumps(line) + "\n").encode())
        self.proc.stdin.flush()

    def done(self):
        self.proc.stdin.write(b"DONE\n")
        self.proc.stdin.flush()
        self.proc.stdin.close()
        self.proc.wait()


def self_launch(gpus: int, argv: list[str]) -> int:
    """`python bench.py --gpus N` started plainly: run the N ranks under torch.distributed.run as a CHILD process (this
    process has not touched a GPU and never will), pass the child's output through and return its exit code.  The
    rendezvous port is one the kernel just handed out, so two jobs on one node do not meet."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    return subprocess.run(cmd, env=env).returncode


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
                    help="tp: which all-reduce the headline leg uses: torch.distributed (RCCL) or the library's own over "
                         "peer-mapped memory, fused with add-RMSNorm; the other one is timed too and reported beside it")
    ap.add_argument("--single-leg", action="store_true", help="tp: time only the --allreduce leg")
    ap.add_argument("--eager", action="store_true", help="no hipGraph (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
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
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    keeper = LineKeeper() if (world > 1 and rank == 0) else None      # started before this process touches a GPU
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
    coll_dev = "cuda" if backend == "nccl" else "cpu"      # where host-side control collectives keep their tensors

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

    def any_rank(flag: bool) -> bool:
        """True on every rank if ``flag`` is set on any: the ranks decide TOGETHER whether a leg failed."""
        if world == 1:
            return flag
        t = torch.tensor([1 if flag else 0], device=coll_dev, dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return bool(t.item())

    def construct(config, tp_size):
        """ModelRunner(config) on every rank, or None on every rank (and the first error text) if any rank failed.
        A constructor that raises has released what it had built (ModelRunner.__init__'s own exception path)."""
        runner, error = None, None
        try:
            runner = ModelRunner(config, rank if tp_size > 1 else 0, None)
        except Exception as e:
            error = repr(e)
        if any_rank(error is not None):
            if runner is not None:
                runner.exit()
            return None, error or "another rank failed to build its runner"
        return runner, None

    def build(tp_size, allreduce="rccl"):
        def config(eager):
            return Config("synthetic", hf_config=hf, quantization=quantization, tensor_parallel_size=tp_size,
                          kvcache_block_size=block_size, num_kvcache_blocks=batch * blocks_per_seq + 1,
                          max_model_len=blocks_per_seq * block_size, max_num_seqs=batch,
                          max_num_batched_tokens=max(8192, blocks_per_seq * block_size),
                          enforce_eager=eager, graph_batch_sizes=[batch], seed=rank if tp_size == 1 else 0,
                          xgmi_allreduce=tp_size > 1 and allreduce == "xgmi")
        r, error = construct(config(args.eager), tp_size)
        if r is None and tp_size > 1 and not args.eager:
            # a collective that cannot be captured on this stack must not cost the whole measurement: every rank
            # agreed that the capture failed (construct), every rank retries eagerly
            print(f"[bench] rank {rank}: building the captured tensor-parallel step failed ({error}); running the "
                  "step eagerly (graph_captured: false)", file=sys.stderr, flush=True)
            from byox_vllm_amd.layers.rotary_embedding import get_rope
            get_rope.cache_clear()
            torch.cuda.empty_cache()
            r, error = construct(config(True), tp_size)
        if r is None:
            raise RuntimeError(error)
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
            t = torch.tensor([dt], device=coll_dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        return dt

    def leg(tp_size, allreduce="rccl"):
        """Build + time one configuration; a failure is agreed on by all ranks -> (None, ..., error text)."""
        runner = seqs = None
        error = elapsed = None
        try:
            if tp_size > 1 and os.environ.get("NV_BENCH_FAIL_TP"):      # test hook: exercises the fallback below
                raise RuntimeError("tensor-parallel leg failed on request (NV_BENCH_FAIL_TP)")
            if tp_size > 1 and os.environ.get("NV_BENCH_ABORT_TP"):     # test hook: the process dies as under a GPU fault
                os.abort()
            if tp_size > 1 and os.environ.get("NV_BENCH_STALL_TP"):     # test hook: a leg that never comes back
                time.sleep(1e6)
            runner, seqs, driver = build(tp_size, allreduce)
            elapsed = timed(driver)
        except Exception as e:
            error = repr(e)
        if any_rank(error is not None):
            if runner is not None:
                try:
                    runner.exit()
                except Exception:
                    pass
            runner = None
            torch.cuda.empty_cache()
            from byox_vllm_amd.layers.rotary_embedding import get_rope
            get_rope.cache_clear()
            return None, None, None, error or "failed on another rank"
        return runner, seqs, elapsed, None

    def release(runner):
        runner.exit()
        torch.cuda.empty_cache()
        from byox_vllm_amd.layers.rotary_embedding import get_rope
        get_rope.cache_clear()

    def make_line(runner, seqs, elapsed, tp_now, tp_error):
        """The JSON line of one timed leg (a dict on rank 0, None elsewhere); every rank times the attention launch."""
        replicas = world // tp_now
        tokens = batch * args.steps * replicas
        bytes_step = step_bytes(runner, seqs, hf, tp_now)
        kern_s, launches = time_attention_kernel(runner, seqs)
        kv_bytes = kv_bytes_per_launch(seqs, hf, tp_now)
        achieved = kv_bytes / kern_s / 1e9
        traffic, traffic_source = pmc_traffic(kv_bytes, hf.head_dim, max(1, hf.num_key_value_heads // tp_now))
        graph_captured = not runner.enforce_eager
        uses_comm = runner.comm is not None
        if rank != 0:
            return None
        return {
            "metric": "decode tokens/s",
            "value": round(tokens / elapsed, 1),
            "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak" if tp_now == 1 else "strong", "vs_baseline": None,
            "dtype": dtype_name, "data": "synthetic (random weights, random KV history, random prompts)",
            "config": {"workload": args.workload, "geometry": model_name, "weight_format": quantization or dtype_name,
                       "batch_per_replica": batch, "context": context, "kv_block_size": block_size,
                       "parallelism": f"{'tp' if tp_now > 1 else 'dp'}{world}", "graph": graph_captured,
                       # single-rank graph steps: step N+1 is launched as soon as step N's tokens are on the host,
                       # before they are returned (K graph executions fall into the K timed steps either way)
                       "run_ahead": bool(runner.run_ahead and graph_captured and tp_now == 1 and runner.graph_has_head),
                       "step": "ModelRunner.run decode: host prepare + hipGraph forward + LM head + sampler + D2H"},
            "ranks": world,
            # ranks that talk through RCCL: 0 when the job's process group is a rehearsal backend or there is one rank
            "rccl_ranks": world if (world > 1 and backend == "nccl") else 0,
            "backend": (("rccl" if backend == "nccl" else backend) if world > 1 else None),
            "graph_captured": graph_captured,
            "allreduce": (("nv_allreduce_add_rmsnorm (peer-mapped memory)" if uses_comm else
                           "torch.distributed all_reduce") if tp_now > 1 else None),
            "tp_error": tp_error,
            "step_hbm_fraction": round(bytes_step / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
            "step_algorithmic_bytes": bytes_step,
            "roofline": {"bound": "hbm", "kernel": "paged_decode_kernel (+ decode_combine_kernel)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "bytes_per_launch": kv_bytes, "avg_launch_us": round(kern_s * 1e6, 2), "launches_timed": launches},
        }

    # N independent replicas of the single-GPU step, no collective on the data path: what the hardware does with this
    # workload when nothing is exchanged.  Under tensor parallelism it is timed first: it is the leg that cannot fail
    # in a new way on a multi-GPU node, and the keeper process holds its line from here on.
    replicas_line, have_replicas = None, False
    if tp > 1 and not args.no_replicas:
        r0, s0, e0, err0 = leg(1)
        have_replicas = r0 is not None        # the ranks agree on this (leg)
        if have_replicas:
            replicas_line = make_line(r0, s0, e0, 1, "the job ended or stalled during the tensor-parallel legs; this is "
                                                    "the line of the replicas leg, printed by the keeper process")
            release(r0)
            if keeper is not None:
                keeper.hold(replicas_line)
        del r0

    tp_error = None
    runner, seqs, elapsed, error = leg(tp, args.allreduce)
    if runner is None:
        # A tensor-parallel step that fails on this stack must not cost the whole measurement: fall back to one
        # replica per rank, say so.
        if tp == 1:
            raise RuntimeError(error)
        tp_error = error
        print(f"[bench] rank {rank}: the tensor-parallel leg failed ({tp_error}); reporting {world} replicas instead",
              file=sys.stderr, flush=True)
        tp = 1
        if not have_replicas:
            runner, seqs, elapsed, error = leg(1)
            if runner is None:
                raise RuntimeError(error)

    if runner is not None:
        line = make_line(runner, seqs, elapsed, tp, tp_error)
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(model_name, quantization, DTYPES[dtype_name], batch, context)
        release(runner)
    elif rank == 0:
        # the replicas were timed above: theirs is the measurement, with the tensor-parallel leg's error in it
        line = dict(replicas_line, tp_error=tp_error)
    del runner
    if tp > 1 and not args.single_leg:
        # the same tensor-parallel job with the other all-reduce, so that one lease of a multi-GPU node compares them
        other = "xgmi" if args.allreduce == "rccl" else "rccl"
        key = "xgmi_allreduce_tok_s" if other == "xgmi" else "rccl_allreduce_tok_s"
        if keeper is not None:   # from here on a crash costs only the comparison
            keeper.hold(dict(line, **{key.replace("tok_s", "error"): "the job ended or stalled during this leg"}))
        r2, _, e2, err2 = leg(tp, other)
        if rank == 0:
            line[key] = round(batch * args.steps / e2, 1) if r2 is not None else None
            if r2 is not None:
                line[key.replace("tok_s", "graph_captured")] = not r2.enforce_eager
            else:
                line[key.replace("tok_s", "error")] = err2
        if r2 is not None:
            release(r2)
        del r2
    if rank == 0:
        if tp > 1 and not args.no_replicas:
            line["replicas_tok_s"] = replicas_line["value"] if replicas_line is not None else None
        print(json.dumps(line), flush=True)
        if keeper is not None:
            keeper.done()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(kv_bytes, head_dim, num_kv_heads):
    """HBM bytes per paged-attention launch from the rocprofv3 PMC passes kept under profiles/ (FETCH_SIZE x 2 on
    gfx950, + WRITE_SIZE): counters cannot be read from inside this process, so the measured traffic / algorithmic
    ratio of the same kernel build and head geometry is applied to this run's byte count.  The file names the
    kernel revision it was collected on; a library whose decode kernel has changed since gets no figure
    -> (bytes or None, where the figure comes from)."""
    from byox_vllm_amd._C import lib
    tag = lib().nv_paged_attn_decode_kernel_tag().decode()
    for name in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("attn_decode_hbm_traffic.json")),
                       reverse=True):
        pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
        if pmc.get("kernel_tag") != tag:
            continue
        for entry in [pmc] + pmc.get("others", []):
            if entry["shape"]["D"] == head_dim and entry["shape"]["Hkv"] == num_kv_heads:
                return int(kv_bytes * entry["traffic_over_algorithmic"]), (
                    f"profiles/{name}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE ratio {entry['traffic_over_algorithmic']} "
                    f"of kernel revision {tag}, applied to this run's bytes")
    return None, f"no PMC collection under profiles/ for kernel revision {tag} at D={head_dim}, Hkv={num_kv_heads}"


if __name__ == "__main__":
    main()
