"""Whole decode path on the GPU: product model + runner + engine against goldens and the CPU oracle."""

import numpy as np
import pytest
import torch

from oracle import layers as L
from oracle import model as OM
from oracle_tokens import check_tokens_against_oracle

pytestmark = pytest.mark.gpu
BS = 16


DT = {"bf16": torch.bfloat16, "fp16": torch.float16}


def _golden_weights(golden, style, dname="bf16"):
    g = golden.npz("model")
    key = f"{style}_{dname}_"
    dt = DT[dname]
    return g, key, {k[len(key) + 2:]: torch.from_numpy(g[k]).to(dt) for k in g.files if k.startswith(key + "w:")}


def make_runner(cfg, weights, quantization=None, num_blocks=12, enforce_eager=True, max_model_len=128, **kw):
    from byox_vllm_amd.config import Config
    from byox_vllm_amd.engine.model_runner import ModelRunner
    from byox_vllm_amd.layers.rotary_embedding import get_rope
    get_rope.cache_clear()
    config = Config("synthetic", hf_config=cfg, state_dict=weights, kvcache_block_size=BS, num_kvcache_blocks=num_blocks,
                    enforce_eager=enforce_eager, max_model_len=max_model_len, max_num_batched_tokens=max(256, max_model_len),
                    quantization=quantization, **kw)
    return ModelRunner(config, 0, None)


def forward(runner, ids, pos, ctx):
    """One forward through the product model with an explicit context (oracle-style metadata)."""
    from byox_vllm_amd.utils.context import reset_context, set_context
    d = lambda x: None if x is None else x.to(runner.device)  # noqa: E731
    set_context(ctx.is_prefill, d(ctx.cu_seqlens_q), d(ctx.cu_seqlens_k), ctx.max_seqlen_q, ctx.max_seqlen_k,
                d(ctx.slot_mapping), d(ctx.context_lens), d(ctx.block_tables))
    with torch.inference_mode():
        logits = runner.model.compute_logits(runner.model(d(ids), d(pos)))
    reset_context()
    return logits.float().cpu()


# Absolute backstop only (bf16 logits of magnitude ~1 carry 2^-8 relative rounding per operation); the real bar is
# relative to the reference's OWN 16-bit error, below.
LOGIT_TOL = 4e-2
ERROR_RATIO = 1.5   # HIP error vs the high-precision truth may be at most this multiple of the reference's own error


def _phases(g, key):
    """The golden run's steps: (name, input ids, lens, block tables, cached) in the order the reference ran them."""
    p0, p1, p2 = (g[key + f"prompt{i}"].tolist() for i in range(3))
    yield "prefill", torch.tensor(p0 + p1), [5, 19], [[3], [7, 1]], None
    lens, tables = [5, 19], [[3], [7, 1]]
    for step in range(3):
        lens = [n + 1 for n in lens]
        for tb, n in zip(tables, lens):
            if len(tb) * BS < n:
                tb.append(9)
        yield f"decode{step}", torch.from_numpy(g[key + f"decode{step}_in"]), list(lens), [list(t) for t in tables], None
    yield "prefix_prefill", torch.tensor(p2[16:]), [22], [[7, 5]], [16]


@pytest.mark.parametrize("style", ["qwen2", "qwen3"])
@pytest.mark.parametrize("dname", ["bf16", "fp16"])
def test_whole_model_vs_reference_goldens(golden, style, dname):
    """Prefill, 3 paged decode steps, prefix-cache prefill against the reference's own 16-bit logits, judged by the
    distance to a high-precision run of the same graph on the same weights (oracle, compute_dtype=float64):
        max|hip - truth| <= 1.5 x max|reference - truth|   and   rms likewise,
    i.e. the HIP path is as close to the exact result as the reference's own bf16/fp16 path is.  (The reference's
    fp32 goldens cannot serve as that truth: in fp32 its add_rms_forward aliases the residual.)"""
    g, key, weights = _golden_weights(golden, style, dname)
    cfg = OM.tiny_config(style, DT[dname])
    runner = make_runner(cfg, weights)
    truth_model = OM.OracleQwen(cfg, weights, BS, 12, compute_dtype=torch.float64)
    report = []
    for name, ids, lens, tables, cached in _phases(g, key):
        if name.startswith("decode"):
            ctx, pos = OM.decode_ctx(lens, tables, BS)
        else:
            ctx, pos = OM.prefill_ctx(lens, tables, BS, cached=cached)
        logits = forward(runner, ids, pos, ctx)
        truth = truth_model.compute_logits(truth_model.forward(ids, pos, ctx), ctx).float()
        ref = torch.from_numpy(g[key + "logits_" + name])
        e_hip, e_ref = (logits - truth).abs(), (ref - truth).abs()
        report.append((name, e_hip.max().item(), e_ref.max().item(), e_hip.pow(2).mean().sqrt().item(),
                       e_ref.pow(2).mean().sqrt().item(), (logits - ref).abs().max().item(), ref.abs().max().item()))
        assert e_hip.max().item() <= ERROR_RATIO * e_ref.max().item(), report[-1]
        assert e_hip.pow(2).mean().sqrt().item() <= ERROR_RATIO * e_ref.pow(2).mean().sqrt().item(), report[-1]
        assert (logits - ref).abs().max().item() < (LOGIT_TOL if dname == "bf16" else LOGIT_TOL / 8), name
        if dname == "fp16":
            # the plain distance between the two 16-bit paths, at the precision of the reference's own CPU-runnable
            # configuration (fp16): within 1e-3 of the reference's logits, i.e. within one fp16 ulp (2^-10) at these
            # magnitudes (measured 0.0005-0.0010); bf16 logits of this size are spaced 0.004-0.008 apart, there the
            # two paths differ by one such step (measured 0.0039-0.0078) and the ratio rule above is the bar
            assert (logits - ref).abs().max().item() <= 1e-3 * max(1.0, ref.abs().max().item()), name
        if name == "prefill":
            assert logits.argmax(-1).tolist() == ref.argmax(-1).tolist()
    for row in report:
        # the last two columns are the plain distance between the two 16-bit paths (HIP vs the reference's own logits)
        # and the size of the logits it is measured on
        print("%s %s %-15s vs fp64 truth: max hip %.4f ref %.4f | rms hip %.5f ref %.5f || max|hip - reference| %.4f "
              "at max|logit| %.2f" % ((style, dname) + row))
    # the paged KV cache holds what the oracle's holds (v = the projection's output: one GEMM away from the
    # normalised embedding, so equal up to one rounding of the model dtype where the summation order matters)
    om = OM.OracleQwen(cfg, weights, BS, 12)
    p0, p1 = g[key + "prompt0"].tolist(), g[key + "prompt1"].tolist()
    ctx, pos = OM.prefill_ctx([5, 19], [[3], [7, 1]], BS)
    om.forward(torch.tensor(p0 + p1), pos, ctx)
    runner2 = make_runner(cfg, weights)
    forward(runner2, torch.tensor(p0 + p1), pos, ctx)
    got_v = runner2.kv_cache[1, 0].cpu()
    blocks = [3, 7, 1]
    mine, theirs = got_v[blocks][:, :3].float(), om.kv[0][1, 0][blocks][:, :3].float()
    assert theirs.abs().max().item() > 0
    ulp = (2.0 ** -7 if dname == "bf16" else 2.0 ** -10) * theirs.abs().clamp_min(1e-3)
    assert ((mine - theirs).abs() <= ulp).all()
    assert (mine == theirs).float().mean().item() > 0.9
    runner.exit()
    runner2.exit()


@pytest.mark.parametrize("quant,bits", [("w8a16", 8), ("w4a16", 4)])
def test_quantised_model_vs_oracle(golden, quant, bits):
    """Weight-only int8/int4 path: oracle = the same model with dequantised dense weights."""
    g, key, weights = _golden_weights(golden, "qwen2")
    cfg = OM.tiny_config("qwen2", torch.bfloat16)
    runner = make_runner(cfg, weights, quantization=quant)
    om = OM.OracleQwen(cfg, OM.fake_quantize(weights, bits), BS, 12)
    ids = torch.tensor(g[key + "prompt0"].tolist() + g[key + "prompt1"].tolist())
    ctx, pos = OM.prefill_ctx([5, 19], [[3], [7, 1]], BS)
    want = om.compute_logits(om.forward(ids, pos, ctx), ctx).float()
    got = forward(runner, ids, pos, ctx)
    assert (got - want).abs().max().item() < LOGIT_TOL
    ctx, pos = OM.decode_ctx([6, 20], [[3], [7, 1]], BS)
    nxt = want.argmax(-1)
    want = om.compute_logits(om.forward(nxt, pos, ctx), ctx).float()
    got = forward(runner, nxt, pos, ctx)
    assert (got - want).abs().max().item() < LOGIT_TOL
    runner.exit()


# (style, prompt seed, tokens per sequence, exact?)  The exact rows have no oracle top-2 gap below the tie tolerance
# (seeds searched on the CPU), so the HIP engine's ids must be the oracle's greedy ids token for token.
GREEDY_CASES = [("qwen2", 3, 12, False), ("qwen2", 147, 12, True), ("qwen3", 3, 12, False), ("qwen3", 377, 3, True)]


@pytest.mark.parametrize("style,seed,max_tokens,exact", GREEDY_CASES)
@pytest.mark.parametrize("eager", [True, False])
def test_engine_greedy_tokens_match_oracle(golden, eager, style, seed, max_tokens, exact):
    """LLM.generate (scheduler + block manager + runner, hipGraph decode when eager=False): every emitted token id
    against the teacher-forced oracle (tests/oracle_tokens.py)."""
    from byox_vllm_amd import LLM, SamplingParams
    from byox_vllm_amd.layers.rotary_embedding import get_rope
    get_rope.cache_clear()
    _, _, weights = _golden_weights(golden, style)
    cfg = OM.tiny_config(style, torch.bfloat16)
    rng = np.random.RandomState(seed)
    prompts = [rng.randint(0, 256, size=n).tolist() for n in (5, 19, 33, 16)]
    llm = LLM("synthetic", hf_config=cfg, state_dict=weights, kvcache_block_size=BS, num_kvcache_blocks=16,
              enforce_eager=eager, max_model_len=128, max_num_batched_tokens=256, max_num_seqs=8)
    try:
        outs = llm.generate(prompts, SamplingParams(temperature=0.0, max_tokens=max_tokens, ignore_eos=True), use_tqdm=False)
    finally:
        llm._exit()
    got = [o["token_ids"] for o in outs]
    assert all(len(x) == max_tokens for x in got)
    check_tokens_against_oracle(cfg, weights, prompts, got, block_size=BS, num_blocks=16, expect_exact=exact,
                                label=f"greedy {style} seed {seed} {'eager' if eager else 'graph'}")


@pytest.mark.parametrize("seed,exact", [(17, False), (20, True)])
@pytest.mark.parametrize("eager", [True, False])
def test_engine_preemption_matches_oracle(golden, eager, seed, exact):
    """Too few KV blocks for the batch: the scheduler preempts and later re-prefills sequences (with prefix-cache
    hits on their own earlier blocks); every token still is the oracle's under the same scheduler (seed 20: a run
    without near-ties, ids identical)."""
    from byox_vllm_amd import LLM, SamplingParams
    from byox_vllm_amd.layers.rotary_embedding import get_rope
    get_rope.cache_clear()
    _, _, weights = _golden_weights(golden, "qwen2")
    cfg = OM.tiny_config("qwen2", torch.bfloat16)
    rng = np.random.RandomState(seed)
    prompts = [rng.randint(0, 256, size=n).tolist() for n in (30, 28, 25)]
    blocks = 9          # 3 sequences x up to 5 blocks of 16 would need 15
    llm = LLM("synthetic", hf_config=cfg, state_dict=weights, kvcache_block_size=BS, num_kvcache_blocks=blocks,
              enforce_eager=eager, max_model_len=128, max_num_batched_tokens=256, max_num_seqs=8)
    try:
        outs = llm.generate(prompts, SamplingParams(temperature=0.0, max_tokens=40, ignore_eos=True), use_tqdm=False)
    finally:
        llm._exit()
    got = [o["token_ids"] for o in outs]
    assert all(len(x) == 40 for x in got)
    check_tokens_against_oracle(cfg, weights, prompts, got, block_size=BS, num_blocks=blocks, expect_exact=exact,
                                label=f"preemption seed {seed} {'eager' if eager else 'graph'}")


@pytest.mark.parametrize("tp", [2, 4, 8])
@pytest.mark.parametrize("collectives", ["gloo", "xgmi-eager", "xgmi-graph"])
def test_engine_tensor_parallel_matches_oracle(golden, tp, collectives, monkeypatch):
    """The whole tensor-parallel runtime on one GPU: worker spawn, shared-memory control messages, sharded weights
    (tp=2: the reference's split; tp=4 on 4/2 heads: KV-head replication, a zero-padded query slot per rank, two
    ranks with an all-zero MLP shard; tp=8: ranks that hold no real query head at all), all-reduce and logits gather.  The ranks share the device, so the process
    group is gloo (NV_DIST_BACKEND; RCCL needs one GPU per rank).  `gloo`: the all-reduces take a host hop and the
    steps run eagerly.  `xgmi-*`: the all-reduces run through nv_allreduce_add_rmsnorm / nv_allreduce on buffers the
    ranks map into each other by hipIpc handles, eagerly and inside every rank's captured decode graph (the path a
    multi-GPU node replays: staging kernel -> graph with in-graph collectives -> LM head and gather outside)."""
    from byox_vllm_amd import LLM, SamplingParams
    from byox_vllm_amd.layers.rotary_embedding import get_rope
    monkeypatch.setenv("NV_DIST_BACKEND", "gloo")
    get_rope.cache_clear()
    _, _, weights = _golden_weights(golden, "qwen2")
    cfg = OM.tiny_config("qwen2", torch.bfloat16)
    rng = np.random.RandomState(11)
    prompts = [rng.randint(0, 256, size=n).tolist() for n in (7, 21, 16)]
    llm = LLM("synthetic", hf_config=cfg, state_dict=weights, kvcache_block_size=BS, num_kvcache_blocks=16,
              enforce_eager=collectives != "xgmi-graph", max_model_len=128, max_num_batched_tokens=256, max_num_seqs=8,
              tensor_parallel_size=tp, xgmi_allreduce=collectives != "gloo")
    try:
        outs = llm.generate(prompts, SamplingParams(temperature=0.0, max_tokens=8, ignore_eos=True), use_tqdm=False)
    finally:
        llm._exit()
    got = [o["token_ids"] for o in outs]
    assert all(len(x) == 8 for x in got)
    # prompt seed 11: no oracle top-2 gap below 0.11, so the ids must be identical
    check_tokens_against_oracle(cfg, weights, prompts, got, block_size=BS, num_blocks=16, expect_exact=True,
                                label=f"tp{tp} {collectives}")


def test_engine_tensor_parallel_step_as_seven_arrays(golden, monkeypatch):
    """Decode steps of a tensor-parallel job travel to the ranks as the packed staging buffer (ModelRunner.run_packed);
    NV_TP_PACKED_STEPS=0 restores the message of seven arrays (run_step, the reference's own shape of it,
    model_runner.py:206-235): same tokens, checked against the oracle like the packed form above."""
    monkeypatch.setenv("NV_TP_PACKED_STEPS", "0")
    test_engine_tensor_parallel_matches_oracle(golden, 2, "xgmi-graph", monkeypatch)


@pytest.mark.parametrize("tp", [4, 8])
def test_engine_tensor_parallel_metric_geometry(tp, monkeypatch):
    """The layer geometry of the metric configuration (Qwen2-0.5B: 14 query / 2 KV heads of 64, hidden 896,
    intermediate 4864 = 38 K-blocks, int4 weights, two layers, a small vocabulary) sharded over 4 and 8 ranks that share
    the GPU: more ranks than KV heads (each KV head on tp/2 ranks with 4+3 / 2+2+2+1 query heads, zero-padded slots),
    o_proj K shards padded to 128, MLP K-blocks dealt 10,10,9,9 / 5,5,5,5,5,5,4,4.  Greedy tokens against the CPU oracle
    on the fake-quantised weights: what bench.py --gpus 4 / 8 runs by default, checked for its arithmetic."""
    from types import SimpleNamespace

    from byox_vllm_amd import LLM, SamplingParams
    from byox_vllm_amd.layers.rotary_embedding import get_rope
    monkeypatch.setenv("NV_DIST_BACKEND", "gloo")
    get_rope.cache_clear()
    cfg = SimpleNamespace(vocab_size=1024, hidden_size=896, intermediate_size=4864, num_hidden_layers=2,
                          num_attention_heads=14, num_key_value_heads=2, head_dim=64, max_position_embeddings=128,
                          rms_norm_eps=1e-6, hidden_act="silu", attention_bias=True, rope_theta=1000000.0,
                          rope_scaling=None, tie_word_embeddings=True, dtype=torch.bfloat16)
    weights = OM.random_weights(cfg, seed=3)
    rng = np.random.RandomState(17)
    prompts = [rng.randint(0, 1024, size=n).tolist() for n in (9, 30, 17)]
    llm = LLM("synthetic", hf_config=cfg, state_dict=weights, quantization="w4a16", kvcache_block_size=BS,
              num_kvcache_blocks=16, enforce_eager=True, max_model_len=128, max_num_batched_tokens=256, max_num_seqs=8,
              tensor_parallel_size=tp)
    try:
        outs = llm.generate(prompts, SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True), use_tqdm=False)
    finally:
        llm._exit()
    got = [o["token_ids"] for o in outs]
    assert all(len(x) == 6 for x in got)
    # prompt seed 17: smallest oracle top-2 gap 0.17, ids identical
    check_tokens_against_oracle(cfg, OM.fake_quantize(weights, 4), prompts, got, block_size=BS, num_blocks=16,
                                expect_exact=True, label=f"metric geometry tp{tp}")


@pytest.mark.parametrize("tp,collectives", [(2, "gloo"), (2, "xgmi-graph"), (4, "xgmi-graph")])
def test_engine_tensor_parallel_under_the_result_store_policy(tp, collectives, monkeypatch):
    """bench.py --gpus N runs the tensor-parallel step at 128 rows, where every rank writes its hand-over results
    through (the result-store policy); the engine tests above run a handful of rows, where the policy is off.  Here
    the window is opened to small steps (NV_RESULT_POLICY_ROWS, read by the worker processes too), so the sharded
    GEMMs' slabs, the fused all-reduce that reads them and the q/k/v and norm rows all take the write-through stores:
    tokens identical to the oracle's."""
    from types import SimpleNamespace

    from byox_vllm_amd import LLM, SamplingParams
    from byox_vllm_amd.layers.rotary_embedding import get_rope
    monkeypatch.setenv("NV_DIST_BACKEND", "gloo")
    monkeypatch.setenv("NV_RESULT_POLICY_ROWS", "1,256")
    get_rope.cache_clear()
    cfg = SimpleNamespace(vocab_size=1024, hidden_size=896, intermediate_size=4864, num_hidden_layers=2,
                          num_attention_heads=14, num_key_value_heads=2, head_dim=64, max_position_embeddings=128,
                          rms_norm_eps=1e-6, hidden_act="silu", attention_bias=True, rope_theta=1000000.0,
                          rope_scaling=None, tie_word_embeddings=True, dtype=torch.bfloat16)
    weights = OM.random_weights(cfg, seed=3)
    rng = np.random.RandomState(17)
    prompts = [rng.randint(0, 1024, size=n).tolist() for n in (9, 30, 17)]
    llm = LLM("synthetic", hf_config=cfg, state_dict=weights, quantization="w4a16", kvcache_block_size=BS,
              num_kvcache_blocks=16, enforce_eager=collectives != "xgmi-graph", max_model_len=128,
              max_num_batched_tokens=256, max_num_seqs=8, tensor_parallel_size=tp, xgmi_allreduce=collectives != "gloo")
    try:
        assert llm.model_runner._decode_policy_for(3) == 1
        outs = llm.generate(prompts, SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True), use_tqdm=False)
    finally:
        llm._exit()
    got = [o["token_ids"] for o in outs]
    check_tokens_against_oracle(cfg, OM.fake_quantize(weights, 4), prompts, got, block_size=BS, num_blocks=16,
                                expect_exact=True, label=f"metric geometry tp{tp} {collectives}, write-through results")


def test_engine_tensor_parallel_qwen2_7b_geometry_on_8_ranks(monkeypatch):
    """BASELINE configuration 5's shards at their own shapes: one Qwen2-7B layer (28 query / 4 KV heads of 128, hidden
    3584, intermediate 18944 = 148 K-blocks dealt 19,19,19,19,18,18,18,18, int4) over 8 ranks sharing the GPU, a small
    vocabulary.  Greedy tokens against the CPU oracle on the fake-quantised weights."""
    from types import SimpleNamespace

    from byox_vllm_amd import LLM, SamplingParams
    from byox_vllm_amd.layers.rotary_embedding import get_rope
    monkeypatch.setenv("NV_DIST_BACKEND", "gloo")
    get_rope.cache_clear()
    cfg = SimpleNamespace(vocab_size=1024, hidden_size=3584, intermediate_size=18944, num_hidden_layers=1,
                          num_attention_heads=28, num_key_value_heads=4, head_dim=128, max_position_embeddings=128,
                          rms_norm_eps=1e-6, hidden_act="silu", attention_bias=True, rope_theta=1000000.0,
                          rope_scaling=None, tie_word_embeddings=False, dtype=torch.bfloat16)
    weights = OM.random_weights(cfg, seed=5)
    rng = np.random.RandomState(23)
    prompts = [rng.randint(0, 1024, size=n).tolist() for n in (11, 25)]
    llm = LLM("synthetic", hf_config=cfg, state_dict=weights, quantization="w4a16", kvcache_block_size=BS,
              num_kvcache_blocks=16, enforce_eager=True, max_model_len=128, max_num_batched_tokens=256, max_num_seqs=8,
              tensor_parallel_size=8)
    try:
        outs = llm.generate(prompts, SamplingParams(temperature=0.0, max_tokens=4, ignore_eos=True), use_tqdm=False)
    finally:
        llm._exit()
    got = [o["token_ids"] for o in outs]
    assert all(len(x) == 4 for x in got)
    # prompt seed 23: smallest oracle top-2 gap 0.125, ids identical
    check_tokens_against_oracle(cfg, OM.fake_quantize(weights, 4), prompts, got, block_size=BS, num_blocks=16,
                                expect_exact=True, label="qwen2-7b geometry tp8")


def test_config5_qwen2_7b_tp8_batch512_captured_graph_with_fused_allreduce(monkeypatch):
    """BASELINE configuration 5 at its own sizes on one GPU: Qwen2-7B layer geometry (two layers, int4), 8 ranks sharing
    the device, **512 sequences** decoding through every rank's captured graph with the peer-memory all-reduce fused
    with add+RMSNorm inside it (M = 512 GEMMs at the shard shapes, nv_allreduce_add_rmsnorm at 512 rows, the
    512-sequence control message through shared memory, KV blocks of 256).  What one GPU cannot show is RCCL and the
    xGMI links themselves.  Rows of a step are independent, so four of the 512 sequences are verified token by token
    against the teacher-forced oracle run on those four alone."""
    from types import SimpleNamespace

    from byox_vllm_amd import LLM, SamplingParams
    from byox_vllm_amd.layers.rotary_embedding import get_rope
    monkeypatch.setenv("NV_DIST_BACKEND", "gloo")
    # 8 ranks on ONE GPU: the all-reduce's workgroups wait for their peers' workgroups, so all of them must be resident
    # together (8 x 32 here; a rank that owns its GPU launches up to 96)
    monkeypatch.setenv("NV_COMM_MAX_BLOCKS", "32")
    get_rope.cache_clear()
    cfg = SimpleNamespace(vocab_size=1024, hidden_size=3584, intermediate_size=18944, num_hidden_layers=2,
                          num_attention_heads=28, num_key_value_heads=4, head_dim=128, max_position_embeddings=512,
                          rms_norm_eps=1e-6, hidden_act="silu", attention_bias=True, rope_theta=1000000.0,
                          rope_scaling=None, tie_word_embeddings=False, dtype=torch.bfloat16)
    weights = OM.random_weights(cfg, seed=5)
    rng = np.random.RandomState(29)
    batch, block, max_tokens = 512, 256, 3
    prompts = [rng.randint(0, 1024, size=int(n)).tolist() for n in rng.randint(16, 33, size=batch)]
    llm = LLM("synthetic", hf_config=cfg, state_dict=weights, quantization="w4a16", kvcache_block_size=block,
              num_kvcache_blocks=batch + 8, enforce_eager=False, graph_batch_sizes=[batch], max_model_len=512,
              max_num_batched_tokens=16384, max_num_seqs=batch, tensor_parallel_size=8, xgmi_allreduce=True)
    try:
        runner = llm.model_runner
        assert runner.comm is not None and not runner.enforce_eager and list(runner.graphs) == [batch]
        replays = []
        real = runner._replay
        runner._replay = lambda bs: (replays.append(bs), real(bs))[1]
        outs = llm.generate(prompts, SamplingParams(temperature=0.0, max_tokens=max_tokens, ignore_eos=True), use_tqdm=False)
    finally:
        llm._exit()
    got = [o["token_ids"] for o in outs]
    assert all(len(x) == max_tokens for x in got)
    assert replays == [batch] * (max_tokens - 1), "the decode steps must have replayed the 512-row graph"
    rows = [0, 137, 300, batch - 1]
    check_tokens_against_oracle(cfg, OM.fake_quantize(weights, 4), [prompts[i] for i in rows], [got[i] for i in rows],
                                block_size=block, num_blocks=8, max_num_batched_tokens=16384,
                                label="config 5 (7B int4 tp8 b512, graph + fused all-reduce), sampled rows")


def test_kv_cache_sized_from_free_memory(golden):
    """num_kvcache_blocks left unset: the runner sizes the cache from the device's memory after its warm-up prefill,
    (total x gpu_memory_utilization - used - peak + current) // block_bytes as nanovllm/engine/model_runner.py:115-150
    does, and decodes through it."""
    from byox_vllm_amd.engine.sequence import Sequence
    from byox_vllm_amd.sampling_params import SamplingParams
    _, _, weights = _golden_weights(golden, "qwen2")
    cfg = OM.tiny_config("qwen2", torch.bfloat16)
    import gc
    torch.cuda.synchronize()                           # blocks freed under work still in flight cannot be returned yet
    gc.collect()                                       # runners of earlier tests that are garbage but not yet collected
    torch.cuda.empty_cache()                           # would otherwise give their memory back inside the constructor
    free, total = torch.cuda.mem_get_info()
    budget = 1 << 30                                   # ask for what is in use now plus 1 GiB
    runner = make_runner(cfg, weights, num_blocks=-1, gpu_memory_utilization=(total - free + budget) / total)
    try:
        block_bytes = 2 * cfg.num_hidden_layers * BS * cfg.num_key_value_heads * cfg.head_dim * 2
        blocks = runner.config.num_kvcache_blocks
        assert runner.kv_cache.shape == (2, cfg.num_hidden_layers, blocks, BS, cfg.num_key_value_heads, cfg.head_dim)
        # the model's own weights and what the warm-up left allocated come out of the 1 GiB; most of it is cache
        # (2 % of slack: memory that earlier tests' libraries hand back in the meantime also counts as free)
        assert 0.5 * budget < blocks * block_bytes <= 1.02 * budget, (blocks, block_bytes)
        rng = np.random.RandomState(2)
        seq = Sequence(rng.randint(0, 256, size=21).tolist(), BS, SamplingParams(temperature=0.0, max_tokens=4))
        seq.block_table = [blocks - 1, 0]              # the last block the sizing produced is usable
        first = runner.run([seq], True)
        seq.append_token(first[0])
        assert len(runner.run([seq], False)) == 1
    finally:
        runner.exit()


def test_graph_replay_equals_eager(golden):
    """A captured decode step (padded batch, staged inputs) returns the same logits as the eager step."""
    _, _, weights = _golden_weights(golden, "qwen2")
    cfg = OM.tiny_config("qwen2", torch.bfloat16)
    from byox_vllm_amd.engine.sequence import Sequence
    from byox_vllm_amd.sampling_params import SamplingParams
    outs = []
    for eager in (True, False):
        runner = make_runner(cfg, weights, enforce_eager=eager, max_num_seqs=8)
        rng = np.random.RandomState(5)
        seqs = []
        for i, n in enumerate((5, 19, 33)):
            s = Sequence(rng.randint(0, 256, size=n).tolist(), BS, SamplingParams(temperature=0.0, max_tokens=4))
            s.block_table = list(range(3 * i, 3 * i + 3))
            seqs.append(s)
        toks = [runner.run(seqs, True)]
        for _ in range(3):
            for s, tkn in zip(seqs, toks[-1]):
                s.append_token(tkn)
            toks.append(runner.run(seqs, False))
        outs.append(toks)
        runner.exit()
    assert outs[0] == outs[1]


@pytest.mark.parametrize("eager", [True, False])
def test_result_store_policy_changes_no_token_and_no_logit(golden, monkeypatch, eager):
    """Decode steps run under the write-through result-store policy; forced off (NV_RESULT_POLICY=0) the same steps give
    the same tokens and bit-identical KV-cache rows."""
    from byox_vllm_amd._C import lib
    from byox_vllm_amd.engine.sequence import Sequence
    from byox_vllm_amd.sampling_params import SamplingParams
    from byox_vllm_amd.engine.model_runner import ModelRunner
    _, _, weights = _golden_weights(golden, "qwen2")
    cfg = OM.tiny_config("qwen2", torch.bfloat16)
    lo, hi = ModelRunner.RESULT_WT_ROWS
    assert lo > 1 and hi < 512          # tiny and very large steps keep plain stores: this test's three rows would too
    monkeypatch.setattr(ModelRunner, "RESULT_WT_ROWS", (1, hi))
    outs = []
    for forced in (None, "0"):
        if forced is None:
            monkeypatch.delenv("NV_RESULT_POLICY", raising=False)
        else:
            monkeypatch.setenv("NV_RESULT_POLICY", forced)
        runner = make_runner(cfg, weights, enforce_eager=eager, max_num_seqs=8)
        assert runner.decode_result_policy == (1 if forced is None else 0)
        assert runner._decode_policy_for(3) == runner.decode_result_policy
        rng = np.random.RandomState(11)
        seqs = []
        for i, n in enumerate((7, 18, 31)):
            s = Sequence(rng.randint(0, 256, size=n).tolist(), BS, SamplingParams(temperature=0.0, max_tokens=4))
            s.block_table = list(range(3 * i, 3 * i + 3))
            seqs.append(s)
        toks = [runner.run(seqs, True)]
        for _ in range(3):
            for s, tkn in zip(seqs, toks[-1]):
                s.append_token(tkn)
            toks.append(runner.run(seqs, False))
        assert lib().nv_get_result_policy() == 0          # the setting does not leak out of a step
        outs.append((toks, runner.kv_cache[:, :, :9].cpu().clone()))
        runner.exit()
    assert outs[0][0] == outs[1][0]
    assert torch.equal(outs[0][1], outs[1][1])


def test_device_side_bookkeeping_equals_host_packing(golden, monkeypatch):
    """Decode steps continued on the device (ids from the previous step's tokens, positions + 1, slots from the device's
    block tables) against the same steps packed on the host every time: identical tokens over block boundaries, a
    batch that shrinks, and a step the host has to re-pack because a table grew."""
    _, _, weights = _golden_weights(golden, "qwen2")
    cfg = OM.tiny_config("qwen2", torch.bfloat16)
    from byox_vllm_amd.engine.block_manager import BlockManager
    from byox_vllm_amd.engine.sequence import Sequence
    from byox_vllm_amd.sampling_params import SamplingParams
    outs, advanced = [], []
    for mode in ("1", "0"):
        monkeypatch.setenv("NV_DEVICE_BOOKKEEPING", mode)
        runner = make_runner(cfg, weights, enforce_eager=False, max_num_seqs=8, num_blocks=32)
        assert runner.device_bookkeeping == (mode == "1")
        count = [0]

        def counted(seqs_, padded, real=runner._continues_previous_step, count=count):
            result = real(seqs_, padded)
            count[0] += result is not None
            return result
        runner._continues_previous_step = counted
        bm = BlockManager(32, BS)
        rng = np.random.RandomState(9)
        seqs = [Sequence(rng.randint(0, 256, size=n).tolist(), BS, SamplingParams(temperature=0.0, max_tokens=64))
                for n in (5, BS - 2, 2 * BS - 1, 33)]
        for s in seqs:
            bm.allocate(s)
        toks = [runner.run(seqs, True)]
        for step in range(2 * BS + 3):
            for s, tkn in zip(seqs, toks[-1]):
                s.append_token(tkn)
                bm.may_append(s)
            if step == BS:                   # one sequence leaves the batch
                bm.deallocate(seqs.pop(1))
                toks[-1] = toks[-1][:1] + toks[-1][2:]
            toks.append(runner.run(seqs, False))
        outs.append(toks)
        advanced.append(count[0])
        runner.exit()
    assert outs[0] == outs[1]
    assert advanced[1] == 0 and advanced[0] >= BS, advanced   # most steps cross no block boundary


def test_steps_launched_ahead_return_the_tokens_of_steps_launched_on_demand(golden, monkeypatch):
    """With NV_RUN_AHEAD (the default) the runner launches a decode step before the caller asks for it whenever no other
    next step is possible.  Sampled tokens (temperature 0.8: the draws depend on the device's step counter), over
    block boundaries, a sequence that reaches max_tokens, a prompt that arrives in mid-run (the step launched ahead is
    dropped by the prefill) and a sequence taken out by the caller without warning (dropped by the next decode
    call): identical to the run that launches every step on demand, and steps were launched ahead and dropped."""
    _, _, weights = _golden_weights(golden, "qwen2")
    cfg = OM.tiny_config("qwen2", torch.bfloat16)
    from byox_vllm_amd.engine.block_manager import BlockManager
    from byox_vllm_amd.engine.sequence import Sequence
    from byox_vllm_amd.sampling_params import SamplingParams
    outs, stats, when = [], [], {}
    for mode in ("1", "0"):
        monkeypatch.setenv("NV_RUN_AHEAD", mode)
        runner = make_runner(cfg, weights, enforce_eager=False, max_num_seqs=8, num_blocks=48)
        assert runner.run_ahead == (mode == "1")
        launches, dropped = [0], [0]

        def counted_replay(padded, real=runner._replay, launches=launches):
            launches[0] += 1
            real(padded)

        def counted_drop(real=runner._drop_step_ahead, runner=runner, dropped=dropped):
            dropped[0] += runner._step_ahead
            real()
        runner._replay, runner._drop_step_ahead = counted_replay, counted_drop
        bm = BlockManager(48, BS)
        rng = np.random.RandomState(21)
        seqs = [Sequence(rng.randint(0, 256, size=n).tolist(), BS, SamplingParams(temperature=0.8, max_tokens=m))
                for n, m in ((5, 200), (BS - 3, 200), (2 * BS - 1, BS + 2), (33, 200))]
        late = Sequence(rng.randint(0, 256, size=11).tolist(), BS, SamplingParams(temperature=0.8, max_tokens=200))
        for s in seqs:
            bm.allocate(s)
        toks = [runner.run(seqs, True)]
        decode_calls = 0
        for step in range(3 * BS):
            for s, tkn in zip(seqs, toks[-1]):
                s.append_token(tkn)
            done = [s for s in seqs if s.num_completion_tokens == s.max_tokens]
            for s in done:                                   # what Scheduler.postprocess does at max_tokens
                bm.deallocate(s)
                seqs.remove(s)
            # both events at a step that has a step launched ahead in flight (found in the first run, repeated in the second)
            if mode == "1" and "arrive" not in when and step >= BS + 5 and runner._step_ahead:
                when["arrive"] = step
            elif mode == "1" and "remove" not in when and step >= 2 * BS + 3 and runner._step_ahead:
                when["remove"] = step
            if step == when.get("arrive"):                   # a prompt arrives: its prefill comes before the next decode
                bm.allocate(late)
                toks.append(runner.run([late], True))
                late.append_token(toks[-1][0])
                seqs.append(late)
            if step == when.get("remove"):                   # the caller drops a sequence the runner could not know about
                bm.deallocate(seqs.pop(0))
            for s in seqs:
                bm.may_append(s)
            toks.append(runner.run(seqs, False))
            decode_calls += 1
        outs.append(toks)
        stats.append((launches[0] - decode_calls, dropped[0]))
        runner.exit()
    assert outs[0] == outs[1]
    assert stats[1] == (0, 0) and set(when) == {"arrive", "remove"}, (stats, when)
    extra, dropped = stats[0]
    assert dropped >= 2 and extra in (dropped, dropped + 1), stats     # the prefill and the removal each dropped one
    print(f"steps launched ahead and dropped: {dropped}, decode launches beyond the calls: {extra}")


@pytest.mark.parametrize("case", ["six requests", "random 80", "random 120"])
def test_generate_with_steps_launched_ahead_equals_generate_on_demand(golden, monkeypatch, case):
    """The whole engine (Scheduler + BlockManager + runner, captured decode graphs) with sampled tokens, sequences that
    end on EOS or at different max_tokens, too few KV blocks for the batch (preemption and re-prefill) and prompts
    that arrive while others decode: NV_RUN_AHEAD on (default) and off produce the same token ids for every request,
    and steps were in fact launched ahead, some of them dropped.  The random cases draw 80 / 120 requests (prompts of
    1-60 tokens, some sharing a prefix of two blocks, 1-60 new tokens, arrival steps spread over the run) for up to 16
    sequences at a time on a block pool that cannot hold them all."""
    from byox_vllm_amd import LLM, SamplingParams
    from byox_vllm_amd.layers.rotary_embedding import get_rope
    _, _, weights = _golden_weights(golden, "qwen2")
    cfg = OM.tiny_config("qwen2", torch.bfloat16)
    if case == "six requests":
        rng = np.random.RandomState(33)
        prompts = [rng.randint(0, 256, size=n).tolist() for n in (9, 30, 17, 28, 5, 12)]
        limits = [40, 25, 60, 33, 48, 20]
        arrive = [0, 0, 0, 6, 19, 30]                      # engine step at which each request is added
        blocks, max_seqs = 14, 8
    else:
        n = int(case.split()[1])
        rng = np.random.RandomState(n)
        shared = rng.randint(0, 256, size=2 * BS).tolist()
        prompts = [(shared if rng.rand() < 0.3 else []) + rng.randint(0, 256, size=rng.randint(1, 61 - 2 * BS)).tolist()
                   for _ in range(n)]
        limits = [int(rng.randint(1, 61)) for _ in range(n)]
        arrive = sorted(int(rng.randint(0, 4 * n)) for _ in range(n))
        blocks, max_seqs = 40, 16

    def run(mode, eos):
        monkeypatch.setenv("NV_RUN_AHEAD", mode)
        get_rope.cache_clear()
        llm = LLM("synthetic", hf_config=cfg, state_dict=weights, kvcache_block_size=BS, num_kvcache_blocks=blocks,
                  enforce_eager=False, max_model_len=128, max_num_batched_tokens=256, max_num_seqs=max_seqs)
        runner = llm.model_runner
        launched, dropped = [0], [0]

        def counted_replay(padded, real=runner._replay):
            launched[0] += 1
            real(padded)

        def counted_drop(real=runner._drop_step_ahead):
            dropped[0] += runner._step_ahead
            real()
        runner._replay, runner._drop_step_ahead = counted_replay, counted_drop
        llm.config.eos = llm.scheduler.eos = eos
        results, order, step, todo = {}, [], 0, list(range(len(prompts)))
        try:
            while todo or not llm._is_finished():
                for i in [i for i in todo if arrive[i] <= step]:
                    llm._add_request(prompts[i], SamplingParams(temperature=0.8, max_tokens=limits[i], ignore_eos=i % 2 == 1))
                    todo.remove(i)
                if not llm._is_finished():
                    finished, _ = llm._step()
                    results.update(dict(finished))
                    order += [k for k, _ in finished]
                step += 1
                assert step < 20000
        finally:
            llm._exit()
        first = min(results)
        return [results[k] for k in sorted(results)], [k - first for k in order], launched[0], dropped[0]

    free_running, _, _, _ = run("0", -1)
    # an EOS id that ends at least one EOS-honouring request in mid-run
    eos = next(t for i in range(0, len(prompts), 2) for t in free_running[i][5:15])
    on_demand, order0, n0, d0 = run("0", eos)
    ahead, order1, n1, d1 = run("1", eos)
    assert d0 == 0 and ahead == on_demand and order1 == order0
    assert any(len(on_demand[i]) < limits[i] for i in range(0, len(prompts), 2))     # the EOS did end somebody early
    assert d1 >= 1 and n1 == n0 + d1, (n0, n1, d0, d1)                    # every extra launch is a dropped step
    print(f"{case}: graph launches on demand {n0}, with steps ahead {n1} ({d1} dropped)")


@pytest.mark.parametrize("quant,bits", [(None, 0), ("w8a16", 8), ("w4a16", 4)])
def test_prefill_of_many_rows_through_the_vendor_library_vs_oracle(golden, quant, bits, monkeypatch):
    """A prefill step of 4 500 tokens: every linear of the step goes through kernels.gemm_many_rows (weight written out
    row-major, hipBLASLt product, SiluAndMul pass for gate/up), the q/k/v projection as an unfused launch.  Last-token
    logits against the oracle (dequantised dense weights), and against the same step with the path switched off."""
    from byox_vllm_amd import kernels
    g, key, weights = _golden_weights(golden, "qwen2")
    cfg = OM.tiny_config("qwen2", torch.bfloat16)
    lens = [900] * 5
    blocks_each = -(-lens[0] // BS)
    tables = [list(range(i * blocks_each, (i + 1) * blocks_each)) for i in range(len(lens))]
    rng = np.random.RandomState(4)
    ids = torch.from_numpy(rng.randint(0, 256, size=sum(lens)))
    ctx, pos = OM.prefill_ctx(lens, tables, BS)
    cfg.max_position_embeddings = 1024          # the fixture's weights do not depend on it; RoPE tables do
    om = OM.OracleQwen(cfg, OM.fake_quantize(weights, bits) if bits else weights, BS, blocks_each * len(lens))
    want = om.compute_logits(om.forward(ids, pos, ctx), ctx).float()
    outs = []
    for rows in (4096, 0):
        monkeypatch.setattr(kernels, "MANY_ROWS", rows)
        calls = []
        real = kernels.gemm_many_rows
        monkeypatch.setattr(kernels, "gemm_many_rows", lambda *a, real=real, calls=calls, **k: (calls.append(1), real(*a, **k))[1])
        runner = make_runner(cfg, weights, quantization=quant, num_blocks=blocks_each * len(lens), max_model_len=1024)
        calls.clear()
        got = forward(runner, ids, pos, ctx)
        runner.exit()
        monkeypatch.setattr(kernels, "gemm_many_rows", real)
        layers = cfg.num_hidden_layers
        assert len(calls) == (4 * layers if rows else 0), (rows, len(calls))   # qkv, o, gate_up, down of every layer
        assert (got - want).abs().max().item() < LOGIT_TOL
        outs.append(got)
    assert (outs[0] - outs[1]).abs().max().item() < LOGIT_TOL
    assert torch.equal(outs[0].argmax(-1), outs[1].argmax(-1)) or (want.topk(2, -1).values.diff(dim=-1).abs().min() < LOGIT_TOL)


def test_engine_long_prompts_through_the_vendor_library_match_oracle(golden):
    """LLM.generate with seven prompts of about a thousand tokens, four of them sharing a 512-token prefix: the first
    prefill step has more than 4096 rows, so its linears run through the vendor-library route, and the prompts that
    arrive later hit the prefix cache.  Greedy tokens against the teacher-forced oracle under the same scheduler."""
    from byox_vllm_amd import LLM, SamplingParams, kernels
    from byox_vllm_amd.layers.rotary_embedding import get_rope
    get_rope.cache_clear()
    _, _, weights = _golden_weights(golden, "qwen2")
    cfg = OM.tiny_config("qwen2", torch.bfloat16)
    cfg.max_position_embeddings = 1100
    rng = np.random.RandomState(12)
    shared = rng.randint(0, 256, size=512).tolist()
    prompts = [rng.randint(0, 256, size=1000).tolist() for _ in range(3)] + \
              [shared + rng.randint(0, 256, size=n).tolist() for n in (470, 500, 430, 488)]
    blocks = 7 * 64 + 8       # the shared prefix is computed once: 3000 + 982 + 500 + 430 + 488 rows in the first step
    calls = []
    real = kernels.gemm_many_rows
    kernels.gemm_many_rows = lambda *a, **k: (calls.append(a[0].size(0)), real(*a, **k))[1]
    try:
        llm = LLM("synthetic", hf_config=cfg, state_dict=weights, kvcache_block_size=BS, num_kvcache_blocks=blocks,
                  enforce_eager=False, max_model_len=1100, max_num_batched_tokens=8192, max_num_seqs=8)
        calls.clear()          # the warm-up step of the constructor took the route too
        try:
            outs = llm.generate(prompts, SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True), use_tqdm=False)
        finally:
            llm._exit()
    finally:
        kernels.gemm_many_rows = real
    assert calls and min(calls) >= kernels.MANY_ROWS, calls
    got = [o["token_ids"] for o in outs]
    check_tokens_against_oracle(cfg, weights, prompts, got, block_size=BS, num_blocks=blocks, max_num_batched_tokens=8192,
                                label="long prompts, vendor-library prefill")
