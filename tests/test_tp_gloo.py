"""Tensor-parallel wiring on 2 CPU ranks (gloo): sharding, bias-on-rank-0, all-reduce, vocab gather, control message.

The HIP kernels need a GPU, so inside these worker processes the two kernel
wrappers the layers call (gemm, embed) are replaced by the CPU oracle's
arithmetic; everything else (weight loaders, which rank adds the bias, the
collectives, the gather order) is the product code under test.
"""

import os
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _cpu_kernels():
    import torch.nn.functional as F

    from byox_vllm_amd import kernels
    from oracle import layers as L

    def gemm(x, w, bias=None, out=None, **kw):
        assert w.fmt in (0, 3)   # unquantised, row-major or streaming layout
        return F.linear(x, w.logical_dense(), bias)

    def embed(ids, weight, vocab_start=0, out=None):
        ok = (ids >= vocab_start) & (ids < vocab_start + weight.size(0))
        return ok.unsqueeze(1) * F.embedding(ok * (ids - vocab_start), weight)

    kernels.gemm, kernels.embed = gemm, embed
    return L


def _worker(rank, world, store, result):
    dist.init_process_group("gloo", init_method=f"file://{store}", world_size=world, rank=rank)
    _cpu_kernels()
    from byox_vllm_amd.layers.embed_head import ParallelLMHead, VocabParallelEmbedding
    from byox_vllm_amd.layers.linear import (MergedColumnParallelLinear, QKVParallelLinear, RowParallelLinear, set_tp)
    from byox_vllm_amd.utils.context import reset_context, set_context
    set_tp(rank, world)
    g = torch.Generator().manual_seed(0)
    h, nh, nkv, d, inter, vocab, T = 32, 4, 2, 8, 48, 64, 5
    x = torch.randn(T, h, generator=g)
    wq, wk, wv = (torch.randn(n * d, h, generator=g) for n in (nh, nkv, nkv))
    bq = torch.randn((nh + 2 * nkv) * d, generator=g)
    wo, bo = torch.randn(h, nh * d, generator=g), torch.randn(h, generator=g)
    wg, wu = torch.randn(inter, h, generator=g), torch.randn(inter, h, generator=g)
    emb = torch.randn(vocab, h, generator=g)
    ids = torch.tensor([0, 31, 32, 63, 40])

    qkv = QKVParallelLinear(h, d, nh, nkv, bias=True)
    for w, sid in ((wq, "q"), (wk, "k"), (wv, "v")):
        qkv.weight_loader(qkv.weight, w, sid)
    for b, sid in zip(bq.split([nh * d, nkv * d, nkv * d]), ("q", "k", "v")):
        qkv.weight_loader(qkv.bias, b, sid)
    y_qkv = qkv(x)                                   # column parallel: this rank's q||k||v slice, bias on every rank

    row = RowParallelLinear(nh * d, h, bias=True)
    row.weight_loader(row.weight, wo)
    row.weight_loader(row.bias, bo)
    attn_out = torch.randn(T, nh * d, generator=g)
    shard = attn_out[:, rank * (nh * d // world):(rank + 1) * (nh * d // world)].contiguous()
    y_row = row(shard)                               # all-reduced: identical on both ranks, bias added once

    mg = MergedColumnParallelLinear(h, [inter, inter])
    mg.weight_loader(mg.weight, wg, 0)
    mg.weight_loader(mg.weight, wu, 1)
    mg.finalize(None)
    y_mg = mg(x)

    ve = VocabParallelEmbedding(vocab, h)
    ve.weight_loader(ve.weight, emb)
    y_emb = ve(ids)                                  # masked lookup + all-reduce

    head = ParallelLMHead(vocab, h)
    head.weight_loader(head.weight, emb)
    set_context(False)
    logits = head(x)                                 # gather to rank 0, concatenated in rank order
    reset_context()

    if rank == 0:
        ref_qkv = torch.nn.functional.linear(x, torch.cat((wq, wk, wv)), bq)
        q_ref, k_ref, v_ref = ref_qkv.split([nh * d, nkv * d, nkv * d], dim=-1)
        want_qkv = torch.cat([t.chunk(world, -1)[0] for t in (q_ref, k_ref, v_ref)], -1)
        ok = {
            "qkv": torch.allclose(y_qkv, want_qkv, atol=1e-5),
            "row": torch.allclose(y_row, torch.nn.functional.linear(attn_out, wo, bo), atol=1e-4),
            "merged": torch.allclose(y_mg, torch.cat([(x @ wg.t()).chunk(world, -1)[0], (x @ wu.t()).chunk(world, -1)[0]], -1), atol=1e-5),
            "embed": torch.allclose(y_emb, emb[ids], atol=1e-6),
            "logits": logits is not None and torch.allclose(logits, x @ emb.t(), atol=1e-4),
        }
        torch.save(ok, result)
    else:
        assert logits is None
    dist.barrier()
    dist.destroy_process_group()


def test_tp2_layers_over_gloo():
    world = 2
    with tempfile.TemporaryDirectory() as tmp:
        store, result = os.path.join(tmp, "store"), os.path.join(tmp, "result.pt")
        mp.spawn(_worker, args=(world, store, result), nprocs=world, join=True)
        ok = torch.load(result, weights_only=True)
    assert all(ok.values()), ok


@pytest.mark.parametrize("tp", [1, 2, 4])
def test_weight_loaders_match_reference_sharding(golden, tp):
    """The product's weight loaders under an explicit TP layout reproduce the reference's shards bit for bit."""
    from byox_vllm_amd.layers import linear as LIN
    from byox_vllm_amd.layers.embed_head import VocabParallelEmbedding
    g = golden.npz("sharding")
    h, nh, nkv, d, inter, vocab = 32, 8, 4, 8, 48, 64
    t = lambda k: torch.from_numpy(g[k])  # noqa: E731
    try:
        for rank in range(tp):
            LIN.set_tp(rank, tp)
            qkv = LIN.QKVParallelLinear(h, d, nh, nkv, bias=True)
            for key, sid in (("wq", "q"), ("wk", "k"), ("wv", "v")):
                qkv.weight_loader(qkv.weight, t(key), sid)
            qkv.weight_loader(qkv.bias, t("bq"), "q")
            mg = LIN.MergedColumnParallelLinear(h, [inter, inter])
            mg.weight_loader(mg.weight, t("wgate"), 0)
            mg.weight_loader(mg.weight, t("wup"), 1)
            row = LIN.RowParallelLinear(nh * d, h)
            row.weight_loader(row.weight, t("wo"))
            ve = VocabParallelEmbedding(vocab, h)
            ve.weight_loader(ve.weight, t("emb"))
            key = f"tp{tp}_r{rank}_"
            assert torch.equal(qkv.weight.data, t(key + "qkv"))
            assert torch.equal(qkv.bias.data[: nh * d // tp], t(key + "qkv_bias_q"))
            assert torch.equal(mg.weight.data, t(key + "merged"))
            assert torch.equal(row.weight.data, t(key + "row"))
            assert torch.equal(ve.weight.data, t(key + "emb"))
    finally:
        LIN._TP_STATE = None


def test_control_message_is_compact_and_complete():
    """Rank 0 ships prepared integer arrays, not Sequence objects: small at batch 512 and sufficient to rebuild the step."""
    import pickle

    from byox_vllm_amd.engine.model_runner import SHM_BYTES, ModelRunner
    from byox_vllm_amd.engine.sequence import Sequence
    from byox_vllm_amd.sampling_params import SamplingParams
    runner = ModelRunner.__new__(ModelRunner)
    runner.block_size, runner.device = 256, torch.device("cpu")
    rng = np.random.RandomState(0)
    seqs = []
    for i in range(512):
        s = Sequence(rng.randint(0, 10000, size=1024 + i % 7).tolist(), 256, SamplingParams(temperature=0.6))
        s.block_table = list(range(5 * i, 5 * i + s.num_blocks))
        seqs.append(s)
    step = runner._prepare_step(seqs, False)
    blob = pickle.dumps(["run_step", step])
    assert len(blob) < 64 * 1024 < SHM_BYTES, len(blob)          # pickled Sequences would be > 4 MB here
    assert step["input_ids"].tolist() == [s.last_token for s in seqs]
    assert step["context_lens"].tolist() == [len(s) for s in seqs]
    assert step["slot_mapping"].tolist() == [s.block_table[-1] * 256 + s.last_block_num_tokens - 1 for s in seqs]
    assert step["block_tables"].shape == (512, 5) and step["temperatures"].dtype == np.float32
    pre = runner._prepare_step(seqs[:3], True)
    assert pre["cu_seqlens_q"].tolist() == [0, 1024, 2049, 3075] and pre["block_tables"] is None


def _attention_block(x, wqkv, bqkv, wo, num_slots, num_kv, d):
    """fp64 attention block on one rank's shard: slot s attends with local KV head s // (slots / kv heads)."""
    qkv = x @ wqkv.t() + bqkv
    q, k, v = qkv.split([num_slots * d, num_kv * d, num_kv * d], dim=-1)
    T = x.size(0)
    q, k, v = q.view(T, num_slots, d), k.view(T, num_kv, d), v.view(T, num_kv, d)
    group = num_slots // num_kv
    outs = []
    for s in range(num_slots):
        p = torch.softmax(q[:, s] @ k[:, s // group].t() * d ** -0.5, dim=-1)
        outs.append(p @ v[:, s // group])
    return torch.cat(outs, -1) @ wo.t()


@pytest.mark.parametrize("H,Hkv,D,tp", [(14, 2, 64, 1), (14, 2, 64, 2), (14, 2, 64, 4), (14, 2, 64, 8),
                                         (28, 4, 128, 2), (28, 4, 128, 4), (28, 4, 128, 8), (8, 2, 64, 4),
                                         (16, 8, 128, 8)])
def test_head_layout_sum_over_ranks_equals_unsharded(H, Hkv, D, tp):
    """KV-head replication and zero-padded query slots (more ranks than KV heads, K shards of o_proj kept at
    multiples of 128) change nothing: the row-parallel sum over all ranks equals the unsharded attention block."""
    from byox_vllm_amd.layers import linear as LIN
    g = torch.Generator().manual_seed(H * 131 + tp)
    hidden, T = 64, 6
    x = torch.randn(T, hidden, generator=g, dtype=torch.float64)
    wq, wk, wv = (torch.randn(n * D, hidden, generator=g, dtype=torch.float64) * 0.2 for n in (H, Hkv, Hkv))
    bq, bk, bv = (torch.randn(n * D, generator=g, dtype=torch.float64) for n in (H, Hkv, Hkv))
    wo = torch.randn(hidden, H * D, generator=g, dtype=torch.float64) * 0.1
    want = _attention_block(x, torch.cat((wq, wk, wv)), torch.cat((bq, bk, bv)), wo, H, Hkv, D)
    total, seen_q = torch.zeros_like(want), []
    old_dtype = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        for rank in range(tp):
            LIN.set_tp(rank, tp)
            qkv = LIN.QKVParallelLinear(hidden, D, H, Hkv, bias=True)
            lay = qkv.layout
            for w, b, sid in ((wq, bq, "q"), (wk, bk, "k"), (wv, bv, "v")):
                qkv.weight_loader(qkv.weight, w, sid)
                qkv.weight_loader(qkv.bias, b, sid)
            o = LIN.RowParallelLinear(H * D, hidden, input_segments=lay.segments(lay.q_heads),
                                      local_input_size=lay.num_heads * D)
            o.weight_loader(o.weight, wo)
            assert (lay.num_heads * D) % 128 == 0 and lay.num_heads % lay.num_kv_heads == 0 and lay.group <= 16
            if Hkv % tp == 0 and (H // tp * D) % 128 == 0:      # the reference's even split, bit for bit
                assert lay.q_heads == list(range(rank * H // tp, (rank + 1) * H // tp))
                assert torch.equal(qkv.weight.data[: lay.num_heads * D], wq.chunk(tp, 0)[rank])
            seen_q += [h for h in lay.q_heads if h >= 0]
            total += _attention_block(x, qkv.weight.data, qkv.bias.data, o.weight.data, lay.num_heads, lay.num_kv_heads, D)
    finally:
        LIN._TP_STATE = None
        torch.set_default_dtype(old_dtype)
    assert sorted(seen_q) == list(range(H))          # every query head lives on exactly one rank
    assert torch.allclose(total, want, rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("inter,tp", [(4864, 2), (4864, 4), (4864, 8), (18944, 8), (18944, 4), (18944, 2),
                                      (768, 4), (640, 3)])
def test_mlp_uneven_shards_sum_over_ranks_equals_unsharded(inter, tp):
    """Intermediate columns are dealt in whole 128-column K-blocks; uneven shares are zero-padded (18944 = 148 blocks
    on 8 ranks -> 19,19,19,19,18,18,18,18)."""
    from byox_vllm_amd.layers import linear as LIN
    g = torch.Generator().manual_seed(inter + tp)
    hidden, T = 32, 4
    x = torch.randn(T, hidden, generator=g, dtype=torch.float64)
    wg, wu = (torch.randn(inter, hidden, generator=g, dtype=torch.float64) * 0.3 for _ in range(2))
    wd = torch.randn(hidden, inter, generator=g, dtype=torch.float64) * 0.1
    silu = torch.nn.functional.silu
    want = (silu(x @ wg.t()) * (x @ wu.t())) @ wd.t()
    total, width = torch.zeros_like(want), set()
    old_dtype = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        for rank in range(tp):
            LIN.set_tp(rank, tp)
            gu = LIN.MergedColumnParallelLinear(hidden, [inter, inter], align=128)
            gu.weight_loader(gu.weight, wg, 0)
            gu.weight_loader(gu.weight, wu, 1)
            down = LIN.RowParallelLinear(inter, hidden, align=128)
            down.weight_loader(down.weight, wd)
            gate, up = (x @ gu.weight.data.t()).chunk(2, -1)
            assert gate.size(1) == down.weight.size(1) and gate.size(1) % 128 == 0
            width.add(gate.size(1))
            total += (silu(gate) * up) @ down.weight.data.t()
    finally:
        LIN._TP_STATE = None
        torch.set_default_dtype(old_dtype)
    assert len(width) == 1                            # same shapes on every rank
    assert torch.allclose(total, want, rtol=1e-10, atol=1e-10)
