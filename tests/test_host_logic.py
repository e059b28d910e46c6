"""Integer/host side of the hot path against the reference's own known-answer tests.

The expected values are the ones asserted in the reference's tests
(tests/test_prepare_funcs.py, tests/test_cache.py, tests/test_scheduler.py),
re-stated against this package's classes.
"""

from collections import deque
from types import SimpleNamespace
from unittest.mock import Mock

import torch

from byox_vllm_amd.engine.block_manager import BlockManager
from byox_vllm_amd.engine.model_runner import ModelRunner, graph_batch_sizes
from byox_vllm_amd.engine.scheduler import Scheduler
from byox_vllm_amd.engine.sequence import Sequence, SequenceStatus
from byox_vllm_amd.utils.context import get_context, reset_context


def make_runner():
    runner = ModelRunner.__new__(ModelRunner)   # as the reference's tests do: no __init__, only block_size
    runner.block_size = 16
    runner.device = torch.device("cpu")
    runner._prepare_block_tables = Mock(return_value=torch.tensor([[0, -1, -1, -1]]))
    return runner


def eq(t, values):
    return t is not None and t.cpu().tolist() == values


def test_prepare_prefill_single_and_multiple():
    runner = make_runner()
    seq = Sequence([1, 2, 3, 4, 5], block_size=16)
    seq.block_table = [0]
    ids, pos = runner._prepare_prefill([seq])
    c = get_context()
    assert c.is_prefill and eq(c.cu_seqlens_q, [0, 5]) and eq(c.cu_seqlens_k, [0, 5])
    assert (c.max_seqlen_q, c.max_seqlen_k) == (5, 5) and eq(c.slot_mapping, [0, 1, 2, 3, 4])
    assert c.context_lens is None and c.block_tables is None
    assert eq(ids, [1, 2, 3, 4, 5]) and eq(pos, [0, 1, 2, 3, 4])
    assert ids.dtype == torch.int64 and c.slot_mapping.dtype == torch.int32 and c.cu_seqlens_q.dtype == torch.int32
    reset_context()
    s0, s1 = Sequence([1, 2, 3], 16), Sequence([4, 5, 6, 7, 8], 16)
    s0.block_table, s1.block_table = [0], [1]
    ids, pos = runner._prepare_prefill([s0, s1])
    c = get_context()
    assert eq(c.cu_seqlens_q, [0, 3, 8]) and eq(c.cu_seqlens_k, [0, 3, 8]) and c.max_seqlen_q == 5
    assert eq(c.slot_mapping, [0, 1, 2, 16, 17, 18, 19, 20]) and c.block_tables is None
    assert eq(ids, [1, 2, 3, 4, 5, 6, 7, 8]) and eq(pos, [0, 1, 2, 0, 1, 2, 3, 4])
    reset_context()


def test_prepare_prefill_with_cached_tokens():
    runner = make_runner()
    seq = Sequence([1, 2, 3, 4, 5, 6], 16)
    seq.num_cached_tokens, seq.block_table = 3, [0, 1]
    ids, pos = runner._prepare_prefill([seq])
    c = get_context()
    assert eq(c.cu_seqlens_q, [0, 3]) and eq(c.cu_seqlens_k, [0, 6]) and (c.max_seqlen_q, c.max_seqlen_k) == (3, 6)
    assert eq(c.slot_mapping, [0, 1, 2, 3, 4, 5]) and eq(c.block_tables, [[0, -1, -1, -1]])
    assert eq(ids, [4, 5, 6]) and eq(pos, [3, 4, 5])
    reset_context()
    s0, s1 = Sequence([1, 2, 3, 4, 5], 16), Sequence([6, 7, 8, 9, 10, 11], 16)
    s0.num_cached_tokens, s1.num_cached_tokens = 2, 1
    s0.block_table, s1.block_table = [0], [1]
    ids, pos = runner._prepare_prefill([s0, s1])
    c = get_context()
    assert eq(c.cu_seqlens_q, [0, 3, 8]) and eq(c.cu_seqlens_k, [0, 5, 11]) and (c.max_seqlen_q, c.max_seqlen_k) == (5, 6)
    assert eq(c.slot_mapping, [0, 1, 2, 3, 4, 16, 17, 18, 19, 20, 21])
    assert eq(ids, [3, 4, 5, 7, 8, 9, 10, 11]) and eq(pos, [2, 3, 4, 1, 2, 3, 4, 5])
    reset_context()


def test_prepare_decode_cases():
    runner = make_runner()
    seq = Sequence([1, 2, 3, 4, 5], 16)
    seq.status, seq.block_table = SequenceStatus.RUNNING, [0, -1, -1, -1]
    runner._prepare_block_tables = Mock(return_value=torch.tensor([[0, -1, -1, -1]], dtype=torch.int32))
    ids, pos = runner._prepare_decode([seq])
    c = get_context()
    assert not c.is_prefill and c.cu_seqlens_q is None and c.cu_seqlens_k is None and c.max_seqlen_q == 0
    assert eq(c.slot_mapping, [-12]) and eq(c.context_lens, [5]) and eq(c.block_tables, [[0, -1, -1, -1]])
    assert eq(ids, [5]) and eq(pos, [4])
    reset_context()
    s0, s1 = Sequence([1, 2, 3], 16), Sequence([4, 5, 6, 7], 16)
    s0.block_table, s1.block_table = [0, -1, -1, -1], [1, -1, -1, -1]
    ids, pos = runner._prepare_decode([s0, s1])
    c = get_context()
    assert eq(c.slot_mapping, [-14, -13]) and eq(c.context_lens, [3, 4]) and eq(ids, [3, 7]) and eq(pos, [2, 3])
    reset_context()
    s0, s1 = Sequence([1, 2, 3, 4], 16), Sequence([5, 6, 7, 8, 9], 16)
    s0.block_table, s1.block_table = [0, -1, -1, -1], [1, -1, -1, -1]
    s0.num_cached_tokens, s1.num_cached_tokens = 2, 3
    ids, pos = runner._prepare_decode([s0, s1])
    c = get_context()
    assert eq(c.slot_mapping, [-13, -12]) and eq(c.context_lens, [4, 5]) and eq(ids, [4, 9]) and eq(pos, [3, 4])
    reset_context()


def test_prepare_block_tables_pads_with_minus_one():
    runner = ModelRunner.__new__(ModelRunner)
    runner.block_size, runner.device = 16, torch.device("cpu")
    s0, s1 = Sequence([1] * 40, 16), Sequence([2] * 5, 16)
    s0.block_table, s1.block_table = [4, 9, 2], [7]
    bt = runner._prepare_block_tables([s0, s1])
    assert bt.dtype == torch.int32 and bt.tolist() == [[4, 9, 2], [7, -1, -1]]
    # real decode slots: last block id * block_size + tokens in last block - 1 (model_runner.py:215)
    runner._prepare_decode([s0, s1])
    assert get_context().slot_mapping.tolist() == [2 * 16 + 7, 7 * 16 + 4]
    reset_context()


def test_block_manager_prefix_cache_trace():
    m = BlockManager(10, 16)
    common = [i for i in range(3) for _ in range(16)]
    seq0 = Sequence(common + [3] * 4, 16)
    m.allocate(seq0)
    assert seq0.block_table == [0, 1, 2, 3] and seq0.num_cached_tokens == 0
    assert m.used_block_ids == {0, 1, 2, 3} and m.free_block_ids == deque([4, 5, 6, 7, 8, 9])
    parent = -1
    for b in (0, 1, 2):
        parent = m._compute_hash(tuple(common[b * 16:(b + 1) * 16]), parent)
        assert m.blocks[b].hash == parent and m.blocks[b].ref_count == 1
    assert m.blocks[3].hash == -1
    seq1 = Sequence(common + [3] * 5, 16)
    m.allocate(seq1)                                  # hit on blocks still in use
    assert seq1.block_table == [0, 1, 2, 4] and seq1.num_cached_tokens == 48 and seq1.num_cached_blocks == 3
    assert [m.blocks[b].ref_count for b in (0, 1, 2)] == [2, 2, 2] and m.free_block_ids == deque([5, 6, 7, 8, 9])
    m.deallocate(seq0)
    assert m.free_block_ids == deque([5, 6, 7, 8, 9, 3])
    m.deallocate(seq1)
    assert m.free_block_ids == deque([5, 6, 7, 8, 9, 3, 4, 2, 1, 0])
    seq2 = Sequence(common + [3] * 6, 16)
    m.allocate(seq2)                                  # hit on freed blocks: revived
    assert seq2.block_table == [0, 1, 2, 5] and seq2.num_cached_tokens == 48
    assert m.used_block_ids == {0, 1, 2, 5} and m.free_block_ids == deque([6, 7, 8, 9, 3, 4])
    m.deallocate(seq2)
    assert m.free_block_ids == deque([6, 7, 8, 9, 3, 4, 5, 2, 1, 0])
    seq3 = Sequence([99] * 160, 16)
    m.allocate(seq3)                                  # miss everywhere: eviction in free-list order
    assert seq3.block_table == [6, 7, 8, 9, 3, 4, 5, 2, 1, 0] and m.free_block_ids == deque([])
    m.deallocate(seq3)
    assert m.free_block_ids == deque([0, 1, 2, 5, 4, 3, 9, 8, 7, 6])


def test_block_manager_stale_hash_mapping_and_append():
    m = BlockManager(10, 16)
    seq0 = Sequence([i for i in range(8) for _ in range(16)], 16)
    m.allocate(seq0)
    m.deallocate(seq0)
    assert len(m.hash_to_block_id) == 8
    seq1 = Sequence([i for i in range(8, 16) for _ in range(16)], 16)
    m.allocate(seq1)
    assert len(m.hash_to_block_id) == 16
    # decode appends: token 129 opens a block, token 144 seals it with a chained hash
    seq1.append_token(5)
    assert m.can_append(seq1)
    m.may_append(seq1)
    assert len(seq1.block_table) == 9
    for _ in range(15):
        seq1.append_token(5)
        m.may_append(seq1)
    last = m.blocks[seq1.block_table[-1]]
    assert last.hash == m._compute_hash([5] * 16, m.blocks[seq1.block_table[-2]].hash)


def _sched(num_blocks=10, block_size=256, **kw):
    cfg = dict(max_num_seqs=512, max_num_batched_tokens=16384, eos=-1, num_kvcache_blocks=num_blocks,
               kvcache_block_size=block_size)
    cfg.update(kw)
    return Scheduler(SimpleNamespace(**cfg))


def test_scheduler_waits_for_blocks():
    s = _sched()
    seq0 = Sequence([i for i in range(3) for _ in range(256)] + [3] * 4, 256)
    s.add(seq0)
    assert s.schedule() == ([seq0], True)
    seq1 = Sequence([i for i in range(7) for _ in range(256)], 256)
    s.add(seq1)
    assert s.schedule() == ([seq0], False)            # seq1 cannot be allocated: seq0 keeps decoding


def test_scheduler_rotates_on_token_budget():
    s = _sched(max_num_batched_tokens=1000)
    seq0 = Sequence([i for i in range(7) for _ in range(256)], 256)
    s.add(seq0)
    assert s.schedule() == ([], False)
    seq1 = Sequence([i for i in range(3) for _ in range(256)], 256)
    seq2 = Sequence([3] * 256, 256)
    s.add(seq1)
    s.add(seq2)
    assert s.waiting == deque([seq0, seq1, seq2])
    assert s.schedule() == ([], False)
    assert s.waiting == deque([seq1, seq2, seq0])
    seqs, is_prefill = s.schedule()
    seq1.append_token(100)
    assert seqs == [seq1] and is_prefill and s.waiting == deque([seq0, seq2])
    assert s.schedule() == ([seq1], False)
    assert s.waiting == deque([seq2, seq0])


def test_scheduler_preempts_itself_when_pool_is_empty():
    s = _sched()
    seq0 = Sequence([i for i in range(7) for _ in range(256)], 256)
    s.add(seq0)
    s.schedule()
    seq0.append_token(100)
    assert s.running == deque([seq0]) and s.block_manager.free_block_ids == deque([7, 8, 9])
    seq1 = Sequence([i for i in range(7, 9) for _ in range(256)], 256)
    s.add(seq1)
    s.schedule()
    seq1.append_token(101)
    assert s.running == deque([seq0, seq1]) and s.block_manager.free_block_ids == deque([9])
    s.schedule()
    assert s.running == deque([seq0]) and s.waiting == deque([seq1])
    assert s.block_manager.free_block_ids == deque([8, 7])
    assert seq0.block_table == [0, 1, 2, 3, 4, 5, 6, 9] and seq1.block_table == []


def test_scheduler_postprocess_eos_and_max_tokens():
    from byox_vllm_amd.sampling_params import SamplingParams
    s = _sched(block_size=16, eos=7)
    a = Sequence([1, 2, 3], 16, SamplingParams(max_tokens=2))
    b = Sequence([4, 5], 16, SamplingParams(max_tokens=8, ignore_eos=True))
    s.add(a)
    s.add(b)
    seqs, _ = s.schedule()
    s.postprocess(seqs, [7, 7])                       # a stops on eos, b ignores it
    assert a.is_finished and not b.is_finished and a.block_table == [] and s.running == deque([b])


def test_graph_batch_sizes_follow_reference():
    assert graph_batch_sizes(512)[:6] == [1, 2, 4, 8, 16, 32] and graph_batch_sizes(512)[-1] == 512
    assert graph_batch_sizes(8) == [1, 2, 4, 8]


def test_sampling_params_and_config_extensions():
    import pytest
    from byox_vllm_amd import SamplingParams
    from byox_vllm_amd.config import Config, qwen2_config
    assert SamplingParams().temperature == 1.0 and SamplingParams(temperature=0.0).temperature == 0.0
    with pytest.raises(AssertionError):
        SamplingParams(temperature=-1.0)
    cfg = Config("synthetic", hf_config=qwen2_config("qwen2-0.5b"), kvcache_block_size=16, max_model_len=999999, max_num_batched_tokens=32768)
    assert cfg.max_model_len == 32768 and cfg.hf_config.head_dim == 64
    with pytest.raises(AssertionError):
        Config("/nonexistent-model-dir")
    with pytest.raises(AssertionError):
        Config("synthetic", hf_config=qwen2_config("qwen2-7b"), tensor_parallel_size=9)


def test_pack_decode_matches_prepare_functions_and_tracks_table_changes():
    """The packed step buffer of the graph path holds exactly what _prepare_decode / _prepare_block_tables /
    _prepare_sample produce, also after block tables change under the table cache."""
    import numpy as np
    from byox_vllm_amd.sampling_params import SamplingParams

    runner = ModelRunner.__new__(ModelRunner)
    runner.block_size, runner.device = 16, torch.device("cpu")
    bm = BlockManager(64, 16)
    seqs = [Sequence(list(range(100, 100 + n)), 16, SamplingParams(temperature=0.5 + 0.1 * i))
            for i, n in enumerate((5, 16, 17, 40, 32))]
    for s in seqs:
        bm.allocate(s)
    host = np.zeros(4096, dtype=np.int32)

    def check():
        n, cols = runner._pack_decode(seqs, host)
        bs = len(seqs)
        ids, pos = runner._prepare_decode(seqs)
        c = get_context()
        assert n == 5 * bs + bs * cols and cols == c.block_tables.size(1)
        assert host[0:bs].tolist() == ids.tolist() and host[bs:2 * bs].tolist() == pos.tolist()
        assert host[2 * bs:3 * bs].tolist() == c.slot_mapping.tolist()
        assert host[3 * bs:4 * bs].tolist() == c.context_lens.tolist()
        assert host[4 * bs:5 * bs].view(np.float32).tolist() == runner._prepare_sample(seqs).tolist()
        assert host[5 * bs:n].reshape(bs, cols).tolist() == c.block_tables.tolist()
        reset_context()

    check()
    check()                                  # second call is served from the table cache
    for s in seqs:                           # one decode step: tokens appended, tables may grow
        s.append_token(7)
        bm.may_append(s)
    check()
    old = list(seqs[3].block_table)          # preemption: same table length afterwards, different block ids
    bm.deallocate(seqs[3])
    filler = Sequence([1] * 16, 16)
    bm.allocate(filler)
    bm.allocate(seqs[3])
    assert len(seqs[3].block_table) == len(old) and seqs[3].block_table != old
    check()
    seqs[0].block_table = list(seqs[0].block_table)   # replaced list object, equal contents
    check()
    seqs.pop()                               # batch composition changes
    check()
    seqs.reverse()                           # same size, other rows: per-row temperatures must follow
    check()


def test_continues_previous_step_decision():
    """When may a decode step be derived on the device from the previous one (NV_STAGE_ADVANCE)?  Only if the same
    Sequence objects come back in the same order, each longer by exactly the token returned for it, with no block
    table touched and the same captured graph; the host logic that decides it, without a GPU."""
    import numpy as np
    from byox_vllm_amd.sampling_params import SamplingParams

    runner = ModelRunner.__new__(ModelRunner)
    runner.block_size, runner.device, runner.device_bookkeeping = 16, torch.device("cpu"), True
    bm = BlockManager(64, 16)
    seqs = [Sequence(list(range(100, 100 + n)), 16, SamplingParams(temperature=0.7)) for n in (5, 14, 20)]
    for s in seqs:
        bm.allocate(s)
    host = np.zeros(1024, dtype=np.int32)

    def full_step(tokens, padded=4):       # what _run_decode_graph records after a step packed on the host
        runner._pack_decode(seqs, host)
        c = runner._pack_cache
        c["padded"], c["tokens"], c["next_ntok"] = padded, list(tokens), [n + 1 for n in c["ntok"].tolist()]

    def advance(tokens):                    # ... and after a step continued on the device
        c = runner._pack_cache
        c["tokens"], c["next_ntok"] = list(tokens), [n + 1 for n in c["next_ntok"]]

    def scheduler_appends(tokens):
        for s, t in zip(seqs, tokens):
            s.append_token(t)
            bm.may_append(s)

    assert runner._continues_previous_step(seqs, 4) is None            # nothing ran yet
    full_step([7, 8, 9])
    assert runner._continues_previous_step(seqs, 4) is None            # the sequences did not move
    scheduler_appends([7, 8, 9])
    assert runner._continues_previous_step(seqs, 4) == [6, 15, 21]
    assert runner._continues_previous_step(seqs, 8) is None            # another captured graph
    advance([1, 2, 3])
    scheduler_appends([1, 2, 3])                                       # the second sequence reaches 16 tokens: no new block yet
    assert runner._continues_previous_step(seqs, 4) == [7, 16, 22]
    advance([4, 5, 6])
    scheduler_appends([4, 5, 6])                                       # 17 tokens: may_append allocated a block
    assert runner._continues_previous_step(seqs, 4) is None
    full_step([1, 1, 1])                                               # re-packed on the host; it returned 1, 1, 1
    scheduler_appends([1, 1, 1])
    assert runner._continues_previous_step(seqs, 4) is not None
    seqs[0].token_ids[-1] = 99                                         # not the token the runner returned
    seqs[0].last_token = 99
    assert runner._continues_previous_step(seqs, 4) is None
    seqs[0].token_ids[-1] = seqs[0].last_token = 1
    assert runner._continues_previous_step(seqs, 4) is not None
    assert runner._continues_previous_step(list(reversed(seqs)), 4) is None     # other order
    assert runner._continues_previous_step(seqs[:2], 4) is None                 # other batch
    seqs[2].block_table.append(60)                                              # a table edited by hand
    assert runner._continues_previous_step(seqs, 4) is None
    seqs[2].block_table.pop()
    runner.device_bookkeeping = False
    assert runner._continues_previous_step(seqs, 4) is None


def test_scheduler_quiet_steps_equal_the_full_pass():
    """The scheduler's shortcut for decode steps in which nothing can change (no queue change, no sequence at a block
    boundary) against the same scheduler walking the running queue every step (the reference's loop,
    scheduler.py:56-91): identical batches, block tables, free lists, hashes and finish order over random workloads with
    arrivals in mid-run, EOS, max_tokens, preemption for lack of blocks and prefix-cache hits."""
    import random
    from types import SimpleNamespace
    from byox_vllm_amd.engine.scheduler import Scheduler
    from byox_vllm_amd.engine.sequence import Sequence
    from byox_vllm_amd.sampling_params import SamplingParams

    def drive(quiet, seed):
        rng = random.Random(seed)
        Sequence.counter = __import__("itertools").count()
        cfg = SimpleNamespace(max_num_seqs=6, max_num_batched_tokens=200, eos=3, num_kvcache_blocks=rng.choice([12, 14, 40]),
                              kvcache_block_size=8, scheduler_quiet_steps=quiet)
        sch = Scheduler(cfg)
        shared = [rng.randrange(4, 50) for _ in range(16)]
        pending = []
        for i in range(10):
            prompt = (shared if rng.random() < 0.4 else []) + [rng.randrange(4, 50) for _ in range(rng.randrange(1, 30))]
            sp = SamplingParams(temperature=1.0, max_tokens=rng.randrange(1, 40), ignore_eos=rng.random() < 0.5)
            pending.append((rng.randrange(0, 40), Sequence(prompt, 8, sp)))
        trace, step, quiet_hits = [], 0, 0
        while (pending or not sch.is_finished()) and step < 2000:
            for item in [p for p in pending if p[0] <= step]:
                sch.add(item[1])
                pending.remove(item)
            if sch.is_finished():
                step += 1
                continue
            before = sch._quiet_left
            seqs, is_prefill = sch.schedule()
            quiet_hits += (not is_prefill) and before > 0
            trace.append((is_prefill, [s.seq_id for s in seqs], [list(s.block_table) for s in seqs],
                          list(sch.block_manager.free_block_ids), sorted(sch.block_manager.hash_to_block_id.items()),
                          [s.seq_id for s in sch.running], [s.seq_id for s in sch.waiting]))
            if seqs:
                toks = [rng.choice([3, 5, 6, 7, 8, 9, 10, 11]) for _ in seqs]
                sch.postprocess(seqs, toks)
            step += 1
        assert step < 2000
        return trace, quiet_hits

    total_hits = 0
    for seed in range(25):
        slow, hits0 = drive(False, seed)
        fast, hits = drive(True, seed)
        assert hits0 == 0
        assert fast == slow, f"seed {seed}"
        total_hits += hits
    assert total_hits > 50          # the shortcut was actually taken


def test_step_launched_ahead_changes_nothing_but_the_order_of_host_work():
    """The runner launches the next decode step before the caller asks for it when no other next step is possible
    (_next_step_is_certain).  Host logic only, against a device emulated in numpy (the staging kernel's two forms, a
    KV cache indexed by slot, a step counter that feeds the "sampler"; a token is a hash of everything its sequence has
    in the cache plus the counter, so a wrong slot, a stale table, a lost or doubled draw or a row overwritten by a
    dropped step changes the tokens).  The real Scheduler drives it over random workloads with arrivals in mid-run,
    EOS, max_tokens, preemption and prefix-cache hits: tokens, finish order and block tables must equal the run that
    never launches ahead, steps must have been launched ahead, and some of them dropped."""
    import itertools
    import random
    import numpy as np
    from byox_vllm_amd import kernels
    from byox_vllm_amd.sampling_params import SamplingParams

    BLOCK, ROWS, COLS = 8, 8, 16

    class Device:
        def __init__(self, runner):
            self.r, self.step, self.prefills = runner, 0, 0
            self.ids, self.pos = np.zeros(ROWS, np.int64), np.zeros(ROWS, np.int64)
            self.slots, self.ctx = np.full(ROWS, -1, np.int64), np.zeros(ROWS, np.int64)
            self.tables, self.tokens = np.full((ROWS, COLS), -1, np.int64), np.zeros(ROWS, np.int64)
            self.kv, self.launched = {}, 0

        def digest(self, table, n, salt):
            h = salt * 1000003
            for pos in range(n):
                tok, at = self.kv[table[pos // BLOCK] * BLOCK + pos % BLOCK]
                assert at == pos
                h = (h * 31 + tok * 7 + pos) % 2147483647
            return [3, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15][h % 12]

        def replay(self, padded):
            host = self.r._host_stage_np
            bs, word = int(host[0]), int(host[1])
            self.launched += 1
            if word & kernels.STAGE_STEP_BACK:
                word &= ~kernels.STAGE_STEP_BACK
                self.step -= 1
            if word & kernels.STAGE_ADVANCE:
                cols = word & ~kernels.STAGE_ADVANCE
                for i in range(bs):
                    self.ids[i], self.pos[i], self.ctx[i] = self.tokens[i], self.pos[i] + 1, self.ctx[i] + 1
                    col = int(self.pos[i]) // BLOCK
                    blk = self.tables[i, col] if col < cols else -1
                    self.slots[i] = blk * BLOCK + self.pos[i] % BLOCK if blk >= 0 else -1
            else:
                p = host[2:]
                self.ids[:bs], self.pos[:bs] = p[:bs], p[bs:2 * bs]
                self.slots[:bs], self.ctx[:bs] = p[2 * bs:3 * bs], p[3 * bs:4 * bs]
                self.tables[:bs, :word] = p[5 * bs:5 * bs + bs * word].reshape(bs, word)
            self.step += 1
            for i in range(bs):
                assert self.slots[i] >= 0
                self.kv[int(self.slots[i])] = (int(self.ids[i]), int(self.pos[i]))
                self.tokens[i] = self.digest(self.tables[i], int(self.ctx[i]), self.step)
            self.r._host_tokens_np[:bs] = self.tokens[:bs]

        def prefill(self, seqs):
            self.prefills += 1
            out = []
            for s in seqs:
                for pos in range(s.num_cached_tokens, len(s)):
                    self.kv[s.block_table[pos // BLOCK] * BLOCK + pos % BLOCK] = (s[pos], pos)
                out.append(self.digest(s.block_table, len(s), -self.prefills))
            return SimpleNamespace(tolist=lambda: out)

    def emulated(ahead, cfg):
        runner = ModelRunner.__new__(ModelRunner)
        runner.block_size, runner.device, runner.config, runner.rank = BLOCK, torch.device("cpu"), cfg, 0
        runner.device_bookkeeping, runner.run_ahead, runner.enforce_eager, runner.world_size = True, ahead, False, 1
        runner.graph_bs, runner.max_graph_bs, runner.graph_has_head, runner.graph_vars = [2, 4, ROWS], ROWS, True, {}
        runner._host_stage_np = np.zeros(2 + ROWS * (5 + COLS), np.int32)
        runner._host_tokens_np = np.zeros(ROWS, np.int64)
        dev = Device(runner)
        runner._replay, runner._wait_for_step = dev.replay, lambda: None
        runner._prepare_prefill, runner._prepare_sample = (lambda seqs: (seqs, None)), (lambda seqs: None)
        runner._run_model = lambda seqs, positions, is_prefill: dev.prefill(seqs)
        runner.sampler = lambda logits, temperatures: logits
        dropped = [0]
        real_drop = runner._drop_step_ahead

        def counting_drop():
            dropped[0] += runner._step_ahead
            real_drop()
        runner._drop_step_ahead = counting_drop
        return runner, dev, dropped

    def drive(ahead, seed):
        rng = random.Random(seed)
        Sequence.counter = itertools.count()
        cfg = SimpleNamespace(max_num_seqs=6, max_num_batched_tokens=200, eos=3, num_kvcache_blocks=rng.choice([14, 18, 48]),
                              kvcache_block_size=BLOCK, scheduler_quiet_steps=True)
        sch = Scheduler(cfg)
        runner, dev, dropped = emulated(ahead, cfg)
        shared = [rng.randrange(4, 50) for _ in range(2 * BLOCK)]
        pending = []
        for i in range(10):
            prompt = (shared if rng.random() < 0.4 else []) + [rng.randrange(4, 50) for _ in range(rng.randrange(1, 30))]
            sp = SamplingParams(temperature=1.0, max_tokens=rng.randrange(1, 60), ignore_eos=rng.random() < 0.5)
            pending.append((rng.randrange(0, 60), Sequence(prompt, BLOCK, sp)))
        trace, step = [], 0
        while (pending or not sch.is_finished()) and step < 3000:
            for item in [p for p in pending if p[0] <= step]:
                sch.add(item[1])
                pending.remove(item)
            step += 1
            if sch.is_finished():
                continue
            seqs, is_prefill = sch.schedule()
            if not seqs:
                continue
            toks = runner.run(seqs, is_prefill)
            trace.append((is_prefill, [s.seq_id for s in seqs], list(toks), [list(s.block_table) for s in seqs]))
            trace.append([s.seq_id for s in sch.postprocess(seqs, toks)])
        assert step < 3000
        decode_steps = sum(1 for t in trace[::2] if not t[0])
        return trace, dev.launched - decode_steps, dropped[0]

    extra_total = dropped_total = 0
    for seed in range(25):
        plain, extra0, dropped0 = drive(False, seed)
        ahead, extra, dropped = drive(True, seed)
        assert extra0 == 0 and dropped0 == 0
        assert ahead == plain, f"seed {seed}"
        assert extra == dropped or extra == dropped + 1, (extra, dropped)   # every unused launch was dropped (the last may be in flight)
        extra_total += extra
        dropped_total += dropped
    assert dropped_total > 5, dropped_total

    # a caller that repeats one step over and over (a timing loop) never asks for the step launched ahead: after a few
    # drops it is no longer launched (1, 2, 4 ... 64 certain steps pass before the next attempt), and the tokens (which
    # move with the step counter) are those of the runner that never launches ahead
    runs = []
    for ahead in (True, False):
        runner, dev, dropped = emulated(ahead, SimpleNamespace(eos=-1))
        seqs = [Sequence(list(range(20, 20 + n)), BLOCK, SamplingParams(temperature=1.0, max_tokens=1000, ignore_eos=True))
                for n in (3, 10)]
        seqs[0].block_table, seqs[1].block_table = [0], [1, 2]
        runner.run(seqs, True)
        runs.append(([runner.run(seqs, False) for _ in range(201)], dropped[0], dev.launched))
    assert runs[0][0] == runs[1][0] and runs[1][1:] == (0, 201)
    assert 3 <= runs[0][1] <= 9 and runs[0][2] == 201 + runs[0][1], runs[0][1:]


def test_next_step_is_certain_counts_down_to_boundaries_and_limits():
    import numpy as np
    from byox_vllm_amd.sampling_params import SamplingParams
    runner = ModelRunner.__new__(ModelRunner)
    runner.block_size, runner.config = 8, SimpleNamespace(eos=3)
    seqs = [Sequence(list(range(10, 10 + n)), 8, SamplingParams(temperature=1.0, max_tokens=m, ignore_eos=ig))
            for n, m, ig in ((5, 100, True), (3, 4, False))]
    cache = {"sure": None}
    # lengths 5 and 3: the first enters a new block when it holds 9 tokens, i.e. the step with 8 tokens is the last
    # certain one to launch from (sure: 3, then 2, 1, 0); the second ends with its 4th completion token (3 + 4 = 7)
    assert runner._next_step_is_certain(seqs, cache, [5, 3], [7, 7]) and cache["sure"] == 2
    assert runner._next_step_is_certain(seqs, cache, [6, 4], [3, 7])          # EOS for a sequence that ignores it
    assert not runner._next_step_is_certain(seqs, cache, [7, 5], [7, 3])      # EOS that ends the second sequence
    assert cache["sure"] == 0
    assert not runner._next_step_is_certain(seqs, cache, [8, 6], [7, 7])      # the first is at its block's end
    cache = {"sure": None}
    assert not runner._next_step_is_certain(seqs, cache, [6, 6], [7, 7])      # the second reaches max_tokens with this token
    cache = {"sure": None}
    assert runner._next_step_is_certain(seqs, cache, [6, 5], [7, 7]) and cache["sure"] == 0
