"""Prefill-first batch scheduler with recompute preemption.

Queue behaviour follows nanovllm/engine/scheduler.py:9-98: waiting requests are
prefilled first (bounded by ``max_num_seqs`` and ``max_num_batched_tokens``; a
request that does not fit rotates to the back of the queue), otherwise every
running request decodes one token, evicting from the back of the running queue
when the block pool is exhausted.
"""

from collections import deque

from byox_vllm_amd.engine.block_manager import BlockManager
from byox_vllm_amd.engine.sequence import Sequence, SequenceStatus
from byox_vllm_amd.utils.logging import logger


class Scheduler:
    def __init__(self, config):
        self.max_num_seqs = config.max_num_seqs
        self.max_num_batched_tokens = config.max_num_batched_tokens
        self.eos = config.eos
        self.block_manager = BlockManager(config.num_kvcache_blocks, config.kvcache_block_size)
        self.waiting: deque[Sequence] = deque()
        self.running: deque[Sequence] = deque()

    def is_finished(self) -> bool:
        return not self.waiting and not self.running

    def add(self, seq: Sequence) -> None:
        self.waiting.append(seq)

    def schedule(self) -> tuple[list[Sequence], bool]:
        batch = self._schedule_prefill()
        if batch:
            return batch, True
        return self._schedule_decode(), False

    def _schedule_prefill(self) -> list[Sequence]:
        batch: list[Sequence] = []
        budget = self.max_num_batched_tokens
        while self.waiting and len(batch) < self.max_num_seqs:
            seq = self.waiting[0]
            if len(seq) > budget:
                self.waiting.rotate(-1)
                break
            if not self.block_manager.can_allocate(seq):
                break
            self.block_manager.allocate(seq)
            budget -= len(seq) - seq.num_cached_tokens
            seq.status = SequenceStatus.RUNNING
            self.waiting.popleft()
            self.running.append(seq)
            batch.append(seq)
        if batch:
            logger.debug("[prefill] scheduled %s", batch)
        return batch

    def _schedule_decode(self) -> list[Sequence]:
        batch: list[Sequence] = []
        while self.running and len(batch) < self.max_num_seqs:
            seq = self.running.popleft()
            evicted_self = False
            while not self.block_manager.can_append(seq):
                if self.running:
                    self._preempt(self.running.pop())
                else:
                    self._preempt(seq)
                    evicted_self = True
                    break
            if not evicted_self:
                self.block_manager.may_append(seq)
                batch.append(seq)
        if batch:
            self.running.extendleft(reversed(batch))
            logger.debug("[decode] scheduled %s", batch)
        return batch

    def _preempt(self, seq: Sequence) -> None:
        logger.debug("preempting %s", seq)
        seq.status = SequenceStatus.WAITING
        self.block_manager.deallocate(seq)
        self.waiting.appendleft(seq)

    def postprocess(self, seqs: list[Sequence], token_ids: list[int]) -> None:
        for seq, token_id in zip(seqs, token_ids):
            seq.append_token(token_id)
            hit_eos = not seq.ignore_eos and token_id == self.eos
            if hit_eos or seq.num_completion_tokens == seq.max_tokens:
                seq.status = SequenceStatus.FINISHED
                self.block_manager.deallocate(seq)
                self.running.remove(seq)
