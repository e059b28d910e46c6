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
        # Quiet decode steps.  After a full pass over the running queue the scheduler knows for how many further steps
        # that pass would repeat itself exactly: as long as the queues do not change, every scheduled sequence grows by
        # one token per step, and none of them reaches a block boundary (length % block_size in {0, 1}: a block to
        # hash or a block to allocate), the pass changes nothing and returns the same batch.  Those steps return the
        # batch of the last pass without walking it (45 us per step for 128 sequences otherwise).
        self.quiet_steps = getattr(config, "scheduler_quiet_steps", True)
        self._quiet_left = 0              # further steps the last full pass stays valid for
        self._quiet_batch: list[Sequence] = []
        self._quiet_len = 0               # length the batch's first sequence must have at the next quiet step

    def is_finished(self) -> bool:
        return not self.waiting and not self.running

    def add(self, seq: Sequence) -> None:
        self.waiting.append(seq)
        self._quiet_left = 0

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
            self._quiet_left = 0
            logger.debug("[prefill] scheduled %s", batch)
        return batch

    def _schedule_decode(self) -> list[Sequence]:
        if self._quiet_left > 0 and len(self._quiet_batch[0]) == self._quiet_len:
            self._quiet_left -= 1
            self._quiet_len += 1
            return self._quiet_batch
        self._quiet_left = 0
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
            if self.quiet_steps:
                # steps until the first sequence of the batch has length % block_size == 0 (each step adds one token)
                bs = self.block_manager.block_size
                self._quiet_left = min((bs - 1 - r) if r else 0 for r in (len(seq) % bs for seq in batch))
                self._quiet_batch = batch
                self._quiet_len = len(batch[0]) + 1
        return batch

    def _preempt(self, seq: Sequence) -> None:
        logger.debug("preempting %s", seq)
        self._quiet_left = 0
        seq.status = SequenceStatus.WAITING
        self.block_manager.deallocate(seq)
        self.waiting.appendleft(seq)

    def postprocess(self, seqs: list[Sequence], token_ids: list[int]) -> list[Sequence]:
        """Append the step's tokens; -> the sequences that finished with it (the reference returns nothing and has the
        engine look for them, llm_engine.py:52)."""
        eos = self.eos
        finished: list[Sequence] = []
        for seq, token_id in zip(seqs, token_ids):
            # Sequence.append_token and num_completion_tokens written out: this loop runs once per sequence and step
            seq.token_ids.append(token_id)
            seq.last_token = token_id
            seq.num_tokens = n = seq.num_tokens + 1
            if (token_id == eos and not seq.ignore_eos) or n - seq.num_prompt_tokens == seq.max_tokens:
                seq.status = SequenceStatus.FINISHED
                self.block_manager.deallocate(seq)
                self.running.remove(seq)
                self._quiet_left = 0
                finished.append(seq)
        return finished
