"""Paged-KV block allocator with a chained-hash prefix cache.

Same observable behaviour as nanovllm/engine/block_manager.py:30-130: the
physical block ids it hands out, the order of the free list and the prefix
hashes are what the reference produces, because those ids end up in the block
tables the attention kernels index with.

Hash of a full block = xxh64(prefix_hash as 8 little-endian bytes (omitted for
the first block) || token ids as int64 little-endian), block_manager.py:109-115.
"""

from collections import deque

import numpy as np
import xxhash

from byox_vllm_amd.engine.sequence import Sequence
from byox_vllm_amd.utils.logging import logger


class Block:
    __slots__ = ("block_id", "ref_count", "hash", "token_ids")

    def __init__(self, block_id: int):
        self.block_id = block_id
        self.ref_count = 0
        self.hash = -1
        self.token_ids: list[int] = []

    def __repr__(self):
        return f"block-{self.block_id}"

    def update(self, hash: int, token_ids: list[int]) -> None:
        self.hash, self.token_ids = hash, token_ids

    def reset(self) -> None:
        self.ref_count, self.hash, self.token_ids = 1, -1, []


class BlockManager:
    def __init__(self, num_blocks: int, block_size: int):
        self.block_size = block_size
        self.blocks = [Block(i) for i in range(num_blocks)]
        self.hash_to_block_id: dict[int, int] = {}
        self.free_block_ids: deque[int] = deque(range(num_blocks))
        self.used_block_ids: set[int] = set()

    # -- hashing ------------------------------------------------------------
    @classmethod
    def _compute_hash(cls, token_ids, prefix: int = -1) -> int:
        state = xxhash.xxh64()
        if prefix != -1:
            state.update(prefix.to_bytes(8, "little"))
        state.update(np.array(token_ids).tobytes())
        return state.intdigest()

    # -- block bookkeeping ----------------------------------------------------
    def _take(self, block: Block) -> Block:
        assert block.ref_count == 0
        block.reset()
        self.free_block_ids.remove(block.block_id)
        self.used_block_ids.add(block.block_id)
        return block

    def _release(self, block: Block) -> None:
        # The hash -> block mapping is left in place on purpose: a freed block
        # can still be hit by a later request with the same prefix.
        assert block.ref_count == 0
        self.used_block_ids.remove(block.block_id)
        self.free_block_ids.append(block.block_id)

    # kept under the reference's names too (tests poke them)
    _allocate_block = _take
    _deallocate_block = _release

    # -- prefill ------------------------------------------------------------
    def can_allocate(self, seq: Sequence) -> bool:
        return len(self.free_block_ids) >= seq.num_blocks

    def allocate(self, seq: Sequence) -> None:
        assert not seq.block_table
        Sequence.table_epoch += 1
        logger.debug("allocating blocks for %s", seq)
        h = -1
        missed = False
        for i in range(seq.num_blocks):
            tokens = seq.block(i)
            full = len(tokens) == self.block_size
            h = self._compute_hash(tokens, h) if full else -1
            hit_id = self.hash_to_block_id.get(h, -1)
            if hit_id == -1 or self.blocks[hit_id].token_ids != tokens:
                missed = True  # once a block misses, every later block misses too
            if missed:
                block = self._take(self.blocks[self.free_block_ids[0]])
            else:
                seq.num_cached_tokens += self.block_size
                block = self.blocks[hit_id]
                if hit_id in self.used_block_ids:
                    block.ref_count += 1
                else:
                    self._take(block)  # hit on a freed block: revive it
            if h != -1:
                block.update(h, tokens)
                self.hash_to_block_id[h] = block.block_id
            seq.block_table.append(block.block_id)

    def deallocate(self, seq: Sequence) -> None:
        logger.debug("deallocating blocks for %s", seq)
        for block_id in reversed(seq.block_table):
            block = self.blocks[block_id]
            block.ref_count -= 1
            if block.ref_count == 0:
                self._release(block)
        seq.num_cached_tokens = 0
        seq.block_table.clear()
        Sequence.table_epoch += 1

    # -- decode ---------------------------------------------------------------
    def can_append(self, seq: Sequence) -> bool:
        needs_new_block = len(seq) % self.block_size == 1
        return len(self.free_block_ids) >= 1 if needs_new_block else True

    def may_append(self, seq: Sequence) -> None:
        table = seq.block_table
        last = self.blocks[table[-1]]
        rem = len(seq) % self.block_size
        if rem == 1:
            # the token just appended opens a new block
            assert last.hash != -1
            Sequence.table_epoch += 1
            table.append(self._take(self.blocks[self.free_block_ids[0]]).block_id)
        elif rem == 0:
            # last block just became full: give it its chained hash
            assert last.hash == -1
            tokens = seq.block(seq.num_blocks - 1)
            prefix = self.blocks[table[-2]].hash if len(table) > 1 else -1
            h = self._compute_hash(tokens, prefix)
            last.update(h, tokens)
            self.hash_to_block_id[h] = last.block_id
        else:
            assert last.hash == -1
