"""Per-request state that feeds block-table indexing.

Behaviour follows nanovllm/engine/sequence.py:14-77: a request owns its token
ids, the list of physical KV block ids (``block_table``) and the number of
prompt tokens already present in the prefix cache.  Every integer the decode
step turns into a slot or a block-table row is derived here.
"""

from enum import Enum, auto
from itertools import count

from byox_vllm_amd.sampling_params import SamplingParams


class SequenceStatus(Enum):
    WAITING = auto()
    RUNNING = auto()
    FINISHED = auto()


class Sequence:
    counter = count()
    # Bumped whenever some block table is re-filled (BlockManager.allocate / deallocate): tells a consumer that
    # caches padded block tables that equal lengths no longer imply equal contents.
    table_epoch = 0

    def __init__(self, token_ids: list[int], block_size: int, sampling_params: SamplingParams | None = None):
        sp = sampling_params if sampling_params is not None else SamplingParams()
        self.seq_id = next(Sequence.counter)
        self.status = SequenceStatus.WAITING
        self.block_size = block_size
        self.token_ids = list(token_ids)
        self.num_prompt_tokens = len(self.token_ids)
        self.num_tokens = self.num_prompt_tokens
        self.last_token = self.token_ids[-1]
        self.num_cached_tokens = 0
        self.block_table: list[int] = []
        self.temperature = sp.temperature
        self.max_tokens = sp.max_tokens
        self.ignore_eos = sp.ignore_eos

    # -- container protocol -------------------------------------------------
    def __len__(self):
        return self.num_tokens

    def __getitem__(self, key):
        return self.token_ids[key]

    def __repr__(self):
        return f"seq-{self.seq_id}"

    # -- derived quantities ---------------------------------------------------
    @property
    def is_finished(self) -> bool:
        return self.status is SequenceStatus.FINISHED

    @property
    def num_completion_tokens(self) -> int:
        return self.num_tokens - self.num_prompt_tokens

    @property
    def prompt_token_ids(self) -> list[int]:
        return self.token_ids[: self.num_prompt_tokens]

    @property
    def completion_token_ids(self) -> list[int]:
        return self.token_ids[self.num_prompt_tokens:]

    @property
    def num_cached_blocks(self) -> int:
        return self.num_cached_tokens // self.block_size

    @property
    def num_blocks(self) -> int:
        return -(-self.num_tokens // self.block_size)

    @property
    def last_block_num_tokens(self) -> int:
        return self.num_tokens - (self.num_blocks - 1) * self.block_size

    def block(self, i: int) -> list[int]:
        """Token ids of logical block ``i`` (the last one may be partial)."""
        assert 0 <= i < self.num_blocks
        lo = i * self.block_size
        return self.token_ids[lo: lo + self.block_size]

    def append_token(self, token_id: int) -> None:
        self.token_ids.append(token_id)
        self.last_token = token_id
        self.num_tokens += 1
