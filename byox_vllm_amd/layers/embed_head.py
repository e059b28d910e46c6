"""Vocab-parallel embedding and LM head (interface of nanovllm/layers/embed_head.py:9-66)."""

import torch
from torch import nn

from byox_vllm_amd import kernels
from byox_vllm_amd.layers.linear import _tp, collectives_on, get_comm, tp_all_reduce, tp_gather
from byox_vllm_amd.layers.quant import LinearWeight
from byox_vllm_amd.utils.context import get_context


class VocabParallelEmbedding(nn.Module):
    def __init__(self, num_embeddings: int, embedding_dim: int):
        super().__init__()
        self.tp_rank, self.tp_size = _tp()
        assert num_embeddings % self.tp_size == 0
        self.num_embeddings = num_embeddings
        self.num_embeddings_per_partition = num_embeddings // self.tp_size
        self.vocab_start_idx = self.num_embeddings_per_partition * self.tp_rank
        self.vocab_end_idx = self.vocab_start_idx + self.num_embeddings_per_partition
        self.weight = nn.Parameter(torch.empty(self.num_embeddings_per_partition, embedding_dim), requires_grad=False)
        self.weight.weight_loader = self.weight_loader

    loads = 0   # bumped by every load into any vocabulary matrix: derived layouts (the LM head's) are rebuilt after it

    def weight_loader(self, param: nn.Parameter, loaded_weight: torch.Tensor):
        shard = param.data.size(0)
        param.data.copy_(loaded_weight.narrow(0, self.tp_rank * shard, shard))
        VocabParallelEmbedding.loads += 1

    def forward(self, x: torch.Tensor):
        # rows outside this rank's vocab shard come back as zeros, so the all-reduce assembles the embedding
        y = kernels.embed(x, self.weight.data, self.vocab_start_idx)
        if collectives_on(self.tp_size):
            if get_comm() is not None:
                return get_comm().all_reduce(y)
            tp_all_reduce(y)
        return y


class ParallelLMHead(VocabParallelEmbedding):
    def __init__(self, num_embeddings: int, embedding_dim: int, bias: bool = False):
        assert not bias
        super().__init__(num_embeddings, embedding_dim)

    def _streaming_weight(self) -> LinearWeight:
        """The vocabulary matrix in the layout nv_gemm streams fastest, built once per loaded weight.  The row-major
        parameter stays: it is what the checkpoint loader fills and what a tied embedding gathers from."""
        cached = getattr(self, "_tiled", None)
        if self.weight is None:          # untied head whose row-major copy was released below
            return cached[1]
        w = self.weight.data
        # rebuilt after every load through a weight_loader or init_random_weights (``loads``) and after an in-place
        # write to the parameter itself (``_version``); writes through ``weight.data`` bypass both by design of torch
        key = (w.data_ptr(), VocabParallelEmbedding.loads, self.weight._version)
        if cached is None or cached[0] != key:
            n, k = w.shape
            packed = LinearWeight.dense_tiled(w) if (n % 16 == 0 and k % 128 == 0) else LinearWeight.dense(w)
            cached = self._tiled = (key, packed)
        return cached[1]

    def release_row_major(self) -> None:
        """An untied head needs only the streaming copy once the checkpoint is loaded (1.1 GB at Qwen2-7B); a tied
        one shares its row-major storage with the embedding, which gathers from it."""
        packed = self._streaming_weight()
        if self.weight is not None and packed.data.data_ptr() != self.weight.data_ptr():
            self.weight = None

    def forward_sample(self, x: torch.Tensor, temperatures: torch.Tensor, seed: int, seed_offset=None, out=None):
        """``forward`` chained with the sampler on one rank (sampler.py:11-15 after embed_head.py:54-66): -> (tokens,
        logits).  The sampler's statistics come out of the GEMM's epilogue; the tokens are those ``Sampler`` draws from
        ``forward``'s logits with the same seed."""
        assert not collectives_on(self.tp_size), "the fused head + sampler is a single-rank path"
        context = get_context()
        if context.is_prefill:
            last = (context.cu_seqlens_q[1:] - 1).long()
            x = x[last].contiguous()
        return kernels.lm_head_sample(x, self._streaming_weight(), temperatures, seed=seed, seed_offset=seed_offset,
                                      out=out)

    def forward(self, x: torch.Tensor):
        context = get_context()
        if context.is_prefill:
            last = (context.cu_seqlens_q[1:] - 1).long()
            x = x[last].contiguous()
        logits = kernels.gemm(x, self._streaming_weight())
        if collectives_on(self.tp_size):
            gathered = tp_gather(logits, self.tp_rank, self.tp_size)
            logits = torch.cat(gathered, -1) if self.tp_rank == 0 else None
        return logits
