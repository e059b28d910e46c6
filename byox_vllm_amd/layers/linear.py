"""Tensor-parallel linear layers on nv_gemm (interface of nanovllm/layers/linear.py:12-147).

Weights are loaded dense through the same per-parameter ``weight_loader``
contract as the reference (column shards narrow dim 0, row shards dim 1, QKV
and gate/up are packed per rank).  ``finalize(quantization)`` then converts the
dense shard to what the kernel consumes (dense, int8 or int4 tiles) and, for
quantised formats, drops the dense copy.
"""

import torch
import torch.distributed as dist
from torch import nn

from byox_vllm_amd import kernels
from byox_vllm_amd.layers.quant import LinearWeight


def divide(numerator, denominator):
    assert numerator % denominator == 0
    return numerator // denominator


def _tp():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


class LinearBase(nn.Module):
    def __init__(self, input_size: int, output_size: int, bias: bool = False, tp_dim: int | None = None):
        super().__init__()
        self.tp_dim = tp_dim
        self.tp_rank, self.tp_size = _tp()
        self.weight = nn.Parameter(torch.empty(output_size, input_size), requires_grad=False)
        self.weight.weight_loader = self.weight_loader
        if bias:
            self.bias = nn.Parameter(torch.empty(output_size), requires_grad=False)
            self.bias.weight_loader = self.weight_loader
        else:
            self.register_parameter("bias", None)
        self.packed: LinearWeight | None = None

    def finalize(self, quantization: str | None = None) -> None:
        """Build the kernel-side weight; quantised layers free the dense parameter."""
        self.packed = LinearWeight.make(self.weight.data, quantization)
        if quantization is not None:
            self.weight = None

    def _gemm(self, x, bias, out=None):
        if self.packed is None:
            self.finalize(None)
        return kernels.gemm(x, self.packed, bias, out=out)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError


class ReplicatedLinear(LinearBase):
    def __init__(self, input_size: int, output_size: int, bias: bool = False):
        super().__init__(input_size, output_size, bias)

    def weight_loader(self, param: nn.Parameter, loaded_weight: torch.Tensor):
        param.data.copy_(loaded_weight)

    def forward(self, x):
        return self._gemm(x, None if self.bias is None else self.bias.data)


class ColumnParallelLinear(LinearBase):
    def __init__(self, input_size: int, output_size: int, bias: bool = False):
        _, tp_size = _tp()
        super().__init__(input_size, divide(output_size, tp_size), bias, 0)

    def weight_loader(self, param: nn.Parameter, loaded_weight: torch.Tensor):
        shard = param.data.size(self.tp_dim)
        param.data.copy_(loaded_weight.narrow(self.tp_dim, self.tp_rank * shard, shard))

    def forward(self, x, out=None):
        return self._gemm(x, None if self.bias is None else self.bias.data, out=out)


class MergedColumnParallelLinear(ColumnParallelLinear):
    def __init__(self, input_size: int, output_sizes: list[int], bias: bool = False):
        self.output_sizes = output_sizes
        super().__init__(input_size, sum(output_sizes), bias)

    def weight_loader(self, param: nn.Parameter, loaded_weight: torch.Tensor, loaded_shard_id: int):
        offset = sum(self.output_sizes[:loaded_shard_id]) // self.tp_size
        size = self.output_sizes[loaded_shard_id] // self.tp_size
        dst = param.data.narrow(self.tp_dim, offset, size)
        dst.copy_(loaded_weight.chunk(self.tp_size, self.tp_dim)[self.tp_rank])


class QKVParallelLinear(ColumnParallelLinear):
    def __init__(self, hidden_size: int, head_size: int, total_num_heads: int,
                 total_num_kv_heads: int | None = None, bias: bool = False):
        _, tp_size = _tp()
        total_num_kv_heads = total_num_kv_heads or total_num_heads
        self.head_size = head_size
        self.num_heads = divide(total_num_heads, tp_size)
        self.num_kv_heads = divide(total_num_kv_heads, tp_size)
        super().__init__(hidden_size, (total_num_heads + 2 * total_num_kv_heads) * head_size, bias)

    def weight_loader(self, param: nn.Parameter, loaded_weight: torch.Tensor, loaded_shard_id: str):
        assert loaded_shard_id in ("q", "k", "v")
        q_rows, kv_rows = self.num_heads * self.head_size, self.num_kv_heads * self.head_size
        offset, size = {"q": (0, q_rows), "k": (q_rows, kv_rows), "v": (q_rows + kv_rows, kv_rows)}[loaded_shard_id]
        dst = param.data.narrow(self.tp_dim, offset, size)
        dst.copy_(loaded_weight.chunk(self.tp_size, self.tp_dim)[self.tp_rank])


class RowParallelLinear(LinearBase):
    def __init__(self, input_size: int, output_size: int, bias: bool = False):
        _, tp_size = _tp()
        super().__init__(divide(input_size, tp_size), output_size, bias, 1)

    def weight_loader(self, param: nn.Parameter, loaded_weight: torch.Tensor):
        if param.dim() == 1:   # a bias is not sharded along the input dimension
            param.data.copy_(loaded_weight)
            return
        shard = param.data.size(self.tp_dim)
        param.data.copy_(loaded_weight.narrow(self.tp_dim, self.tp_rank * shard, shard))

    def forward(self, x, out=None):
        bias = self.bias.data if (self.bias is not None and self.tp_rank == 0) else None
        y = self._gemm(x, bias, out=out)
        if self.tp_size > 1:
            dist.all_reduce(y)
        return y
