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


_TP_STATE = None   # (rank, size, process group) once the runner has fixed the tensor-parallel layout


def set_tp(rank: int, size: int, group=None) -> None:
    """Fix the tensor-parallel layout explicitly (a data-parallel job has a process group but TP size 1)."""
    global _TP_STATE
    _TP_STATE = (rank, size, group)


def tp_group():
    return _TP_STATE[2] if _TP_STATE is not None else None


def _tp():
    if _TP_STATE is not None:
        return _TP_STATE[0], _TP_STATE[1]
    if dist.is_available() and dist.is_initialized():   # the reference's rule: TP = the default group
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

    def _gemm(self, x, bias, out=None, **kw):
        if self.packed is None:
            self.finalize(None)
        return kernels.gemm(x, self.packed, bias, out=out, **kw)

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

    def forward(self, x, out=None, partial_out=False):
        """partial_out: hand the split-K slabs (and the bias) to a fused consumer instead of reducing here."""
        return self._gemm(x, None if self.bias is None else self.bias.data, out=out, partial_out=partial_out)


class MergedColumnParallelLinear(ColumnParallelLinear):
    def __init__(self, input_size: int, output_sizes: list[int], bias: bool = False):
        self.output_sizes = output_sizes
        self.fused_silu = False
        super().__init__(input_size, sum(output_sizes), bias)

    def finalize(self, quantization: str | None = None, fuse_silu: bool = False) -> None:
        """fuse_silu: store gate/up rows interleaved so forward_silu_mul() can apply SiluAndMul in the GEMM epilogue."""
        from byox_vllm_amd.layers.quant import interleave_gate_up
        self.fused_silu = fuse_silu and self.bias is None and len(self.output_sizes) == 2
        w = interleave_gate_up(self.weight.data) if self.fused_silu else self.weight.data
        self.packed = LinearWeight.make(w, quantization)
        if quantization is not None or self.fused_silu:
            self.weight = None

    def forward(self, x, out=None, partial_out=False):
        assert not self.fused_silu, "weights are stored for the fused epilogue: call forward_silu_mul"
        return super().forward(x, out=out, partial_out=partial_out)

    def forward_silu_mul(self, x, out=None):
        """silu(gate) * up in one launch -> [T, intermediate / tp]."""
        assert self.fused_silu
        return self._gemm(x, None, out=out, silu_mul=True)

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

    def forward(self, x, out=None, partial_out=False):
        """partial_out (single rank only): return the split-K slabs for the following add_rmsnorm to sum."""
        bias = self.bias.data if (self.bias is not None and self.tp_rank == 0) else None
        if partial_out and self.tp_size == 1 and bias is None:
            return self._gemm(x, None, partial_out=True)
        y = self._gemm(x, bias, out=out)
        if self.tp_size > 1:
            dist.all_reduce(y, group=tp_group())
        return y
