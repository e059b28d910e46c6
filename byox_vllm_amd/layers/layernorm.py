"""RMSNorm on the HIP kernels (interface of nanovllm/layers/layernorm.py:5-50)."""

import torch
from torch import nn

from byox_vllm_amd import kernels


class RMSNorm(nn.Module):
    def __init__(self, hidden_size: int, eps: float = 1e-6) -> None:
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(hidden_size), requires_grad=False)

    def rms_forward(self, x: torch.Tensor) -> torch.Tensor:
        return kernels.rms_norm(x.contiguous(), self.weight.data, self.eps)

    def add_rms_forward(self, x, residual: torch.Tensor):
        # x may be a kernels.Partials (split-K slabs of the preceding linear).  The residual is updated in place.
        if isinstance(x, kernels.PendingReduce):   # tensor parallel: all-reduce + add + norm in one launch; the
            from byox_vllm_amd.layers.linear import get_comm   # residual stays sharded by rows (layers/comm.py)
            return get_comm().all_reduce_add_rms_norm(x.value, residual, self.weight.data, self.eps)
        if isinstance(x, kernels.Partials):
            return kernels.add_rms_norm(x, residual, self.weight.data, self.eps, residual_out=residual)
        return kernels.add_rms_norm(x, residual, self.weight.data, self.eps, out=x, residual_out=residual)

    def forward(self, x: torch.Tensor, residual: torch.Tensor | None = None):
        if residual is None:
            return self.rms_forward(x)
        return self.add_rms_forward(x, residual)
