"""SiLU-and-multiply (interface of nanovllm/layers/activation.py:6-13)."""

import torch
from torch import nn

from byox_vllm_amd import kernels


class SiluAndMul(nn.Module):
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return kernels.silu_mul(x)
