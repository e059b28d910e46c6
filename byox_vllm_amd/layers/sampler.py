"""Token sampler (interface of nanovllm/layers/sampler.py:5-15).

Gumbel-max sampling as in the reference, with two additions: temperature 0
means argmax, and the Exp(1) draws can be injected (``noise``) so a test can
replay the reference's random numbers.
"""

import torch
from torch import nn

from byox_vllm_amd import kernels


class Sampler(nn.Module):
    def __init__(self, seed: int = 0):
        super().__init__()
        self.seed = seed
        self.calls = 0

    def forward(self, logits: torch.Tensor, temperatures: torch.Tensor, noise: torch.Tensor | None = None):
        self.calls += 1
        return kernels.sample(logits, temperatures, noise=noise, seed=(self.seed << 32) + self.calls)
