"""Weight-only quantisation formats and their dequantisation (test oracle).

The reference has no quantisation code (its linears are dense F.linear,
nanovllm/layers/linear.py:49,70,144), so the format is defined by this build
and the oracle for a quantised linear is "dequantise to the model dtype on the
CPU, then the reference's F.linear".  Parity unpinned: nothing in the reference
can pin these numbers.

Formats (group size 128 along the input dimension, symmetric):
* w4a16: u = clamp(round(w / s), -8, 7) + 8 in [0, 15]; s = max|w| / 7 per
  (row, group), stored in the model dtype; logical packing is two values per
  byte along `in`, low nibble = even index.  Dequantised w = (u - 8) * s.
* w8a16: q = clamp(round(w / s), -127, 127) as int8; s = max|w| / 127.
"""

import torch

GROUP = 128


def _group_scales(w: torch.Tensor, qmax: int, dtype):
    n, k = w.shape
    assert k % GROUP == 0
    g = w.float().view(n, k // GROUP, GROUP)
    s = (g.abs().amax(dim=-1) / qmax).clamp_min(1e-8).to(dtype)   # [N, K/128]
    return g, s


def quantize_w4(w: torch.Tensor, dtype=None):
    """-> (packed uint8 [N, K/2], scales [N, K/128] in the model dtype)."""
    dtype = dtype or w.dtype
    g, s = _group_scales(w, 7, dtype)
    u = torch.clamp(torch.round(g / s.float().unsqueeze(-1)), -8, 7).to(torch.int16) + 8
    u = u.view(w.size(0), -1).to(torch.uint8)
    packed = u[:, 0::2] | (u[:, 1::2] << 4)
    return packed.contiguous(), s.contiguous()


def dequantize_w4(packed: torch.Tensor, scales: torch.Tensor) -> torch.Tensor:
    n = packed.size(0)
    u = torch.stack((packed & 0xF, packed >> 4), dim=-1).view(n, -1).float() - 8.0
    w = u.view(n, -1, GROUP) * scales.float().unsqueeze(-1)
    return w.view(n, -1).to(scales.dtype)


def quantize_w8(w: torch.Tensor, dtype=None):
    """-> (int8 [N, K], scales [N, K/128] in the model dtype)."""
    dtype = dtype or w.dtype
    g, s = _group_scales(w, 127, dtype)
    q = torch.clamp(torch.round(g / s.float().unsqueeze(-1)), -127, 127).to(torch.int8)
    return q.view(w.size(0), -1).contiguous(), s.contiguous()


def dequantize_w8(q: torch.Tensor, scales: torch.Tensor) -> torch.Tensor:
    n = q.size(0)
    w = q.float().view(n, -1, GROUP) * scales.float().unsqueeze(-1)
    return w.view(n, -1).to(scales.dtype)


def quantized_linear(x, qweight, scales, bias=None, bits=4):
    """Oracle for the dequant-GEMM: dequantise to the model dtype, then F.linear."""
    w = dequantize_w4(qweight, scales) if bits == 4 else dequantize_w8(qweight, scales)
    return torch.nn.functional.linear(x, w.to(x.dtype), bias)
