"""Weight containers for nv_gemm: dense, or weight-only int8 / int4 (group 128).

The reference has no quantised linears (nanovllm/layers/linear.py is dense
F.linear); the format below is this build's own.  Quantisation is symmetric
with one scale per (output row, 128 input columns), scales kept in the model
dtype:  int4: u = clamp(round(w/s), -8, 7) + 8, s = max|w|/7;  int8:
q = clamp(round(w/s), -127, 127), s = max|w|/127.

Device layout ("tiled"): the GEMM kernel gives each wavefront a 16-row tile of
W and walks K in blocks of 128.  For tile nt and block kb, lane (c = lane % 16,
g = lane // 16) needs W[16 nt + c][128 kb + 32 ks + 8 g + e] for ks in 0..3,
e in 0..7.  Those 32 values are stored contiguously per lane:
  int4: 4 dwords (dword ks, nibble e)  -> tensor int32 [N/16, K/128, 64, 4]
  int8: 32 bytes (byte 8 ks + e)       -> tensor int8  [N/16, K/128, 64, 32]
so one or two 16-byte loads per lane fetch a whole K-block, fully coalesced.
Scales are stored transposed, [K/128, N].
"""

import torch

from byox_vllm_amd._C import W_DENSE, W_DENSE_TILED, W_INT4, W_INT8

GROUP = 128


def _tile_index_view(t: torch.Tensor) -> torch.Tensor:
    """[N, K] -> [N/16, K/128, g(4), c(16), ks(4), e(8)] in lane-major order."""
    n, k = t.shape
    return t.view(n // 16, 16, k // GROUP, 4, 4, 8).permute(0, 2, 4, 1, 3, 5)


def _untile(t: torch.Tensor, n: int, k: int) -> torch.Tensor:
    """Inverse of _tile_index_view: [N/16, K/128, 4, 16, 4, 8] -> [N, K]."""
    return t.permute(0, 3, 1, 4, 2, 5).reshape(n, k)


def _scales(w: torch.Tensor, qmax: int):
    n, k = w.shape
    assert k % GROUP == 0 and n % 16 == 0, f"weight [{n}, {k}] needs N % 16 == 0 and K % 128 == 0"
    g = w.float().view(n, k // GROUP, GROUP)
    s = (g.abs().amax(dim=-1) / qmax).clamp_min(1e-8).to(w.dtype)
    return g, s


def interleave_gate_up(w: torch.Tensor) -> torch.Tensor:
    """[gate; up] rows -> rows 4i..4i+3 = (gate 2i, gate 2i+1, up 2i, up 2i+1): the row order
    nv_gemm's silu(gate)*up epilogue expects (a lane of the MFMA result owns 4 consecutive rows)."""
    gate, up = w.chunk(2, 0)
    inter = gate.size(0)
    assert inter % 2 == 0
    g = gate.reshape(inter // 2, 2, -1)
    u = up.reshape(inter // 2, 2, -1)
    return torch.cat((g, u), dim=1).reshape(2 * inter, -1).contiguous()


class LinearWeight:
    """What nv_gemm consumes: ``fmt``, ``data`` (device tensor), ``scales`` ([K/128, N] or None)."""

    def __init__(self, fmt, data, scales, out_features, in_features, dtype):
        self.fmt, self.data, self.scales = fmt, data, scales
        self.out_features, self.in_features, self.dtype = out_features, in_features, dtype

    @property
    def nbytes(self) -> int:
        n = self.data.numel() * self.data.element_size()
        return n + (self.scales.numel() * self.scales.element_size() if self.scales is not None else 0)

    @classmethod
    def dense(cls, w: torch.Tensor):
        assert w.dim() == 2 and w.is_contiguous()
        return cls(W_DENSE, w, None, w.size(0), w.size(1), w.dtype)

    @classmethod
    def dense_tiled(cls, w: torch.Tensor):
        """Unquantised weights re-ordered for streaming: [N/16][K/128][k-step 4][lane 64][8 elements], lane (c, g) of
        k-step ks holding W[16 nt + c][128 kb + 32 ks + 8 g + e].  Every 16-byte-per-lane load of a wavefront is then
        one contiguous KB and a tile's K-blocks follow each other in memory (row-major rows are 2 K bytes apart and
        give 64-byte pieces).  A copy: the row-major parameter stays what the checkpoint loader and a tied
        embedding use."""
        n, k = w.shape
        assert k % GROUP == 0 and n % 16 == 0, f"weight [{n}, {k}] needs N % 16 == 0 and K % 128 == 0"
        tiled = w.view(n // 16, 16, k // GROUP, 4, 4, 8).permute(0, 2, 3, 4, 1, 5).contiguous()
        return cls(W_DENSE_TILED, tiled.view(n // 16, k // GROUP, 4, 64, 8), None, n, k, w.dtype)

    def logical_dense(self):
        """Row-major [N, K] view of a dense or dense-tiled weight."""
        if self.fmt == W_DENSE:
            return self.data
        assert self.fmt == W_DENSE_TILED
        n, k = self.out_features, self.in_features
        return self.data.view(n // 16, k // GROUP, 4, 4, 16, 8).permute(0, 4, 1, 2, 3, 5).reshape(n, k)

    @classmethod
    def int4(cls, w: torch.Tensor):
        n, k = w.shape
        g, s = _scales(w, 7)
        v = torch.clamp(torch.round(g / s.float().unsqueeze(-1)), -8, 7).to(torch.int8).view(n, k)
        return cls.from_quantized(v, s, "w4a16")

    @classmethod
    def int8(cls, w: torch.Tensor):
        n, k = w.shape
        g, s = _scales(w, 127)
        q = torch.clamp(torch.round(g / s.float().unsqueeze(-1)), -127, 127).to(torch.int8).view(n, k)
        return cls.from_quantized(q, s, "w8a16")

    @classmethod
    def from_quantized(cls, values: torch.Tensor, scales: torch.Tensor, quantization: str):
        """Kernel-side weight from already quantised values: ``values`` int8 [N, K] (int4: -8..7, int8: -127..127),
        ``scales`` [N, K/128] in the model dtype.  What int4()/int8() end in, and what a checkpoint quantised offline
        (utils/quantize.py) is loaded through, so both give the same bytes."""
        n, k = values.shape
        assert values.dtype == torch.int8 and scales.shape == (n, k // GROUP) and k % GROUP == 0 and n % 16 == 0
        if quantization == "w8a16":
            tiled = _tile_index_view(values).reshape(n // 16, k // GROUP, 64, 32).contiguous()
            return cls(W_INT8, tiled, scales.t().contiguous(), n, k, scales.dtype)
        assert quantization == "w4a16"
        u = values.to(torch.int32) + 8
        shifts = torch.arange(0, 32, 4, device=values.device, dtype=torch.int32)
        tiled = (_tile_index_view(u) << shifts).sum(dim=-1, dtype=torch.int64)
        tiled = (tiled & 0xFFFFFFFF).to(torch.int64)
        tiled = torch.where(tiled >= 2 ** 31, tiled - 2 ** 32, tiled).to(torch.int32)
        return cls(W_INT4, tiled.reshape(n // 16, k // GROUP, 64, 4).contiguous(), scales.t().contiguous(), n, k, scales.dtype)

    def quantized_values(self):
        """-> (int8 [N, K] values, scales [N, K/128]): the inverse of from_quantized."""
        if self.fmt == W_INT8:
            return self.logical_int8()
        packed, scales = self.logical_int4()
        n = packed.size(0)
        u = torch.stack((packed & 0xF, packed >> 4), dim=-1).view(n, -1)
        return (u.to(torch.int16) - 8).to(torch.int8), scales

    @classmethod
    def make(cls, w: torch.Tensor, quantization: str | None):
        if quantization is None:   # unquantised: the streaming layout when the shape allows it
            n, k = w.shape
            return cls.dense_tiled(w.contiguous()) if (n % 16 == 0 and k % GROUP == 0) else cls.dense(w.contiguous())
        return cls.int4(w) if quantization == "w4a16" else cls.int8(w)

    # -- logical (untiled) views, used by tests to compare with the CPU oracle's format ----
    def logical_int4(self):
        """-> (uint8 [N, K/2] two values per byte, low nibble = even index; scales [N, K/128])."""
        assert self.fmt == W_INT4
        n, k = self.out_features, self.in_features
        d = self.data.view(n // 16, k // GROUP, 4, 16, 4, 1).to(torch.int64) & 0xFFFFFFFF
        shifts = torch.arange(0, 32, 4, device=d.device, dtype=torch.int64)
        u = _untile((d >> shifts) & 0xF, n, k).to(torch.uint8)
        return (u[:, 0::2] | (u[:, 1::2] << 4)).contiguous(), self.scales.t().contiguous()

    def logical_int8(self):
        assert self.fmt == W_INT8
        n, k = self.out_features, self.in_features
        q = _untile(self.data.view(n // 16, k // GROUP, 4, 16, 4, 8), n, k)
        return q.contiguous(), self.scales.t().contiguous()
