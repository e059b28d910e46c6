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


K_ALIGN = 128   # nv_gemm walks K in blocks of 128 (= one quantisation group): every K shard is a multiple of it


def split_units(total: int, tp_size: int, tp_rank: int) -> tuple[int, int, int]:
    """Deal ``total`` units to ``tp_size`` ranks as evenly as possible -> (first unit, units owned, padded units).
    Every rank allocates ``padded`` units (the largest share); the tail of a smaller share is zero.  When
    ``total`` divides, this is the reference's even chunking (linear.py:62-64,140-145)."""
    base, extra = divmod(total, tp_size)
    count = base + (1 if tp_rank < extra else 0)
    start = tp_rank * base + min(tp_rank, extra)
    return start, count, base + (1 if extra else 0)


class HeadLayout:
    """Which query / key-value heads one tensor-parallel rank owns, and in which local slots.

    * ``num_kv_heads % tp == 0``: the reference's split (qwen3.py:30-34): contiguous heads, nothing replicated.
    * ``tp % num_kv_heads == 0`` (more ranks than KV heads, e.g. 28/4 heads on 8 ranks): each KV head is
      replicated on ``tp / num_kv_heads`` consecutive ranks, which share its query group as evenly as possible
      (7 heads -> 4 + 3).
    Local query slots are grouped per local KV head (slot s attends with KV head s // group) and padded with
    all-zero heads so that every rank has the same shape and ``slots * head_dim`` is a multiple of 128 (the K
    extent of o_proj).  A zero head has zero q rows/bias and zero o_proj columns: it adds exactly 0 to the
    row-parallel sum, so the all-reduced result equals the unsharded layer's."""

    def __init__(self, num_heads: int, num_kv_heads: int, head_dim: int, tp_size: int, tp_rank: int):
        assert num_heads % num_kv_heads == 0, "query heads must be a multiple of KV heads"
        group = num_heads // num_kv_heads
        if num_kv_heads % tp_size == 0:
            per = num_kv_heads // tp_size
            self.kv_heads = list(range(tp_rank * per, (tp_rank + 1) * per))
            real = [list(range(kv * group, (kv + 1) * group)) for kv in self.kv_heads]
            widest = group
        else:
            assert tp_size % num_kv_heads == 0, (
                f"{num_heads}/{num_kv_heads} heads on {tp_size} ranks: KV heads must divide or be divided by the rank count")
            rep = tp_size // num_kv_heads
            kv, sub = divmod(tp_rank, rep)
            start, count, widest = split_units(group, rep, sub)
            self.kv_heads = [kv]
            real = [list(range(kv * group + start, kv * group + start + count))]
        slots = widest
        if (num_heads * head_dim) % K_ALIGN == 0:    # (toy shapes that nv_gemm could not run anyway stay unpadded)
            while (slots * len(self.kv_heads) * head_dim) % K_ALIGN:
                slots += 1
        assert slots <= 16, "a local GQA group wider than 16 heads does not fit the attention kernel"
        self.group = slots                                        # query slots per local KV head
        self.q_heads = [(g[j] if j < len(g) else -1) for g in real for j in range(slots)]   # -1 = zero head
        self.num_heads = len(self.q_heads)
        self.num_kv_heads = len(self.kv_heads)
        self.head_dim = head_dim

    def segments(self, heads):
        """[(source row of the full tensor, rows, destination row)] for the listed heads (runs merged; -1 skipped)."""
        d, out = self.head_dim, []
        for slot, h in enumerate(heads):
            if h < 0:
                continue
            if out and out[-1][0] + out[-1][1] == h * d and out[-1][2] + out[-1][1] == slot * d:
                out[-1] = (out[-1][0], out[-1][1] + d, out[-1][2])
            else:
                out.append((h * d, d, slot * d))
        return out


def copy_segments(dst: torch.Tensor, src: torch.Tensor, dim: int, segments) -> None:
    """dst = 0 except dst[dst_lo : dst_lo + n] = src[src_lo : src_lo + n] along ``dim`` for every segment."""
    dst.zero_()
    for src_lo, n, dst_lo in segments:
        dst.narrow(dim, dst_lo, n).copy_(src.narrow(dim, src_lo, n))


_TP_STATE = None   # (rank, size, process group) once the runner has fixed the tensor-parallel layout


_SINGLE_RANK_COLLECTIVES = False


def set_tp(rank: int, size: int, group=None, single_rank_collectives: bool = False) -> None:
    """Fix the tensor-parallel layout explicitly (a data-parallel job has a process group but TP size 1).
    ``single_rank_collectives``: issue the row-parallel all-reduces, the embedding's all-reduce and the logits gather
    even with one rank, so that a machine with a single GPU executes the RCCL calls of the tensor-parallel path
    (sums over one rank: results unchanged)."""
    global _TP_STATE, _SINGLE_RANK_COLLECTIVES
    _TP_STATE = (rank, size, group)
    _SINGLE_RANK_COLLECTIVES = bool(single_rank_collectives) and size == 1


def collectives_on(tp_size: int) -> bool:
    return tp_size > 1 or _SINGLE_RANK_COLLECTIVES


_COMM = None   # layers.comm.XgmiComm of this rank, or None: collectives then go through torch.distributed (RCCL)


def set_comm(comm) -> None:
    global _COMM
    _COMM = comm


def get_comm():
    return _COMM


def tp_group():
    return _TP_STATE[2] if _TP_STATE is not None else None


def _host_hop(group) -> bool:
    """True when the group's backend cannot take device tensors reliably (gloo rehearsal of the TP runtime)."""
    return dist.get_backend(group) != "nccl"


def tp_all_reduce(t: torch.Tensor) -> None:
    """Sum ``t`` over the tensor-parallel group in place (RCCL over xGMI; the reference's dist.all_reduce,
    linear.py:146, embed_head.py:40).  Under a gloo group the data takes a hop through host memory in fp32."""
    group = tp_group()
    if t.is_cuda and _host_hop(group):
        host = t.float().cpu()
        dist.all_reduce(host, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, group=group)


def tp_gather(t: torch.Tensor, tp_rank: int, tp_size: int):
    """Gather ``t`` from every rank onto the group's first rank -> list of tensors there, None elsewhere
    (embed_head.py:62-65 of the reference)."""
    group = tp_group()
    dst = dist.get_global_rank(group, 0) if group is not None else 0
    if t.is_cuda and _host_hop(group):
        host = t.float().cpu()
        parts = [torch.empty_like(host) for _ in range(tp_size)] if tp_rank == 0 else None
        dist.gather(host, parts, dst, group=group)
        return [p.to(device=t.device, dtype=t.dtype) for p in parts] if tp_rank == 0 else None
    parts = [torch.empty_like(t) for _ in range(tp_size)] if tp_rank == 0 else None
    dist.gather(t, parts, dst, group=group)
    return parts


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
        """Build the kernel-side weight.  The row-major parameter is freed whenever the kernel-side weight is a copy of
        it (quantised, or unquantised in the streaming layout): one copy of every matrix stays in device memory."""
        self.packed = self._pack(lambda t: t, quantization)
        if self.weight is not None and self.packed.data.data_ptr() != self.weight.data_ptr():
            self.weight = None

    def _pack(self, reorder, quantization):
        """The kernel-side weight of this rank's shard: quantised here from the dense parameter, or taken as it was
        quantised offline (load_quantized); ``reorder`` permutes rows (gate/up interleaving)."""
        self._finish_quantized()
        q = getattr(self, "_quantized", None)
        if q is not None:
            assert quantization == q[2], f"checkpoint holds {q[2]} weights, the engine was asked for {quantization}"
            self._quantized = None
            return LinearWeight.from_quantized(reorder(q[0]), reorder(q[1]), q[2])
        return LinearWeight.make(reorder(self.weight.data), quantization)

    def load_quantized(self, values: torch.Tensor, scales: torch.Tensor, quantization: str, *shard_id) -> None:
        """Shard an offline-quantised matrix (values int8 [N, K], scales [N, K/128], both of the FULL layer) exactly as
        weight_loader shards the dense one: the loaders only move rows and 128-aligned column blocks around, so they
        are run once on the values and once on the scales spread to one per element; rows and columns the rank pads
        with zeros get the zero weight and the scale a zero group quantises to."""
        q = getattr(self, "_quantized", None)
        if q is None:
            # staged where the dense parameter lives (the model's device): the kernel-side weight built from these
            # in _pack must be device memory, whatever device the checkpoint reader handed the tensors on
            shape, device = self.weight.shape, self.weight.device
            q = self._quantized = (torch.zeros(shape, dtype=torch.int8, device=device),
                                   torch.zeros(shape, dtype=scales.dtype, device=device), quantization)
        self.weight_loader(q[0], values, *shard_id)
        self.weight_loader(q[1], scales.repeat_interleave(128, dim=1), *shard_id)

    def _finish_quantized(self) -> None:
        """One scale per element -> one per local group of 128.  A local group normally lies inside one group of the
        full matrix (plus zero padding) and keeps that group's scale and values.  Where the shard boundaries are not
        128-aligned in the full matrix (64-wide heads dealt unevenly) a local group can straddle two full groups with
        different scales: such groups are re-quantised from their dequantised values, as load-time quantisation of the
        dense shard would do."""
        q = getattr(self, "_quantized", None)
        if q is None or q[1].size(1) != q[0].size(1):
            return
        values, per_elem, quantization = q
        n, k = values.shape
        g = per_elem.float().view(n, k // 128, 128)
        top = g.amax(-1)
        low = torch.where(g == 0, torch.full_like(g, float("inf")), g).amin(-1)
        mixed = (top != low) & (top != 0)
        scales = top.to(per_elem.dtype)
        # an all-zero group's scale: what quantising a zero group gives (clamp_min(1e-8) rounded to the model dtype)
        scales[top == 0] = torch.tensor(1e-8, dtype=torch.float32, device=per_elem.device).to(per_elem.dtype)
        if mixed.any():
            qmax = 7 if quantization == "w4a16" else 127
            w = (values.float().view(n, k // 128, 128) * g)[mixed].to(per_elem.dtype).float()   # the dense weight's value
            s_new = (w.abs().amax(-1) / qmax).clamp_min(1e-8).to(per_elem.dtype)
            v_new = torch.clamp(torch.round(w / s_new.float().unsqueeze(-1)), -8 if qmax == 7 else -127, qmax)
            values = values.clone()
            values.view(n, k // 128, 128)[mixed] = v_new.to(torch.int8)
            scales[mixed] = s_new
        self._quantized = (values, scales, quantization)

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
    def __init__(self, input_size: int, output_sizes: list[int], bias: bool = False, align: int = 1):
        """align: shard every packed matrix in units of ``align`` rows (the MLP passes 128 so that the matching
        row-parallel K shards are whole K-blocks); uneven shares are zero-padded to the largest one."""
        self.output_sizes = output_sizes
        self.fused_silu = False
        tp_rank, tp_size = _tp()
        self.align = align
        # per packed matrix: (first source row, rows owned, rows allocated)
        self.shards = []
        for size in output_sizes:
            assert size % align == 0, f"output size {size} is not a multiple of {align}"
            start, count, padded = split_units(size // align, tp_size, tp_rank)
            self.shards.append((start * align, count * align, padded * align))
        LinearBase.__init__(self, input_size, sum(p for _, _, p in self.shards), bias, 0)

    def finalize(self, quantization: str | None = None, fuse_silu: bool = False) -> None:
        """fuse_silu: store gate/up rows interleaved so forward_silu_mul() can apply SiluAndMul in the GEMM epilogue."""
        from byox_vllm_amd.layers.quant import interleave_gate_up
        self.fused_silu = fuse_silu and self.bias is None and len(self.output_sizes) == 2
        self.packed = self._pack(interleave_gate_up if self.fused_silu else (lambda t: t), quantization)
        if self.weight is not None and self.packed.data.data_ptr() != self.weight.data_ptr():
            self.weight = None

    def forward(self, x, out=None, partial_out=False):
        assert not self.fused_silu, "weights are stored for the fused epilogue: call forward_silu_mul"
        return super().forward(x, out=out, partial_out=partial_out)

    def forward_silu_mul(self, x, out=None):
        """silu(gate) * up in one launch -> [T, intermediate / tp]."""
        assert self.fused_silu
        return self._gemm(x, None, out=out, silu_mul=True)

    def weight_loader(self, param: nn.Parameter, loaded_weight: torch.Tensor, loaded_shard_id: int):
        offset = sum(p for _, _, p in self.shards[:loaded_shard_id])
        start, count, padded = self.shards[loaded_shard_id]
        copy_segments(param.data.narrow(self.tp_dim, offset, padded), loaded_weight, self.tp_dim, [(start, count, 0)])


class QKVParallelLinear(ColumnParallelLinear):
    def __init__(self, hidden_size: int, head_size: int, total_num_heads: int,
                 total_num_kv_heads: int | None = None, bias: bool = False):
        tp_rank, tp_size = _tp()
        total_num_kv_heads = total_num_kv_heads or total_num_heads
        self.head_size = head_size
        self.layout = HeadLayout(total_num_heads, total_num_kv_heads, head_size, tp_size, tp_rank)
        self.num_heads = self.layout.num_heads          # local query slots (zero heads included)
        self.num_kv_heads = self.layout.num_kv_heads
        LinearBase.__init__(self, hidden_size, (self.num_heads + 2 * self.num_kv_heads) * head_size, bias, 0)

    def weight_loader(self, param: nn.Parameter, loaded_weight: torch.Tensor, loaded_shard_id: str):
        assert loaded_shard_id in ("q", "k", "v")
        q_rows, kv_rows = self.num_heads * self.head_size, self.num_kv_heads * self.head_size
        offset, size = {"q": (0, q_rows), "k": (q_rows, kv_rows), "v": (q_rows + kv_rows, kv_rows)}[loaded_shard_id]
        heads = self.layout.q_heads if loaded_shard_id == "q" else self.layout.kv_heads
        copy_segments(param.data.narrow(self.tp_dim, offset, size), loaded_weight, self.tp_dim, self.layout.segments(heads))


class RowParallelLinear(LinearBase):
    def __init__(self, input_size: int, output_size: int, bias: bool = False, align: int = 1,
                 input_segments=None, local_input_size: int | None = None):
        """Input columns are sharded in units of ``align`` (uneven shares zero-padded, see split_units), or taken
        from explicit ``input_segments`` [(source column, columns, local column)] of width ``local_input_size``
        (o_proj follows the attention block's HeadLayout)."""
        tp_rank, tp_size = _tp()
        if input_segments is None:
            assert input_size % align == 0, f"input size {input_size} is not a multiple of {align}"
            start, count, padded = split_units(input_size // align, tp_size, tp_rank)
            input_segments, local_input_size = [(start * align, count * align, 0)], padded * align
        self.input_segments = input_segments
        super().__init__(local_input_size, output_size, bias, 1)

    def weight_loader(self, param: nn.Parameter, loaded_weight: torch.Tensor):
        if param.dim() == 1:   # a bias is not sharded along the input dimension
            param.data.copy_(loaded_weight)
            return
        copy_segments(param.data, loaded_weight, self.tp_dim, self.input_segments)

    def forward(self, x, out=None, partial_out=False):
        """partial_out (single rank only): return the split-K slabs for the following add_rmsnorm to sum."""
        bias = self.bias.data if (self.bias is not None and self.tp_rank == 0) else None
        if partial_out and not collectives_on(self.tp_size) and bias is None:
            return self._gemm(x, None, partial_out=True)
        if partial_out and self.tp_size > 1 and _COMM is not None:
            # the sum over ranks is taken by the consumer: the all-reduce fused with the following add-RMSNorm
            return kernels.PendingReduce(self._gemm(x, None, partial_out=True) if bias is None else self._gemm(x, bias))
        y = self._gemm(x, bias, out=out)
        if collectives_on(self.tp_size):
            tp_all_reduce(y)
        return y
