"""All-reduce over peer-mapped device memory, fused with residual add + RMSNorm (nv_comm_* / nv_allreduce*).

Stands where the reference calls ``dist.all_reduce`` after a row-parallel linear (nanovllm/layers/linear.py:145-146)
and then ``RMSNorm.add_rms_forward`` (layers/layernorm.py:28-40).  One ``XgmiComm`` per rank; the ranks exchange the
buffers' inter-process handles once (here through torch.distributed's object all-gather, any host channel would do)
and afterwards talk through device memory only, so the calls can live inside a captured decode graph.
"""

import ctypes as C

import torch

from byox_vllm_amd import kernels
from byox_vllm_amd._C import check, dtype_code, lib, ptr, stream


class XgmiComm:
    def __init__(self, rank: int, world: int, max_rows: int, hidden: int):
        self.rank, self.world, self.max_rows, self.hidden = rank, world, max_rows, hidden
        handle = C.c_void_p()
        check(lib().nv_comm_init(C.byref(handle), rank, world, max_rows, hidden), "nv_comm_init")
        self._h = handle

    # -- wiring ---------------------------------------------------------------------------------------------------
    def export_handle(self) -> bytes:
        buf = C.create_string_buffer(64)
        check(lib().nv_comm_export(self._h, buf), "nv_comm_export")
        return buf.raw

    def open_peer(self, peer: int, handle: bytes) -> None:
        check(lib().nv_comm_open_peer(self._h, peer, C.create_string_buffer(handle, 64)), "nv_comm_open_peer")

    def set_peer(self, peer: int, other: "XgmiComm") -> None:
        """Peer living in the same process (tests; one process driving several devices)."""
        check(lib().nv_comm_set_peer(self._h, peer, lib().nv_comm_local_base(other._h)), "nv_comm_set_peer")

    def connect(self, group=None) -> None:
        """Exchange the buffers' handles over ``group`` (a torch.distributed group whose ranks are the comm's)."""
        import torch.distributed as dist
        handles = [None] * self.world
        dist.all_gather_object(handles, self.export_handle(), group=group)
        for peer, h in enumerate(handles):
            if peer != self.rank:
                self.open_peer(peer, h)
        dist.barrier(group=group)

    @property
    def memory_kind(self) -> str:
        return {2: "uncached", 1: "fine-grained", 0: "plain"}[lib().nv_comm_memory_kind(self._h)]

    def error(self) -> int:
        return lib().nv_comm_error(self._h)

    def close(self) -> None:
        if self._h is not None:
            lib().nv_comm_destroy(self._h)
            self._h = None

    def owned_rows(self, rows: int) -> tuple[int, int]:
        per = -(-rows // self.world)
        return min(rows, self.rank * per), min(rows, (self.rank + 1) * per)

    # -- data path ------------------------------------------------------------------------------------------------
    def _source(self, x):
        if isinstance(x, kernels.Partials):
            assert x.bias is None
            return None, x.slabs, x.n, x.rows, x.cols, x.dtype, x.slabs.device
        assert x.is_contiguous() and x.dim() == 2
        return x, None, 0, x.size(0), x.size(1), x.dtype, x.device

    def all_reduce(self, x, out=None):
        """out = sum over ranks of x (a tensor [rows, hidden] or the Partials of a split-K GEMM)."""
        xt, slabs, n, rows, cols, dt, dev = self._source(x)
        out = torch.empty(rows, cols, dtype=dt, device=dev) if out is None else out
        check(lib().nv_allreduce(self._h, ptr(xt), ptr(slabs), n, ptr(out), rows, cols, dtype_code(dt), stream()),
              "nv_allreduce")
        return out

    def all_reduce_add_rms_norm(self, x, residual, weight, eps, out=None):
        """-> (RMSNorm(sum over ranks of x + residual) * weight for all rows, residual updated in place in the rows
        this rank owns)."""
        xt, slabs, n, rows, cols, dt, dev = self._source(x)
        assert residual.is_contiguous() and residual.shape == (rows, cols) and residual.dtype == dt == weight.dtype
        out = torch.empty(rows, cols, dtype=dt, device=dev) if out is None else out
        check(lib().nv_allreduce_add_rmsnorm(self._h, ptr(xt), ptr(slabs), n, ptr(residual), ptr(weight), ptr(out),
                                             ptr(residual), rows, cols, eps, dtype_code(dt), stream()),
              "nv_allreduce_add_rmsnorm")
        return out, residual
