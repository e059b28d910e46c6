"""ctypes binding of libnv_hip.so (the C ABI declared in include/nv_hip.h).

There is no fallback: if the library is missing, or a call returns an error
code, a RuntimeError is raised.  Tensors are passed as raw device pointers;
PyTorch only owns the memory and the stream.
"""

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# NV_HIP_LIB points at another build of the same library (A/B measurements); the default is the in-tree build.
LIB_PATH = os.environ.get("NV_HIP_LIB") or os.path.join(_HERE, "libnv_hip.so")

NV_BF16, NV_F16 = 0, 1
W_DENSE, W_INT8, W_INT4, W_DENSE_TILED = 0, 1, 2, 3
GEMM_SILU_MUL, GEMM_PARTIAL_OUT = 1, 2

_p, _i, _i64, _f, _sz, _u64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t, C.c_uint64

_PROTOS = {
    "nv_version": (C.c_int, []),
    "nv_last_error": (C.c_char_p, []),
    "nv_rmsnorm": (_i, [_p, _p, _p, _i64, _i, _i64, _i64, _f, _i, _p]),
    "nv_add_rmsnorm": (_i, [_p, _p, _p, _p, _p, _i64, _i, _f, _i, _p]),
    "nv_add_rmsnorm_splitk": (_i, [_p, _i, _p, _p, _p, _p, _i64, _i, _f, _i, _p]),
    "nv_rope_kv_store_splitk": (_i, [_p, _i, _p, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _f, _i64, _i, _i, _i, _i, _p]),
    "nv_rope_kv_store": (_i, [_p, _i64, _p, _p, _p, _p, _p, _p, _p, _f, _i64, _i, _i, _i, _i, _p]),
    "nv_kv_store": (_i, [_p, _i64, _p, _i64, _p, _p, _p, _i64, _i, _i, _p]),
    "nv_paged_attn_decode_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "nv_paged_attn_decode_pick_splits": (_i, [_i, _i, _i]),
    "nv_paged_attn_decode_kernel_tag": (C.c_char_p, []),
    "nv_set_result_policy": (_i, [_i]),
    "nv_get_result_policy": (_i, []),
    "nv_paged_attn_decode": (_i, [_p, _i64, _p, _p, _p, _i, _p, _p, _i64, _p, _sz, _i, _i, _i, _i, _i, _f, _i, _i, _p]),
    "nv_paged_attn_decode_static": (_i, [_p, _i64, _p, _p, _p, _i, _p, _p, _i64, _i, _i, _i, _i, _i, _f, _i, _i, _p]),
    "nv_attn_prefill": (_i, [_p, _i64, _p, _p, _i64, _p, _p, _p, _i, _p, _i64, _i, _i, _i, _i, _i, _i, _f, _i, _p]),
    "nv_silu_mul": (_i, [_p, _p, _i64, _i, _i, _p]),
    "nv_gemm_lt_available": (_i, []),
    "nv_gemm_lt": (_i, [_p, _i64, _p, _p, _p, _i64, _i, _i, _i, _i, _p, _sz, _p]),
    "nv_weight_expand": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _p]),
    "nv_gemm_workspace_bytes": (_sz, [_i64, _i, _i]),
    "nv_gemm_pick_split_k": (_i, [_i64, _i, _i, _i]),
    "nv_gemm": (_i, [_p, _i64, _p, _p, _p, _p, _i64, _i64, _i, _i, _i, _i, _p, _sz, _i, _i, _p]),
    "nv_gemm_rope_kv_store_is_faster": (_i, [_i64, _i, _i, _i, _i, _i]),
    "nv_gemm_rope_kv_store": (_i, [_p, _i64, _p, _p, _p, _p, _i64, _i64, _i, _i, _p, _p, _p, _p, _p, _p, _p, _f, _i, _i, _i, _i, _p]),
    "nv_comm_buffer_bytes": (_sz, [_i, _i, _i]),
    "nv_comm_init": (_i, [C.POINTER(C.c_void_p), _i, _i, _i, _i]),
    "nv_comm_memory_kind": (_i, [_p]),
    "nv_comm_local_base": (_p, [_p]),
    "nv_comm_export": (_i, [_p, _p]),
    "nv_comm_open_peer": (_i, [_p, _i, _p]),
    "nv_comm_set_peer": (_i, [_p, _i, _p]),
    "nv_comm_error": (_i, [_p]),
    "nv_comm_destroy": (_i, [_p]),
    "nv_allreduce": (_i, [_p, _p, _p, _i, _p, _i64, _i, _i, _p]),
    "nv_allreduce_add_rmsnorm": (_i, [_p, _p, _p, _i, _p, _p, _p, _p, _i64, _i, _f, _i, _p]),
    "nv_embed": (_i, [_p, _p, _p, _i64, _i, _i64, _i64, _i, _p]),
    "nv_sample_workspace_bytes": (_sz, [_i, _i]),
    "nv_lm_head_sample_workspace_bytes": (_sz, [_i64, _i]),
    "nv_lm_head_sample": (_i, [_p, _i64, _p, _i, _p, _i64, _p, _p, _p, _sz, _i64, _i, _i, _u64, _p, _i, _p]),
    "nv_sample": (_i, [_p, _i64, _p, _p, _p, _p, _sz, _i, _i, _u64, _p, _i, _p]),
    "nv_stage_decode_packed": (_i, [_p, _i, _i, _i, _p, _p, _p, _p, _p, _i, _p, _p, _p]),
    "nv_stage_decode_packed_advance": (_i, [_p, _i, _p, _p, _p, _p, _p, _i, _p, _p, _p, _i, _p]),
    "nv_stage_decode_inputs": (_i, [_p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
}

_lib = None


def exported_symbols():
    return list(_PROTOS)


def lib():
    """Load the shared library once; fail loudly if it was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for the HIP path.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(code: int, what: str) -> None:
    if code != 0:
        raise RuntimeError(f"{what} failed ({code}): {lib().nv_last_error().decode()}")


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.bfloat16:
        return NV_BF16
    if dt == torch.float16:
        return NV_F16
    raise TypeError(f"the HIP path computes in bf16 or fp16, got {dt}")


def ptr(t):
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream() -> int:
    """The current HIP stream of the current device as an integer handle.  torch.cuda.current_stream() builds a Python
    Stream object on every call (about 8 us, once per launch); the raw getter is a plain C call."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def require_gpu(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{what}: tensor is on {t.device}; byox_vllm_amd has no CPU compute path")
