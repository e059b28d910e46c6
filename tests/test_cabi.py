"""The C-ABI library builds, loads and exports every symbol include/nv_hip.h declares (no compute without a GPU)."""

import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "nv_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nv_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as entry
    entry.build()
    from byox_vllm_amd import _C
    handle = ctypes.CDLL(_C.LIB_PATH)
    declared = _declared()
    assert len(declared) >= 16
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in nv_hip.h but not exported"
    assert sorted(_C.exported_symbols()) == declared, "ctypes prototypes and header disagree"


def test_version_and_error_string_without_gpu():
    from byox_vllm_amd import _C
    lib = _C.lib()
    assert lib.nv_version() >= 100
    # argument validation happens on the host, before any launch: callable without a GPU
    code = lib.nv_rmsnorm(None, None, None, 1, 128, 128, 128, 1e-6, 0, None)
    assert code == -1 and b"null pointer" in lib.nv_last_error()
    assert lib.nv_gemm_workspace_bytes(128, 896, 7) == 7 * 128 * 896 * 4
    assert lib.nv_paged_attn_decode_workspace_bytes(128, 14, 64, 1) == 0
    assert lib.nv_paged_attn_decode_pick_splits(128, 2, 1024) >= 1


def test_product_has_no_cpu_path():
    import pytest
    import torch
    from byox_vllm_amd import kernels
    with pytest.raises(RuntimeError, match="no CPU compute path"):
        kernels.silu_mul(torch.zeros(2, 16, dtype=torch.bfloat16))
