"""Offline weight-only quantisation of a checkpoint (the reference has no quantised path: nanovllm/layers/linear.py is
dense F.linear, nanovllm/utils/loader.py:13-29 loads dense safetensors; format: byox_vllm_amd/layers/quant.py).

    python -m byox_vllm_amd.utils.quantize SRC_DIR DST_DIR {w4a16,w8a16}

Every layer linear (q/k/v/o/gate/up/down_proj) of the HF-named safetensors under SRC_DIR is written to DST_DIR as
``<name>.qweight`` (int8 values: int4 in -8..7, int8 in -127..127, the whole unsharded matrix) and ``<name>.scales``
([N, K/128], model dtype); everything else (embedding, norms, biases, LM head) is copied; ``config.json`` gains
``"quantization"``.  utils.loader.load_model shards those tensors like the dense ones, so a rank ends up with the bytes
it would have produced by quantising its dense shard at load time, without ever holding the dense model.
"""

import json
import os
import shutil
import sys
from glob import glob

import torch

from byox_vllm_amd.layers.quant import LinearWeight

QUANTISED = ("q_proj.weight", "k_proj.weight", "v_proj.weight", "o_proj.weight", "gate_proj.weight", "up_proj.weight",
             "down_proj.weight")


def quantize_tensor(w: torch.Tensor, quantization: str):
    """-> (int8 values [N, K], scales [N, K/128]); N is padded to a multiple of 16 only inside this call."""
    n, k = w.shape
    pad = (-n) % 16
    if pad:
        w = torch.cat((w, w.new_zeros(pad, k)), 0)
    lw = LinearWeight.int4(w) if quantization == "w4a16" else LinearWeight.int8(w)
    values, scales = lw.quantized_values()
    return values[:n].contiguous(), scales[:n].contiguous()


def quantize_checkpoint(src: str, dst: str, quantization: str) -> None:
    from safetensors import safe_open
    from safetensors.torch import save_file
    assert quantization in ("w4a16", "w8a16")
    os.makedirs(dst, exist_ok=True)
    for file in sorted(glob(os.path.join(src, "*.safetensors"))):
        out = {}
        with safe_open(file, "pt", "cpu") as f:
            for name in f.keys():  # noqa: SIM118
                t = f.get_tensor(name)
                if name.endswith(QUANTISED) and t.dim() == 2 and t.size(1) % 128 == 0:
                    base = name[: -len("weight")]
                    out[base + "qweight"], out[base + "scales"] = quantize_tensor(t, quantization)
                else:
                    out[name] = t
        save_file(out, os.path.join(dst, os.path.basename(file)))
    for extra in os.listdir(src):
        if not extra.endswith(".safetensors") and os.path.isfile(os.path.join(src, extra)) and extra != "config.json":
            shutil.copy(os.path.join(src, extra), os.path.join(dst, extra))
    cfg_path = os.path.join(src, "config.json")
    cfg = json.load(open(cfg_path)) if os.path.exists(cfg_path) else {}
    cfg["quantization"] = quantization
    with open(os.path.join(dst, "config.json"), "w") as f:
        json.dump(cfg, f, indent=1)


if __name__ == "__main__":
    quantize_checkpoint(sys.argv[1], sys.argv[2], sys.argv[3])
    print("wrote", sys.argv[2])
