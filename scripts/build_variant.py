"""Build a measurement variant of the library: python scripts/build_variant.py NAME SOURCE[,SOURCE...] -DFLAG[=V] ...

The named csrc sources (e.g. attn, or gemm_bf16,gemm_bf16_q,gemm_bf16_wt for every bf16 GEMM) are recompiled with the extra -D flags into objects under /tmp and linked
with the in-tree objects of all other sources into byox_vllm_amd/abl_NAME.so (git-ignored; it travels to the GPU box).
Select it at run time with NV_HIP_LIB=byox_vllm_amd/abl_NAME.so, so one GPU call can time several variants of a
kernel against the shipped build inside the same process environment.
"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "byox_vllm_amd", "csrc")


def main():
    name, sources, flags = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
    os.makedirs(f"/tmp/variant_{name}", exist_ok=True)
    objs, procs = [], []
    for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
        base = os.path.basename(src)[:-4]
        if base in sources:
            obj = f"/tmp/variant_{name}/{base}.o"
            cmd = ["hipcc", "-O3", "-std=c++17", "-fno-slp-vectorize", "--offload-arch=gfx950", "-fPIC", *flags, "-c", src, "-o", obj]
            procs.append((src, subprocess.Popen(cmd)))
        else:
            obj = src[:-4] + ".o"
            assert os.path.exists(obj), f"{obj} missing: run __graft_entry__.build() first"
        objs.append(obj)
    for src, p in procs:
        if p.wait() != 0:
            raise SystemExit(f"hipcc failed on {src}")
    out = os.path.join(ROOT, "byox_vllm_amd", f"abl_{name}.so")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    print("built", out)


if __name__ == "__main__":
    main()
