"""VGPR / spill / occupancy table of one .hip file's kernels: python scripts/kernel_resources.py <file.hip> [filter]."""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-c", src, "-o", "/tmp/_kr.o",
                      "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = {}
for line in out.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+: +(.*?): +(.*?) \[-Rpass", line) or re.search(r"remark: +(.*?): +(.*?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": v}
    cur[k] = v
    if k.startswith("LDS Size") and flt in cur["name"]:
        print(f'{cur["name"][:64]:64s} VGPR {cur.get("VGPRs")} AGPR {cur.get("AGPRs")} spill {cur.get("VGPRs Spill")} '
              f'scratch {cur.get("ScratchSize [bytes/lane]")} occ {cur.get("Occupancy [waves/SIMD]")} LDS {v}')
