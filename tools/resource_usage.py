#!/usr/bin/env python3
"""Per-kernel VGPR / SGPR / scratch / LDS / occupancy of one .hip source (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "rust-renderer_amd/csrc/kernels.hip")
pat = sys.argv[2] if len(sys.argv) > 2 else "."
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-I", os.path.join(ROOT, "include"),
       "-I", os.path.join(ROOT, "rust-renderer_amd/csrc"), "-c", src, "-o", "/tmp/_ru.o", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: (?:\s*)(Function Name|VGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().split("(")[0]
        rows[cur] = {}
    elif cur:
        rows[cur][k.split(" ")[0]] = v
print("%-70s %5s %5s %8s %4s %7s" % ("kernel", "VGPR", "SGPR", "scratch", "occ", "LDS"))
for k, r in rows.items():
    if re.search(pat, k):
        print("%-70s %5s %5s %8s %4s %7s" % (k[:70], r.get("VGPRs"), r.get("SGPRs"), r.get("ScratchSize"), r.get("Occupancy"), r.get("LDS")))
