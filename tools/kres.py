#!/usr/bin/env python3
"""VGPRs / scratch bytes per kernel of one csrc file (hipcc -Rpass-analysis=kernel-resource-usage).   usage: python tools/kres.py gru_persist.hip [name filter]"""
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "m3f.pytorch_amd", "csrc", sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(root, "include"),
                      "-fno-gpu-rdc", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                     capture_output=True, text=True, cwd=os.path.dirname(src))
if out.returncode != 0:
    sys.exit("hipcc failed:\n" + "\n".join(l for l in out.stderr.splitlines() if "error" in l)[:4000])
out = out.stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(anonymous namespace\)::|m3t_gru::|void ", "", cur).split("(")[0]
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(SGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|VGPRs Spill|SGPRs Spill): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).split(" [")[0]] = int(m.group(2))
for k, v in rows.items():
    if flt in k:
        print("%-60s VGPR %3d AGPR %3d scratch %4d  SGPR %3d  LDS %6d" % (k[:60], v.get("VGPRs", -1), v.get("AGPRs", -1), v.get("ScratchSize", -1), v.get("SGPRs", -1), v.get("LDS Size", -1)))
