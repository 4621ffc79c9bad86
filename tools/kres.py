#!/usr/bin/env python3
"""Dev helper: per-kernel VGPR / scratch / LDS table from hipcc -Rpass-analysis=kernel-resource-usage.
usage: python tools/kres.py hcatgnet_amd/csrc/fused.hip [filter]"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
                      "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + sys.argv[3:],
                     capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur.replace("(anonymous namespace)::", "").replace("void ", ""))
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|VGPRs Spill|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).split(" [")[0].replace("VGPRs Spill", "Spill").split(" ")[0]] = int(m.group(2))
    if "error" in line:
        print(line)
for k, v in rows.items():
    if flt in k:
        print(f"{k:50s} vgpr {v.get('VGPRs', 0):4d} agpr {v.get('AGPRs', 0):4d} scratch {v.get('ScratchSize', 0):5d} lds {v.get('LDS', 0):7d} spill {v.get('Spill', 0)} occ {v.get('Occupancy', 0)}")
