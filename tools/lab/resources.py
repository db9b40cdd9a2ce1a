#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output: one line per kernel (name, SGPR, VGPR, AGPR, spill, LDS, occupancy)."""
import re, subprocess, sys
rows, cur = [], None
for line in open(sys.argv[1], errors="replace"):
    m = re.search(r"remark: +(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (.*?) \[-Rpass", line)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    elif cur is not None:
        cur[k] = v
for r in rows:
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", r["name"]], capture_output=True, text=True).stdout.strip()
    except Exception:
        name = r["name"]
    name = re.sub(r"void |pxr::|\(.*\)$", "", name)
    print(f"{name[:110]:110s} sgpr {r.get('TotalSGPRs','?'):>3} vgpr {r.get('VGPRs','?'):>3} agpr {r.get('AGPRs','?'):>3} scratch {r.get('ScratchSize [bytes/lane]','?'):>4} vspill {r.get('VGPRs Spill','?'):>3} sspill {r.get('SGPRs Spill','?'):>3} occ {r.get('Occupancy [waves/SIMD]','?')}")
