#!/usr/bin/env python3
"""Compile-time resource table of every kernel of both translation units (hipcc -Rpass-analysis=kernel-resource-usage; no GPU needed):
VGPRs / AGPRs / scratch bytes per lane / SGPR and VGPR spills / waves per SIMD.   python tools/kernel_resources.py"""
import os
import re
import subprocess

CS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mopa_rl_amd", "csrc")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -mfma -Rpass-analysis=kernel-resource-usage -c -o /dev/null".split()
rows, cur = {}, None
for tu in ("mopa_hip.hip", "mopa_envdyn.hip"):
    txt = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [tu], cwd=CS, capture_output=True, text=True).stderr
    for line in txt.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            rows[cur] = {}
        for k, pat in (("v", "VGPRs:"), ("a", "AGPRs:"), ("scr", r"ScratchSize \[bytes/lane\]:"), ("ss", "SGPRs Spill:"), ("vs", "VGPRs Spill:"), ("occ", r"Occupancy \[waves/SIMD\]:")):
            m = re.search(pat + r" (\d+)", line)
            if m and cur:
                rows[cur][k] = int(m.group(1))
names = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.split("\n")
out = set()
for n, r in zip(names, rows.values()):
    out.add(f"{n[:64]:66s} VGPR {r.get('v', 0):3d}  AGPR {r.get('a', 0):3d}  scratch {r.get('scr', 0):4d} B  SGPR spills {r.get('ss', 0):3d}  VGPR spills {r.get('vs', 0):3d}  waves/SIMD {r.get('occ')}")
print("\n".join(sorted(out)))
