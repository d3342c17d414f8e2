#!/usr/bin/env python
"""Compile one csrc/*.hip for gfx950 with -Rpass-analysis=kernel-resource-usage and print name / VGPRs / scratch / LDS per
kernel (build container; no GPU needed):  python tools/kernel_resources.py conv3d.hip [filter]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "multimodal-registration_amd", "csrc", sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:] 
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-c", src, "-o", "/tmp/_kr.o",
                    "-Rpass-analysis=kernel-resource-usage"] + extra, capture_output=True, text=True)
cur = {}
rows = []
for ln in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    for key in ("VGPRs", "AGPRs", "SGPRs", "ScratchSize [bytes/lane]", "LDS Size [bytes/block]", "Occupancy [waves/SIMD]"):
        m = re.search(re.escape(key) + r": (\d+)", ln)
        if m and cur is not None:
            cur[key.split(" ")[0]] = int(m.group(1))
for c in rows:
    if flt in c["name"]:
        print(f"{c.get('VGPRs', '?'):>4} vgpr {c.get('ScratchSize', '?'):>4} scratch {c.get('Occupancy', '?')} occ  {c['name'][:150]}")
if r.returncode:
    print(r.stderr[-3000:])
