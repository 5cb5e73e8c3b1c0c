#!/usr/bin/env python3
"""Kernel resource usage of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel:
   python tools/kres.py mpntrackseg_amd/csrc/edge_chain_bf16.hip [name filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage",
                    "-c", src, "-o", "/dev/null"] + sys.argv[3:], capture_output=True, text=True)
cur = None
rows = []
for line in r.stderr.splitlines():
    m = re.search(r"remark: .*?Function Name: (\S+)", line) or re.search(r"Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    for key in ("VGPRs", "AGPRs", "SGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]", "VGPR Spill", "SGPR Spill"):
        m = re.search(re.escape(key) + r": (\d+)", line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
for c in rows:
    name = subprocess.run(["c++filt", c["name"]], capture_output=True, text=True).stdout.strip()
    if flt and flt not in name:
        continue
    print("%-110s vgpr %3s agpr %3s sgpr %3s scratch %4s occ %s lds %6s" % (name[:110], c.get("VGPRs"), c.get("AGPRs"), c.get("SGPRs"),
          c.get("ScratchSize [bytes/lane]"), c.get("Occupancy [waves/SIMD]"), c.get("LDS Size [bytes/block]")))
if r.returncode != 0:
    print(r.stderr[-3000:])
