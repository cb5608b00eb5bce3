#!/usr/bin/env python3
"""Print VGPR/AGPR/scratch/occupancy/LDS of every kernel in a .hip file (hipcc -Rpass-analysis)."""
import re
import subprocess
import sys

src = sys.argv[1]
out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-c", src, "-o", "/dev/null", "-I", "include",
                      "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:], capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::", "", name)[:70]
    print(f"{name:70s} vgpr {v.get('VGPRs', 0):4d} agpr {v.get('AGPRs', 0):4d} scratch {v.get('ScratchSize', 0):4d} "
          f"occ {v.get('Occupancy', 0)} lds {v.get('LDS Size', 0):6d} spill {v.get('VGPRs Spill', 0)}")
