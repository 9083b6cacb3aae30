#!/usr/bin/env python3
"""Registers, scratch and occupancy of every kernel of zrk_hot.hip as the compiler reports them
(-Rpass-analysis=kernel-resource-usage), one line per kernel.   python tools/kernel_resources.py [filter]"""
import re
import subprocess
import sys
from pathlib import Path

root = Path(__file__).resolve().parent.parent
src = root / "zrk_modulation_amd" / "csrc" / "zrk_hot.hip"
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", f"-I{root / 'include'}", "-c", str(src),
       "-o", "/tmp/zrk_res.o", "-Rpass-analysis=kernel-resource-usage"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
pat = sys.argv[1] if len(sys.argv) > 1 else ""
cur = None
rows = []
for line in err.splitlines():
    m = re.search(r"remark:\s*([A-Za-z /\[\]]+?): (.*?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        name = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        cur = {"name": name}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
for r in rows:
    if pat in r["name"]:
        print(f"{r['name']:<62} VGPR {r.get('VGPRs', '?'):>4} AGPR {r.get('AGPRs', '?'):>3} SGPR {r.get('TotalSGPRs', '?'):>4} (spilt {r.get('SGPRs Spill', '?')}) "
              f"scratch {r.get('ScratchSize [bytes/lane]', '?'):>4} occupancy {r.get('Occupancy [waves/SIMD]', '?'):>2} LDS {r.get('LDS Size [bytes/block]', '?'):>6}")
