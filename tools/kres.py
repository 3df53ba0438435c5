#!/usr/bin/env python3
"""Per-kernel register / scratch / spill summary of rt_kernels.hip as compiled with the shipped flags
(`make asm` output parsed; no GPU needed).  Usage: tools/kres.py [extra make vars, e.g. WAVES=5]"""
import os
import re
import subprocess
import sys

csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hslu_i", "ba_raytracing", "f2501_raytracer_amd", "csrc")
out = subprocess.run(["make", "-C", csrc, "-B", "asm"] + sys.argv[1:], capture_output=True, text=True)
txt = out.stderr + out.stdout
rows, cur = [], None
for line in txt.splitlines():
    m = re.search(r"remark: +Function Name: (\S+)", line)
    if m:
        cur = {"name": re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", m.group(1)).split("E1")[0][:28]}
        rows.append(cur)
        continue
    m = re.search(r"remark: +([A-Za-z \[\]/]+): (\S+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = m.group(2)
keys = ["TotalSGPRs", "VGPRs", "SGPRs Spill", "VGPRs Spill", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"]
print(f"{'kernel':30s}" + "".join(f"{k.split(' [')[0]:>14s}" for k in keys))
for r in rows:
    print(f"{r['name']:30s}" + "".join(f"{r.get(k, '-'):>14s}" for k in keys))
if out.returncode:
    print(txt[-3000:])
    sys.exit(out.returncode)
