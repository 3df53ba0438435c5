#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table of the shipped kernels: `make asm` remarks -> one line per kernel.

    python tools/kernel_resources.py            (runs `make asm` in csrc/ and prints the table)
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hslu_i", "ba_raytracing", "f2501_raytracer_amd", "csrc")


def table(extra=()):
    out = subprocess.run(["make", "-C", CSRC, "-B", "asm", *extra], capture_output=True, text=True)
    txt = out.stdout + out.stderr
    rows, cur = [], None
    for line in txt.splitlines():
        m = re.search(r"remark: (.*?) \[-Rpass-analysis", line)
        if not m:
            if "error" in line:
                print(line, file=sys.stderr)
            continue
        k, _, v = m.group(1).partition(":")
        if k.strip() == "Function Name":
            cur = {"name": re.sub(r"^_ZN\d+_GLOBAL__N_1\d+|E\d+RtDev.*$|EPKf.*$", "", v.strip())}
            rows.append(cur)
        elif cur is not None:
            cur[k.strip()] = v.strip()
    return rows


if __name__ == "__main__":
    rows = table(sys.argv[1:])
    cols = ["VGPRs", "SGPRs", "ScratchSize [bytes/lane]", "SGPRs Spill", "VGPRs Spill", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"]
    print("| kernel | VGPRs | SGPRs | scratch B/lane | SGPR spills | VGPR spills | waves/SIMD | LDS B |")
    print("|---|---|---|---|---|---|---|---|")
    for r in rows:
        print("| " + r["name"] + " | " + " | ".join(r.get(c, "?") for c in cols) + " |")
