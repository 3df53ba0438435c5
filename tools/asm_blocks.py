#!/usr/bin/env python3
"""Basic-block instruction mix of one kernel in a hipcc -S listing (static; for reasoning about the
hot loops).  Usage: asm_blocks.py file.s kernel_substring [min_valu]"""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
minv = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and key in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
blocks, cur = [], None
order = {}
for i in range(start, end + 1):
    l = lines[i].strip()
    m = re.match(r"^(\.LBB\d+_\d+):", l) or re.match(r"^; (%bb\.\d+):", l)
    if m or cur is None:
        cur = dict(name=m.group(1) if m else "entry", line=i + 1, valu=0, salu=0, smem=0, vmem=0, lds=0, trans=0, pk=0, br=[], ins=[])
        order[cur["name"]] = len(blocks)
        blocks.append(cur)
        if m:
            continue
    if not l or l.startswith(";") or l.startswith("."):
        continue
    op = l.split()[0]
    cur["ins"].append(l)
    if op.startswith("v_"):
        cur["valu"] += 1
        if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)", op):
            cur["trans"] += 1
        if op.startswith("v_pk_"):
            cur["pk"] += 1
    elif op.startswith("s_load") or op.startswith("s_buffer"):
        cur["smem"] += 1
    elif op.startswith("s_cbranch") or op.startswith("s_branch"):
        cur["br"].append(l.split()[-1])
        cur["salu"] += 1
    elif op.startswith("s_"):
        cur["salu"] += 1
    elif op.startswith("ds_"):
        cur["lds"] += 1
    elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
        cur["vmem"] += 1
tot = dict(valu=0, salu=0, smem=0)
for b in blocks:
    for k in tot:
        tot[k] += b[k]
print(f"{len(blocks)} blocks, static VALU {tot['valu']} SALU {tot['salu']} SMEM {tot['smem']}")
for idx, b in enumerate(blocks):
    back = [t for t in b["br"] if t in order and order[t] <= idx]
    if b["valu"] >= minv or back:
        print(f"{idx:4d} {b['name']:12s} L{b['line']:6d} valu {b['valu']:4d} (trans {b['trans']:2d} pk {b['pk']:2d}) salu {b['salu']:3d} smem {b['smem']:2d} vmem {b['vmem']:2d} lds {b['lds']:2d}"
              f"  -> {','.join(b['br'])}{'   BACK ' + ','.join(f'{t}({order[t]})' for t in back) if back else ''}")
