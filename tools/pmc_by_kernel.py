#!/usr/bin/env python3
"""Aggregates a rocprofv3 --pmc counter_collection.csv by kernel (every rt_* kernel; sums over dispatches) and prints the
derived figures per kernel: VALU-issue fraction on the SIMD-32 convention (2 cycles per wave64 VALU instruction; kernel
cycles = GRBM_GUI_ACTIVE / 8 XCDs), wait shares of the wave-cycles, HBM bytes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE).
Usage: pmc_by_kernel.py <rocprofv3 output dir> [frames the run rendered]"""
import collections
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
frames = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.defaultdict(int))
meta = {}
for r in csv.DictReader(open(f)):
    m = re.search(r"(rt_\w+)", r["Kernel_Name"])
    if not m:
        continue
    name = m.group(1)
    agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
    n[name][r["Counter_Name"]] += 1
    meta[name] = (r.get("VGPR_Count"), r.get("Scratch_Size"), r.get("LDS_Block_Size"))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", kv[1].get("SQ_WAVE_CYCLES", 0))):
    disp = max(n[k].values())
    line = f"{k:26s} {disp / frames:6.1f} dispatches/frame  VGPR {meta[k][0]} scratch {meta[k][1]} LDS {meta[k][2]} | per frame: "
    line += ", ".join(f"{a} {b / frames:.4g}" for a, b in sorted(v.items()))
    print(line)
    if "SQ_INSTS_VALU" in v and "GRBM_GUI_ACTIVE" in v:
        cyc = v["GRBM_GUI_ACTIVE"] / 8.0
        print(f"   VALU-issue fraction {v['SQ_INSTS_VALU'] * 2.0 / (1024.0 * cyc):.3f}   kernel cycles/frame {cyc / frames:.4g}   SALU/VALU "
              f"{v.get('SQ_INSTS_SALU', 0) / v['SQ_INSTS_VALU']:.3f}   SMEM/VALU {v.get('SQ_INSTS_SMEM', 0) / v['SQ_INSTS_VALU']:.3f}")
    if "SQ_ACTIVE_INST_VALU" in v and "SQ_THREAD_CYCLES_VALU" in v:
        print("   VALU lane utilisation %.3f" % (v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64)))
    if "SQ_WAVE_CYCLES" in v:
        print("   " + "  ".join("%s/SQ_WAVE_CYCLES %.3f" % (c, v[c] / v["SQ_WAVE_CYCLES"]) for c in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES") if c in v))
    if "FETCH_SIZE" in v or "WRITE_SIZE" in v:
        fetch, write = v.get("FETCH_SIZE", 0) * 1024 * 2, v.get("WRITE_SIZE", 0) * 1024
        print(f"   HBM per frame: fetched {fetch / frames / 1e6:.1f} MB (FETCH_SIZE x2), written {write / frames / 1e6:.1f} MB")
