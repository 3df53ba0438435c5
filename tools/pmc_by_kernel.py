#!/usr/bin/env python3
"""Aggregates a rocprofv3 --pmc counter_collection.csv by render kernel (sums over dispatches)."""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(int)
for r in csv.DictReader(open(f)):
    for name in ("rt_primary", "rt_trace", "rt_shade", "rt_resolve"):
        if name in r["Kernel_Name"]:
            agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
            n[name] += 1
for k, v in agg.items():
    print(k, n[k] // max(1, len(v)), "dispatches", {a: f"{b:.4g}" for a, b in v.items()})
    if "SQ_ACTIVE_INST_VALU" in v and "SQ_THREAD_CYCLES_VALU" in v:
        print("   VALU lane utilisation %.3f" % (v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64)))
    if "SQ_WAVE_CYCLES" in v:
        for c in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
            if c in v:
                print("   %s / SQ_WAVE_CYCLES = %.3f" % (c, v[c] / v["SQ_WAVE_CYCLES"]))
