#!/usr/bin/env python3
"""Condenses a tools/profile.sh output directory (gpurun_out/prof_<tag>) into profiles/<tag>_*.csv/.md:
kernel_stats.csv verbatim (rocprofv3 --kernel-trace --stats) + per-dispatch PMC means of the render kernel,
with the derived figures DESIGN.md quotes (HBM bytes with the gfx950 FETCH_SIZE x2 correction of
MI355X_MICROARCH.md section HBM, VALUBusy, VALU lane utilisation, effective clock)."""
import collections
import csv
import glob
import os
import shutil
import sys

tag = sys.argv[1]
src = os.path.join("gpurun_out", "prof_" + tag)
os.makedirs("profiles", exist_ok=True)
ks = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
shutil.copy(ks, os.path.join("profiles", f"{tag}_kernel_stats.csv"))
rows = list(csv.DictReader(open(ks)))
kern = [r for r in rows if "rt_primary_kernel" in r["Name"]][0]
avg_ms = float(kern["AverageNs"]) / 1e6
pmc = collections.OrderedDict()
meta = {}
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv"))):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "rt_primary_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}
    for k, v in agg.items():
        pmc[k] = sum(v) / len(v)
with open(os.path.join("profiles", f"{tag}_pmc.csv"), "w") as fh:
    fh.write("counter,mean_per_dispatch\n")
    for k, v in pmc.items():
        fh.write(f"{k},{v:.6g}\n")
out = [f"# rocprofv3 summary {tag}: rt_primary_kernel", "",
       f"* calls {kern['Calls']}, average {avg_ms:.3f} ms (min {float(kern['MinNs'])/1e6:.3f}, max {float(kern['MaxNs'])/1e6:.3f}), {kern['Percentage']} % of GPU time",
       f"* launch: {meta}"]
if "FETCH_SIZE" in pmc:
    fetch = pmc["FETCH_SIZE"] * 1024 * 2  # KiB -> B, x2 gfx950 correction for wide coalesced reads (upper bound here)
    write = pmc.get("WRITE_SIZE", 0) * 1024
    out.append(f"* HBM traffic per launch: FETCH_SIZE {pmc['FETCH_SIZE']:.1f} KiB (x2 corrected: {fetch/1e6:.2f} MB), "
               f"WRITE_SIZE {pmc.get('WRITE_SIZE', 0):.1f} KiB ({write/1e6:.2f} MB) -> {(fetch+write)/1e6:.2f} MB, "
               f"{(fetch+write)/avg_ms/1e6:.3f} GB/s")
if "GRBM_GUI_ACTIVE" in pmc:
    clk = pmc["GRBM_GUI_ACTIVE"] / 8 / (avg_ms * 1e-3) / 1e9
    out.append(f"* effective clock {clk:.2f} GHz (GRBM_GUI_ACTIVE / 8 XCDs / time)")
    if "SQ_ACTIVE_INST_VALU" in pmc:
        vb = pmc["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (pmc["GRBM_GUI_ACTIVE"] / 8)
        out.append(f"* VALUBusy = SQ_ACTIVE_INST_VALU*4 / 1024 SIMDs / cycles = {100*vb:.1f} %"
                   + (" (the guide's 4-cycles-per-instruction convention; > 100 % means VALU instructions retire faster than that)" if vb > 1.0 else ""))
if "SQ_THREAD_CYCLES_VALU" in pmc and "SQ_ACTIVE_INST_VALU" in pmc:
    out.append(f"* VALU lane utilisation = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU*64) = "
               f"{100*pmc['SQ_THREAD_CYCLES_VALU']/(pmc['SQ_ACTIVE_INST_VALU']*64):.1f} %")
if "SQ_WAVE_CYCLES" in pmc:
    for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
        if k in pmc:
            out.append(f"* {k} / SQ_WAVE_CYCLES = {100*pmc[k]/pmc['SQ_WAVE_CYCLES']:.1f} %")
for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_WAVES"):
    if k in pmc:
        out.append(f"* {k} {pmc[k]:.4g}")
open(os.path.join("profiles", f"{tag}_summary.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
