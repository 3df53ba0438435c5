#!/usr/bin/env python3
"""Condenses a tools/profile.sh output directory (gpurun_out/prof_<tag>_<workload>) into
profiles/<tag>_<workload>_{kernel_stats.csv,pmc.csv,summary.md}: kernel_stats.csv verbatim (rocprofv3 --kernel-trace
--stats) and the PMC counters of the workload's dominant kernel PER FRAME (summed over the kernel's dispatches of one
frame: rt_shade_kernel runs once per ray-tree level), with the derived figures DESIGN.md quotes: HBM bytes with the
gfx950 FETCH_SIZE x2 correction (MI355X_MICROARCH.md, HBM), the VALU-issue fraction on the SIMD-32 convention
(2 cycles per wave64 VALU instruction, 4 per transcendental), VALU lane utilisation, effective clock.
The pmc.csv records the build id of the library and the kernel name; bench.py only quotes it for that build."""
import collections
import csv
import glob
import os
import shutil
import sys

tag, wl = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "c3")
kname = {"c1": "rt_primary_kernel", "c2": "rt_primary_kernel", "c3": "rt_primary_kernel"}.get(wl, "rt_shade_kernel")
src = os.path.join("gpurun_out", f"prof_{tag}_{wl}")
os.makedirs("profiles", exist_ok=True)
build_id = open(os.path.join(src, "build_id.txt")).read().split()[0]
ks = sorted(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)[-1]  # (the newest run into this directory)
shutil.copy(ks, os.path.join("profiles", f"{tag}_{wl}_kernel_stats.csv"))
rows = list(csv.DictReader(open(ks)))
nf = os.path.join(src, "n_frames.txt")
n_frames = int(open(nf).read().split()[0]) if os.path.exists(nf) else 4  # frames per profiled run (tools/profile.sh)
kern = [r for r in rows if kname in r["Name"]][0]
calls = int(kern["Calls"])
per_frame = calls / n_frames
avg_ms = float(kern["AverageNs"]) / 1e6
frame_ms = float(kern["TotalDurationNs"]) / 1e6 / n_frames
pmc = collections.OrderedDict()
meta = {}
newest = {}
for f in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):  # one file per pass: the newest run's
    d = os.path.basename(os.path.dirname(os.path.dirname(f)))
    if d not in newest or os.path.getmtime(f) > os.path.getmtime(newest[d]):
        newest[d] = f
for f in sorted(newest.values()):
    agg = collections.defaultdict(float)
    n_disp = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if kname in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            n_disp[r["Counter_Name"]] += 1
            meta = {k: r[k] for k in ("Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}
    for k, v in agg.items():
        frames = max(1.0, n_disp[k] / per_frame)
        pmc[k] = v / frames
with open(os.path.join("profiles", f"{tag}_{wl}_pmc.csv"), "w") as fh:
    fh.write("counter,mean_per_dispatch\n")  # (column name kept; values are per FRAME = per dispatch when the kernel runs once)
    fh.write(f"build_id,{build_id}\nkernel,{kname}\ndispatches_per_frame,{per_frame:g}\n")
    for k, v in pmc.items():
        fh.write(f"{k},{v:.6g}\n")
out = [f"# rocprofv3 summary {tag} / {wl}: {kname} (build {build_id})", "",
       f"* {calls} calls over {n_frames} frames ({per_frame:g} per frame), average {avg_ms:.3f} ms per call "
       f"(min {float(kern['MinNs'])/1e6:.3f}, max {float(kern['MaxNs'])/1e6:.3f}), {frame_ms:.3f} ms per frame, {kern['Percentage']} % of GPU time",
       f"* launch: {meta}"]
for r in rows:
    if r["Name"] != kern["Name"] and float(r["Percentage"]) >= 0.5:
        out.append(f"* also: {r['Name'][:60]} {r['Calls']} calls, {float(r['TotalDurationNs'])/1e6/n_frames:.3f} ms per frame, {r['Percentage']} %")
if "FETCH_SIZE" in pmc:
    fetch = pmc["FETCH_SIZE"] * 1024 * 2  # KiB -> B, x2 gfx950 correction for wide coalesced reads (upper bound here)
    write = pmc.get("WRITE_SIZE", 0) * 1024
    out.append(f"* HBM traffic per frame: FETCH_SIZE {pmc['FETCH_SIZE']:.1f} KiB (x2 corrected: {fetch/1e6:.2f} MB), "
               f"WRITE_SIZE {pmc.get('WRITE_SIZE', 0):.1f} KiB ({write/1e6:.2f} MB) -> {(fetch+write)/1e6:.2f} MB, "
               f"{(fetch+write)/frame_ms/1e6:.3f} GB/s")
if "GRBM_GUI_ACTIVE" in pmc:
    cycles = pmc["GRBM_GUI_ACTIVE"] / 8
    out.append(f"* effective clock {cycles / (frame_ms * 1e-3) / 1e9:.2f} GHz (GRBM_GUI_ACTIVE / 8 XCDs / kernel time)")
    if "SQ_INSTS_VALU" in pmc:
        trans = pmc.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
        issue = (pmc["SQ_INSTS_VALU"] - trans) * 2 + trans * 4
        out.append(f"* VALU-issue fraction = (SQ_INSTS_VALU x 2 cyc" + (f", {trans:.3g} transcendentals x 4 cyc" if trans else "")
                   + f") / (1024 SIMDs x {cycles:.4g} cycles) = {100 * issue / (1024 * cycles):.1f} %  (SIMD-32: a wave64 VALU instruction issues over 2 cycles)")
        for k in ("SQ_INSTS_SALU", "SQ_INSTS_SMEM"):
            if k in pmc:
                out.append(f"* {k} / SQ_INSTS_VALU = {100 * pmc[k] / pmc['SQ_INSTS_VALU']:.1f} %")
if "SQ_THREAD_CYCLES_VALU" in pmc and "SQ_ACTIVE_INST_VALU" in pmc:
    out.append(f"* VALU lane utilisation = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU*64) = "
               f"{100*pmc['SQ_THREAD_CYCLES_VALU']/(pmc['SQ_ACTIVE_INST_VALU']*64):.1f} %")
if "SQ_WAVE_CYCLES" in pmc:
    for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
        if k in pmc:
            out.append(f"* {k} / SQ_WAVE_CYCLES = {100*pmc[k]/pmc['SQ_WAVE_CYCLES']:.1f} %")
if "SQC_DCACHE_REQ" in pmc and "SQC_DCACHE_HITS" in pmc:
    out.append(f"* scalar cache hit rate {100 * pmc['SQC_DCACHE_HITS'] / pmc['SQC_DCACHE_REQ']:.1f} %")
for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_WAVES"):
    if k in pmc:
        out.append(f"* {k} {pmc[k]:.4g}")
open(os.path.join("profiles", f"{tag}_{wl}_summary.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
