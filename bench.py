#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its configs[2] workload.

A "step" = one full frame of the hot path: semesterbild (text.obj, 14 521 mesh triangles + slabs +
9 spheres, 5 lights) at high_resolution (1620x1350) with anti_aliasing (16 rays/px) and
soft_shadows (10-light clouds -> 50 shadow rays per shaded hit).  Scene, BVH, sample tables and the
framebuffer are resident in HBM before the timed region.  At N > 1 the frame is tile-partitioned
(48x48 RENDER_STRIDE tiles, lattice ownership) over N ranks, one process per GPU; every rank renders
its tiles into a rank-compact staging buffer and the packed pixels are gathered to rank 0 with ONE
ncclSend / ncclRecv group (C ABI: rt_comm_* / rt_render_gather_device) inside the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `value` = (primary + reflection + refraction rays of one frame as the
reference casts them, all ranks) / (max-over-ranks seconds per frame), in Mray/s.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

ALG_BYTES_PER_RAY = 64.0  # 32 B ray in + 32 B hit out
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_SIMD = 256 * 4          # 256 CUs x 4 SIMD-32
VALU_CYC = 2.0            # v_fma_f32 wave64 on a SIMD-32: 2 cycles (MI355X_MICROARCH.md, cycle constants)
TRANS_CYC = 4.0           # v_exp / v_log / v_rcp / v_rsq / v_sqrt: twice the plain issue cost (8 vs 4 for one wave alone)

WORKLOADS = {
    # BASELINE.json configs[2]
    "c3": dict(features=["high_resolution", "anti_aliasing", "soft_shadows"], model="text", scene="semesterbild",
               name="semesterbild@high_resolution+anti_aliasing+soft_shadows (text.obj)"),
    # configs[0]: the reference's own CPU-runnable case (here also on the GPU)
    "c1": dict(features=[], model=None, scene="test_scene",
               name="test_scene@768x640, no AA / reflections"),
    # configs[1]
    "c2": dict(features=["medium_resolution"], model=None, scene="test_scene_spheres",
               name="test_scene spheres-only@medium_resolution, no secondary rays"),
    # configs[3]
    "c4": dict(features=["high_resolution", "realistic", "extreme_quality"], model="text", scene="semesterbild",
               depth=8, name="semesterbild@high_resolution+realistic+extreme_quality depth 8 (text.obj)"),
    # configs[4]: as c4 at 4K (the scene is aspect dependent, lib.rs:73-79)
    "c5": dict(features=["realistic", "extreme_quality"], model="text", scene="semesterbild", depth=8, size=(3840, 2160),
               name="semesterbild@3840x2160+realistic+extreme_quality depth 8 (text.obj)"),
}
DOMINANT_KERNEL = {"c1": "rt_primary_kernel", "c2": "rt_primary_kernel", "c3": "rt_primary_kernel",
                   "c4": "rt_shade_kernel", "c5": "rt_shade_kernel"}


def build_workload(key):
    from hslu_i.ba_raytracing.f2501_raytracer_amd import RenderConfig, scenes

    w = WORKLOADS[key]
    size = w.get("size", (None, None))
    cfg = RenderConfig.from_features(w["features"], depth_override=w.get("depth"), width_override=size[0],
                                     height_override=size[1])
    if w["scene"] == "semesterbild":
        flat = scenes.semesterbild(cfg, w["model"]).flatten()
    elif w["scene"] == "test_scene":
        flat = scenes.test_scene(cfg).flatten()
    else:
        flat = scenes.test_scene(cfg).flatten().without_triangles()
    return cfg, flat, w["name"]


def pmc_summary(workload, build_id):
    """Per-launch PMC means of the workload's dominant kernel, from the newest committed profiles/*_<workload>_pmc.csv
    that was collected on THIS build of the kernels (tools/profile.sh: separate rocprofv3 --pmc passes of this very
    command; the file records the build id of librt_hip.so).  A summary of another build is refused."""
    import csv
    import glob

    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{workload}_pmc.csv")), reverse=True):
        rows = list(csv.DictReader(open(f)))
        vals = {r["counter"]: r["mean_per_dispatch"] for r in rows}
        if vals.get("build_id") != build_id:
            continue
        out = {k: float(v) for k, v in vals.items() if k not in ("build_id", "kernel")}
        out["_file"] = os.path.basename(f)
        out["_kernel"] = vals.get("kernel", "")
        return out
    return None


def cpu_info():
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return model


def cpu_baseline(cfg, flat, budget_s=18.0):
    """The reference's simd_render CPU path restated (kind "port": oracle/rt_simd_baseline.c -- 8-lane AVX2 packets
    of the 8/16/24 samples of a pixel, brute-force scan of every object per ray like the reference, rows of 48x48
    tiles handed to a thread pool) timed on a bounded window of the same workload: one thread first (also sizes the
    sample), then every core this process is granted."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib

    cores = oracle_lib.host_cores()
    cx, cy = cfg.width // 2, cfg.height // 2
    # one row segment of 8 pixels on one thread
    probe = (cx - 4, cy, 8, 1)
    # (timed inside the C call: the Python wrapper rebuilds the sample tables on every call)
    _, _, s1 = oracle_lib.render(flat, cfg, window=probe, n_threads=1, aux=False, impl="simd")
    dt1 = max(s1["kernel_ms"] * 1e-3, 1e-3)
    rays1 = s1["rays_primary"] + s1["rays_reflection"] + s1["rays_refraction"]
    px_per_s_thread = 8 / dt1
    # sample for ~budget_s on all cores: rows of 48-pixel tile segments (the work item of the pool)
    n_px = float(np.clip(budget_s * px_per_s_thread * cores, 48 * cores, 48 * 48 * 64))
    tiles_x = int(max(1, min(cfg.width // 48 - 1, np.ceil(np.sqrt(n_px / 2304.0)))))
    w = 48 * tiles_x
    h = int(max(1, min(cfg.height, np.ceil(n_px / w))))
    win = (max(0, cx - w // 2), max(0, cy - h // 2), w, h)
    _, _, st = oracle_lib.render(flat, cfg, window=win, n_threads=cores, aux=False, impl="simd")
    dt = max(st["kernel_ms"] * 1e-3, 1e-3)
    rays = st["rays_primary"] + st["rays_reflection"] + st["rays_refraction"]
    return {
        "value": rays / dt / 1e6, "unit": "Mray/s", "cores": cores, "kind": "port",
        "sample": f"{win[2]}x{win[3]} px window at frame centre, {rays} rays + {st['rays_shadow']} shadow rays in {dt:.1f} s; "
                  f"8-lane AVX2 packets, 48x48 tiles, brute-force scan of {flat.n_objects} objects per ray (restated "
                  f"simd_render: the reference is Rust nightly and cannot be built offline)",
        "mshadow_per_s": st["rays_shadow"] / dt / 1e6,
        "one_thread": {"value": rays1 / dt1 / 1e6, "unit": "Mray/s", "sample": f"8 px, {dt1:.2f} s"},
        "parallel_efficiency": (rays / dt) / (rays1 / dt1) / cores,
        "cpu_model": cpu_info(), "cpus_visible": os.cpu_count(), "cpus_granted": cores,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)  # ~1 s of frames: enough for a GPU-busy sampler to see them
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default=os.environ.get("RT_BENCH_BACKEND", "nccl"), choices=["nccl", "gloo"],
                    help="nccl (= RCCL, the real path); gloo only to rehearse N > 1 on a one-GPU box")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from hslu_i.ba_raytracing.f2501_raytracer_amd import _abi, _lib
    from hslu_i.ba_raytracing.f2501_raytracer_amd.distributed import HostGather, RcclGather
    from hslu_i.ba_raytracing.f2501_raytracer_amd.renderer import DeviceScene

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    cfg, flat, wname = build_workload(args.workload)
    lib = _lib.load()
    ds = DeviceScene(flat, device=local_rank)
    p, keep = _abi.make_params(cfg, n_ranks=world, rank=rank)
    npix = cfg.width * cfg.height
    fb = torch.zeros(npix, dtype=torch.int32, device=dev)
    stream = torch.cuda.Stream(device=dev)
    # N > 1: the library's own RCCL communicator (torch.distributed only carries its 128-byte id)
    rccl = RcclGather(world, rank, local_rank) if (world > 1 and args.backend == "nccl") else None
    host_gather = HostGather(cfg, world, rank) if (world > 1 and args.backend == "gloo") else None
    gather_ms_host = []

    def frame(ev0=None, ev1=None):
        with torch.cuda.stream(stream):
            if ev0 is not None:
                ev0.record(stream)
            if rccl is not None:
                rccl.render_gather(ds, p, fb.data_ptr(), stream.cuda_stream)
            else:
                _lib.check(lib.rt_render_device(ds.handle, C.byref(p), C.c_void_p(fb.data_ptr()), None,
                                                C.c_void_p(stream.cuda_stream)))
            if ev1 is not None:
                ev1.record(stream)
            if host_gather is not None:  # rehearsal: the same staging layout through gloo on the host
                stream.synchronize()
                t = time.perf_counter()
                host = fb.cpu().numpy().view(np.uint32)
                host_gather.run(host)
                if rank == 0:
                    fb.copy_(torch.from_numpy(host.view(np.int32)))
                gather_ms_host.append((time.perf_counter() - t) * 1e3)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(max(args.warmup, 1 if world > 1 else 0)):  # N > 1: the first gather connects the peers (set-up, not a step)
        frame()
    barrier()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    infos = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        frame(*evs[i])
    barrier()
    elapsed = time.perf_counter() - t0
    frame_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))  # device time of one step on the launch stream
    if rccl is not None:
        infos.append(rccl.last())

    st = _abi.rt_stats()
    _lib.check(lib.rt_render_collect_stats(ds.handle, C.byref(st)))
    own = dict(rank=rank, rays=st.rays_primary + st.rays_reflection + st.rays_refraction, rays_traced=st.rays_traced,
               frame_ms=frame_ms)
    if infos:
        gi = infos[-1]
        own.update(kernel_ms=gi["render_ms"], gather_ms=gi["gather_ms"], bytes_sent=gi["bytes_sent"],
                   bytes_received=gi["bytes_received"], tiles_owned=gi["tiles_owned"], rccl_ranks=gi["n_ranks"],
                   rccl_rank=gi["rank"], transport={0: "none", 1: "rccl", 2: "local"}[gi["transport"]])
    elif host_gather is not None:
        own.update(kernel_ms=frame_ms, gather_ms=float(np.mean(gather_ms_host[-args.steps:])), bytes_sent=host_gather.bytes_sent,
                   transport="gloo-host (rehearsal)")
    else:
        own.update(kernel_ms=frame_ms, gather_ms=0.0, bytes_sent=0, transport="none")
    counts = torch.tensor([st.rays_primary, st.rays_reflection, st.rays_refraction, st.rays_shadow,
                           st.pixels_written, st.rays_traced], dtype=torch.int64, device=dev)
    tmax = torch.tensor([elapsed, frame_ms], dtype=torch.float64, device=dev)
    per_rank = [own]
    if world > 1:
        if args.backend == "gloo":
            counts, tmax = counts.cpu(), tmax.cpu()
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        per_rank = [None] * world
        dist.all_gather_object(per_rank, own)
    counts = counts.tolist()
    elapsed, _ = tmax.tolist()
    rays = counts[0] + counts[1] + counts[2]
    sec_per_step = elapsed / args.steps

    if rank == 0:
        mrays = rays / sec_per_step / 1e6
        kernel_ms = own["kernel_ms"]
        build_id = lib.rt_build_id().decode()
        # ---- roofline of the dominant kernel ------------------------------------------------------------------------
        # (1) the HBM bound SURVEY 8(d) prescribes: algorithmic 64 B per ray actually traced by this rank's launches
        achieved = own["rays_traced"] * ALG_BYTES_PER_RAY / (kernel_ms * 1e-3) / 1e9
        pmc = pmc_summary(args.workload, build_id) if world == 1 else None
        traffic = None
        if pmc and "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
            # KiB -> B; FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md, HBM)
            traffic = pmc["FETCH_SIZE"] * 1024.0 * 2.0 + pmc["WRITE_SIZE"] * 1024.0
        # (2) the bound that applies: vector-instruction issue.  Instructions of the dominant kernel per launch x
        # issue cycles (2 per plain VALU instruction on a SIMD-32, 4 per transcendental) / (1024 SIMDs x cycles)
        valu = None
        if pmc and "SQ_INSTS_VALU" in pmc and "GRBM_GUI_ACTIVE" in pmc:
            cycles = pmc["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
            trans = pmc.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
            issue = (pmc["SQ_INSTS_VALU"] - trans) * VALU_CYC + trans * TRANS_CYC
            valu = {
                "bound": "valu_issue", "kernel": pmc["_kernel"], "achieved": issue / cycles, "peak": float(N_SIMD),
                "unit": "SIMD issue cycles per cycle", "frac": issue / (N_SIMD * cycles),
                "insts_valu": pmc["SQ_INSTS_VALU"], "insts_valu_trans": trans if "SQ_INSTS_VALU_TRANS_F32" in pmc else None,
                "insts_salu": pmc.get("SQ_INSTS_SALU"), "insts_smem": pmc.get("SQ_INSTS_SMEM"),
                "kernel_cycles": cycles, "source": pmc["_file"],
            }
        out = {
            "metric": "Mray/s (primary+secondary), semesterbild@high_resolution" if args.workload == "c3"
                      else "Mray/s (primary+secondary)",
            "value": mrays, "unit": "Mray/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": sec_per_step * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": wname, "width": cfg.width, "height": cfg.height,
                "rays_per_frame": rays, "rays_traced_per_frame": counts[5],
                "shadow_rays_per_frame": counts[3], "pixels_written": counts[4],
                "mshadow_per_s": counts[3] / sec_per_step / 1e6,
                "objects": flat.n_objects, "lights": int(flat.lights.shape[0]) * cfg.point_light_multiplicator,
                "parallelism": f"tiles{cfg.render_stride}x{cfg.render_stride}/{world}gpu" + ("+rccl_gather" if world > 1 else ""),
                "bvh": ds.bvh_info(), "build_id": build_id,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": pmc["_file"] if (pmc and traffic is not None) else None,
                "kernel": DOMINANT_KERNEL[args.workload], "kernel_ms": kernel_ms,
                "algorithmic_bytes": own["rays_traced"] * ALG_BYTES_PER_RAY,
                "note": "64 B per TRACED ray (SURVEY 8d ray-stream model); no ray touches HBM in this kernel, so this "
                        "fraction cannot rank it -- the applicable bound is `valu_issue` below",
                "valu_issue": valu if valu is not None else
                {"bound": "valu_issue", "frac": None,
                 "note": f"no profiles/*_{args.workload}_pmc.csv was collected on build {build_id} (tools/profile.sh)"},
            },
            "ranks": per_rank,
        }
        if world == 1:
            # device -> host copy of the packed frame (the boundary's rt_render pays it; never part of `value`)
            host = torch.empty(npix, dtype=torch.int32).pin_memory()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            d2h = []
            for _ in range(4):
                e0.record()
                host.copy_(fb, non_blocking=True)
                e1.record()
                torch.cuda.synchronize(dev)
                d2h.append(e0.elapsed_time(e1))
            out["d2h_ms"] = float(np.median(d2h[1:]))
            out["config"]["mray_per_s_incl_d2h"] = rays / (sec_per_step + out["d2h_ms"] * 1e-3) / 1e6
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg, flat)
            out["cpu_baseline"]["gpu_over_cpu"] = mrays / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        if rccl is not None:
            rccl.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
