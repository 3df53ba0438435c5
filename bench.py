#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its configs[2] workload.

A "step" = one full frame of the hot path: semesterbild (text.obj, 14 521 mesh triangles + slabs +
9 spheres, 5 lights) at high_resolution (1620x1350) with anti_aliasing (16 rays/px) and
soft_shadows (10-light clouds -> 50 shadow rays per shaded hit).  Scene, BVH, sample tables and the
framebuffer are resident in HBM before the timed region.  At N > 1 the frame is tile-partitioned
(48x48 RENDER_STRIDE tiles, permuted ownership) over N ranks, one process per GPU, and the packed
pixels are gathered to rank 0 over RCCL inside the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `value` = (primary + reflection + refraction rays of one frame, all
ranks) / (max-over-ranks seconds per frame), in Mray/s.
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


def pmc_traffic_bytes():
    """HBM bytes per launch of the render kernel from the newest committed rocprofv3 PMC summary
    (profiles/*_pmc.csv, written by tools/summarize_profile.py from separate --pmc passes of this very
    command): FETCH_SIZE (KiB) x 2 (gfx950 correction, MI355X_MICROARCH.md section HBM) + WRITE_SIZE (KiB)."""
    import csv
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.csv")))
    if not files:
        return None, None
    vals = {r["counter"]: float(r["mean_per_dispatch"]) for r in csv.DictReader(open(files[-1]))}
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        return None, None
    return vals["FETCH_SIZE"] * 1024.0 * 2.0 + vals["WRITE_SIZE"] * 1024.0, os.path.basename(files[-1])


def cpu_baseline(cfg, flat, budget_s=20.0):
    """The oracle (kind "port": brute-force linear scan like the reference, scalar fp32, pthreads over
    rows) timed on a bounded window of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib

    threads = os.cpu_count() or 1
    # probe one pixel per thread to size the sample for ~budget_s of wall time on all host threads
    side = max(4, int(np.ceil(np.sqrt(threads))))
    cx, cy = cfg.width // 2 - side // 2, cfg.height // 2 - side // 2
    t0 = time.time()
    oracle_lib.render(flat, cfg, window=(cx, cy, side, side), n_threads=threads, aux=False)
    dt = max(time.time() - t0, 1e-3)
    px_per_s = side * side / dt
    n_px = float(np.clip(budget_s * px_per_s, 256, 65536))
    w = int(min(256, max(16, np.sqrt(n_px))))
    h = int(max(16, min(256, n_px / w)))
    win = (cfg.width // 2 - w // 2, cfg.height // 2 - h // 2, w, h)
    t0 = time.time()
    _, _, st = oracle_lib.render(flat, cfg, window=win, n_threads=threads, aux=False)
    dt = time.time() - t0
    rays = st["rays_primary"] + st["rays_reflection"] + st["rays_refraction"]
    return {
        "value": rays / dt / 1e6, "unit": "Mray/s", "cores": threads, "kind": "port",
        "sample": f"{win[2]}x{win[3]} px window at frame centre, {rays} rays + {st['rays_shadow']} shadow rays "
                  f"in {dt:.1f} s (brute-force scan of {flat.n_objects} objects per ray, as the reference)",
        "mshadow_per_s": st["rays_shadow"] / dt / 1e6,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default=os.environ.get("RT_BENCH_BACKEND", "nccl"), choices=["nccl", "gloo"],
                    help="nccl (= RCCL, the real path); gloo only to rehearse N > 1 on a one-GPU box")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from hslu_i.ba_raytracing.f2501_raytracer_amd import _abi, _lib
    from hslu_i.ba_raytracing.f2501_raytracer_amd.distributed import TileGather, owned_pixel_indices
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
    gather = TileGather(cfg, world, rank, dev, host_staging=(args.backend == "gloo")) if world > 1 else None

    def frame(ev0=None, ev1=None):
        with torch.cuda.stream(stream):
            if ev0 is not None:
                ev0.record(stream)
            _lib.check(lib.rt_render_device(ds.handle, C.byref(p), C.c_void_p(fb.data_ptr()), None,
                                            C.c_void_p(stream.cuda_stream)))
            if ev1 is not None:
                ev1.record(stream)
            if gather is not None:
                gather.run(fb, stream)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if gather is not None:
        # set-up, not a step: create the RCCL communicator / connections before anything is timed
        with torch.cuda.stream(stream):
            gather.run(fb, stream)
        barrier()
    for _ in range(args.warmup):
        frame()
    barrier()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        frame(*evs[i])
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))

    st = _abi.rt_stats()
    _lib.check(lib.rt_render_collect_stats(ds.handle, C.byref(st)))
    counts = torch.tensor([st.rays_primary, st.rays_reflection, st.rays_refraction, st.rays_shadow,
                           st.pixels_written], dtype=torch.int64, device=dev)
    tmax = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=dev)
    if world > 1:
        if args.backend == "gloo":
            counts, tmax = counts.cpu(), tmax.cpu()
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    counts = counts.tolist()
    elapsed, kernel_ms_max = tmax.tolist()
    rays = counts[0] + counts[1] + counts[2]
    sec_per_step = elapsed / args.steps

    if rank == 0:
        mrays = rays / sec_per_step / 1e6
        # roofline of the dominant (only) kernel: algorithmic bytes = rays of THIS rank's launch x 64 B
        own_rays = st.rays_primary + st.rays_reflection + st.rays_refraction
        achieved = own_rays * ALG_BYTES_PER_RAY / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "Mray/s (primary+secondary), semesterbild@high_resolution" if args.workload == "c3"
                      else "Mray/s (primary+secondary)",
            "value": mrays, "unit": "Mray/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": sec_per_step * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": wname, "width": cfg.width, "height": cfg.height,
                "rays_per_frame": rays, "shadow_rays_per_frame": counts[3], "pixels_written": counts[4],
                "mshadow_per_s": counts[3] / sec_per_step / 1e6,
                "objects": flat.n_objects, "lights": int(flat.lights.shape[0]) * cfg.point_light_multiplicator,
                "parallelism": f"tiles{cfg.render_stride}x{cfg.render_stride}/{world}gpu" + ("+rccl_gather" if world > 1 else ""),
                "bvh": ds.bvh_info(),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic_bytes()[0] if (args.workload == "c3" and world == 1) else None,
                "traffic_source": pmc_traffic_bytes()[1] if (args.workload == "c3" and world == 1) else None,
                "kernel": "rt_primary_kernel", "kernel_ms": kernel_ms,
                "note": "algorithmic 64 B/ray ray-stream model (SURVEY 8d); the kernel is bound by vector-instruction issue, "
                        "see DESIGN.md and profiles/",
            },
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg, flat)
            out["cpu_baseline"]["gpu_over_cpu"] = mrays / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
