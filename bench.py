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

Frames are enqueued on two streams alternately (two frames in flight, each into its own frame buffer): a launch cannot
end before its longest wavefront does, and the head of the next frame fills the compute units that drain leaves idle
(a frame with secondary rays is a chain of such launches, one per ray-tree level).  Every timed step is a complete
frame; the K steps are bracketed by barrier + synchronise.

Rank 0 prints ONE JSON line.  `value` = (primary + reflection + refraction rays of one frame as the
reference casts them, all ranks) / (max-over-ranks seconds per frame), in Mray/s; `value_traced` = the rays the GPU
traced (repeated AA samples once) over the same time.
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
# HIP maps its streams onto a few hardware queues (4 by default); two streams that share one run their work one after
# the other.  This process uses two render streams, the library's gather stream and whatever torch.distributed creates:
# give them queues of their own, so that two frames in flight really overlap.  (Must be set before HIP initialises.)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402

# Device memory a scene may spend on its optional acceleration tables (rt_scene_desc.device_budget_bytes).  The library's
# default is 128 MiB, under which semesterbild keeps its receiver flags (22 MB) but not the per-cell candidate lists (0.86 GB
# for 2 % of a frame); the bench opts into them and says so (config.scene_budget_bytes, config.scene_memory; the frame time
# under the library's default budget is reported beside it as `lean_scene`).
SCENE_BUDGET = 2 << 30
ALG_BYTES_PER_RAY = 64.0  # 32 B ray in + 32 B hit out
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_VECTOR_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector (non-matrix) peak, 256 CUs x 128 lanes x 2 flop x 2.4 GHz
# SURVEY.md section 8(d) flop model: literal sphere test 35, literal triangle test 85; the reference scans every object
# for every ray AND for every shadow ray (raytracer.rs:48,180), so its brute-force equivalent per ray is 35 S + 85 T
FLOP_SPHERE, FLOP_TRIANGLE = 35.0, 85.0
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
    # the reference's own feature values beside the two BASELINE configs above (SURVEY 8a notes / 8d):
    # configs[2] with the mesh the reference's feature set would load (no high_quality_model / medium_resolution ->
    # text_lowres.obj, 1 639 triangles, src/main.rs:31-35)
    "c3lowres": dict(features=["high_resolution", "anti_aliasing", "soft_shadows"], model="text_lowres", scene="semesterbild",
                     name="semesterbild@high_resolution+anti_aliasing+soft_shadows (text_lowres.obj)"),
    # configs[3] at the depth `realistic + extreme_quality` means in the reference: 21 / 21 (raytracer_renderer.rs:55-73)
    "c4d21": dict(features=["high_resolution", "realistic", "extreme_quality"], model="text", scene="semesterbild",
                  depth=21, name="semesterbild@high_resolution+realistic+extreme_quality depth 21 (text.obj)"),
    # configs[4]: as c4 at 4K (the scene is aspect dependent, lib.rs:73-79)
    "c5": dict(features=["realistic", "extreme_quality"], model="text", scene="semesterbild", depth=8, size=(3840, 2160),
               name="semesterbild@3840x2160+realistic+extreme_quality depth 8 (text.obj)"),
}
DOMINANT_KERNEL = {"c1": "rt_primary_kernel", "c2": "rt_primary_kernel", "c3": "rt_primary_kernel", "c3lowres": "rt_primary_kernel",
                   "c4": "rt_shade_kernel", "c4d21": "rt_shade_kernel", "c5": "rt_shade_kernel"}


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
    tiles handed to a thread pool) timed on a bounded sample of the same workload: one thread first (also sizes the
    sample), then every core this process is granted.  The sample is a set of whole 48x48 tiles SPREAD OVER THE FRAME
    (the tiles rank 0 of R would own under the lattice interleave of rt_tile_owner, R chosen for ~budget_s seconds), so
    that it sees background, walls, text and spheres in the frame's own proportions.  Built on this machine with
    -O3 -march=native when a C compiler is here, else the travelling x86-64-v3 build."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib

    build = oracle_lib.use_native_build()
    cores = oracle_lib.host_cores()
    cx, cy = cfg.width // 2, cfg.height // 2
    # one row segment of 8 pixels on one thread
    probe = (cx - 4, cy, 8, 1)
    # (timed inside the C call: the Python wrapper rebuilds the sample tables on every call)
    _, _, s1 = oracle_lib.render(flat, cfg, window=probe, n_threads=1, aux=False, impl="simd")
    dt1 = max(s1["kernel_ms"] * 1e-3, 1e-3)
    rays1 = s1["rays_primary"] + s1["rays_reflection"] + s1["rays_refraction"]
    px_per_s_thread = 8 / dt1
    # ~budget_s on all cores, in whole tiles (the probe sits where every pixel hits: the frame average is about half as expensive)
    n_px = float(np.clip(budget_s * px_per_s_thread * cores * 2.2, 2304, cfg.width * cfg.height))
    n_tiles_frame = -(-cfg.width // 48) * -(-cfg.height // 48)
    share = int(max(1, round(n_tiles_frame * 2304 / n_px)))
    _, _, st = oracle_lib.render(flat, cfg, n_ranks=share, rank=0, n_threads=cores, aux=False, impl="simd")
    dt = max(st["kernel_ms"] * 1e-3, 1e-3)
    rays = st["rays_primary"] + st["rays_reflection"] + st["rays_refraction"]
    n_tiles = -(-n_tiles_frame // share)
    return {
        "value": rays / dt / 1e6, "unit": "Mray/s", "cores": cores, "kind": "port",
        "sample": f"{n_tiles} of the frame's {n_tiles_frame} 48x48 tiles, spread over the frame (lattice interleave 1/{share}): "
                  f"{rays} rays + {st['rays_shadow']} shadow rays in {dt:.1f} s; 8-lane AVX2 packets, rows of 48x48 tiles, brute-force "
                  f"scan of {flat.n_objects} objects per ray (restated simd_render: the reference is Rust nightly and cannot "
                  f"be built offline)",
        "build": build,
        "mshadow_per_s": st["rays_shadow"] / dt / 1e6,
        "per_core": rays / dt / 1e6 / cores,
        # (only sizes the sample above: 8 adjacent pixels where every ray hits -- NOT comparable with the spread sample, so no
        # parallel-efficiency figure is derived from it)
        "sizing_probe_one_thread": {"value": rays1 / dt1 / 1e6, "unit": "Mray/s", "sample": f"8 px at the frame centre, {dt1:.2f} s"},
        "cpu_model": cpu_info(), "cpus_visible": os.cpu_count(), "cpus_granted": cores,
    }


def boundary_costs(cfg, flat, lib, _abi, _lib, device):
    """What a drop-in pays around the hot path, end to end: scene preparation (host BVH + 8 octant copies + threaded
    copy + receiver-cell tables + upload), the first rt_render call (adds the sample-table uploads and, with soft
    shadows, rt_flags_kernel) and a steady-state rt_render call (host buffer in, packed pixels out over PCIe), plus the
    Python mirror's per-call flatten() + fingerprint() (RaytracerRenderer.render re-reads the scene like the reference's
    render(&buffer, &scene)).  The reference's analogue is ONE call, src/main.rs:330-334."""
    from hslu_i.ba_raytracing.f2501_raytracer_amd import scenes as _scenes  # noqa: F401
    from hslu_i.ba_raytracing.f2501_raytracer_amd.renderer import DeviceScene

    t = time.perf_counter()
    ds = DeviceScene(flat, device=device, budget=SCENE_BUDGET)
    scene_create_ms = (time.perf_counter() - t) * 1e3
    p, keep = _abi.make_params(cfg)
    host = np.zeros(cfg.width * cfg.height, np.uint32)
    st = _abi.rt_stats()
    calls, kern, setup = [], [], []
    for _ in range(4):
        t = time.perf_counter()
        _lib.check(lib.rt_render(ds.handle, C.byref(p), host.ctypes.data, None, C.byref(st)))
        calls.append((time.perf_counter() - t) * 1e3)
        kern.append(st.kernel_ms)
        setup.append(st.setup_ms)
    t = time.perf_counter()
    fp = flat.fingerprint()
    fingerprint_ms = (time.perf_counter() - t) * 1e3
    ds.close()
    del fp
    return {
        "scene_create_ms": scene_create_ms, "first_call_ms": calls[0], "steady_call_ms": float(np.median(calls[1:])),
        "first_call_device_ms": kern[0], "steady_call_device_ms": float(np.median(kern[1:])),
        # rt_stats.setup_ms: device time of what the call enqueues BEFORE its render kernels -- on the first call the sample-table
        # uploads and rt_flags_kernel with the per-cell lists (once per scene and light-cloud size); kernel_ms never contains it
        "first_call_setup_ms": setup[0], "steady_call_setup_ms": float(np.median(setup[1:])), "fingerprint_ms": fingerprint_ms,
        "note": "rt_render with a pageable host buffer: H2D of the fill + frame + D2H of the packed pixels; first_call_setup_ms = "
                "table uploads + rt_flags_kernel (receiver flags and per-cell candidate lists)",
    }


def time_workload(key, lib, _abi, _lib, DeviceScene, torch, dev, device_index, streams, steps=8, warmup=4):
    """ms per frame and Mray/s of another workload: `steps` frames back to back through rt_render_device on two streams
    used alternately, HBM-resident frame buffers, wall clock between two synchronisations."""
    cfg, flat, name = build_workload(key)
    ds = DeviceScene(flat, device=device_index, budget=SCENE_BUDGET)
    p, keep = _abi.make_params(cfg)
    fbs = [torch.zeros(cfg.width * cfg.height, dtype=torch.int32, device=dev) for _ in range(2)]
    n = [0]  # (the caller's two streams: every further stream would have to share a hardware queue with one of them)

    def frame():  # two frames in flight, like the headline measurement
        i = n[0] & 1
        n[0] += 1
        _lib.check(lib.rt_render_device(ds.handle, C.byref(p), C.c_void_p(fbs[i].data_ptr()), None, C.c_void_p(streams[i].cuda_stream)))

    for _ in range(warmup):
        frame()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        frame()
    torch.cuda.synchronize(dev)
    sec = (time.perf_counter() - t0) / steps
    # ... and one frame after the other on ONE stream (what a single frame costs)
    t0 = time.perf_counter()
    for _ in range(max(2, steps // 2)):
        n[0] = 0
        frame()
        torch.cuda.synchronize(dev)
    sec_alone = (time.perf_counter() - t0) / max(2, steps // 2)
    st = _abi.rt_stats()
    _lib.check(lib.rt_render_collect_stats(ds.handle, C.byref(st)))
    rays = st.rays_primary + st.rays_reflection + st.rays_refraction
    out = {"workload": name, "steps": steps, "frames_in_flight": 2, "ms_per_step": sec * 1e3, "value": rays / sec / 1e6, "unit": "Mray/s",
           "ms_per_frame_alone": sec_alone * 1e3, "value_alone": rays / sec_alone / 1e6,
           "value_traced": st.rays_traced / sec / 1e6, "mshadow_per_s": st.rays_shadow / sec / 1e6, "rays_per_frame": rays,
           "queue_bytes": int(st.queue_bytes), "notes": int(st.notes)}
    ds.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)  # ~1 s of frames: enough for a GPU-busy sampler to see them
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-boundary-costs", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="default run (config 3, one GPU): skip the short timings of configs 4 and 5 appended as `other_workloads`")
    ap.add_argument("--in-flight", type=int, default=0, choices=[0, 1, 2, 3, 4],
                    help="frames in flight (streams used alternately).  0 = 2: the head of a frame fills the CUs the drain of the one "
                         "before leaves idle (frames with secondary rays: every ray-tree level is a launch with a drain of its own)")
    ap.add_argument("--sub-frames", type=int, default=0, choices=[0, 1, 2],
                    help="rt_tuning.sub_frames: chains a frame with secondary rays is split into.  0 = the library's choice (two while "
                         "no other frame of the scene is running, one otherwise); profiles use 1 (chains stretch each other's launches)")
    ap.add_argument("--phases", type=int, default=0, choices=[0, 1, 2],
                    help="rt_tuning.phases: 0 = the library's choice, 1 = fused kernels, 2 = phase-split pipeline (csrc/rt_phases.h)")
    ap.add_argument("--levels", type=int, default=0, choices=[0, 1, 2, 3],
                    help="rt_tuning.levels: 0 = the library's choice, 1 = trace/sort/shade per level, 2 = all levels traced first + one shade "
                         "launch, 3 = levels traced back to back, each shaded on one of two streams as soon as it is sorted")
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
    t_scene = time.perf_counter()
    ds = DeviceScene(flat, device=local_rank, budget=SCENE_BUDGET)
    first_scene_create_ms = (time.perf_counter() - t_scene) * 1e3  # (the first scene of a process: includes HIP initialisation)
    p, keep = _abi.make_params(cfg, n_ranks=world, rank=rank, tuning=dict(sub_frames=args.sub_frames, phases=args.phases, levels=args.levels))
    npix = cfg.width * cfg.height
    n_fly = args.in_flight or 2
    if args.backend == "gloo" and world > 1:
        n_fly = 1  # (the host rehearsal synchronises every frame)
    # one frame buffer and one stream per frame in flight (a displayed sequence is double buffered anyway)
    fbs = [torch.zeros(npix, dtype=torch.int32, device=dev) for _ in range(n_fly)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_fly)]
    # N > 1: the library's own RCCL communicator (torch.distributed only carries its 128-byte id)
    rccl = RcclGather(world, rank, local_rank) if (world > 1 and args.backend == "nccl") else None
    host_gather = HostGather(cfg, world, rank) if (world > 1 and args.backend == "gloo") else None
    gather_ms_host = []

    def frame(i, ev0=None, ev1=None):
        stream, fb = streams[i % n_fly], fbs[i % n_fly]
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

    for i in range(max(args.warmup, 1 if world > 1 else 0)):  # N > 1: the first gather connects the peers (set-up, not a step)
        frame(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        frame(i)
    barrier()
    elapsed = time.perf_counter() - t0
    # The dominant kernel's launch duration, for the roofline: frames one after the other on ONE stream, each bracketed by
    # events on that stream (with two frames in flight a launch shares the GPU with its neighbour and its duration says
    # nothing about the kernel).  Outside the timed region.
    n_iso = max(1, min(args.steps, 20))
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_iso)]
    n_fly_timed, n_fly = n_fly, 1
    for i in range(n_iso):
        frame(0, *evs[i])
    barrier()
    n_fly = n_fly_timed
    frame_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))  # device time of one step alone on its stream
    infos = []
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
    elapsed, frame_ms_slowest = tmax.tolist()
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
        # (3) flops, SURVEY 8(d): what the reference's brute-force scan would have spent on this frame's rays (every ray
        # AND every shadow ray tests every object, raytracer.rs:48,180), and what the kernels actually executed (fp32
        # vector instructions of the dominant kernel from the PMC summary x the lanes active in them; fma = 2 flop)
        per_scan = FLOP_SPHERE * flat.n_spheres + FLOP_TRIANGLE * flat.n_triangles
        brute = (rays + counts[3]) * per_scan
        flops = {
            "peak_tflops": FP32_VECTOR_PEAK_TFLOPS, "unit": "Gflop/s",
            "brute_force_equivalent": brute / sec_per_step / 1e9,
            "brute_force_equivalent_frac_of_peak": brute / sec_per_step / 1e12 / FP32_VECTOR_PEAK_TFLOPS,
            "flop_per_scan": per_scan,
            "note": "brute-force equivalent = (rays + shadow rays, as the reference casts them) x (35 S + 85 T); above the "
                    "vector peak means the BVH and the beam tests removed that work, not that the ALUs did it",
            "bvh_actual": None, "bvh_actual_frac_of_peak": None,
        }
        if pmc and all(k in pmc for k in ("SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_ADD_F32")):
            lanes = 64.0
            if pmc.get("SQ_ACTIVE_INST_VALU") and pmc.get("SQ_THREAD_CYCLES_VALU"):
                lanes = min(64.0, pmc["SQ_THREAD_CYCLES_VALU"] / pmc["SQ_ACTIVE_INST_VALU"])
            wave_flop = 2.0 * pmc["SQ_INSTS_VALU_FMA_F32"] + pmc["SQ_INSTS_VALU_MUL_F32"] + pmc["SQ_INSTS_VALU_ADD_F32"] + \
                pmc.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
            actual = wave_flop * lanes  # per frame of the dominant kernel
            flops.update(bvh_actual=actual / (kernel_ms * 1e-3) / 1e9,
                         bvh_actual_frac_of_peak=actual / (kernel_ms * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS,
                         bvh_actual_flop_per_frame=actual, active_lanes_per_valu_inst=lanes, source=pmc["_file"])
        out = {
            "metric": ("Mray/s (primary+secondary, rays counted as the reference casts them), semesterbild@high_resolution"
                       if args.workload == "c3" else "Mray/s (primary+secondary, rays counted as the reference casts them)"),
            "value": mrays, "unit": "Mray/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": sec_per_step * 1e3, "higher_is_better": True, "scaling": "strong",
            # `value` / `ms_per_step` are PIPELINED throughput (frames_in_flight frames on alternating streams).  One frame alone on
            # one stream (device time between HIP events on that stream, mean of the isolated frames after the timed region;
            # slowest rank at N > 1, gather included):
            "ms_per_frame_alone": frame_ms_slowest, "value_alone": rays / (frame_ms_slowest * 1e-3) / 1e6,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # the rays the GPU actually traced (bit-identical repeats of the AA sample table are traced once)
            "value_traced": counts[5] / sec_per_step / 1e6,
            "config": {
                "workload": wname, "width": cfg.width, "height": cfg.height,
                "rays_per_frame": rays, "rays_traced_per_frame": counts[5],
                "shadow_rays_per_frame": counts[3], "pixels_written": counts[4],
                "mshadow_per_s": counts[3] / sec_per_step / 1e6,
                "objects": flat.n_objects, "lights": int(flat.lights.shape[0]) * cfg.point_light_multiplicator,
                "parallelism": f"tiles{cfg.render_stride}x{cfg.render_stride}/{world}gpu" + ("+rccl_gather" if world > 1 else ""),
                "frames_in_flight": n_fly_timed,
                "bvh": ds.bvh_info(), "build_id": build_id,
                "notes": int(st.notes), "queue_bytes": int(st.queue_bytes),
                "first_scene_create_ms_incl_hip_init": first_scene_create_ms,
                "scene_budget_bytes": SCENE_BUDGET, "scene_memory": ds.memory_info(), "phases": args.phases,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": pmc["_file"] if (pmc and traffic is not None) else None,
                "kernel": DOMINANT_KERNEL[args.workload], "kernel_ms": kernel_ms,
                "kernel_ms_note": f"mean of {n_iso} launches alone on one stream, HIP events on that stream, after the timed region",
                "algorithmic_bytes": own["rays_traced"] * ALG_BYTES_PER_RAY,
                "note": "64 B per TRACED ray (SURVEY 8d ray-stream model); no ray touches HBM in this kernel, so this "
                        "fraction cannot rank it -- the applicable bound is `valu_issue` below",
                "valu_issue": valu if valu is not None else
                {"bound": "valu_issue", "frac": None,
                 "note": f"no profiles/*_{args.workload}_pmc.csv was collected on build {build_id} (tools/profile.sh)"},
                "flops": flops,
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
                host.copy_(fbs[0], non_blocking=True)
                e1.record()
                torch.cuda.synchronize(dev)
                d2h.append(e0.elapsed_time(e1))
            out["d2h_ms"] = float(np.median(d2h[1:]))
            out["config"]["mray_per_s_incl_d2h"] = rays / (sec_per_step + out["d2h_ms"] * 1e-3) / 1e6
            if not args.no_boundary_costs:
                out["boundary"] = boundary_costs(cfg, flat, lib, _abi, _lib, local_rank)
        if world == 1 and args.workload == "c3" and not args.no_boundary_costs:
            # the same frame under the library's DEFAULT scene budget (128 MiB: receiver flags, no per-cell candidate lists)
            lean = DeviceScene(flat, device=local_rank)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
            for a, b in ev:
                a.record(streams[0])
                _lib.check(lib.rt_render_device(lean.handle, C.byref(p), C.c_void_p(fbs[0].data_ptr()), None, C.c_void_p(streams[0].cuda_stream)))
                b.record(streams[0])
            torch.cuda.synchronize(dev)
            out["lean_scene"] = {"ms_per_frame_alone": float(np.mean([a.elapsed_time(b) for a, b in ev[2:]])), "scene_memory": lean.memory_info(),
                                 "note": "rt_scene_desc.device_budget_bytes = 0 (library default)"}
            lean.close()
        if world == 1 and args.workload == "c3" and not args.no_other_workloads:
            # the other BASELINE configs under the same clock (a few frames each; their own bench lines: --workload c4 / c5)
            out["other_workloads"] = {k: time_workload(k, lib, _abi, _lib, DeviceScene, torch, dev, local_rank, (streams * 2)[:2]) for k in ("c4", "c5")}
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
