#!/usr/bin/env python3
"""Quick kernel timing for A/B work (not the official metric; the per-wave work statistics need
RT_HIP_LIB=.../librt_hip_stats.so, built by `make STATS=1` in csrc): renders named workloads (optionally a
window) through rt_render_device with HBM-resident buffers and prints kernel ms + ray rates.
Usage: perf_ab.py [c3|c4|c2|default][:x0,y0,w,h] ... [--reps N]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from hslu_i.ba_raytracing.f2501_raytracer_amd import RenderConfig, _abi, _lib, scenes
from hslu_i.ba_raytracing.f2501_raytracer_amd.config import DEFAULT_FEATURES
from hslu_i.ba_raytracing.f2501_raytracer_amd.renderer import DeviceScene

reps = 3
args = [a for a in sys.argv[1:]]
if "--reps" in args:
    i = args.index("--reps")
    reps = int(args[i + 1])
    del args[i:i + 2]
lib = _lib.load()
dev = torch.device("cuda", 0)
for spec in args or ["c3"]:
    spec, _, rk = spec.partition("@")  # name[:window][@rank/n_ranks]
    name, _, win = spec.partition(":")
    rank, n_ranks = (int(v) for v in rk.split("/")) if rk else (0, 1)
    window = tuple(int(v) for v in win.split(",")) if win else None
    if name.startswith("sem="):
        # sem=feat+feat[/depth][/model]  e.g. sem=high_resolution+realistic+high_quality/8/text
        parts = name[4:].split("/")
        kw = {}
        if len(parts) > 1 and parts[1]:
            kw["depth_override"] = int(parts[1])
        cfg = RenderConfig.from_features(parts[0].split("+"), **kw)
        flat = scenes.semesterbild(cfg, parts[2] if len(parts) > 2 else None).flatten()
    elif name == "default":
        cfg = RenderConfig.from_features(DEFAULT_FEATURES)
        flat = scenes.semesterbild(cfg).flatten()
    else:
        cfg, flat, _ = bench.build_workload(name)
    if os.environ.get("RT_AB_LIGHTS"):  # cost model experiments: keep only the first k lights
        import dataclasses
        flat = dataclasses.replace(flat, lights=flat.lights[: int(os.environ["RT_AB_LIGHTS"])])
    # RT_AB_BVH="max_leaf=8,tri_cost=1.0": rt_bvh_tuning fields
    bvh = {k: (float(v) if "." in v else int(v)) for k, v in (kv.split("=") for kv in os.environ.get("RT_AB_BVH", "").split(",") if kv)}
    ds = DeviceScene(flat, 0, bvh=bvh, budget=int(os.environ.get('RT_AB_BUDGET_MB', '2048')) << 20)  # (like bench.py)
    # RT_AB_TUNING="no_aa_dedup=1,chunk_log2=20": rt_tuning fields
    tuning = {k: int(v, 0) for k, v in (kv.split("=") for kv in os.environ.get("RT_AB_TUNING", "").split(",") if kv)}
    p, keep = _abi.make_params(cfg, window=window, n_ranks=n_ranks, rank=rank, tuning=tuning)
    if os.environ.get("RT_AB_TILE"):  # experiments: ownership tile size of the multi-GPU partition
        p.tile_size = int(os.environ["RT_AB_TILE"])
    fb = torch.zeros(cfg.width * cfg.height, dtype=torch.int32, device=dev)
    times = []
    for r in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.rt_render_device(ds.handle, C.byref(p), C.c_void_p(fb.data_ptr()), None, None))
        e1.record()
        torch.cuda.synchronize()
        if r > 0:
            times.append(e0.elapsed_time(e1))
    st = _abi.rt_stats()
    _lib.check(lib.rt_render_collect_stats(ds.handle, C.byref(st)))
    rays = st.rays_primary + st.rays_reflection + st.rays_refraction
    ms = float(np.median(times))
    print(f"{spec:28s} kernel {ms:10.3f} ms  (min {min(times):.3f})  rays {rays:>11d} (traced {st.rays_traced})  shadow {st.rays_shadow:>13d}  "
          f"simd_eff {st.wave_ray_lanes/max(1, 64*st.wave_ray_passes):.3f}  {rays/ms/1e3:8.1f} Mray/s  {st.rays_shadow/ms/1e6:7.2f} Gshadow/s  checksum {int(fb.to(torch.int64).sum()) & 0xFFFFFFFF:08x}")
    sp = max(1, st.wave_shadow_passes)
    print(f"{'':28s} per wave-pass: nearest nodes {st.wave_nearest_nodes/max(1,st.wave_ray_passes):.1f} tris {st.wave_nearest_tris/max(1,st.wave_ray_passes):.1f} | "
          f"shadow passes {st.wave_shadow_passes} nodes {st.wave_shadow_nodes/sp:.1f} tris {st.wave_shadow_tris/sp:.1f} exact {st.wave_shadow_tris_exact/sp:.2f}")
    if os.environ.get("RT_HIP_LIB", "").endswith("_prof4.so"):
        # make PROFILE=4 build: lane occupancy per class of (wavefront, light) set.  Counter order (wave_flush, RT_PROFILE):
        # wave_ray_lanes = prof[6], nearest_nodes = prof[0], nearest_tris = prof[1], shadow_nodes = prof[2], shadow_tris = prof[3],
        # nearest_tris_exact = prof[4], shadow_tris_exact = prof[5]
        n_no, l_no, n_tr, l_tr = st.wave_nearest_nodes, st.wave_nearest_tris, st.wave_shadow_nodes, st.wave_shadow_tris
        walked, listed, pre = st.wave_nearest_tris_exact, st.wave_shadow_tris_exact, st.wave_ray_lanes
        print(f"{'':28s} sets with nothing to test: {n_no} ({l_no/max(1,n_no):.1f} lanes each) | sets traced: {n_tr} "
              f"({l_tr/max(1,n_tr):.1f} lanes each) | candidates by BVH walk: {walked} sets, by per-cell lists: {listed} sets "
              f"({pre/max(1,listed):.1f} slots per union)")
    elif os.environ.get("RT_HIP_LIB", "").endswith("_prof5.so"):
        # make PROFILE=5 build: what regrouping lanes could save in the sets that are traced with a shared candidate list
        sets, lanes, pairs, own, needy, sph, lens = (st.wave_nearest_nodes, st.wave_nearest_tris, st.wave_shadow_nodes, st.wave_shadow_tris,
                                                     st.wave_nearest_tris_exact, st.wave_shadow_tris_exact, st.wave_ray_lanes)
        print(f"{'':28s} LIST sets {sets}: {lanes/max(1,sets):.1f} lanes, list length {lens/max(1,sets):.2f}; (lane, candidate) pairs tested per sample "
              f"{pairs} -> surviving the lane's own beam {own} ({100.0*own/max(1,pairs):.1f} %); lanes with a candidate of their own "
              f"{needy} of {lanes} ({100.0*needy/max(1,lanes):.1f} %); (lane, sphere) pairs {sph}")
    elif os.environ.get("RT_HIP_LIB", "").endswith("_prof3.so"):
        # make PROFILE=3 build: outcome of the candidate sets that had something to test
        h = [st.wave_nearest_nodes, st.wave_nearest_tris, st.wave_shadow_nodes, st.wave_shadow_tris, st.wave_nearest_tris_exact,
             st.wave_shadow_tris_exact]
        tot = max(1, sum(h))
        names = ["tris: all lit", "tris: all occluded", "tris: mixed", "spheres only: all lit", "spheres only: all occluded", "spheres only: mixed"]
        print(f"{'':28s} set outcomes: " + "  ".join(f"{n} {100.0*v/tot:.1f}%" for n, v in zip(names, h)))
    elif os.environ.get("RT_HIP_LIB", "").endswith("_prof2.so"):
        # make PROFILE=2 build: histogram of the (wavefront, light) candidate sets
        h = [st.wave_nearest_nodes, st.wave_nearest_tris, st.wave_shadow_nodes, st.wave_shadow_tris, st.wave_nearest_tris_exact,
             st.wave_shadow_tris_exact]
        tot = max(1, sum(h))
        names = ["empty", "spheres only", "1-4 tris", "5-16 tris", ">16 tris", "overflow"]
        print(f"{'':28s} candidate sets: " + "  ".join(f"{n} {100.0*v/tot:.1f}%" for n, v in zip(names, h))
              + f"  | spheres kept per set {st.wave_ray_lanes/tot:.2f}")
    elif os.environ.get("RT_HIP_LIB", "").endswith("_prof.so"):
        # make PROFILE=1 build: the work counters carry summed wave-level shader-clock cycles per region
        reg = dict(nearest=st.wave_nearest_nodes, collect=st.wave_nearest_tris, setup=st.wave_shadow_nodes,
                   spheres=st.wave_shadow_tris, triangles=st.wave_nearest_tris_exact, lighting=st.wave_shadow_tris_exact)
        whole = max(1, st.wave_ray_lanes)
        print(f"{'':28s} region share of process_ray wave-cycles: " + "  ".join(f"{k} {100.0*v/whole:.1f}%" for k, v in reg.items())
              + f"  | cycles/shadow pass {whole/sp:.0f}")
    ds.close()
