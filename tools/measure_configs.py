#!/usr/bin/env python3
"""Fills BASELINE.md section 5: every BASELINE.json config on 1 GPU (+ the CPU oracle where it finishes)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bench
import oracle_lib

threads = os.cpu_count() or 1
for key in ("c1", "c2"):
    cfg, flat, name = bench.build_workload(key)
    t0 = time.time()
    _, _, st = oracle_lib.render(flat, cfg, n_threads=threads, aux=False)
    dt = time.time() - t0
    rays = st["rays_primary"]
    print(f"CPU oracle {key} full frame: {dt*1e3:.1f} ms on {threads} threads, {rays/dt/1e6:.3f} Mray/s, "
          f"{st['rays_shadow']/dt/1e6:.3f} Mshadow/s")
