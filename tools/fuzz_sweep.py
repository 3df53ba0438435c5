#!/usr/bin/env python3
"""One-off wider fuzz of the conservative shortcuts against the brute-force oracle (the committed parity suite runs
seeds 1..10): random soups with all features, several windows per seed.  Usage: fuzz_sweep.py first last"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_parity_gpu as T  # noqa: E402
from hslu_i.ba_raytracing.f2501_raytracer_amd import RenderConfig  # noqa: E402

sys.stdout.reconfigure(line_buffering=True)  # (a redirected log must show progress: a silent GPU job is taken for hung)
first, last = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, last + 1):
    feats = ["realistic", "anti_aliasing", "soft_shadows"] if seed % 3 else ["anti_aliasing", "high_quality"]
    cfg = RenderConfig.from_features(feats, width_override=160, height_override=128, n_cloud_sets=16,
                                     depth_override=3 if seed % 3 else None, cloud_seed=seed)
    flat = T.random_scene(seed, n_spheres=3 + seed % 12, n_tris=200 + 37 * (seed % 40), n_lights=2 + seed % 3, cfg=cfg)
    win = ((11 * seed) % 96, (5 * seed) % 80, 64, 48)
    try:
        T.compare(cfg, flat, win)
    except AssertionError as e:
        bad += 1
        print(f"seed {seed}: FAIL {str(e)[:200]}", flush=True)
print(f"{last - first + 1 - bad} of {last - first + 1} seeds agree with the oracle")
sys.exit(1 if bad else 0)
