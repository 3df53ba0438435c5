#!/usr/bin/env python3
"""One-off wider fuzz of the conservative shortcuts against the brute-force oracle (the committed parity suite runs
seeds 1..10): random soups with all features, several windows per seed.  Usage: fuzz_sweep.py first last [variant]   (variant 2: odd frame sizes, 1-6 lights, degenerate object counts, edge windows)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_parity_gpu as T  # noqa: E402
from hslu_i.ba_raytracing.f2501_raytracer_amd import RenderConfig  # noqa: E402



def run(first, last, variant=1):
    """Seeds first..last of generator `variant` against the oracle; returns the number of seeds that disagree."""
    bad = 0
    for seed in range(first, last + 1):
        if variant == 2:
            # odd frame sizes (ragged last wavefronts and tiles everywhere), 1..6 lights, scenes from a handful of objects to thousands,
            # spheres only / triangles only now and then, windows of odd sizes touching the frame's edges
            import numpy as np
            r = np.random.default_rng(1000 + seed)
            W, H = int(r.integers(33, 200)), int(r.integers(17, 150))
            H = max(H, W // 5 + 1)  # (random_scene places objects in a box that needs some height)
            feats = [["realistic", "anti_aliasing", "soft_shadows"], ["anti_aliasing", "high_quality"], ["soft_shadows", "reflections"],
                     ["realistic"], ["anti_aliasing", "soft_shadows", "refractions"]][seed % 5]
            cfg = RenderConfig.from_features(feats, width_override=W, height_override=H, n_cloud_sets=int(r.integers(4, 33)),
                                             depth_override=int(r.integers(1, 6)) if ("realistic" in feats or "reflections" in feats or "refractions" in feats) else None,
                                             cloud_seed=seed)
            n_s = 0 if seed % 7 == 0 else int(r.integers(1, 41))
            n_t = 0 if seed % 11 == 0 else int(r.integers(1, 3000 if seed % 4 == 0 else 400))
            if n_s + n_t == 0:
                n_s = 3
            flat = T.random_scene(seed, n_spheres=n_s, n_tris=n_t, n_lights=int(r.integers(1, 7)), cfg=cfg)
            ww, wh = int(r.integers(1, min(W, 72) + 1)), int(r.integers(1, min(H, 56) + 1))
            win = (int(r.integers(0, W - ww + 1)), int(r.integers(0, H - wh + 1)), ww, wh)
            try:
                T.compare(cfg, flat, win)
            except AssertionError as e:
                bad += 1
                print(f"seed {seed} (variant 2: {W}x{H} {feats} spheres {n_s} tris {n_t} window {win}): FAIL {str(e)[:200]}", flush=True)
            continue
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
    return bad


if __name__ == "__main__":
    sys.stdout.reconfigure(line_buffering=True)  # (a redirected log must show progress: a silent GPU job is taken for hung)
    sys.exit(1 if run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 1) else 0)
