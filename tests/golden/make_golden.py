#!/usr/bin/env python3
"""Generates the golden render fixtures tests/golden/*.npz with the CPU oracle.

The reference cannot be run (Rust nightly) and holds no golden vectors for this path,
so these fixtures are outputs of oracle/rt_oracle.c on seeded inputs: they pin the oracle against
regressions and give the GPU tests a committed expectation.  Inputs are rebuilt from the recorded
feature list / window / seeds (see tests/test_oracle_golden.py).  Run: python make_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np

import oracle_lib
from hslu_i.ba_raytracing.f2501_raytracer_amd import RenderConfig, scenes

CASES = {
    # BASELINE config 1 (reference's own CPU-runnable case): test_scene 768x640, no AA / secondary
    "c1_test_scene": dict(scene="test_scene", features=[], window=(300, 220, 128, 96), kw={}),
    # BASELINE config 2: spheres only, medium resolution
    "c2_spheres_only": dict(scene="test_scene_spheres", features=["medium_resolution"], window=(380, 250, 160, 120), kw={}),
    # everything on: AA (random + rotation table), soft shadows, reflections + refractions
    "test_scene_everything": dict(scene="test_scene", features=["realistic", "high_quality", "anti_aliasing_randomness",
                                                               "anti_aliasing_rotation_scale"],
                                  window=(430, 170, 40, 32), kw=dict(n_cloud_sets=32, depth_override=4)),
    # BASELINE config 3 on the low-res mesh
    "c3_semesterbild_lowres": dict(scene="semesterbild:text_lowres", features=["high_resolution", "anti_aliasing", "soft_shadows"],
                                   window=(420, 330, 24, 16), kw=dict(n_cloud_sets=64)),
    # BASELINE config 4 flavour: reflections + refractions through the glass sphere, depth 8
    "c4_semesterbild_realistic": dict(scene="semesterbild:text_lowres", features=["high_resolution", "realistic"],
                                      window=(1236, 990, 32, 24), kw=dict(depth_override=8)),
}


def build(case):
    cfg = RenderConfig.from_features(case["features"], **case["kw"])
    s = case["scene"]
    if s == "test_scene":
        flat = scenes.test_scene(cfg).flatten()
    elif s == "test_scene_spheres":
        flat = scenes.test_scene(cfg).flatten().without_triangles()
    else:
        flat = scenes.semesterbild(cfg, s.split(":")[1]).flatten()
    return cfg, flat


def crop(cfg, win, a):
    x0, y0, w, h = win
    return a.reshape((cfg.height, cfg.width) + a.shape[1:])[y0:y0 + h, x0:x0 + w].copy()


if __name__ == "__main__":
    for name, case in CASES.items():
        cfg, flat = build(case)
        argb, planes, st = oracle_lib.render(flat, cfg, window=case["window"])
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"),
            meta=json.dumps(dict(case, stats={k: v for k, v in st.items() if k.startswith(("rays", "pixels"))})),
            argb=crop(cfg, case["window"], argb), hit_id=crop(cfg, case["window"], planes["hit_id"]),
            hit_t=crop(cfg, case["window"], planes["hit_t"]), rgb=crop(cfg, case["window"], planes["rgb"]))
        print(name, st)
