#!/usr/bin/env python3
"""Renders a scene on GPU 0 and writes a PNG (FileOutput::render_buffer analogue, reference
src/output/file.rs:27-49).  Usage: render_png.py OUT.png scene feature[,feature...] [WxH] [model] [depth]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image

from hslu_i.ba_raytracing.f2501_raytracer_amd import RenderConfig, scenes
from hslu_i.ba_raytracing.f2501_raytracer_amd.config import DEFAULT_FEATURES
from hslu_i.ba_raytracing.f2501_raytracer_amd.renderer import ImageBuffer, RaytracerRenderer

out, scene_name, feats = sys.argv[1], sys.argv[2], sys.argv[3]
features = list(DEFAULT_FEATURES) if feats == "default" else [f for f in feats.split(",") if f]
kw = {}
if len(sys.argv) > 4 and sys.argv[4] != "-":
    w, h = sys.argv[4].split("x")
    kw.update(width_override=int(w), height_override=int(h))
model = sys.argv[5] if len(sys.argv) > 5 and sys.argv[5] != "-" else None
if len(sys.argv) > 6:
    kw.update(depth_override=int(sys.argv[6]))
cfg = RenderConfig.from_features(features, **kw)
scene = scenes.semesterbild(cfg, model) if scene_name == "semesterbild" else scenes.test_scene(cfg)
buf = ImageBuffer.new(cfg.width, cfg.height)
r = RaytracerRenderer(cfg)
t = time.time()
r.render(buf, scene)
print(f"{scene_name} {cfg.width}x{cfg.height} features={sorted(cfg.features)} objects={scene.num_objects()} "
      f"wall={time.time() - t:.3f}s stats={r.last_stats}")
Image.fromarray(buf.as_rgb8()).save(out)
