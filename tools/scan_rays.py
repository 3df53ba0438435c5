import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import bench
from hslu_i.ba_raytracing.f2501_raytracer_amd.renderer import ImageBuffer, RaytracerRenderer
cfg, flat, _ = bench.build_workload("c4d21")
r = RaytracerRenderer(cfg, device=0)
buf = ImageBuffer.new(cfg.width, cfg.height)
for y in range(986, 996, 2):
    row = []
    for x in range(1236, 1260, 2):
        r.render(buf, flat, window=(x, y, 1, 1))
        st = r.last_stats
        row.append(st["rays_primary"] + st["rays_reflection"] + st["rays_refraction"])
    print(y, row)
