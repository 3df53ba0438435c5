"""Anti-aliasing sample table and soft-shadow light-cloud sets (seeded).

Reference: `ANTIALIASING_SAMPLES` / `get_antialiasing_simpling_directions` /
`bundle_rays_for_simd_antialiased_raytracing` (`src/renderer/raytracer_renderer.rs:105-127,
876-916,1021-1138`) and `PointLight::to_point_light_cloud` (`src/scene/lighting/light.rs:183-225`).

The reference draws both from an *unseeded* `fast_poisson` 1.0.2 Bridson sampler, so
its images are not reproducible run to run.  Here the same constructions are driven by a seeded
generator; the Bridson algorithm is restated from the published method (fast_poisson is not
vendored in the reference tree): first point uniform in the box, then candidates in the annulus
[r, 2r) around a random active point, `k` attempts per active point before it is retired.
"""
from __future__ import annotations

import math
from typing import List, Sequence

import numpy as np

from .config import RenderConfig
from .f32math import F, Vec3


def bridson_points(dims: Sequence[float], radius: float, k: int, max_points: int, rng: np.random.Generator) -> np.ndarray:
    """First `max_points` points of a Bridson Poisson-disk sequence in the box [0,dims)."""
    nd = len(dims)
    dims = np.asarray(dims, np.float64)
    cell = radius / math.sqrt(nd)
    grid_shape = tuple(int(math.ceil(d / cell)) for d in dims)
    grid = {}
    pts: List[np.ndarray] = []
    active: List[int] = []

    def cell_of(p):
        return tuple(int(c) for c in np.floor(p / cell))

    def ok(p) -> bool:
        if np.any(p < 0) or np.any(p >= dims):
            return False
        c = cell_of(p)
        rng_cells = [range(max(ci - 2, 0), min(ci + 3, gs)) for ci, gs in zip(c, grid_shape)]
        for idx in np.ndindex(*[len(r) for r in rng_cells]):
            key = tuple(r[i] for r, i in zip(rng_cells, idx))
            j = grid.get(key)
            if j is not None and np.sum((pts[j] - p) ** 2) < radius * radius:
                return False
        return True

    def add(p):
        grid[cell_of(p)] = len(pts)
        active.append(len(pts))
        pts.append(p)

    add(rng.random(nd) * dims)
    while active and len(pts) < max_points:
        ai = int(rng.integers(0, len(active)))
        base = pts[active[ai]]
        for _ in range(k):
            dist = radius * (1.0 + rng.random())
            v = rng.standard_normal(nd)
            v /= np.linalg.norm(v)
            cand = base + v * dist
            if ok(cand):
                add(cand)
                break
        else:
            active[ai] = active[-1]
            active.pop()
    return np.asarray(pts[:max_points], np.float64)


def aa_sample_table(cfg: RenderConfig) -> np.ndarray:
    """ANTIALIASING_SAMPLES, raytracer_renderer.rs:105-127: [0,0], 8x[1,1], then Poisson2D points
    (anti_aliasing_randomness) or [1,1]s; truncated to total8 = spp.next_multiple_of(8)."""
    total = -(-cfg.samples_per_pixel // 8) * 8
    samples = [[0.0, 0.0]] + [[1.0, 1.0]] * 8
    if cfg.has("anti_aliasing_randomness"):
        rng = np.random.default_rng(cfg.aa_seed)
        pts = bridson_points([1.2, 1.2], 3.0 / float(np.float32(total)), total, total - 1, rng)
        samples.extend(pts.tolist())
    else:
        samples.extend([[1.0, 1.0]] * total)
    return np.asarray(samples[:total], np.float32)


def aa_directions(cfg: RenderConfig) -> List[Vec3]:
    """get_antialiasing_simpling_directions, raytracer_renderer.rs:876-916 (t,l,b,r,tl,tr,bl,br)."""
    if cfg.has("anti_aliasing_rotation_scale"):
        ang = F(math.atan(float(F(0.5))))
        s, c = F(math.sin(float(ang))), F(math.cos(float(ang)))
        x, y = Vec3.unit_x(), Vec3.unit_y()
        x_r = x.mul_add(Vec3.broadcast(c), y * Vec3.broadcast(s))
        y_r = x.mul_add(Vec3.broadcast(-s), y * Vec3.broadcast(c))
    else:
        x_r, y_r = Vec3.unit_x(), Vec3.unit_y()
    t, l, r, b = -y_r, -x_r, x_r, y_r
    tl, tr, bl, br = t + l, t + r, b + l, b + r
    return [v.normalized() for v in (t, l, b, r, tl, tr, bl, br)]


def aa_offsets(cfg: RenderConfig) -> np.ndarray:
    """Per-sample (dx, dy) added to the pixel's scene coordinate,
    bundle_rays_for_simd_antialiased_raytracing (Vec3x8 path), raytracer_renderer.rs:1021-1138.
    The direction cycle restarts in every 8-lane chunk (:1111-1116)."""
    total = cfg.aa_total_rays
    table = aa_sample_table(cfg)
    dirs = aa_directions(cfg)
    if cfg.has("anti_aliasing_rotation_scale"):
        scale = F(np.sqrt(F(5.0))) / F(2.05)
    else:
        scale = F(0.85)
    out = np.zeros((total, 2), np.float32)
    for k in range(total):
        px = F(table[k, 0]) * cfg.fw * scale
        py = F(table[k, 1]) * cfg.fh * scale
        d = dirs[k % 8]
        out[k, 0] = px * d.x
        out[k, 1] = py * d.y
    return out


def cloud_sets(cfg: RenderConfig) -> np.ndarray:
    """`n_cloud_sets` independent N-point sets, each built like to_point_light_cloud<N>
    (light.rs:183-225): first N points of Poisson3D([R,R,R], r = 4/N, k = N), padded with
    uniform*R; in "pixel units" (the kernel multiplies by fw, fh, fd)."""
    n = cfg.point_light_multiplicator
    if n <= 1:
        return np.zeros((1, 1, 3), np.float32)
    rng = np.random.default_rng(cfg.cloud_seed)
    R = float(F(1.725) + F(n) / F(20.0))
    sets = np.zeros((cfg.n_cloud_sets, n, 3), np.float32)
    for s in range(cfg.n_cloud_sets):
        pts = bridson_points([R, R, R], 4.0 / n, n, n, rng)
        if pts.shape[0] < n:
            pad = rng.random((n - pts.shape[0], 3)) * R
            pts = np.concatenate([pts, pad], axis=0)
        sets[s] = pts.astype(np.float32)
    return sets
