"""The 8-lane AVX2 packet baseline (oracle/rt_simd_baseline.c, bench.py's cpu_baseline) against the scalar parity
oracle (oracle/rt_oracle.c): same op sequence per lane, so hit ids, distances and un-quantised colours are bit-identical
-- with and without anti-aliasing (the two packet shapes of the reference's simd_render path), soft shadows, secondary
rays, ragged tiles and tile ownership."""
import numpy as np
import pytest

import oracle_lib
from hslu_i.ba_raytracing.f2501_raytracer_amd import RenderConfig, scenes


def both(cfg, flat, win, **kw):
    a = oracle_lib.render(flat, cfg, window=win, n_threads=4, **kw)
    b = oracle_lib.render(flat, cfg, window=win, n_threads=4, impl="simd", **kw)
    return a, b


def assert_same(a, b):
    (argb_a, pa, sa), (argb_b, pb, sb) = a, b
    assert np.array_equal(pa["hit_id"], pb["hit_id"])
    assert np.array_equal(pa["hit_t"].view(np.uint32), pb["hit_t"].view(np.uint32))
    assert np.array_equal(pa["rgb"].view(np.uint32), pb["rgb"].view(np.uint32)), np.abs(pa["rgb"] - pb["rgb"]).max()
    assert np.array_equal(argb_a, argb_b)
    for k in ("rays_primary", "rays_reflection", "rays_refraction", "rays_shadow", "pixels_written", "rays_traced"):
        assert sa[k] == sb[k], (k, sa[k], sb[k])


@pytest.mark.parametrize("features,kw,win", [
    ([], {}, (190, 140, 61, 53)),                                                      # 8-pixel packets, ragged tails
    (["reflections", "refractions"], dict(depth_override=5), (300, 180, 50, 50)),      # packet recursion with lane masks
    (["anti_aliasing", "soft_shadows"], dict(n_cloud_sets=16), (330, 240, 20, 14)),    # sample packets, per-pixel clouds
    (["realistic", "extreme_quality"], dict(n_cloud_sets=8, depth_override=3), (401, 263, 9, 7)),  # 24 samples = 3 packets
    (["backface_culling", "reflections"], {}, (250, 200, 40, 30)),
    (["soft_shadows"], dict(n_cloud_sets=8), (350, 250, 21, 9)),                       # no AA: 8 pixels, 8 different clouds
])
def test_packets_equal_scalar_oracle_on_test_scene(features, kw, win):
    cfg = RenderConfig.from_features(features, **kw)
    flat = scenes.test_scene(cfg).flatten()
    assert_same(*both(cfg, flat, win))


def test_packets_equal_scalar_oracle_on_semesterbild_and_tile_ownership():
    cfg = RenderConfig.from_features(["high_resolution", "anti_aliasing", "soft_shadows"], n_cloud_sets=16)
    flat = scenes.semesterbild(cfg, "text_lowres").flatten()
    assert_same(*both(cfg, flat, (420, 330, 6, 4)))
    cfg1 = RenderConfig.from_features([], width_override=200, height_override=150)
    flat1 = scenes.test_scene(cfg1).flatten()
    for rank in range(3):
        assert_same(*both(cfg1, flat1, None, n_ranks=3, rank=rank))
