"""The phase-split form of the render loop (rt_tuning.phases = RT_PHASES_SPLIT; csrc/rt_phases.h: hit -> classify ->
one kernel per class of (wavefront, light) set -> resolve) against the CPU oracle, and against the fused kernels:
every (hit point, light) share reaches the pixel as the same fixed-point term in both forms, so frames with secondary
rays must be the same bits; frames without (the fused kernel sums floats in the reference's lane order there) agree to
fp32 reassociation."""
import numpy as np
import pytest

import bench
import oracle_lib
from hslu_i.ba_raytracing.f2501_raytracer_amd import RenderConfig, _abi, scenes
from test_parity_gpu import RGB_TOL, gpu_render, random_scene, window_mask

pytestmark = pytest.mark.gpu

SPLIT = dict(phases=_abi.RT_PHASES_SPLIT)
FUSED = dict(phases=_abi.RT_PHASES_FUSED)
COUNTS = ("rays_primary", "rays_reflection", "rays_refraction", "rays_shadow", "pixels_written", "rays_traced")


def compare_split(cfg, flat, win, **tuning):
    argb_g, pg, sg = gpu_render(cfg, flat, win, **SPLIT, **tuning)
    argb_o, po, so = oracle_lib.render(flat, cfg, window=win)
    m = window_mask(cfg, win)
    assert np.array_equal(argb_g != 0, argb_o != 0)
    assert not (argb_g[~m] != 0).any()
    assert np.array_equal(pg["hit_id"], po["hit_id"])
    hit = m & (po["hit_id"] >= 0)
    assert np.array_equal(pg["hit_t"][hit].view(np.uint32), po["hit_t"][hit].view(np.uint32)), "hit t not bit-exact"
    d = np.abs(pg["rgb"] - po["rgb"]).max(axis=1)
    assert int((d > RGB_TOL).sum()) == 0, f"max {d.max():.3e}"
    for k in COUNTS[:5]:
        assert sg[k] == so[k], (k, sg[k], so[k])
    return float(d.max())


def test_split_config1_full_frame_hard_shadows():
    """N = 1: every set is of the WALK class (one BVH walk per shadow ray)."""
    cfg = RenderConfig.from_features([])
    flat = scenes.test_scene(cfg).flatten()
    compare_split(cfg, flat, None)


def test_split_config2_spheres_only():
    cfg = RenderConfig.from_features(["medium_resolution"])
    flat = scenes.test_scene(cfg).flatten().without_triangles()
    compare_split(cfg, flat, None)


def test_split_aa_soft_shadows_and_everything_on_windows():
    cfg = RenderConfig.from_features(["anti_aliasing", "soft_shadows"])
    flat = scenes.test_scene(cfg).flatten()
    compare_split(cfg, flat, (250, 180, 120, 80))
    cfg = RenderConfig.from_features(["realistic", "high_quality", "anti_aliasing_randomness", "anti_aliasing_rotation_scale"],
                                     n_cloud_sets=32, depth_override=4)
    flat = scenes.test_scene(cfg).flatten()
    compare_split(cfg, flat, (427, 171, 45, 37))
    compare_split(cfg, flat, (427, 171, 45, 37), shadow_candidate_cap=_abi.RT_CAND_CAP_NONE)


def test_split_linear_scan_and_backface_culling():
    cfg = RenderConfig.from_features(["realistic"], depth_override=3)
    flat = scenes.test_scene(cfg).flatten()
    win = (300, 200, 64, 48)
    a, p, s = gpu_render(cfg, flat, win, traversal=_abi.RT_TRAVERSAL_LINEAR, **SPLIT)
    b, q, t = gpu_render(cfg, flat, win, traversal=_abi.RT_TRAVERSAL_LINEAR, **FUSED)
    assert np.array_equal(a, b) and np.array_equal(p["rgb"].view(np.uint32), q["rgb"].view(np.uint32))
    assert all(s[k] == t[k] for k in COUNTS)
    cfg = RenderConfig.from_features(["realistic", "backface_culling", "soft_shadows"], depth_override=3, n_cloud_sets=16)
    compare_split(cfg, scenes.test_scene(cfg).flatten(), win)


@pytest.mark.parametrize("seed", [1, 4, 7])
def test_split_random_scenes_all_features(seed):
    cfg = RenderConfig.from_features(["realistic", "anti_aliasing", "soft_shadows"], width_override=160, height_override=128,
                                     depth_override=4, n_cloud_sets=16)
    flat = random_scene(seed, 5, 120, 3, cfg)
    compare_split(cfg, flat, (20, 16, 96, 80))


def test_split_config3_windows_vs_oracle_and_fused_full_frame():
    cfg, flat, _ = bench.build_workload("c3")
    for win in ((400, 380, 24, 16), (548, 418, 24, 16), (330, 700, 24, 16)):
        compare_split(cfg, flat, win)
    a, p, s = gpu_render(cfg, flat, **SPLIT)
    b, q, t = gpu_render(cfg, flat, **FUSED)
    assert np.array_equal(p["hit_id"], q["hit_id"]) and np.array_equal(p["hit_t"].view(np.uint32), q["hit_t"].view(np.uint32))
    assert all(s[k] == t[k] for k in COUNTS), (s, t)
    d = np.abs(p["rgb"] - q["rgb"]).max()
    assert d <= 2e-6, d  # (fixed-point per-light terms against the fused kernel's float chain over all lights)
    for sh in (16, 8, 0):
        assert np.abs(((a >> sh) & 0xFF).astype(np.int32) - ((b >> sh) & 0xFF).astype(np.int32)).max() <= 1
    # the split frame is deterministic and independent of how lanes are packed: every AA repeat traced, 3-rank union
    a2, p2, _ = gpu_render(cfg, flat, **SPLIT)
    assert np.array_equal(a, a2)
    a3, p3, s3 = gpu_render(cfg, flat, no_aa_dedup=1, **SPLIT)
    assert np.array_equal(a3, a) and np.array_equal(p3["rgb"].view(np.uint32), p["rgb"].view(np.uint32))
    acc = np.zeros_like(a)
    for rank in range(3):
        ar, _, _ = gpu_render(cfg, flat, n_ranks=3, rank=rank, aux=False, **SPLIT)
        assert not (acc[ar != 0] != 0).any()
        acc |= ar
    assert np.array_equal(acc, a)
    print(f"config 3: split {s['kernel_ms']:.2f} ms, fused {t['kernel_ms']:.2f} ms; max |dRGB| split vs fused {d:.2e}")


@pytest.mark.parametrize("key", ["c4", "c5"])
def test_split_streaming_frames_equal_the_fused_frames_bit_for_bit(key):
    cfg, flat, _ = bench.build_workload(key)
    a, p, s = gpu_render(cfg, flat, **SPLIT)
    b, q, t = gpu_render(cfg, flat, **FUSED)
    assert all(s[k] == t[k] for k in COUNTS), (s, t)
    assert np.array_equal(p["rgb"].view(np.uint32), q["rgb"].view(np.uint32))
    assert np.array_equal(a, b)
    a1, _, s1 = gpu_render(cfg, flat, aux=False, sub_frames=1, **SPLIT)
    assert np.array_equal(a1, a) and all(s1[k] == s[k] for k in COUNTS)
    a3, _, s3 = gpu_render(cfg, flat, aux=False, chunk_log2=20, **SPLIT)
    assert np.array_equal(a3, a) and all(s3[k] == s[k] for k in COUNTS)
    print(f"{key}: split {s['kernel_ms']:.1f} ms, fused {t['kernel_ms']:.1f} ms")


MERGED = dict(levels=_abi.RT_LEVELS_MERGED)
PIPELINED = dict(levels=_abi.RT_LEVELS_PIPELINED)
CHAINED = dict(levels=_abi.RT_LEVELS_CHAINED)


@pytest.mark.parametrize("MODE", [MERGED, PIPELINED], ids=["merged", "pipelined"])
def test_merged_levels_against_the_oracle_and_the_chained_schedule(MODE):
    """rt_tuning.levels = RT_LEVELS_MERGED: rt_trace_spawn_kernel appends a ray's children where its hit is found, the levels are traced
    back to back and ONE shade launch handles the hits of all levels in one hit-point order.  The pixel sums are the same integers
    whatever the order: same bits as the chained schedule, and green against the oracle."""
    cfg = RenderConfig.from_features(["realistic", "high_quality", "anti_aliasing_randomness", "anti_aliasing_rotation_scale"],
                                     n_cloud_sets=32, depth_override=4)
    flat = scenes.test_scene(cfg).flatten()
    win = (427, 171, 45, 37)
    a, p, s = gpu_render(cfg, flat, win, **MODE)
    b, q, t = gpu_render(cfg, flat, win, **CHAINED)
    assert np.array_equal(a, b) and np.array_equal(p["rgb"].view(np.uint32), q["rgb"].view(np.uint32))
    assert all(s[k] == t[k] for k in COUNTS)
    argb_o, po, so = oracle_lib.render(flat, cfg, window=win)
    assert np.array_equal(p["hit_id"], po["hit_id"]) and float(np.abs(p["rgb"] - po["rgb"]).max()) <= RGB_TOL
    assert all(s[k] == so[k] for k in COUNTS[:5])
    # deep trees, forced small batches (queues that must grow), one chain / two chains, linear scan, culling
    cfg = RenderConfig.from_features(["realistic", "anti_aliasing", "soft_shadows"], depth_override=7, n_cloud_sets=16)
    flat = scenes.test_scene(cfg).flatten()
    win = (300, 200, 96, 64)
    ref, pr, sr = gpu_render(cfg, flat, win, **CHAINED)
    for kw in (dict(), dict(chunk_log2=10), dict(sub_frames=1), dict(sub_frames=2), dict(shadow_candidate_cap=_abi.RT_CAND_CAP_NONE)):
        a, p, s = gpu_render(cfg, flat, win, **MODE, **kw)
        assert np.array_equal(a, ref) and np.array_equal(p["rgb"].view(np.uint32), pr["rgb"].view(np.uint32)), kw
        assert all(s[k] == sr[k] for k in COUNTS), kw
    a, p, s = gpu_render(cfg, flat, win, traversal=_abi.RT_TRAVERSAL_LINEAR, **MODE)
    b, q, t = gpu_render(cfg, flat, win, traversal=_abi.RT_TRAVERSAL_LINEAR, **CHAINED)
    assert np.array_equal(a, b) and all(s[k] == t[k] for k in COUNTS)


@pytest.mark.parametrize("key,MODE", [("c4", MERGED), ("c4d21", MERGED), ("c5", MERGED), ("c4", PIPELINED)],
                         ids=["c4-merged", "c4d21-merged", "c5-merged", "c4-pipelined"])
def test_merged_levels_full_frames_equal_the_chained_frames_bit_for_bit(key, MODE):
    cfg, flat, _ = bench.build_workload(key)
    a, p, s = gpu_render(cfg, flat, **MODE)
    b, q, t = gpu_render(cfg, flat, **CHAINED)
    assert all(s[k] == t[k] for k in COUNTS), (s, t)
    assert np.array_equal(p["rgb"].view(np.uint32), q["rgb"].view(np.uint32))
    assert np.array_equal(a, b)
    a3 = np.zeros_like(a)
    for rank in range(3):
        ar, _, _ = gpu_render(cfg, flat, n_ranks=3, rank=rank, aux=False, **MODE)
        a3 |= ar
    assert np.array_equal(a3, a)
    print(f"{key}: {MODE} {s['kernel_ms']:.1f} ms, chained {t['kernel_ms']:.1f} ms (first frames of their shapes: verified, rendered twice if the queues had to grow)")
