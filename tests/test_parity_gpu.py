"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Bars: hit ids and pixel indices bit-exact; hit distance bit-exact (same fp32 op sequence, correctly
rounded div/sqrt on both sides); un-quantised RGB within 1e-4 (BASELINE.json north_star); packed
pixels may differ by 1 LSB where tanhf/powf differ in the last ulp.
"""
import numpy as np
import pytest

import oracle_lib
from hslu_i.ba_raytracing.f2501_raytracer_amd import RenderConfig, _abi, scenes
from hslu_i.ba_raytracing.f2501_raytracer_amd.renderer import ImageBuffer, RaytracerRenderer

pytestmark = pytest.mark.gpu

RGB_TOL = 1e-4


def window_mask(cfg, win):
    m = np.zeros((cfg.height, cfg.width), bool)
    if win is None:
        m[:] = True
    else:
        x0, y0, w, h = win
        m[y0:y0 + h, x0:x0 + w] = True
    return m.ravel()


def gpu_render(cfg, flat, win=None, traversal=_abi.RT_TRAVERSAL_BVH, n_ranks=1, rank=0):
    buf = ImageBuffer.new(cfg.width, cfg.height)
    r = RaytracerRenderer(cfg, device=0, traversal=traversal)
    planes = r.render(buf, flat, window=win, aux=True, n_ranks=n_ranks, rank=rank)
    return buf.buffer.copy(), planes, r.last_stats


def compare(cfg, flat, win, traversal=_abi.RT_TRAVERSAL_BVH, max_bad_px=0):
    argb_g, pg, sg = gpu_render(cfg, flat, win, traversal)
    argb_o, po, so = oracle_lib.render(flat, cfg, window=win)
    m = window_mask(cfg, win)
    # pixel indices: exactly the same pixels written, nothing outside the window touched
    assert np.array_equal(argb_g != 0, argb_o != 0)
    assert not (argb_g[~m] != 0).any()
    assert np.array_equal(pg["hit_id"], po["hit_id"]), \
        f"{(pg['hit_id'] != po['hit_id']).sum()} hit ids differ"
    hit = m & (po["hit_id"] >= 0)
    assert np.array_equal(pg["hit_t"][hit].view(np.uint32), po["hit_t"][hit].view(np.uint32)), "hit t not bit-exact"
    d = np.abs(pg["rgb"] - po["rgb"]).max(axis=1)
    bad = int((d > RGB_TOL).sum())
    assert bad <= max_bad_px, f"{bad} pixels exceed {RGB_TOL} (max {d.max():.3e})"
    # packed pixels: at most 1 LSB per channel
    for sh in (16, 8, 0):
        a = ((argb_g >> sh) & 0xFF).astype(np.int32)
        b = ((argb_o >> sh) & 0xFF).astype(np.int32)
        assert np.abs(a - b).max() <= 1
    for k in ("rays_primary", "rays_reflection", "rays_refraction", "rays_shadow", "pixels_written"):
        assert sg[k] == so[k], (k, sg[k], so[k])
    print(f"max |dRGB| vs oracle = {float(d.max()):.3e}")
    return float(d.max())


def test_config1_test_scene_full_frame():
    """BASELINE config 1 workload on the GPU: test_scene 768x640, no AA / secondary rays."""
    cfg = RenderConfig.from_features([])
    flat = scenes.test_scene(cfg).flatten()
    compare(cfg, flat, None)


def test_config2_spheres_only_medium_resolution():
    """BASELINE config 2: test_scene spheres only, 1140x950, no secondary rays."""
    cfg = RenderConfig.from_features(["medium_resolution"])
    flat = scenes.test_scene(cfg).flatten().without_triangles()
    compare(cfg, flat, None)


def test_linear_scan_mode_matches_oracle():
    cfg = RenderConfig.from_features([])
    flat = scenes.test_scene(cfg).flatten()
    compare(cfg, flat, (200, 150, 160, 128), traversal=_abi.RT_TRAVERSAL_LINEAR)


def test_reflections_refractions_window():
    cfg = RenderConfig.from_features(["realistic"])
    flat = scenes.test_scene(cfg).flatten()
    compare(cfg, flat, (280, 160, 192, 160))


def test_aa_soft_shadows_window():
    cfg = RenderConfig.from_features(["anti_aliasing", "soft_shadows"], n_cloud_sets=64)
    flat = scenes.test_scene(cfg).flatten()
    compare(cfg, flat, (300, 200, 96, 96))


def test_everything_on_window_ragged():
    """AA + rotation/random table + soft shadows + reflections/refractions; window not aligned to the
    16x16 workgroup tile and touching the frame corner."""
    cfg = RenderConfig.from_features(["realistic", "high_quality", "anti_aliasing_randomness",
                                      "anti_aliasing_rotation_scale"], n_cloud_sets=32, depth_override=4)
    flat = scenes.test_scene(cfg).flatten()
    compare(cfg, flat, (cfg.width - 37, cfg.height - 29, 37, 29))


def test_backface_culling_flag():
    cfg = RenderConfig.from_features(["backface_culling", "reflections"])
    flat = scenes.test_scene(cfg).flatten()
    compare(cfg, flat, (250, 200, 128, 96))


@pytest.mark.parametrize("model", ["text_lowres"])
def test_semesterbild_config3_windows(model):
    """BASELINE config 3 (semesterbild, high_resolution + anti_aliasing + soft_shadows) on windows the
    brute-force oracle finishes in seconds."""
    cfg = RenderConfig.from_features(["high_resolution", "anti_aliasing", "soft_shadows"], n_cloud_sets=64)
    flat = scenes.semesterbild(cfg, model).flatten()
    for win in ((420, 330, 32, 24), (800, 560, 24, 24), (1180, 1010, 32, 16)):
        compare(cfg, flat, win)


def test_semesterbild_realistic_window():
    cfg = RenderConfig.from_features(["high_resolution", "realistic"], depth_override=8)
    flat = scenes.semesterbild(cfg, "text_lowres").flatten()
    compare(cfg, flat, (700, 480, 48, 32))


def test_bvh_equals_linear_full_frame_semesterbild():
    """Size-independent property at full size: the BVH must not change any result of the linear scan."""
    cfg = RenderConfig.from_features(["high_resolution"])
    flat = scenes.semesterbild(cfg, "text").flatten()
    a1, p1, s1 = gpu_render(cfg, flat, None, _abi.RT_TRAVERSAL_BVH)
    a2, p2, s2 = gpu_render(cfg, flat, (0, 0, cfg.width, 160), _abi.RT_TRAVERSAL_LINEAR)
    m = window_mask(cfg, (0, 0, cfg.width, 160))
    assert np.array_equal(p1["hit_id"][m], p2["hit_id"][m])
    assert np.array_equal(a1[m], a2[m])


def test_tile_partition_union_equals_full():
    """Multi-GPU sharding property: the union of the ranks' tiles is the full frame, tiles disjoint."""
    cfg = RenderConfig.from_features(["reflections"])
    flat = scenes.test_scene(cfg).flatten()
    full, _, sf = gpu_render(cfg, flat)
    acc = np.zeros_like(full)
    written = 0
    for rank in range(3):
        part, _, sp = gpu_render(cfg, flat, n_ranks=3, rank=rank)
        assert not ((acc != 0) & (part != 0)).any()
        acc |= part
        written += sp["pixels_written"]
    assert np.array_equal(acc, full)
    assert written == sf["pixels_written"]


def test_empty_scene_and_prefilled_buffer():
    """Miss pixels are never written (image_buffer.rs:27-37): an empty scene leaves the fill intact."""
    from hslu_i.ba_raytracing.f2501_raytracer_amd import Scene
    cfg = RenderConfig.from_features([])
    buf = ImageBuffer.new_with_color(cfg.width, cfg.height, 0x12345678)
    RaytracerRenderer(cfg).render(buf, Scene.new().flatten())
    assert (buf.buffer == 0x12345678).all()


def test_error_paths():
    import ctypes as C
    from hslu_i.ba_raytracing.f2501_raytracer_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    d = _abi.rt_scene_desc()
    assert lib.rt_scene_create(C.byref(d), 0, C.byref(h)) == _abi.RT_ERR_INVALID_ARG  # abi_version 0
    assert b"abi_version" in lib.rt_last_error()
    cfg = RenderConfig.from_features([])
    with pytest.raises(ValueError):
        RaytracerRenderer(cfg).render(ImageBuffer.new(10, 10), scenes.test_scene(cfg))


# ---- committed golden fixtures (tests/golden/*.npz, oracle outputs) -----------------------------
import os  # noqa: E402
import sys  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden  # noqa: E402
from test_oracle_golden import check_against_fixture  # noqa: E402


@pytest.mark.parametrize("name", sorted(make_golden.CASES))
def test_gpu_reproduces_golden_fixture(name):
    case = make_golden.CASES[name]
    cfg, flat = make_golden.build(case)
    argb, planes, st = gpu_render(cfg, flat, tuple(case["window"]))
    check_against_fixture(name, argb, planes, st, cfg, rgb_tol=RGB_TOL)


def test_default_features_match_reference_output_png_statistically():
    """The reference's only result artefact: output.png = semesterbild at the default feature set,
    1140x950.  Stochastic reference -> compare 4x4 box-filtered images statistically."""
    from PIL import Image
    from hslu_i.ba_raytracing.f2501_raytracer_amd.config import DEFAULT_FEATURES
    cfg = RenderConfig.from_features(DEFAULT_FEATURES)
    scene = scenes.semesterbild(cfg)
    buf = ImageBuffer.new(cfg.width, cfg.height)
    RaytracerRenderer(cfg).render(buf, scene)
    img = buf.as_rgb8().astype(np.float32)
    h4, w4 = cfg.height // 4 * 4, cfg.width // 4 * 4
    box = img[:h4, :w4].reshape(h4 // 4, 4, w4 // 4, 4, 3).mean(axis=(1, 3))
    ref = np.asarray(Image.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                                             "reference_output_box4.png")).convert("RGB")).astype(np.float32)
    assert ref.shape == box.shape
    d = np.abs(box - ref)
    mse = float(((box - ref) ** 2).mean())
    psnr = 10 * np.log10(255.0 ** 2 / mse)
    iou = float(((box.sum(2) > 1) & (ref.sum(2) > 1)).sum() / ((box.sum(2) > 1) | (ref.sum(2) > 1)).sum())
    print(f"vs output.png (4x4 box): mean abs {d.mean():.3f}/255, PSNR {psnr:.1f} dB, silhouette IoU {iou:.5f}")
    assert d.mean() < 1.0 and psnr > 38.0 and iou > 0.995


@pytest.mark.parametrize("n_tris", [1, 2, 3, 5])
def test_tiny_triangle_counts_single_leaf_bvh(n_tris):
    """BVH edge cases: a scene whose whole triangle set is one leaf (root with an empty second child),
    with spheres, shadows and transmissive triangles."""
    from hslu_i.ba_raytracing.f2501_raytracer_amd.scene import FlatScene
    cfg = RenderConfig.from_features(["reflections", "refractions"], width_override=96, height_override=80)
    full = scenes.test_scene(cfg).flatten()
    keep = [0, 1, 2, 40, 60][:n_tris]
    flat = FlatScene(full.sphere_center, full.sphere_r_sq, full.sphere_r_inv, full.sphere_material,
                     full.tri_v1[keep], full.tri_e1[keep], full.tri_e2[keep], full.tri_normal[keep],
                     full.tri_material[keep], full.materials, full.lights)
    compare(cfg, flat, None)


def test_candidate_overflow_falls_back_to_per_sample_walk(monkeypatch):
    """Soft shadows share one BVH walk per (wavefront, light); when the candidate list overflows the
    kernel must fall back to a walk per sample with identical results.  RT_CAND_MAX (experiment knob
    of the library) forces the overflow."""
    cfg = RenderConfig.from_features(["high_resolution", "anti_aliasing", "soft_shadows"], n_cloud_sets=64)
    flat = scenes.semesterbild(cfg, "text_lowres").flatten()
    win = (420, 330, 64, 48)
    a_ref, p_ref, s_ref = gpu_render(cfg, flat, win)
    for cap in ("0", "3"):
        monkeypatch.setenv("RT_CAND_MAX", cap)
        a, p, s_ = gpu_render(cfg, flat, win)
        # every occlusion decision is identical; the colour differs only by the rounding of the fast
        # arrival path (sets with nothing to test skip the IEEE normalisation of the light direction)
        assert np.array_equal(p["hit_id"], p_ref["hit_id"]) and np.array_equal(p["hit_t"], p_ref["hit_t"])
        assert np.abs(p["rgb"] - p_ref["rgb"]).max() <= 2e-6
        ch = lambda v: np.stack([(v >> sh) & 0xFF for sh in (0, 8, 16, 24)]).astype(np.int32)
        assert np.abs(ch(a) - ch(a_ref)).max() <= 1
        assert s_["rays_shadow"] == s_ref["rays_shadow"]
    monkeypatch.delenv("RT_CAND_MAX")


def test_extreme_quality_24_samples_ragged_workgroups():
    """24 rays/pixel (extreme_quality): a 256-thread workgroup holds 10 pixels (240 lanes), so pixel
    groups straddle wavefronts and 4x4 tiles; N_cloud = 28."""
    cfg = RenderConfig.from_features(["extreme_quality", "reflections", "refractions"], n_cloud_sets=16, depth_override=3)
    flat = scenes.test_scene(cfg).flatten()
    compare(cfg, flat, (401, 263, 37, 23))


def test_config5_4k_aspect_window():
    """BASELINE config 5 geometry: 3840x2160 changes the scene itself (16:9 constants, lib.rs:73-92)."""
    cfg = RenderConfig.from_features(["realistic", "anti_aliasing", "soft_shadows"], width_override=3840,
                                     height_override=2160, n_cloud_sets=32, depth_override=4)
    flat = scenes.semesterbild(cfg, "text_lowres").flatten()
    compare(cfg, flat, (1800, 900, 40, 24))
    compare(cfg, flat, (2900, 1700, 24, 16))


def test_window_smaller_than_a_tile_and_single_pixel():
    cfg = RenderConfig.from_features(["anti_aliasing"])
    flat = scenes.test_scene(cfg).flatten()
    compare(cfg, flat, (123, 77, 1, 1))
    compare(cfg, flat, (0, 0, 3, 2))
    compare(cfg, flat, (cfg.width - 1, cfg.height - 1, 1, 1))


def random_scene(seed, n_spheres, n_tris, n_lights, cfg):
    """Seeded random soup: spheres and triangles of mixed sizes (incl. slivers and near-degenerate ones),
    diffuse / metallic / transmissive materials, lights inside the view volume."""
    from hslu_i.ba_raytracing.f2501_raytracer_amd.scene import FlatScene
    rng = np.random.default_rng(seed)
    f32 = np.float32
    sh = float(cfg.scene_height)
    sd = float(cfg.scene_depth)
    mats = []
    for _ in range(12):
        kind = rng.integers(0, 4)
        col = rng.uniform(0.05, 1.0, 3)
        if kind == 0:    # diffuse
            mats.append([*col, 0, rng.uniform(0, 0.6), 1, 0, 0, 0])
        elif kind == 1:  # metallic
            mats.append([*col, rng.uniform(0.1, 1.0), rng.uniform(0, 0.5), 0, 0, 0, 0])
        elif kind == 2:  # transmissive
            mats.append([*col, 0, rng.uniform(0, 0.5), rng.uniform(1.1, 1.9), rng.uniform(0.55, 1.0), rng.uniform(0, 0.3), 1])
        else:            # transmissive + metallic
            mats.append([*col, rng.uniform(0.05, 0.4), rng.uniform(0, 0.5), rng.uniform(1.1, 1.9), rng.uniform(0.55, 1.0), 0, 1])
    mats = np.asarray(mats, f32)
    box_lo, box_hi = np.array([0.05, 0.05, 0.1]), np.array([0.95, sh - 0.05, sd])
    sc = rng.uniform(box_lo, box_hi, (n_spheres, 3)).astype(f32)
    sr = rng.uniform(0.02, 0.12, n_spheres).astype(f32)
    v1 = rng.uniform(box_lo, box_hi, (n_tris, 3)).astype(f32)
    scale = np.where(rng.random(n_tris) < 0.2, 0.3, 0.05)[:, None]
    e1 = (rng.normal(0, 1, (n_tris, 3)) * scale).astype(f32)
    e2 = (rng.normal(0, 1, (n_tris, 3)) * scale).astype(f32)
    sliver = rng.random(n_tris) < 0.25
    e2[sliver] = (e1[sliver] * rng.uniform(0.3, 1.0, (sliver.sum(), 1)) + rng.normal(0, 2e-3, (sliver.sum(), 3))).astype(f32)
    tiny = rng.random(n_tris) < 0.05
    e1[tiny] *= f32(1e-3)
    nrm = np.cross(e1, e2)
    nrm = (nrm / np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)).astype(f32)
    odd = rng.random(n_tris) < 0.3  # non-unit stored normals, like the OBJ path
    nrm[odd] *= rng.uniform(0.2, 1.0, (odd.sum(), 1)).astype(f32)
    lights = np.zeros((n_lights, 7), f32)
    lights[:, :3] = rng.uniform([0.1, 0.05, 0.0], [0.9, sh * 0.6, sd * 0.5], (n_lights, 3))
    lights[:, 3:6] = rng.uniform(0.4, 1.0, (n_lights, 3))
    lights[:, 6] = rng.uniform(0.2, 0.8, n_lights)
    return FlatScene(sc, (sr * sr).astype(f32), (1 / sr).astype(f32), rng.integers(0, len(mats), n_spheres).astype(np.uint32),
                     v1, e1, e2, nrm, rng.integers(0, len(mats), n_tris).astype(np.uint32), mats, lights)


@pytest.mark.parametrize("seed", list(range(1, 11)))
def test_random_scenes_all_features(seed):
    """Fuzz: every conservative shortcut of the kernel (padded BVH boxes, triangle pre-filter, shared
    soft-shadow candidates with beam rejection, sphere culling, ray streaming) against the brute-force
    oracle on random geometry with all features on."""
    cfg = RenderConfig.from_features(["realistic", "anti_aliasing", "soft_shadows"], width_override=160,
                                     height_override=128, n_cloud_sets=16, depth_override=3, cloud_seed=seed)
    flat = random_scene(seed, n_spheres=6 + seed, n_tris=300 + 100 * seed, n_lights=3, cfg=cfg)
    compare(cfg, flat, ((13 * seed) % 100, (7 * seed) % 80, 56, 40))


def test_more_than_32_spheres():
    """The per-(wavefront, light) sphere mask and the lane-parallel sphere pre-selection cover spheres 0..31; the
    ones beyond are tested unconditionally.  40 spheres, soft shadows, secondary rays."""
    cfg = RenderConfig.from_features(["realistic", "anti_aliasing", "soft_shadows"], width_override=160,
                                     height_override=128, n_cloud_sets=8, depth_override=2, cloud_seed=5)
    flat = random_scene(23, n_spheres=40, n_tris=120, n_lights=2, cfg=cfg)
    compare(cfg, flat, (40, 30, 64, 48))


def test_random_scene_shadows_only_dense():
    """Same, many small triangles and big light clouds, no secondary rays (the headline kernel path)."""
    cfg = RenderConfig.from_features(["anti_aliasing", "high_quality"], width_override=192, height_override=160,
                                     n_cloud_sets=8, cloud_seed=7)
    flat = random_scene(11, n_spheres=4, n_tris=2500, n_lights=4, cfg=cfg)
    compare(cfg, flat, (60, 50, 48, 40))


def test_umbra_penumbra_and_horizon_classification():
    """Beam-level shortcuts of the soft-shadow path against the brute-force oracle where they all trigger: a floor
    under an opaque and a glass occluder (full umbra, penumbra bands, filtered light), a light below the floor's
    horizon, a light almost in the floor plane (grazing), and a sphere resting on the floor (rays leaving their
    own sphere)."""
    from hslu_i.ba_raytracing.f2501_raytracer_amd.scene import FlatScene
    f32 = np.float32
    cfg = RenderConfig.from_features(["anti_aliasing", "soft_shadows"], width_override=200, height_override=160,
                                     n_cloud_sets=8, cloud_seed=3)
    sh, sd = float(cfg.scene_height), float(cfg.scene_depth)
    mats = np.asarray([[0.8, 0.8, 0.7, 0, 0.2, 1, 0, 0, 0],       # floor
                       [0.9, 0.2, 0.2, 0, 0.4, 1, 0, 0, 0],       # opaque occluder
                       [0.3, 0.9, 0.5, 0, 0.3, 1.5, 0.7, 0.1, 1],  # glass occluder
                       [0.4, 0.5, 0.9, 0.3, 0.5, 1, 0, 0, 0]], f32)
    # the "floor" is a slanted plane rising with depth (so that it fills the lower half of the image)
    c0, a0, b0 = np.asarray([0.0, 0.85 * sh, 0.05 * sd]), np.asarray([1.0, 0.0, 0.0]), np.asarray([0.0, -0.45 * sh, 0.9 * sd])
    up = np.cross(a0, b0)
    up /= np.linalg.norm(up)  # towards the camera

    def on_plane(u, v, lift=0.0):
        return c0 + u * a0 + v * b0 + lift * up

    quads = [  # (corner, e1, e2, material)
        (c0, a0, b0, 0),
        (on_plane(0.25, 0.35, 0.12), 0.25 * a0, 0.25 * b0, 1),
        (on_plane(0.60, 0.40, 0.08), 0.20 * a0 + 0.01 * up, 0.20 * b0, 2),
    ]
    v1, e1, e2, nrm, mid = [], [], [], [], []
    for c, a, b, m in quads:
        c, a, b = np.asarray(c, f32), np.asarray(a, f32), np.asarray(b, f32)
        n = np.cross(a, b)
        n = (n / np.linalg.norm(n)).astype(f32)
        for (p0, u, v) in ((c, a, b), (c + a + b, -a, -b)):
            v1.append(p0), e1.append(u), e2.append(v), nrm.append(n), mid.append(m)
    sc = np.asarray([on_plane(0.5, 0.78, 0.05)], f32)
    sr = np.asarray([0.05], f32)
    lights = np.asarray([[*on_plane(0.45, 0.45, 0.5), 1, 1, 1, 0.7],       # above the occluders
                         [*on_plane(0.15, 0.30, 0.004), 1, 0.9, 0.8, 0.5],  # grazing: almost in the floor plane
                         [*on_plane(0.70, 0.50, -0.2), 0.8, 0.9, 1, 0.6]], f32)  # below the floor's horizon
    flat = FlatScene(sc, (sr * sr).astype(f32), (1 / sr).astype(f32), np.asarray([3], np.uint32),
                     np.asarray(v1, f32), np.asarray(e1, f32), np.asarray(e2, f32), np.asarray(nrm, f32),
                     np.asarray(mid, np.uint32), mats, lights)
    compare(cfg, flat, (30, 70, 140, 60))


def test_config3_full_size_properties(monkeypatch):
    """BASELINE.json configs[2] at its full size (1620x1350, 16 rays/px, 5 x 10 shadow rays per hit, text.obj):
    size-independent properties instead of the (hours-long) brute-force oracle --
    determinism, the multi-GPU tile partition, and the whole soft-shadow machinery (shared candidate lists, beam
    rejection, umbra / horizon / arrival shortcuts) against the plain per-sample BVH walk of the same kernel."""
    cfg = RenderConfig.from_features(["high_resolution", "anti_aliasing", "soft_shadows"])
    flat = scenes.semesterbild(cfg, "text").flatten()
    a0, p0, s0 = gpu_render(cfg, flat)
    assert s0["rays_primary"] == cfg.width * cfg.height * cfg.aa_total_rays
    n_hits = s0["rays_shadow"] // (5 * cfg.point_light_multiplicator)
    assert s0["rays_shadow"] == n_hits * 5 * cfg.point_light_multiplicator and 0 < n_hits <= s0["rays_primary"]
    a1, _, _ = gpu_render(cfg, flat)
    assert np.array_equal(a0, a1), "render is not deterministic"
    acc = np.zeros_like(a0)
    for rank in range(3):
        ar, _, _ = gpu_render(cfg, flat, n_ranks=3, rank=rank)
        assert not (acc[ar != 0] != 0).any(), "tiles of two ranks overlap"
        acc |= ar
    assert np.array_equal(acc, a0), "union of the ranks' tiles differs from the single-GPU frame"
    monkeypatch.setenv("RT_CAND_MAX", "0")  # every (wavefront, light) overflows: one BVH walk per sample, no shortcuts
    a2, p2, s2 = gpu_render(cfg, flat)
    monkeypatch.delenv("RT_CAND_MAX")
    assert np.array_equal(p2["hit_id"], p0["hit_id"]) and np.array_equal(p2["hit_t"], p0["hit_t"])
    assert s2["rays_shadow"] == s0["rays_shadow"]
    d = np.abs(p2["rgb"] - p0["rgb"]).max()
    assert d <= 2e-6, d
    ch = lambda v: np.stack([(v >> sh) & 0xFF for sh in (0, 8, 16, 24)]).astype(np.int32)
    assert np.abs(ch(a2) - ch(a0)).max() <= 1


def test_progressive_bands_equal_one_render():
    """`render_progressive` (the reference's tile-by-tile fill of the shared buffer) ends with the same buffer as one
    `render`, and reports every band once."""
    cfg = RenderConfig.from_features(["reflections", "anti_aliasing"])
    flat = scenes.test_scene(cfg).flatten()
    full, _, _ = gpu_render(cfg, flat)
    buf = ImageBuffer.new(cfg.width, cfg.height)
    seen = []
    r = RaytracerRenderer(cfg, device=0)
    n = r.render_progressive(buf, flat, on_tiles=lambda b, w: seen.append((w, int((b.buffer != 0).sum()))))
    assert n == len(seen) == -(-cfg.height // cfg.render_stride)
    assert all(seen[i][1] <= seen[i + 1][1] for i in range(len(seen) - 1))  # the picture only grows
    assert np.array_equal(buf.buffer, full)


def test_ray_streaming_with_tiny_chunks(monkeypatch):
    """Forces many primary batches and multi-chunk queue levels (RT_CHUNK_LOG2 = 10 -> 1024 rays per
    launch): the deepest-first drain and the queue-capacity invariant must give the same image."""
    cfg = RenderConfig.from_features(["realistic", "anti_aliasing"], depth_override=5)
    flat = scenes.test_scene(cfg).flatten()
    win = (300, 200, 96, 64)
    a_ref, p_ref, s_ref = gpu_render(cfg, flat, win)
    monkeypatch.setenv("RT_CHUNK_LOG2", "10")
    a, p, s_ = gpu_render(cfg, flat, win)
    monkeypatch.delenv("RT_CHUNK_LOG2")
    assert np.array_equal(a, a_ref) and np.array_equal(p["rgb"], p_ref["rgb"])
    for k in ("rays_primary", "rays_reflection", "rays_refraction", "rays_shadow"):
        assert s_[k] == s_ref[k]
    compare(cfg, flat, win)
