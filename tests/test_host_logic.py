"""Host-side logic: constants table, scene builders, sample tables, tiling.  CPU only.

Mirrors the reference's own (few) unit tests where they touch the path: test_chunked_access
(src/image_buffer.rs:327-347, exactly-once tile coverage), test_gcd / test_multiple_gcd
(src/helpers.rs:200-225, RENDER_STRIDE arithmetic), test_geometry_collection
(src/geometry/render_geometry.rs:201-231)."""
import os

import numpy as np
import pytest

from hslu_i.ba_raytracing.f2501_raytracer_amd import (BoundedPlane, ColorType, Material, PointLight, RenderConfig, Rotor3,
                                                      Scene, SphereData, TransmissionProperties, TriangleData, Vec3,
                                                      maximize_value, sampling, scenes)
from hslu_i.ba_raytracing.f2501_raytracer_amd.config import DEFAULT_FEATURES, expand_features
from hslu_i.ba_raytracing.f2501_raytracer_amd.distributed import owned_pixel_indices, tile_owner_map
from hslu_i.ba_raytracing.f2501_raytracer_amd.f32math import F, gcd, lcm

F32 = np.float32


def test_constants_table_5_6_aspect():
    """SURVEY.md 8(a) row K, computed there with numpy float32 from src/lib.rs:30-92."""
    for feats, w, fw, fh in (([], 768, 0.0013020834, 0.0013020833),
                             (["medium_resolution"], 1140, 0.000877193, 0.00087719294),
                             (["high_resolution"], 1620, 0.000617284, 0.0006172839)):
        c = RenderConfig.from_features(feats)
        assert c.width == w
        assert c.scene_height == F32(0.8333333) and c.scene_depth == F32(0.9166666)
        assert c.average_scene_dimension == F32(0.9166667)
        assert c.fw == F32(fw) and c.fh == F32(fh) and c.fd == F32(fh)
        f = c.focus
        assert (f.x, f.y, f.z) == (F32(0.5), F32(0.41666666), F32(-1.7416666))
        assert abs(float(c.eps_distance) - 1.0928e-5) < 1e-9
        assert c.render_stride == 48


def test_constants_table_16_9_aspect():
    c = RenderConfig.from_features(["realistic", "extreme_quality"], width_override=3840, height_override=2160)
    assert (c.scene_height, c.scene_depth, c.average_scene_dimension) == (F32(0.5625), F32(0.78125), F32(0.78125))
    f = c.focus
    assert (f.x, f.y, f.z) == (F32(0.5), F32(0.28125), F32(-1.484375))
    assert c.fw == c.fh == c.fd == F32(0.00026041668)
    assert abs(float(c.eps_distance) - 9.313e-6) < 1e-9
    assert c.render_stride == 48


def test_feature_table():
    """raytracer_renderer.rs:55-93 and Cargo.toml:62-83."""
    base = RenderConfig.from_features([])
    assert (base.max_depth_reflection, base.max_depth_refraction) == (9, 8)
    assert (base.point_light_multiplicator, base.samples_per_pixel, base.aa_total_rays) == (1, 9, 16)
    assert RenderConfig.from_features(["soft_shadows"]).point_light_multiplicator == 10
    hq = RenderConfig.from_features(["high_quality"])
    assert (hq.max_depth_reflection, hq.max_depth_refraction, hq.point_light_multiplicator) == (13, 18, 19)
    assert hq.has("anti_aliasing") and hq.has("soft_shadows") and hq.has("high_quality_model")
    xq = RenderConfig.from_features(["extreme_quality"])
    assert (xq.max_depth_reflection, xq.max_depth_refraction, xq.point_light_multiplicator) == (21, 21, 28)
    assert (xq.samples_per_pixel, xq.aa_total_rays) == (24, 24)
    d = expand_features(DEFAULT_FEATURES)
    assert {"reflections", "refractions", "anti_aliasing", "soft_shadows", "scene_backface_culling"} <= d
    with pytest.raises(ValueError):
        expand_features(["no_such_feature"])
    assert RenderConfig.from_features(["medium_resolution"]).model_path().endswith("text.obj")
    assert RenderConfig.from_features(["high_resolution"]).model_path().endswith("text_lowres.obj")


def test_gcd_lcm():  # reference src/helpers.rs:200-225
    assert gcd(48, 18) == 6 and gcd(17, 5) == 1 and gcd(0, 7) == 7
    assert lcm(4, 6) == 12 and lcm(48, lcm(8, gcd(1620, 16))) == 48


def test_test_scene_counts_and_stale_light():
    cfg = RenderConfig.from_features([])
    s = scenes.test_scene(cfg)
    assert (len(s.spheres), len(s.triangles), len(s.scene_lights)) == (4, 3 + 7 * 12, 6)
    assert float(s.scene_lights[3].position.x) == pytest.approx(1.0 - 80.0)  # examples/test_scene.rs:324
    f = s.flatten()
    assert f.n_objects == 91 and f.lights.shape == (6, 7)
    assert f.without_triangles().n_objects == 4


def test_semesterbild_counts_and_mesh_normals():
    cfg = RenderConfig.from_features(["high_resolution"])
    s = scenes.semesterbild(cfg, "text")
    assert (len(s.spheres), len(s.triangles), len(s.scene_lights)) == (9, 14521 + 48, 5)
    f = s.flatten()
    n = np.linalg.norm(f.tri_normal[:14521], axis=1)
    # 12 978 of 14 521 mesh normals are non-unit (0.25/0.25/0.5 lerp), min ~0.16
    assert int((np.abs(n - 1) > 1e-3).sum()) == 12978
    assert 0.15 < float(n.min()) < 0.17
    low = scenes.semesterbild(cfg)  # high_resolution without high_quality_model -> text_lowres
    assert len(low.triangles) == 1639 + 48
    culled = scenes.semesterbild(RenderConfig.from_features(["high_resolution", "scene_backface_culling"]), "text")
    assert 0 < len(s.triangles) - len(culled.triangles) < 50


def test_packed_mesh_equals_reference_obj_if_present():
    ref = "/root/reference/data/obj/text/text_lowres.obj"
    if not os.path.exists(ref):
        pytest.skip("reference tree not present (GPU box)")
    from hslu_i.ba_raytracing.f2501_raytracer_amd.obj import load_obj_scene
    a = load_obj_scene(ref, None).flatten()
    b = load_obj_scene(scenes.mesh_path(RenderConfig.from_features([]), "text_lowres"), None).flatten()
    for k in ("tri_v1", "tri_e1", "tri_e2", "tri_normal", "materials"):
        assert np.array_equal(getattr(a, k), getattr(b, k)), k


def test_bounded_plane_is_a_closed_slab():
    bp = BoundedPlane.with_material(-Vec3.unit_z(), Vec3.new(0.5, 0.5, 1.0), Vec3.unit_y(), 1.0, 0.5, 0.1,
                                    Material.diffuse(ColorType.new(1, 1, 1)))
    tris = bp.to_basic_geometries()
    assert len(tris) == 12
    pts = np.array([v.to_list() for t in tris for v in (t.vertex1, t.vertex2, t.vertex3)])
    assert np.allclose(pts.min(0), [0.0, 0.25, 0.95], atol=1e-6) and np.allclose(pts.max(0), [1.0, 0.75, 1.05], atol=1e-6)
    normals = {tuple(np.round(t.normal.to_list(), 6)) for t in tris}
    assert len(normals) == 6
    with pytest.raises(AssertionError):
        BoundedPlane.with_material(Vec3.unit_z(), Vec3.new(0, 0, 0), Vec3.unit_z(), 1, 1, 0.1, Material.diffuse(ColorType.new(1, 1, 1)))


def test_geometry_collection_buckets():  # reference render_geometry.rs:201-231
    s = Scene.new()
    s.add_geometry(SphereData.new(Vec3.new(0, 0, 0), 1.0, ColorType.new(1, 0, 0)))
    s.add_geometry(TriangleData.new(Vec3.new(0, 0, 0), Vec3.new(1, 0, 0), Vec3.new(0, 1, 0), ColorType.new(0, 1, 0)))
    assert (len(s.spheres), len(s.triangles), s.num_objects()) == (1, 1, 2)
    t = s.triangles[0]
    assert t.normal.to_list() == [0.0, 0.0, 1.0]
    with pytest.raises(TypeError):
        s.add_geometry("sphere")


def test_transmission_mask_and_materials():
    assert not TransmissionProperties.none().mask() and not TransmissionProperties.default().mask()
    assert TransmissionProperties.new(0.5, 1.5).mask()
    assert not TransmissionProperties.new(0.0, 1.5).mask()  # |opacity| <= eps -> not transmissive
    m = Material.new(ColorType.new(0.1, 0.2, 0.3), 0.4, 0.5, TransmissionProperties.new_with_boost(0.6, 1.7, 0.8))
    assert m.row() == pytest.approx((0.1, 0.2, 0.3, 0.4, 0.5, 1.7, 0.6, 0.8, 1.0))


def test_maximize_value():
    w = maximize_value(ColorType.new(0.25, 0.25, 0.25))
    assert w.to_tuple() == pytest.approx((1, 1, 1), abs=1e-6)
    c = maximize_value(ColorType.new(0.825, 0.675, 0.5))
    assert float(c.red) == pytest.approx(1.0, abs=1e-6) and 0 < float(c.blue) < float(c.green) < 1
    l = PointLight.new(Vec3.new(0, 0, 0), ColorType.new(1.0, 1.0, 1.0), 0.3)
    assert l.color.to_tuple() == pytest.approx((1, 1, 1)) and float(l.intensity) == pytest.approx(0.3)


def test_rotor_conventions():
    """quarter turns: from_rotation_xy rotates x towards y (ultraviolet's plane convention)."""
    h = float(np.pi / 2)
    v = Vec3.unit_x().rotated_by(Rotor3.from_rotation_xy(h))
    assert v.to_list() == pytest.approx([0, 1, 0], abs=1e-6)
    v = Vec3.unit_y().rotated_by(Rotor3.from_rotation_yz(h))
    assert v.to_list() == pytest.approx([0, 0, 1], abs=1e-6)
    v = Vec3.unit_x().rotated_by(Rotor3.from_rotation_xz(h))
    assert v.to_list() == pytest.approx([0, 0, 1], abs=1e-6)
    r = Rotor3.from_euler_angles(-0.04, 0.125, 0.51)
    assert float(Vec3.new(0.3, -0.2, 0.9).rotated_by(r).mag()) == pytest.approx(float(Vec3.new(0.3, -0.2, 0.9).mag()), rel=1e-6)


def test_aa_table_deterministic_mode():
    """raytracer_renderer.rs:105-127 without anti_aliasing_randomness: [0,0] then [1,1]s; offsets follow
    the t,l,b,r,tl,tr,bl,br cycle restarted per 8-lane chunk, scale 0.85."""
    cfg = RenderConfig.from_features(["high_resolution", "anti_aliasing"])
    tab = sampling.aa_sample_table(cfg)
    assert tab.shape == (16, 2) and tab[0].tolist() == [0, 0] and (tab[1:] == 1).all()
    off = sampling.aa_offsets(cfg)
    s = F32(0.85)
    assert off[0].tolist() == [0, 0]
    assert off[1, 0] == -(F32(1) * cfg.fw * s) and off[1, 1] == 0           # l
    assert off[2, 1] == F32(1) * cfg.fh * s and off[2, 0] == 0              # b
    assert off[8, 0] == 0 and off[8, 1] == -(F32(1) * cfg.fh * s)           # chunk 2 restarts with t
    assert RenderConfig.from_features(["extreme_quality"]).aa_total_rays == 24


def test_aa_table_random_mode_is_seeded():
    cfg = RenderConfig.from_features(["anti_aliasing_randomness", "anti_aliasing_rotation_scale"])
    a, b = sampling.aa_offsets(cfg), sampling.aa_offsets(cfg)
    assert np.array_equal(a, b) and a.shape == (16, 2)
    tab = sampling.aa_sample_table(cfg)
    assert (tab[9:] >= 0).all() and (tab[9:] <= 1.2).all()
    d = sampling.aa_directions(cfg)
    assert all(abs(float(v.mag()) - 1) < 1e-6 for v in d)


def test_cloud_sets():
    cfg = RenderConfig.from_features(["soft_shadows"], n_cloud_sets=8)
    cs = sampling.cloud_sets(cfg)
    assert cs.shape == (8, 10, 3)
    R = 1.725 + 10 / 20
    assert (cs >= 0).all() and (cs <= R).all()
    # Bridson: points of a set are at least r = 4/N apart
    for s in cs:
        d = np.linalg.norm(s[:, None] - s[None], axis=-1) + np.eye(10) * 9
        assert d.min() >= 4 / 10 - 1e-5
    assert np.array_equal(cs, sampling.cloud_sets(cfg))
    assert sampling.cloud_sets(RenderConfig.from_features([])).shape == (1, 1, 3)


@pytest.mark.parametrize("n_ranks", [1, 2, 3, 4, 8])
def test_tiles_cover_every_pixel_exactly_once(n_ranks):
    """reference test_chunked_access (image_buffer.rs:327-347) for the multi-GPU partition."""
    cfg = RenderConfig.from_features(["high_resolution"])
    seen = np.zeros(cfg.width * cfg.height, np.int32)
    counts = []
    for r in range(n_ranks):
        idx = owned_pixel_indices(cfg, n_ranks, r)
        seen[idx] += 1
        counts.append(idx.shape[0])
    assert (seen == 1).all()
    assert max(counts) <= 1.02 * (sum(counts) / n_ranks)
    own = tile_owner_map(cfg, n_ranks)
    assert own.shape == (29, 34)  # 34 x 29 tiles at 1620x1350
    if n_ranks > 1:  # neighbouring tiles never share a rank
        assert (own[:, 1:] != own[:, :-1]).all() and (own[1:, :] != own[:-1, :]).all()


def test_output_encoder_matches_oracle_pack(oracle, tmp_path):
    """WindowColorEncoder::to_output (window.rs:105-109) host mirror == the oracle's pack, and FileOutput
    writes the buffer's RGB8 rows (file.rs:27-49)."""
    import ctypes as C
    from PIL import Image
    from hslu_i.ba_raytracing.f2501_raytracer_amd.output import FileOutput, WindowColorEncoder
    from hslu_i.ba_raytracing.f2501_raytracer_amd.renderer import ImageBuffer
    rng = np.random.default_rng(5)
    cases = [(1.0, 0.5, 0.0), (0.5 / 255, 1.5 / 255, 2.5 / 255), (2.0, -1.0, 1.0)] + [tuple(rng.uniform(-0.2, 1.2, 3)) for _ in range(200)]
    for rgb in cases:
        want = oracle.rt_oracle_pack(*[C.c_float(c) for c in rgb])
        assert WindowColorEncoder.to_output(rgb) == want
    assert WindowColorEncoder.to_output((1.0, 0.5, 0.0)) == 0xFFFF8000
    assert WindowColorEncoder.from_output(0xFF336699) == pytest.approx((0.2, 0.4, 0.6))
    buf = ImageBuffer.new(4, 2)
    buf.buffer[:] = [0xFFFF0000, 0xFF00FF00, 0xFF0000FF, 0, 0xFF102030, 0xFFFFFFFF, 0, 0xFF808080]
    path = str(tmp_path / "o.png")
    FileOutput.new(path).render_buffer(buf)
    img = np.asarray(Image.open(path))
    assert img.shape == (2, 4, 3) and img[0, 0].tolist() == [255, 0, 0] and img[1, 0].tolist() == [0x10, 0x20, 0x30]


def test_exact_reciprocal_sqrt_near_one():
    """csrc/rt_kernels.hip normalize_unit: for s within 1024 ulp of 1.0f the correctly rounded 1 / sqrt(s) (IEEE sqrt, then
    IEEE division -- what normalize() computes and the oracle's vnormalize does) is a closed form of the integer
    distance of s from 1.0f.  Exhaustive over the range the kernel uses (and exact well beyond it)."""
    one = int(np.float32(1.0).view(np.uint32))
    for k in range(-2800, 2801):
        s = np.uint32(one + k).view(np.float32)
        ref = int((np.float32(1.0) / np.sqrt(s)).view(np.uint32))
        if k >= 0:
            rb = 0x3F800000 - (k >> 1) * 2
        else:
            rb = 0x3F800000 + (((((-k) + 1) >> 1) + 1) >> 1)
        assert ref == rb, (k, hex(ref), hex(rb))
