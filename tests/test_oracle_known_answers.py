"""Pins the oracle's functions to analytic known answers (tests/golden/known_answers.json, made by
tests/golden/make_known_answers.py in float64 from textbook formulas).  CPU only."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

from hslu_i.ba_raytracing.f2501_raytracer_amd import _abi
from hslu_i.ba_raytracing.f2501_raytracer_amd.scene import FlatScene

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "known_answers.json")) as fh:
    KA = json.load(fh)

FP = C.POINTER(C.c_float)


def f3(v):
    a = np.asarray(v, np.float32)
    return a, a.ctypes.data_as(FP)


def one_object_scene(sphere=None, tri=None):
    z3 = np.zeros((0, 3), np.float32)
    mats = np.asarray([[1, 1, 1, 0, 0, 1, 0, 0, 0]], np.float32)
    if sphere is not None:
        c, r = sphere
        r = np.float32(r)
        flat = FlatScene(np.asarray([c], np.float32), np.asarray([r * r], np.float32), np.asarray([1 / r], np.float32),
                         np.zeros(1, np.uint32), z3, z3, z3, z3, np.zeros(0, np.uint32), mats, np.zeros((0, 7), np.float32))
    else:
        v1, v2, v3 = (np.asarray(v, np.float32) for v in tri)
        e1, e2 = v2 - v1, v3 - v1
        n = np.cross(e1, e2)
        n = (n / np.linalg.norm(n)).astype(np.float32)
        flat = FlatScene(z3, np.zeros(0, np.float32), np.zeros(0, np.float32), np.zeros(0, np.uint32),
                         v1[None], e1[None], e2[None], n[None], np.zeros(1, np.uint32), mats, np.zeros((0, 7), np.float32))
    return _abi.make_scene_desc(flat)


@pytest.mark.parametrize("case", KA["sphere"])
def test_sphere(oracle, case):
    desc, keep = one_object_scene(sphere=(case["c"], case["r"]))
    o, op = f3(case["o"])
    d, dp = f3(case["d"])
    out = np.zeros(7, np.float32)
    hit = oracle.rt_oracle_sphere(C.byref(desc), 0, op, dp, 0, out.ctypes.data_as(FP))
    assert bool(hit) == case["hit"]
    if case["hit"]:
        assert out[0] == pytest.approx(case["t"], rel=2e-5, abs=2e-6)
        np.testing.assert_allclose(out[1:4], case["p"], atol=5e-6)
        np.testing.assert_allclose(out[4:7], case["n"], atol=2e-5)
        assert abs(float(np.linalg.norm(out[4:7])) - 1.0) < 1e-6


@pytest.mark.parametrize("case", KA["triangle"])
def test_triangle(oracle, case):
    desc, keep = one_object_scene(tri=(case["v1"], case["v2"], case["v3"]))
    o, op = f3(case["o"])
    d, dp = f3(case["d"])
    out = np.zeros(4, np.float32)
    hit = oracle.rt_oracle_triangle(C.byref(desc), 0, op, dp, 0, out.ctypes.data_as(FP))
    assert bool(hit) == case["hit"]
    if case["hit"]:
        assert out[0] == pytest.approx(case["t"], rel=5e-5, abs=5e-6)
        np.testing.assert_allclose(out[1:4], case["p"], atol=2e-5)


@pytest.mark.parametrize("case", KA["fresnel"])
def test_fresnel(oracle, case):
    m, mp = f3(case["mat"])
    n, np_ = f3(case["n"])
    v, vp = f3(case["v"])
    out = np.zeros(3, np.float32)
    oracle.rt_oracle_fresnel(mp, np_, vp, C.c_float(case["other"]), out.ctypes.data_as(FP))
    np.testing.assert_allclose(out, case["refl"], rtol=2e-5, atol=1e-6)


def test_fresnel_normal_incidence_is_f0(oracle):
    """F(0) = ((1 - n)/(1 + n))^2 = 0.04 for glass n = 1.5 in vacuum."""
    m, mp = f3([1, 1, 1, 0, 0, 1.5, 0.9, 0, 1])
    n, np_ = f3([0, 0, 1])
    out = np.zeros(3, np.float32)
    oracle.rt_oracle_fresnel(mp, np_, np_, C.c_float(1.0), out.ctypes.data_as(FP))
    np.testing.assert_allclose(out, [0.04] * 3, rtol=1e-6)


@pytest.mark.parametrize("case", KA["atten"])
def test_attenuation(oracle, case):
    t = math.inf if case["t"] == "inf" else case["t"]
    assert oracle.rt_oracle_atten(C.c_float(t)) == pytest.approx(case["a"], rel=1e-6, abs=1e-12)


@pytest.mark.parametrize("case", KA["pack"])
def test_pixel_pack(oracle, case):
    rgb = [math.nan if c == "nan" else c for c in case["rgb"]]
    assert oracle.rt_oracle_pack(*[C.c_float(c) for c in rgb]) == case["argb"]


@pytest.mark.parametrize("case", KA["refract"])
def test_refract(oracle, case):
    i, ip = f3(case["i"])
    n, np_ = f3(case["n"])
    out = np.zeros(3, np.float32)
    oracle.rt_oracle_refract(ip, np_, C.c_float(case["eta"]), out.ctypes.data_as(FP))
    np.testing.assert_allclose(out, case["out"], atol=2e-6)


def test_nearest_ties_go_to_later_object(oracle):
    """raytracer.rs:193-213: `<=` means the later object wins an exact tie -- two identical spheres."""
    import oracle_lib
    from hslu_i.ba_raytracing.f2501_raytracer_amd import RenderConfig
    cfg = RenderConfig.from_features([], width_override=16, height_override=16)
    z3 = np.zeros((0, 3), np.float32)
    c = [[0.5, 0.5, 0.5], [0.5, 0.5, 0.5]]
    flat = FlatScene(np.asarray(c, np.float32), np.full(2, 0.04, np.float32), np.full(2, 5.0, np.float32),
                     np.zeros(2, np.uint32), z3, z3, z3, z3, np.zeros(0, np.uint32),
                     np.asarray([[1, 0, 0, 0, 0, 1, 0, 0, 0]], np.float32), np.asarray([[0.5, 0.1, 0, 1, 1, 1, 1]], np.float32))
    _, planes, _ = oracle_lib.render(flat, cfg)
    ids = planes["hit_id"]
    assert (ids == 1).sum() > 0 and (ids == 0).sum() == 0
