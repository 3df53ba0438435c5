"""Pipeline-level known answers in float64.

`tests/golden/known_answers.json` pins single functions.  This file pins the PIPELINE: a float64 model of one pixel of
the reference's Whitted loop, written from the reference's formulas (`single_raytrace`,
`calculate_lighting` + `PointLight::calculate_contribution_at`, `has_any_intersection` with its opacity / filter
chain, `calculate_reflection`, `calculate_refractions`, `compute_fresnel`, distance attenuation) and NOT from
oracle/rt_oracle.c -- textbook ray/sphere and Moeller-Trumbore ray/triangle tests instead of the matrix-inverse form,
numpy float64 throughout.  The fp32 oracle must agree with it on pixels chosen to exercise each term: lit diffuse +
specular, a shadow filtered through glass (opacity chain + absorption filter), an opaque shadow, one bounce of
reflection on a metallic sphere, and refraction through a glass sphere with Fresnel weights.  CPU only.
"""
import numpy as np
import pytest

import oracle_lib
from hslu_i.ba_raytracing.f2501_raytracer_amd import RenderConfig
from hslu_i.ba_raytracing.f2501_raytracer_amd.scene import FlatScene

EPS = float(np.finfo(np.float32).eps)


def norm(v):
    return v / np.sqrt(v @ v)


class Model:
    """float64 restatement of the reference formulas for spheres + triangles + point lights (no AA, N_cloud = 1)."""

    def __init__(self, flat, cfg, reflections, refractions):
        self.f, self.cfg = flat, cfg
        self.refl, self.refr = reflections, refractions
        self.eps_d = float(cfg.eps_distance)
        self.air = float(cfg.air_ior)

    def mat(self, row):
        m = self.f.materials[row].astype(np.float64)
        tr = m[8] != 0 and not (abs(m[6]) <= EPS)
        return dict(color=m[0:3], metallic=m[3], shininess=m[4], ior=m[5], opacity=m[6], boost=m[7], tr=tr)

    def hits(self, o, d):
        """all valid (t, id, n, material row) along the ray"""
        out = []
        f = self.f
        for i in range(f.n_spheres):
            c, r2 = f.sphere_center[i].astype(np.float64), float(f.sphere_r_sq[i])
            v = o - c
            b, cc = 2.0 * (d @ v), v @ v - r2
            disc = b * b - 4.0 * cc
            if not disc >= 0:
                continue
            s = np.sqrt(disc)
            t0, t1 = (-b - s) / 2, (-b + s) / 2
            if t0 >= 0 and (not t1 >= 0 or t0 < t1):
                t = t0
            elif t1 >= 0:
                t = t1
            else:
                continue
            out.append((t, i, norm(o + d * t - c), int(f.sphere_material[i])))
        for i in range(f.n_triangles):
            v1, e1, e2 = (a[i].astype(np.float64) for a in (f.tri_v1, f.tri_e1, f.tri_e2))
            pv = np.cross(d, e2)
            det = e1 @ pv
            if abs(det) <= EPS:
                continue
            tv = o - v1
            u = (tv @ pv) / det
            qv = np.cross(tv, e1)
            v = (d @ qv) / det
            t = (e2 @ qv) / det
            if t <= EPS or u < 0 or v < 0 or u + v >= 1:
                continue
            out.append((t, f.n_spheres + i, f.tri_normal[i].astype(np.float64), int(f.tri_material[i])))
        return out

    def nearest(self, o, d):
        best = None
        for h in self.hits(o, d):
            if best is None or h[0] <= best[0]:
                best = h
        return best

    def fresnel(self, m, n, v, other):
        if not m["tr"]:
            return np.full(3, m["metallic"])
        nv = n @ v
        c = abs(nv)
        inside = nv < 0
        eta = m["ior"] / other if inside else other / m["ior"]
        sin2 = eta * eta * (1 - c * c)
        tir = (inside and sin2 > 1) or m["metallic"] > 0
        f0 = ((other - m["ior"]) / (other + m["ior"])) ** 2
        f0v = f0 * (1 - m["metallic"]) + m["color"] * m["metallic"]
        F = f0v + (1 - f0v) * (1 - c) ** 5
        if tir:
            return np.full(3, m["metallic"] if m["metallic"] > 0 else 1.0)
        return F

    def shadow(self, o, d, tmax):
        op, filt, occ = 1.0, np.ones(3), False
        for t, _, n, row in sorted(self.hits(o, d), key=lambda h: h[1]):
            if not t <= tmax:
                continue
            m = self.mat(row)
            io = 0.0
            if m["tr"]:
                io = m["opacity"] * (1 - self.fresnel(m, n, -d, 1.0)[0])
            op = min(max(op - (1 - io), 0.0), 1.0)
            if not m["tr"] and abs(op) <= EPS:
                occ = True
            filt = filt - m["color"] * (1 - min(max(m["opacity"] if m["tr"] else 1.0, 0.0), 1 - EPS))
            if occ:
                break
        return occ, op, filt

    @staticmethod
    def atten(t):
        return min(max(1.0 / (1 + abs(t) + 0.1 * t * t), 0.0), 1.0)

    def lighting(self, p, n, m, view):
        amb = m["color"] * 0.08
        direct, spec = np.zeros(3), np.zeros(3)
        for L in self.f.lights.astype(np.float64):
            lp, lc, lI = L[0:3], L[3:6], L[6]
            ld = norm(lp - p)
            so = p + ld * self.eps_d
            occ, op, filt = self.shadow(so, ld, np.sqrt((lp - so) @ (lp - so)))
            if occ:
                continue
            v = lp - p
            dist = np.sqrt(v @ v) + EPS
            c = (v @ n) / dist
            sig = min(max((np.tanh(0.95 * (EPS + dist + dist * dist)) + 1) / 2, 0.0), 1.0)
            ccol = m["color"] * lc if c > 0 else np.zeros(3)
            cint = c * lI * sig if c > 0 else 0.0
            Lc = ccol / filt
            diff = max(n @ ld, 0.0)
            s = 0.0
            if m["shininess"] > 0:
                rr = norm(ld - 2 * (ld @ n) * n)
                s = max(rr @ view, 0.0) ** max(m["shininess"] * 512, 1.0)
            if diff > 0:
                direct = direct + m["color"] * Lc * (diff * cint * op)
                if m["shininess"] > 0:
                    spec = spec + lc * (cint * op * s)
        return amb + direct, spec

    def trace(self, o, d_raw, n_start, depth):
        if depth == 0:
            return None
        d = norm(d_raw)
        h = self.nearest(o, d)
        if h is None:
            return None
        t, _, n, row = h
        m = self.mat(row)
        p = o + d * t
        direct, spec = self.lighting(p, n, m, d)
        a = self.atten(t)
        direct, spec = direct * a, spec * a
        refl, refr = np.zeros(3), np.zeros(3)
        T, R = m["tr"], m["metallic"] > 0 or m["tr"]
        if self.refl and R:
            c = d @ n
            ins = c < 0
            inorm = -n if ins else n
            n2 = m["ior"] if ins else self.air
            eta = n2 / n_start if ins else n_start / n2
            tir = eta * eta * (1 - c * c) >= 1
            if m["metallic"] > 0 or (T and tir):
                r = norm(d - 2 * (d @ n) * n)
                Rf = self.fresnel(m, inorm, -d, n_start)
                ch = self.trace(p + r * self.eps_d, r, n_start, (depth - 1) if depth is not None else self.cfg.max_depth_reflection)
                if ch is not None:
                    refl = ch[0] * self.atten(ch[1]) * Rf
        if self.refr and T:
            c = d @ n
            ins = c <= 0
            inorm = -n if ins else n
            n2 = m["ior"] if ins else self.air
            eta = n2 / n_start if ins else n_start / n2
            Tr = 1 - self.fresnel(m, inorm, d, 1 / eta)
            nn, e = -inorm, 1 / eta
            ndi = nn @ d
            k = 1 - e * e * (1 - ndi * ndi)
            if k >= 0:
                q = norm(d * e - nn * (e * ndi + np.sqrt(k)))
                ch = self.trace(p + q * self.eps_d, q, n2, max(depth - 1, 0) if depth is not None else self.cfg.max_depth_refraction)
                if ch is not None:
                    refr = ch[0] * (1 + m["boost"]) * Tr
        return (refl + refr + spec) if T else (direct + refl + spec), t

    def pixel(self, gx, gy):
        cfg = self.cfg
        c = np.array([gx * float(cfg.fw), gy * float(cfg.fh), 0.0])
        focus = np.array([float(cfg.focus.x), float(cfg.focus.y), float(cfg.focus.z)])
        r = self.trace(c, c - focus, self.air, None)
        return None if r is None else r[0]


def build_scene(cfg):
    f32 = np.float32
    sh, sd = float(cfg.scene_height), float(cfg.scene_depth)
    mats = np.asarray([
        [0.8, 0.7, 0.6, 0.0, 0.3, 1.0, 0.0, 0.0, 0],      # 0 wall: diffuse + specular
        [0.6, 0.9, 0.7, 0.0, 0.2, 1.5, 0.7, 0.1, 1],      # 1 glass: transmissive, opacity 0.7, boost 0.1
        [0.9, 0.3, 0.2, 0.0, 0.0, 1.0, 0.0, 0.0, 0],      # 2 opaque diffuse
        [0.9, 0.9, 0.95, 0.8, 0.5, 1.0, 0.0, 0.0, 0],     # 3 metallic mirror
    ], f32)
    # wall: two big triangles at z = 0.8 sd facing the camera (normal -z)
    z = 0.8 * sd
    quad = [(-1.0, -1.0, z), (3.0, -1.0, z), (-1.0, 3.0, z)], [(3.0, 3.0, z), (-1.0, 3.0, z), (3.0, -1.0, z)]
    v1 = np.asarray([q[0] for q in quad], f32)
    e1 = np.asarray([np.subtract(q[1], q[0]) for q in quad], f32)
    e2 = np.asarray([np.subtract(q[2], q[0]) for q in quad], f32)
    nrm = np.asarray([[0, 0, -1], [0, 0, -1]], f32)
    sc = np.asarray([[0.30, 0.45 * sh, 0.45 * sd], [0.62, 0.40 * sh, 0.40 * sd], [0.80, 0.70 * sh, 0.35 * sd]], f32)
    sr = np.asarray([0.11, 0.07, 0.08], f32)
    lights = np.asarray([[0.45, 0.15 * sh, 0.0, 1.0, 0.95, 0.9, 0.8], [0.1, 0.8 * sh, 0.1 * sd, 0.9, 1.0, 1.0, 0.5]], f32)
    return FlatScene(sc, (sr * sr).astype(f32), (1 / sr).astype(f32), np.asarray([1, 2, 3], np.uint32),
                     v1, e1, e2, nrm, np.zeros(2, np.uint32), mats, lights)


@pytest.mark.parametrize("features", [[], ["reflections"], ["refractions"], ["reflections", "refractions"]])
def test_oracle_matches_float64_pipeline_model(features):
    cfg = RenderConfig.from_features(features, width_override=96, height_override=80, depth_override=3)
    flat = build_scene(cfg)
    model = Model(flat, cfg, "reflections" in features, "refractions" in features)
    argb, planes, st = oracle_lib.render(flat, cfg, n_threads=4)
    rgb = planes["rgb"].reshape(cfg.height, cfg.width, 3)
    ids = planes["hit_id"].reshape(cfg.height, cfg.width)
    seen = set()
    worst = 0.0
    for gy in range(2, cfg.height, 7):
        for gx in range(1, cfg.width, 5):
            want = model.pixel(gx, gy)
            if want is None:
                assert ids[gy, gx] == -1
                continue
            seen.add(int(ids[gy, gx]))
            d = float(np.abs(rgb[gy, gx] - want).max())
            worst = max(worst, d)
            assert d <= 5e-5, (gx, gy, int(ids[gy, gx]), rgb[gy, gx], want)
    # the sample covers the glass sphere, the opaque sphere, the mirror and the wall
    assert {0, 1, 2}.issubset(seen) and (3 in seen or 4 in seen), seen
    print(f"{features}: max |dRGB| oracle(fp32) vs float64 model = {worst:.2e} over {len(seen)} objects")
