#!/usr/bin/env python3
"""Generates tests/golden/known_answers.json: analytic known answers for the single-function
probes of the oracle, computed INDEPENDENTLY of oracle/rt_oracle.c in float64 from textbook
formulas (analytic ray/sphere roots, Moeller-Trumbore barycentrics, Schlick, GLSL refract).

The reference has no golden vectors for this path, so these pin the oracle's
functions to the mathematics the reference's formulas implement.  Run: python make_known_answers.py
"""
import json
import math
import os

import numpy as np

rng = np.random.default_rng(20250704)
cases = {"sphere": [], "triangle": [], "fresnel": [], "atten": [], "pack": [], "refract": []}


def unit(v):
    v = np.asarray(v, np.float64)
    return v / np.linalg.norm(v)


# ---- spheres: ray o + t d against centre c radius r ------------------------------------------
def sphere_truth(o, d, c, r):
    v = o - c
    b = 2 * d.dot(v)
    cc = v.dot(v) - r * r
    disc = b * b - 4 * cc
    if disc < 0:
        return None
    s = math.sqrt(disc)
    t0, t1 = (-b - s) / 2, (-b + s) / 2
    if t0 >= 0:
        t = t0
    elif t1 >= 0:
        t = t1
    else:
        return None
    p = o + t * d
    return t, p, (p - c) / np.linalg.norm(p - c)


fixed = [
    # head-on hit at exactly t = 2 (centre 3 away, radius 1)
    ([0, 0, 0], [0, 0, 1], [0, 0, 3], 1.0),
    # origin inside the sphere: the far root is taken
    ([0, 0, 3], [0, 0, 1], [0, 0, 3], 1.0),
    # sphere behind the ray: miss
    ([0, 0, 0], [0, 0, 1], [0, 0, -3], 1.0),
    # clear miss
    ([0, 0, 0], [0, 0, 1], [5, 0, 3], 1.0),
]
for o, d, c, r in fixed:
    o, d, c = np.asarray(o, np.float64), unit(d), np.asarray(c, np.float64)
    tr = sphere_truth(o, d, c, r)
    cases["sphere"].append(dict(o=o.tolist(), d=d.tolist(), c=c.tolist(), r=r,
                                hit=tr is not None,
                                t=None if tr is None else tr[0],
                                p=None if tr is None else tr[1].tolist(),
                                n=None if tr is None else tr[2].tolist()))
for _ in range(40):
    c = rng.uniform(-1, 1, 3)
    r = float(rng.uniform(0.1, 0.6))
    o = rng.uniform(-2, 2, 3)
    d = unit(c + rng.normal(0, 0.5, 3) * r - o)
    tr = sphere_truth(o, d, c, r)
    # keep only numerically well-separated cases (no grazing)
    v = o - c
    disc = (2 * d.dot(v)) ** 2 - 4 * (v.dot(v) - r * r)
    if abs(disc) < 1e-3:
        continue
    cases["sphere"].append(dict(o=o.tolist(), d=d.tolist(), c=c.tolist(), r=r, hit=tr is not None,
                                t=None if tr is None else tr[0],
                                p=None if tr is None else tr[1].tolist(),
                                n=None if tr is None else tr[2].tolist()))


# ---- triangles: Moeller-Trumbore -------------------------------------------------------------
def tri_truth(o, d, v1, v2, v3):
    e1, e2 = v2 - v1, v3 - v1
    pv = np.cross(d, e2)
    det = e1.dot(pv)
    if abs(det) < 1e-12:
        return None, None
    tv = o - v1
    u = tv.dot(pv) / det
    qv = np.cross(tv, e1)
    v = d.dot(qv) / det
    t = e2.dot(qv) / det
    return (t, u, v), det


tri = (np.array([0.0, 0, 2]), np.array([1.0, 0, 2]), np.array([0.0, 1, 2]))
fixed_t = [
    ([0.25, 0.25, 0], [0, 0, 1]),     # interior hit, t = 2
    ([0.25, 0.25, 4], [0, 0, -1]),    # from behind (no culling): hit, t = 2
    ([2.0, 2.0, 0], [0, 0, 1]),       # outside: miss
    ([0.25, 0.25, 3], [0, 0, 1]),     # triangle behind the origin: miss (t < 0)
    ([0.25, 0.25, 0], [1, 0, 0]),     # parallel: miss (|det| <= eps)
    ([0.6, 0.6, 0], [0, 0, 1]),       # u + v = 1.2 > 1: miss
    ([-0.1, 0.3, 0], [0, 0, 1]),      # u < 0: miss
]
for o, d in fixed_t:
    o, d = np.asarray(o, np.float64), unit(d)
    r, det = tri_truth(o, d, *tri)
    hit = r is not None and r[0] > 1.2e-7 and r[1] >= 0 and r[2] >= 0 and r[1] + r[2] < 1
    cases["triangle"].append(dict(o=o.tolist(), d=d.tolist(), v1=tri[0].tolist(), v2=tri[1].tolist(), v3=tri[2].tolist(),
                                  hit=bool(hit), t=r[0] if hit else None,
                                  p=(o + r[0] * d).tolist() if hit else None))
for _ in range(60):
    v1, v2, v3 = rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 3)
    o = rng.uniform(-2, 2, 3) + np.array([0, 0, -3.0])
    bary = rng.uniform(-0.4, 1.2, 2)
    target = v1 + bary[0] * (v2 - v1) + bary[1] * (v3 - v1)
    d = unit(target - o)
    r, det = tri_truth(o, d, v1, v2, v3)
    if r is None:
        continue
    t, u, v = r
    # skip borderline classifications so fp32 rounding cannot flip them
    if min(abs(u), abs(v), abs(1 - u - v), abs(t)) < 1e-3 or abs(det) < 1e-3:
        continue
    hit = t > 0 and u >= 0 and v >= 0 and u + v < 1
    cases["triangle"].append(dict(o=o.tolist(), d=d.tolist(), v1=v1.tolist(), v2=v2.tolist(), v3=v3.tolist(),
                                  hit=bool(hit), t=t if hit else None, p=(o + t * d).tolist() if hit else None))


# ---- Fresnel (Schlick as written in material.rs:468-525) ---------------------------------------
def fresnel_truth(mat, n, v, other):
    r, g, b, metallic, shin, ior, op, boost, has = mat
    transmissive = has != 0 and abs(op) > 1.1920929e-7
    if not transmissive:
        return [metallic] * 3
    ndv = float(np.dot(n, v))
    c = abs(ndv)
    inside = ndv < 0
    eta = ior / other if inside else other / ior
    sin2 = eta * eta * (1 - c * c)
    tir = (inside and sin2 > 1) or metallic > 0
    f0 = ((other - ior) / (other + ior)) ** 2
    out = []
    for col in (r, g, b):
        f0v = f0 * (1 - metallic) + col * metallic
        F = f0v + (1 - f0v) * (1 - c) ** 5
        out.append((metallic if metallic > 0 else 1.0) if tir else F)
    return out


glass = [1.0, 0.8, 1.0, 0.0, 0.15, 1.5, 0.99, 0.025, 1.0]
metal_glass = [0.75, 0.5, 1.0, 0.2, 0.3, 1.5, 0.78, 0.0, 1.0]
opaque_metal = [1.0, 1.0, 1.0, 0.95, 0.23, 0.0, 0.0, 0.0, 0.0]
diffuse = [0.5, 0.75, 0.75, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0]
nz = [0.0, 0.0, 1.0]
for mat, n, v, other in [
    (glass, nz, nz, 1.0),                               # normal incidence: F = f0 = ((1-1.5)/(1+1.5))^2 = 0.04
    (glass, nz, unit([0.6, 0, 0.8]).tolist(), 1.000293),
    (glass, nz, unit([0.0, 0.98, -0.2]).tolist(), 1.000293),  # inside, grazing: TIR -> 1
    (glass, nz, unit([0.0, 0.1, -1.0]).tolist(), 1.0),         # inside, steep: no TIR
    (metal_glass, nz, unit([0.3, 0.2, 0.9]).tolist(), 1.0),    # metallic > 0 -> metallic
    (opaque_metal, nz, nz, 1.0),                               # non transmissive -> metallic
    (diffuse, nz, nz, 1.0),                                    # non transmissive, metallic 0 -> 0
]:
    cases["fresnel"].append(dict(mat=mat, n=n, v=v, other=other, refl=fresnel_truth(mat, np.asarray(n), np.asarray(v), other)))

# ---- attenuation 1/(1+|t|+0.1 t^2) -------------------------------------------------------------
for t in [0.0, 1.0, -1.0, 0.5, 2.0, 10.0, 1e9]:
    cases["atten"].append(dict(t=t, a=min(max(1.0 / (1.0 + abs(t) + 0.1 * t * t), 0.0), 1.0)))
cases["atten"].append(dict(t="inf", a=0.0))

# ---- pixel pack: clamp, *255, round half to even, 0xFFRRGGBB ------------------------------------
for rgb, want in [
    ((1.0, 0.5, 0.0), 0xFFFF8000),       # 127.5 -> 128 (half to even)
    ((0.0, 0.0, 0.0), 0xFF000000),
    ((2.0, -1.0, 1.0), 0xFFFF00FF),      # clamped
    ((0.5 / 255, 1.5 / 255, 2.5 / 255), 0xFF000202),  # .5 ties: 0.5->0, 1.5->2, 2.5->2
    ((0.2, 0.4, 0.6), 0xFF336699),
]:
    cases["pack"].append(dict(rgb=list(rgb), argb=want))
cases["pack"].append(dict(rgb=["nan", 0.25, 0.75], argb=0xFF0040BF))

# ---- refract (GLSL) ------------------------------------------------------------------------------
def refract_truth(i, n, eta):
    ndi = float(np.dot(n, i))
    k = 1 - eta * eta * (1 - ndi * ndi)
    if k < 0:
        return [0.0, 0.0, 0.0]
    return (eta * i - (eta * ndi + math.sqrt(k)) * n).tolist()


for i, n, eta in [
    (unit([0, 0, 1]), np.array([0, 0, -1.0]), 1 / 1.5),
    (unit([0.5, 0, 1]), np.array([0, 0, -1.0]), 1 / 1.5),
    (unit([1, 0, 0.2]), np.array([0, 0, -1.0]), 1.5),   # TIR -> zero vector
    (unit([0.3, -0.4, 0.7]), unit([0.1, 0.1, -1.0]), 1.13 / 1.000293),
]:
    cases["refract"].append(dict(i=i.tolist(), n=n.tolist(), eta=eta, out=refract_truth(i, n, eta)))

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "known_answers.json")
with open(out, "w") as fh:
    json.dump(cases, fh, indent=1)
print({k: len(v) for k, v in cases.items()})
