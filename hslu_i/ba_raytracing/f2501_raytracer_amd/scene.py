"""Host-side mirror of the reference's scene API (same names, argument meaning, error behaviour).

Reference: `src/scene/scene.rs` (Scene), `src/geometry/basic/{sphere,triangle}.rs`,
`src/geometry/composite/bounded_plane.rs`, `src/raytracing/material.rs`,
`src/scene/lighting/light.rs` (PointLight), `src/color.rs:124-131` (maximize_value).

A `Scene` here is plain host data; `Scene.flatten()` produces the SoA arrays of
`rt_scene_desc` (include/rt_hip.h) with the canonical object order: all spheres in insertion
order, then all triangles in insertion order (hit id = index in that order).
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .f32math import EPSILON, F, Rotor3, Similarity3, Vec3


class ColorType:
    """palette::LinSrgb<f32> (reference src/helpers.rs:12)."""

    __slots__ = ("red", "green", "blue")

    def __init__(self, red, green, blue):
        self.red, self.green, self.blue = F(red), F(green), F(blue)

    @staticmethod
    def new(red, green, blue) -> "ColorType":
        return ColorType(red, green, blue)

    def to_tuple(self):
        return (self.red, self.green, self.blue)


def _srgb_encode(x: np.float32) -> np.float32:
    x = F(x)
    if x <= F(0.0031308):
        return F(12.92) * x
    return F(1.055) * F(np.power(x, F(1.0 / 2.4), dtype=np.float32)) - F(0.055)


def _srgb_decode(x: np.float32) -> np.float32:
    x = F(x)
    if x <= F(0.04045):
        return x / F(12.92)
    return F(np.power((x + F(0.055)) / F(1.055), F(2.4), dtype=np.float32))


def maximize_value(color: ColorType) -> ColorType:
    """reference src/color.rs:124-131: linear -> sRGB -> HSV, V = 1 -> sRGB -> linear."""
    r, g, b = (_srgb_encode(c) for c in color.to_tuple())
    mx, mn = max(r, g, b), min(r, g, b)
    if mx <= F(0):
        # hue undefined, saturation 0, value forced to 1 -> white
        return ColorType(1, 1, 1)
    delta = mx - mn
    sat = delta / mx
    if delta == F(0):
        hue = F(0)
    elif mx == r:
        hue = F(60) * (((g - b) / delta) % F(6))
    elif mx == g:
        hue = F(60) * ((b - r) / delta + F(2))
    else:
        hue = F(60) * ((r - g) / delta + F(4))
    # HSV(h, s, 1) -> RGB
    c = sat  # v * s with v = 1
    hp = F(hue / F(60))
    x = c * (F(1) - F(abs(float(hp % F(2)) - 1.0)))
    m = F(1) - c
    sector = int(np.floor(float(hp))) % 6
    rgb = [(c, x, 0), (x, c, 0), (0, c, x), (0, x, c), (x, 0, c), (c, 0, x)][sector]
    enc = [F(v) + m for v in rgb]
    return ColorType(*(_srgb_decode(v) for v in enc))


@dataclass
class TransmissionProperties:
    """reference src/raytracing/material.rs:15-74"""

    refraction_index: np.float32
    opacity: np.float32
    has_opacity: bool
    boost: np.float32

    @staticmethod
    def new(opacity, refraction_index) -> "TransmissionProperties":
        return TransmissionProperties(F(refraction_index), F(opacity), True, F(0))

    @staticmethod
    def new_with_boost(opacity, refraction_index, boost) -> "TransmissionProperties":
        return TransmissionProperties(F(refraction_index), F(opacity), True, F(boost))

    @staticmethod
    def none() -> "TransmissionProperties":
        return TransmissionProperties(F(0), F(0), False, F(0))

    @staticmethod
    def default() -> "TransmissionProperties":
        return TransmissionProperties(F(1), F(0), False, F(0))

    def mask(self) -> bool:
        return bool(self.has_opacity and not (abs(self.opacity - F(0)) <= EPSILON))


@dataclass
class Material:
    """reference src/raytracing/material.rs:78-174"""

    color: ColorType
    metallic: np.float32
    shininess: np.float32
    transmission: TransmissionProperties

    @staticmethod
    def new(color: ColorType, metallic, shininess, transmission: TransmissionProperties) -> "Material":
        return Material(color, F(metallic), F(shininess), transmission)

    @staticmethod
    def diffuse(color: ColorType) -> "Material":
        return Material(color, F(0), F(0), TransmissionProperties.default())

    @staticmethod
    def translucent(color: ColorType, opacity, refraction_index) -> "Material":
        return Material(color, F(0), F(0), TransmissionProperties.new(opacity, refraction_index))

    def row(self) -> Tuple[float, ...]:
        t = self.transmission
        return (
            float(self.color.red), float(self.color.green), float(self.color.blue),
            float(self.metallic), float(self.shininess), float(t.refraction_index),
            float(t.opacity), float(t.boost), 1.0 if t.has_opacity else 0.0,
        )


@dataclass
class SphereData:
    """reference src/geometry/basic/sphere.rs:20-49"""

    center: Vec3
    r_sq: np.float32
    r_inv: np.float32
    material: Material

    @staticmethod
    def new(center: Vec3, radius, color: ColorType) -> "SphereData":
        return SphereData.with_material(center, radius, Material.diffuse(color))

    @staticmethod
    def with_material(center: Vec3, radius, material: Material) -> "SphereData":
        radius = F(radius)
        return SphereData(center, radius * radius, F(1.0) / radius, material)


@dataclass
class TriangleData:
    """reference src/geometry/basic/triangle.rs:22-102"""

    vertex1: Vec3
    vertex2: Vec3
    vertex3: Vec3
    edge1: Vec3
    edge2: Vec3
    normal: Vec3
    material: Material

    @staticmethod
    def new(v1: Vec3, v2: Vec3, v3: Vec3, color: ColorType) -> "TriangleData":
        return TriangleData.with_material(v1, v2, v3, Material.diffuse(color))

    @staticmethod
    def with_material(v1: Vec3, v2: Vec3, v3: Vec3, material: Material) -> "TriangleData":
        e1, e2 = v2 - v1, v3 - v1
        return TriangleData(v1, v2, v3, e1, e2, e1.cross(e2).normalized(), material)

    @staticmethod
    def with_material_and_normal(v1: Vec3, v2: Vec3, v3: Vec3, normal: Vec3, material: Material) -> "TriangleData":
        return TriangleData(v1, v2, v3, v2 - v1, v3 - v1, normal, material)


@dataclass
class PointLight:
    """reference src/scene/lighting/light.rs:162-181; colour is value-maximised at construction."""

    position: Vec3
    color: ColorType
    intensity: np.float32

    @staticmethod
    def new(position: Vec3, color: ColorType, intensity) -> "PointLight":
        return PointLight(position, maximize_value(color), F(intensity))

    def into(self) -> "PointLight":  # `.into()` -> SceneLightSource::PointLight
        return self


class BoundedPlane:
    """reference src/geometry/composite/bounded_plane.rs: a slab -> 12 triangles."""

    def __init__(self, center, up, left, normal, width, height, depth, material):
        self.center, self.up, self.left, self.normal = center, up, left, normal
        self.width, self.height, self.depth = F(width), F(height), F(depth)
        self.material = material

    @staticmethod
    def new(normal, center, up, width, height, depth, color: ColorType) -> "BoundedPlane":
        return BoundedPlane.with_material(normal, center, up, width, height, depth, Material.diffuse(color))

    @staticmethod
    def with_material(normal: Vec3, center: Vec3, up: Vec3, width, height, depth, material: Material) -> "BoundedPlane":
        # same panics as bounded_plane.rs:58-73
        assert F(width) > 0, "width must be positive"
        assert F(height) > 0, "height must be positive"
        assert abs(normal.dot(up) - F(0)) <= EPSILON, "up must be orthogonal to normal"
        return BoundedPlane(center, up, normal.cross(up).normalized(), normal, width, height, depth, material)

    def triangulate(self):
        x = Vec3.broadcast(self.width / F(2.0)) * (-self.left)
        y = Vec3.broadcast(self.height / F(2.0)) * self.up
        c = self.center
        p0, p1, p2, p3 = (-x) + y, x + y, (-x) - y, x - y
        return ((c + p1, c + p0, c + p3), (c + p2, c + p3, c + p0))

    def to_basic_geometries(self) -> List[TriangleData]:
        t1, t2 = self.triangulate()
        tris: List[TriangleData] = []
        half = F(0.5)
        for depth_offset, normal in ((-(self.depth * half), -self.normal), (self.depth * half, self.normal)):
            off = self.normal * Vec3.broadcast(depth_offset)
            tris.append(TriangleData.with_material_and_normal(t1[0] + off, t1[1] + off, t1[2] + off, normal, self.material))
            tris.append(TriangleData.with_material_and_normal(t2[0] + off, t2[1] + off, t2[2] + off, normal, self.material))
        for d, dir_offset, width, normal in (
            (self.up, self.height, self.width, self.up),
            (self.left, self.width, self.height, self.left),
            (-self.up, self.height, self.width, -self.up),
            (-self.left, self.width, self.height, -self.left),
        ):
            plate_center = d.mul_add(Vec3.broadcast(dir_offset * half), self.center)
            a, b = BoundedPlane.with_material(normal, plate_center, self.normal, width, self.depth, F(0), self.material).triangulate()
            tris.append(TriangleData.with_material_and_normal(a[0], a[1], a[2], normal, self.material))
            tris.append(TriangleData.with_material_and_normal(b[0], b[1], b[2], normal, self.material))
        return tris


class Scene:
    """reference src/scene/scene.rs:24-178"""

    def __init__(self):
        self.spheres: List[SphereData] = []
        self.triangles: List[TriangleData] = []
        self.scene_lights: List[PointLight] = []

    @staticmethod
    def new() -> "Scene":
        return Scene()

    @staticmethod
    def with_capacities(scene_objects: int, scene_lights: int = 0) -> "Scene":
        return Scene()

    def add_sphere(self, sphere: SphereData) -> None:
        self.spheres.append(sphere)

    def add_triangle(self, triangle: TriangleData) -> None:
        self.triangles.append(triangle)

    def add_geometry(self, geometry) -> None:
        if isinstance(geometry, SphereData):
            self.add_sphere(geometry)
        elif isinstance(geometry, TriangleData):
            self.add_triangle(geometry)
        else:
            raise TypeError(f"not a RenderGeometry: {type(geometry)!r}")

    def add_light(self, light: PointLight) -> None:
        self.scene_lights.append(light)

    def merge(self, other: "Scene") -> None:
        self.spheres.extend(other.spheres)
        self.triangles.extend(other.triangles)
        self.scene_lights.extend(other.scene_lights)

    def num_objects(self) -> int:
        return len(self.spheres) + len(self.triangles)

    # ---- scene.rs:43-134 ----------------------------------------------------------------
    @staticmethod
    def from_obj(path: str, transform: Optional[Similarity3] = None, continue_on_material_failure: bool = True) -> "Scene":
        from .obj import load_obj_scene

        return load_obj_scene(path, transform, continue_on_material_failure)

    # ---- scene.rs:136-155 ---------------------------------------------------------------
    @staticmethod
    def backface_culling(scene: "Scene", view_direction: Vec3) -> "Scene":
        out = Scene()
        out.scene_lights = list(scene.scene_lights)
        out.spheres = list(scene.spheres)
        for t in scene.triangles:
            if t.material.transmission.mask():
                out.triangles.append(t)
                continue
            # abs_diff_ne(dot, 1.0, 0.01)
            if abs(t.normal.dot(view_direction) - F(1.0)) > F(0.01):
                out.triangles.append(t)
        return out

    # ---- flattening to rt_scene_desc ----------------------------------------------------
    def flatten(self) -> "FlatScene":
        mats: List[Tuple[float, ...]] = []
        index: Dict[Tuple[float, ...], int] = {}

        def mat_id(m: Material) -> int:
            row = m.row()
            i = index.get(row)
            if i is None:
                i = len(mats)
                index[row] = i
                mats.append(row)
            return i

        ns, nt = len(self.spheres), len(self.triangles)
        sc = np.zeros((ns, 3), np.float32)
        srs = np.zeros((ns,), np.float32)
        sri = np.zeros((ns,), np.float32)
        sm = np.zeros((ns,), np.uint32)
        for i, s in enumerate(self.spheres):
            sc[i] = s.center.to_list()
            srs[i], sri[i], sm[i] = s.r_sq, s.r_inv, mat_id(s.material)
        tv1 = np.zeros((nt, 3), np.float32)
        te1 = np.zeros((nt, 3), np.float32)
        te2 = np.zeros((nt, 3), np.float32)
        tn = np.zeros((nt, 3), np.float32)
        tm = np.zeros((nt,), np.uint32)
        for i, t in enumerate(self.triangles):
            tv1[i], te1[i], te2[i], tn[i] = t.vertex1.to_list(), t.edge1.to_list(), t.edge2.to_list(), t.normal.to_list()
            tm[i] = mat_id(t.material)
        lights = np.zeros((len(self.scene_lights), 7), np.float32)
        for i, l in enumerate(self.scene_lights):
            lights[i] = l.position.to_list() + [float(c) for c in l.color.to_tuple()] + [float(l.intensity)]
        materials = np.asarray(mats, np.float32).reshape(-1, 9)
        return FlatScene(sc, srs, sri, sm, tv1, te1, te2, tn, tm, materials, lights)


@dataclass
class FlatScene:
    """SoA arrays in the layout of `rt_scene_desc` (include/rt_hip.h)."""

    sphere_center: np.ndarray
    sphere_r_sq: np.ndarray
    sphere_r_inv: np.ndarray
    sphere_material: np.ndarray
    tri_v1: np.ndarray
    tri_e1: np.ndarray
    tri_e2: np.ndarray
    tri_normal: np.ndarray
    tri_material: np.ndarray
    materials: np.ndarray
    lights: np.ndarray

    @property
    def n_spheres(self) -> int:
        return int(self.sphere_center.shape[0])

    @property
    def n_triangles(self) -> int:
        return int(self.tri_v1.shape[0])

    @property
    def n_objects(self) -> int:
        return self.n_spheres + self.n_triangles

    def without_triangles(self) -> "FlatScene":
        """BASELINE config 2: "spheres-only intersect"."""
        z3 = np.zeros((0, 3), np.float32)
        return FlatScene(self.sphere_center, self.sphere_r_sq, self.sphere_r_inv, self.sphere_material,
                         z3, z3, z3, z3, np.zeros((0,), np.uint32), self.materials, self.lights)

    def fingerprint(self) -> bytes:
        """Content hash of every array (the renderer's device-scene cache key: a mutated or different scene must
        never render with a stale device copy)."""
        import hashlib

        h = hashlib.blake2b(digest_size=16)
        c = self.contiguous()
        for name in ("sphere_center", "sphere_r_sq", "sphere_r_inv", "sphere_material", "tri_v1", "tri_e1", "tri_e2",
                     "tri_normal", "tri_material", "materials", "lights"):
            a = getattr(c, name)
            h.update(name.encode())
            h.update(str(a.shape).encode())
            h.update(a.tobytes())
        return h.digest()

    def contiguous(self) -> "FlatScene":
        f = lambda a, dt: np.ascontiguousarray(a, dtype=dt)
        return FlatScene(
            f(self.sphere_center, np.float32), f(self.sphere_r_sq, np.float32), f(self.sphere_r_inv, np.float32),
            f(self.sphere_material, np.uint32), f(self.tri_v1, np.float32), f(self.tri_e1, np.float32),
            f(self.tri_e2, np.float32), f(self.tri_normal, np.float32), f(self.tri_material, np.uint32),
            f(self.materials, np.float32), f(self.lights, np.float32),
        )
