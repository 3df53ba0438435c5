"""Strict fp32 vector / rotor arithmetic for host-side scene construction.

The reference builds its scenes with `ultraviolet` 0.10 `Vec3`, `Rotor3`, `Isometry3` and
`Similarity3` (reference `src/main.rs:30-296`, `src/scene/scene.rs:43-134`,
`src/geometry/composite/bounded_plane.rs`).  Those types are not vendored in the reference tree,
so the formulas here are restated from the crate; every operation is
rounded to fp32 after each step, like the Rust code.
"""
from __future__ import annotations

import math

import numpy as np

F = np.float32
EPSILON = F(1.1920929e-7)  # f32::EPSILON, reference src/float_ext.rs:44-45


def f32(x) -> np.float32:
    return F(x)


def fma(a, b, c) -> np.float32:
    """fused a*b+c rounded once to fp32 (exact in float64 for fp32 inputs up to the final round
    in all but pathological double-rounding cases, which do not matter for scene set-up)."""
    return F(np.float64(a) * np.float64(b) + np.float64(c))


class Vec3:
    """ultraviolet::Vec3 (fp32)."""

    __slots__ = ("x", "y", "z")

    def __init__(self, x, y, z):
        self.x, self.y, self.z = F(x), F(y), F(z)

    # constructors -------------------------------------------------------------------------
    @staticmethod
    def new(x, y, z) -> "Vec3":
        return Vec3(x, y, z)

    @staticmethod
    def unit_x() -> "Vec3":
        return Vec3(1, 0, 0)

    @staticmethod
    def unit_y() -> "Vec3":
        return Vec3(0, 1, 0)

    @staticmethod
    def unit_z() -> "Vec3":
        return Vec3(0, 0, 1)

    @staticmethod
    def broadcast(v) -> "Vec3":
        return Vec3(v, v, v)

    # arithmetic ---------------------------------------------------------------------------
    def __add__(self, o):
        return Vec3(self.x + o.x, self.y + o.y, self.z + o.z)

    def __sub__(self, o):
        return Vec3(self.x - o.x, self.y - o.y, self.z - o.z)

    def __neg__(self):
        return Vec3(-self.x, -self.y, -self.z)

    def __mul__(self, o):
        if isinstance(o, Vec3):
            return Vec3(self.x * o.x, self.y * o.y, self.z * o.z)
        o = F(o)
        return Vec3(self.x * o, self.y * o, self.z * o)

    __rmul__ = __mul__

    def dot(self, o) -> np.float32:
        # x.mul_add(ox, y.mul_add(oy, z*oz))
        return fma(self.x, o.x, fma(self.y, o.y, self.z * o.z))

    def mag_sq(self) -> np.float32:
        return self.dot(self)

    def mag(self) -> np.float32:
        return F(np.sqrt(self.mag_sq()))

    def normalized(self) -> "Vec3":
        r = F(1.0) / self.mag()
        return self * r

    def cross(self, o) -> "Vec3":
        return Vec3(
            (self.y * o.z) + (-self.z * o.y),
            (self.z * o.x) + (-self.x * o.z),
            (self.x * o.y) + (-self.y * o.x),
        )

    def mul_add(self, mul: "Vec3", add: "Vec3") -> "Vec3":
        return Vec3(fma(self.x, mul.x, add.x), fma(self.y, mul.y, add.y), fma(self.z, mul.z, add.z))

    def lerp(self, end: "Vec3", t) -> "Vec3":
        t = F(t)
        return self * (F(1.0) - t) + end * t

    def rotated_by(self, rotor: "Rotor3") -> "Vec3":
        return rotor.rotate_vec(self)

    def to_list(self):
        return [float(self.x), float(self.y), float(self.z)]

    def __iter__(self):
        return iter((self.x, self.y, self.z))

    def __repr__(self):
        return f"Vec3({self.x!r}, {self.y!r}, {self.z!r})"


class Rotor3:
    """ultraviolet::Rotor3 {s, bv:{xy,xz,yz}} (fp32)."""

    __slots__ = ("s", "xy", "xz", "yz")

    def __init__(self, s, xy, xz, yz):
        self.s, self.xy, self.xz, self.yz = F(s), F(xy), F(xz), F(yz)

    @staticmethod
    def identity() -> "Rotor3":
        return Rotor3(1, 0, 0, 0)

    @staticmethod
    def from_angle_plane(angle, plane_xy, plane_xz, plane_yz) -> "Rotor3":
        half = F(angle) * F(0.5)
        s, c = F(math.sin(float(half))), F(math.cos(float(half)))
        return Rotor3(c, F(plane_xy) * -s, F(plane_xz) * -s, F(plane_yz) * -s)

    @staticmethod
    def from_rotation_xy(angle) -> "Rotor3":
        return Rotor3.from_angle_plane(angle, 1, 0, 0)

    @staticmethod
    def from_rotation_xz(angle) -> "Rotor3":
        return Rotor3.from_angle_plane(angle, 0, 1, 0)

    @staticmethod
    def from_rotation_yz(angle) -> "Rotor3":
        return Rotor3.from_angle_plane(angle, 0, 0, 1)

    @staticmethod
    def from_euler_angles(roll, pitch, yaw) -> "Rotor3":
        # roll: xy plane, pitch: yz plane, yaw: xz plane; applied roll -> pitch -> yaw
        return (
            Rotor3.from_rotation_xz(yaw) * Rotor3.from_rotation_yz(pitch) * Rotor3.from_rotation_xy(roll)
        )

    def __mul__(self, q: "Rotor3") -> "Rotor3":
        a = self
        return Rotor3(
            a.s * q.s - a.xy * q.xy - a.xz * q.xz - a.yz * q.yz,
            a.xy * q.s + a.s * q.xy + a.yz * q.xz - a.xz * q.yz,
            a.xz * q.s + a.s * q.xz - a.yz * q.xy + a.xy * q.yz,
            a.yz * q.s + a.s * q.yz + a.xz * q.xy - a.xy * q.xz,
        )

    def rotate_vec(self, v: Vec3) -> Vec3:
        s, xy, xz, yz = self.s, self.xy, self.xz, self.yz
        fx = s * v.x + xy * v.y + xz * v.z
        fy = s * v.y - xy * v.x + yz * v.z
        fz = s * v.z - xz * v.x - yz * v.y
        fw = xy * v.z - xz * v.y + yz * v.x
        return Vec3(
            s * fx + xy * fy + xz * fz + yz * fw,
            s * fy - xy * fx - xz * fw + yz * fz,
            s * fz + xy * fw - xz * fx - yz * fy,
        )


class Isometry3:
    """ultraviolet::Isometry3: rotate then translate."""

    def __init__(self, translation: Vec3, rotation: Rotor3):
        self.translation, self.rotation = translation, rotation

    @staticmethod
    def new(translation: Vec3, rotation: Rotor3) -> "Isometry3":
        return Isometry3(translation, rotation)

    def transform_vec(self, v: Vec3) -> Vec3:
        return self.rotation.rotate_vec(v) + self.translation


class Similarity3:
    """ultraviolet::Similarity3: rotate, scale, translate."""

    def __init__(self, translation: Vec3, rotation: Rotor3, scale):
        self.translation, self.rotation, self.scale = translation, rotation, F(scale)

    @staticmethod
    def new(translation: Vec3, rotation: Rotor3, scale) -> "Similarity3":
        return Similarity3(translation, rotation, scale)

    @staticmethod
    def identity() -> "Similarity3":
        return Similarity3(Vec3(0, 0, 0), Rotor3.identity(), 1.0)

    def transform_vec(self, v: Vec3) -> Vec3:
        return self.rotation.rotate_vec(v) * self.scale + self.translation


def gcd(a: int, b: int) -> int:
    """reference src/helpers.rs gcd (Euclid)."""
    while b:
        a, b = b, a % b
    return a


def lcm(a: int, b: int) -> int:
    return a // gcd(a, b) * b
