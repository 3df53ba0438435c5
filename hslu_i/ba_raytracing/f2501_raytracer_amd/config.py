"""Render configuration: the reference's compile-time feature/const table as runtime values.

Reference: `src/lib.rs:30-92` (resolution, scene units, focus, factors, air ior) and
`src/renderer/raytracer_renderer.rs:55-93` (recursion depths, light-cloud size, samples per
pixel), `Cargo.toml:62-83` (feature implications).  Every value is computed in fp32 in the same
order as the Rust `const` expressions.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import FrozenSet, Iterable, Optional

import numpy as np

from .f32math import EPSILON, F, Vec3

# Cargo.toml:62-83 feature implications
_IMPLIES = {
    "anti_aliasing_rotation_scale": ["anti_aliasing"],
    "anti_aliasing_randomness": ["anti_aliasing"],
    "realistic": ["reflections", "light_reflections", "refractions"],
    "high_quality": ["anti_aliasing", "soft_shadows", "high_quality_model"],
    "extreme_quality": ["high_quality"],
}
KNOWN_FEATURES = frozenset(
    [
        "simd_render", "anti_aliasing", "anti_aliasing_rotation_scale", "anti_aliasing_randomness",
        "high_resolution", "medium_resolution", "soft_shadows", "reflections", "light_reflections",
        "refractions", "backface_culling", "scene_backface_culling", "save_rendering_image",
        "realistic", "high_quality_model", "high_quality", "extreme_quality",
    ]
)
# Cargo.toml:64
DEFAULT_FEATURES = (
    "realistic", "save_rendering_image", "scene_backface_culling", "anti_aliasing_randomness",
    "anti_aliasing_rotation_scale", "medium_resolution", "high_quality",
)


def expand_features(features: Iterable[str]) -> FrozenSet[str]:
    out = set()
    todo = list(features)
    while todo:
        f = todo.pop()
        if f not in KNOWN_FEATURES:
            raise ValueError(f"unknown feature {f!r}")
        if f in out:
            continue
        out.add(f)
        todo.extend(_IMPLIES.get(f, ()))
    return frozenset(out)


@dataclass(frozen=True)
class RenderConfig:
    """All constants the render path reads, derived from a feature set + optional resolution
    override (the reference's compile-time `WINDOW_WIDTH`/`WINDOW_HEIGHT` env, lib.rs:50-71)."""

    features: FrozenSet[str] = field(default_factory=frozenset)
    width_override: Optional[int] = None
    height_override: Optional[int] = None
    # runtime-only knobs without reference counterpart
    depth_override: Optional[int] = None  # BASELINE config 4 quotes "recursion depth 8"
    cloud_seed: int = 1
    aa_seed: int = 1
    n_cloud_sets: int = 1024
    scene_scale: float = 1.0  # SCENE_WIDTH (lib.rs:73 fixes it to 1.0): tests render scaled copies of a scene

    @staticmethod
    def from_features(features: Iterable[str] = (), **kw) -> "RenderConfig":
        return RenderConfig(features=expand_features(features), **kw)

    def has(self, f: str) -> bool:
        return f in self.features

    # ---- lib.rs:30-71 -------------------------------------------------------------------
    @property
    def width(self) -> int:
        if self.width_override:
            return int(self.width_override)
        return 1620 if self.has("high_resolution") else (1140 if self.has("medium_resolution") else 768)

    @property
    def height(self) -> int:
        if self.height_override:
            return int(self.height_override)
        return 1350 if self.has("high_resolution") else (950 if self.has("medium_resolution") else 640)

    # ---- lib.rs:73-92 -------------------------------------------------------------------
    @property
    def aspect(self) -> np.float32:
        return F(self.height) / F(self.width)

    @property
    def scene_width(self) -> np.float32:
        return F(self.scene_scale)

    @property
    def scene_height(self) -> np.float32:
        return self.scene_width * self.aspect

    @property
    def scene_depth(self) -> np.float32:
        return (self.scene_width + self.scene_height) / F(2.0)

    @property
    def average_scene_dimension(self) -> np.float32:
        return (self.scene_width + self.scene_height + self.scene_depth) / F(3.0)

    @property
    def window_scene_depth(self) -> int:
        return (self.width + self.height) // 2

    @property
    def fw(self) -> np.float32:
        return self.scene_width / F(self.width)

    @property
    def fh(self) -> np.float32:
        return self.scene_height / F(self.height)

    @property
    def fd(self) -> np.float32:
        return self.scene_depth / F(self.window_scene_depth)

    @property
    def focus(self) -> Vec3:
        return Vec3(self.scene_width / F(2.0), self.scene_height / F(2.0), F(-1.9) * self.scene_depth)

    @property
    def air_ior(self) -> np.float32:
        return F(1.000293)

    @property
    def eps_distance(self) -> np.float32:
        # vector.rs:697-700: default_epsilon() * from_element(100.0 * AVERAGE_SCENE_DIMENSION)
        return EPSILON * (F(100.0) * self.average_scene_dimension)

    # ---- raytracer_renderer.rs:55-93 ----------------------------------------------------
    @property
    def max_depth_reflection(self) -> int:
        if self.depth_override is not None:
            return int(self.depth_override)
        if self.has("high_quality"):
            return 21 if self.has("extreme_quality") else 13
        return 9

    @property
    def max_depth_refraction(self) -> int:
        if self.depth_override is not None:
            return int(self.depth_override)
        if self.has("high_quality"):
            return 21 if self.has("extreme_quality") else 18
        return 8

    @property
    def point_light_multiplicator(self) -> int:
        if self.has("soft_shadows"):
            if self.has("high_quality"):
                return 28 if self.has("extreme_quality") else 19
            return 10
        return 1

    @property
    def samples_per_pixel(self) -> int:
        return 24 if self.has("extreme_quality") else 9

    @property
    def aa_total_rays(self) -> int:
        """get_total_rays::<Vec3x8>() = spp.next_multiple_of(8), raytracer_renderer.rs:1018-1020"""
        return -(-self.samples_per_pixel // 8) * 8

    @property
    def ambient(self) -> np.float32:
        return F(0.08)  # raytracer_renderer.rs:754

    @property
    def render_stride(self) -> int:
        """RENDER_STRIDE = lcm(16*3, lcm(8, gcd(W, 16))), renderer/mod.rs:84-90"""
        from .f32math import gcd, lcm

        return lcm(16 * 3, lcm(8, gcd(self.width, 16)))

    def model_path(self) -> str:
        """src/main.rs:31-35"""
        if self.has("high_quality_model") or self.has("medium_resolution"):
            return "data/obj/text/text.obj"
        return "data/obj/text/text_lowres.obj"
