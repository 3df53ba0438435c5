"""The two scenes named by BASELINE.json, built through the mirrored scene API.

* `semesterbild(cfg)`  -- reference `src/main.rs:30-320` (identical to examples/semesterbild.rs)
* `test_scene(cfg)`    -- reference `examples/test_scene.rs:22-343` (incl. the stale light x =
                          SCENE_WIDTH - 80.0 kept literal)

Mesh input: the reference loads `data/obj/text/text.obj` / `text_lowres.obj`.  Those data files
are packed (vertex/normal/index arrays only) into `data/*.npz` next to this module by
`tools/pack_obj.py`, because the reference tree does not exist on the GPU box.
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np

from .config import RenderConfig
from .f32math import F, Isometry3, Rotor3, Similarity3, Vec3
from .scene import (BoundedPlane, ColorType, Material, PointLight, Scene, SphereData,
                    TransmissionProperties, TriangleData)

_DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def mesh_path(cfg: RenderConfig, model: Optional[str] = None) -> str:
    """model: None -> what src/main.rs:31-35 would pick; "text" / "text_lowres" to force."""
    if model is None:
        model = os.path.splitext(os.path.basename(cfg.model_path()))[0]
    return os.path.join(_DATA_DIR, model + ".npz")


def semesterbild(cfg: RenderConfig, model: Optional[str] = None) -> Scene:
    SW, SH, SD, AVG = cfg.scene_width, cfg.scene_height, cfg.scene_depth, cfg.average_scene_dimension
    from .obj import load_obj_scene

    scene = load_obj_scene(
        mesh_path(cfg, model),
        Similarity3.new(
            Vec3.new(F(0.0135) * SW, F(0.145) * SH, F(0.885) * SD),
            Rotor3.from_euler_angles(0.0, -0.015, 0.0),
            F(1.226) * AVG,
        ),
    )

    def sphere(cx, cy, cz, r, color, metallic, shininess, tp):
        scene.add_sphere(SphereData.with_material(
            Vec3.new(F(cx) * SW, F(cy) * SH, F(cz) * SD), F(r) * AVG,
            Material.new(ColorType.new(*color), metallic, shininess, tp)))

    TP = TransmissionProperties
    sphere(0.475, 0.385, 0.595, 0.291, (1.0, 0.8, 1.0), 0.0, 0.15, TP.new_with_boost(0.99, 1.5, 0.025))
    sphere(0.8, 0.76, 0.2, 0.07, (0.75, 0.5, 1.0), 0.2, 0.3, TP.new(0.78, 1.5))
    sphere(0.76, 0.76, 0.4, 0.07, (0.75, 0.9, 0.8), 0.2, 0.35, TP.new(0.6, 1.8))
    sphere(0.73, 0.7, 0.52, 0.065, (0.75, 0.9, 0.8), 0.0, 0.7, TP.new(0.78, 1.3))
    sphere(0.69, 0.76, 0.3, 0.07, (0.88, 0.9, 0.88), 0.0, 0.1, TP.new_with_boost(1.0, 1.42, 0.125))
    sphere(0.1, 0.68, 0.3, 0.07, (0.88, 0.9, 0.88), 0.2, 0.7, TP.none())
    sphere(0.35, 0.76, 0.25, 0.07, (0.9, 0.2, 0.3), 0.0, 0.01, TP.none())
    sphere(0.2, 0.87, 0.5, 0.07, (0.88, 0.5, 0.7), 0.4, 0.2, TP.none())
    sphere(0.5, 0.87, 0.46, 0.075, (1.0, 1.0, 1.0), 0.95, 0.23, TP.none())

    rotor = Rotor3.from_euler_angles(-0.04, 0.125, 0.51)
    isometry = Isometry3.new(Vec3.new(F(0.25) * SW, F(0.002) * SH, F(0.037) * SD), rotor)

    back = BoundedPlane.with_material(
        -(Vec3.unit_z().rotated_by(rotor)),
        isometry.transform_vec(Vec3.new(SW * F(0.5), (SH * F(1.1)) * F(0.5), SD)),
        Vec3.unit_y().rotated_by(rotor), SW, SH * F(1.1), F(0.01) * SD,
        Material.new(ColorType.new(0.5, 0.75, 0.75), 0.0, 0.0, TP.none())).to_basic_geometries()
    bottom = BoundedPlane.with_material(
        Vec3.unit_y().rotated_by(rotor),
        isometry.transform_vec(Vec3.new(SW * F(0.5), SH + F(0.001), SD * F(0.5))),
        Vec3.unit_z().rotated_by(rotor), SW, SD, F(0.012) * SD,
        Material.new(ColorType.new(0.75, 0.5, 0.75), 0.0, 0.7, TP.new(0.675, 1.13))).to_basic_geometries()
    bottom2 = BoundedPlane.with_material(
        Vec3.unit_y().rotated_by(rotor),
        isometry.transform_vec(Vec3.new(SW * F(0.5), SH + F(0.09), SD * F(0.5))),
        Vec3.unit_z().rotated_by(rotor), SW, SD, F(0.01) * SD,
        Material.new(ColorType.new(0.75, 0.5, 0.75), 0.0, 0.7, TP.none())).to_basic_geometries()
    right = BoundedPlane.with_material(
        -(Vec3.unit_x().rotated_by(rotor)),
        isometry.transform_vec(Vec3.new(SW, (SH * F(1.1)) * F(0.5), SD * F(0.5))),
        -(Vec3.unit_z().rotated_by(rotor)), SH * F(1.1), SD, F(0.01) * SD,
        Material.new(ColorType.new(0.875, 0.85, 0.61), 0.55, 0.325, TP.none())).to_basic_geometries()
    for tris in (back, bottom, bottom2, right):
        for t in tris:
            scene.add_triangle(t)

    scene.add_light(PointLight.new(Vec3.new(SW / F(1.2), 0.0, F(0.015) * SD), ColorType.new(0.825, 0.675, 0.5), 1.0).into())
    scene.add_light(PointLight.new(Vec3.new(SW / F(2.4), SH * F(0.1), F(0.08) * SD), ColorType.new(0.825, 0.675, 0.65), 0.675).into())
    scene.add_light(PointLight.new(Vec3.new(SW, SH, F(0.01) * SD), ColorType.new(0.825, 0.35, 0.8), 0.435).into())
    scene.add_light(PointLight.new(
        isometry.transform_vec(Vec3.new(SW * F(0.5), SH + F(0.05), SD * F(0.75))),
        ColorType.new(1.0, 1.0, 1.0), 0.2775).into())
    scene.add_light(PointLight.new(Vec3.new(F(0.2) * SW, SH * F(0.67), F(0.95) * SD), ColorType.new(0.825, 0.5, 0.7), 0.26).into())

    if cfg.has("scene_backface_culling"):
        scene = Scene.backface_culling(scene, Vec3.unit_z())
    return scene


def test_scene(cfg: RenderConfig) -> Scene:
    SW, SH, SD = cfg.scene_width, cfg.scene_height, cfg.scene_depth
    TP = TransmissionProperties
    scene = Scene.with_capacities(20)
    scene.add_sphere(SphereData.new(
        Vec3.new(SW / F(2.5), SH / F(2.75), F(0.170) * SD), F(0.070) * SD,
        ColorType.new(F(255.0) / F(255.0), F(0.0) / F(255.0), F(0.0) / F(255.0))))
    scene.add_sphere(SphereData.with_material(
        Vec3.new(SW / F(2.5), SH / F(1.5), F(0.170) * SD), F(0.070) * SD,
        Material.new(ColorType.new(1.0, 0.0, 0.0), 0.8, 0.0, TP.none())))
    scene.add_sphere(SphereData.with_material(
        Vec3.new(F(1.9) * (SW / F(2.5)), SH / F(2.8), F(0.160) * SD), F(0.088) * SD,
        Material.new(ColorType.new(F(250.0) / F(255.0), F(255.0) / F(255.0), F(245.0) / F(255.0)), 0.01, 0.2, TP.new(0.85, 1.5))))
    scene.add_sphere(SphereData.with_material(
        Vec3.new(SW / F(2.5), F(2.1) * (SH / F(2.5)), F(0.5) * SD), F(0.250) * SD,
        Material.new(ColorType.new(F(254.0) / F(255.0), 1.0, 1.0), 0.5, 0.05, TP.none())))

    scene.add_triangle(TriangleData.with_material(
        Vec3.new(SW * F(0.05), SH * F(0.2), F(0.2) * SD), Vec3.new(SW * F(0.3), SH * F(0.5), F(0.2) * SD),
        Vec3.new(SW * F(0.25), SH * F(0.15), F(0.15) * SD),
        Material.new(ColorType.new(0.5, 0.7, 0.8), 0.001, 0.2, TP.new(0.999, 1.8))))
    scene.add_triangle(TriangleData.with_material(
        Vec3.new(SW * F(0.55), SH * F(0.45), F(0.2) * SD), Vec3.new(SW * F(0.7), SH * F(0.72), F(0.2) * SD),
        Vec3.new(SW * F(0.65), SH * F(0.35), F(0.14) * SD),
        Material.new(ColorType.new(0.7, 0.7, 0.8), 0.1, 0.3, TP.none())))
    scene.add_triangle(TriangleData.with_material(
        Vec3.new(SW * F(0.7), SH * F(0.90), F(0.2) * SD), Vec3.new(SW * F(0.55), SH * F(0.65), F(0.2) * SD),
        Vec3.new(SW * F(0.65), SH * F(0.55), F(0.14) * SD),
        Material.new(ColorType.new(0.7, 0.7, 0.8), 0.1, 0.3, TP.new(1.0, 1.5))))

    def rotated_pair(rot: Rotor3):
        return (-Vec3.unit_z()).rotated_by(rot), Vec3.unit_y().rotated_by(rot)

    n, up = rotated_pair(Rotor3.from_rotation_yz(-0.555))
    for t in BoundedPlane.with_material(
            n, Vec3.new(SW * F(0.5), SH * F(0.45), F(0.270) * SD), up, SW * F(0.55), SH * F(0.55), F(0.01) * SD,
            Material.new(ColorType.new(0.6, 0.7, 0.5), 0.075, 0.07, TP.new_with_boost(1.0, 1.5, 0.5))).to_basic_geometries():
        scene.add_triangle(t)
    n, up = rotated_pair(Rotor3.from_rotation_xz(-0.9955))
    for t in BoundedPlane.with_material(
            n, Vec3.new(SW * F(0.82), SH * F(0.57), F(0.110) * SD), up, SW * F(0.318), SH * F(0.35), F(0.007) * SD,
            Material.new(ColorType.new(0.99, 0.99, 0.99), 1.0, 0.2, TP.none())).to_basic_geometries():
        scene.add_triangle(t)

    wall = lambda col: Material.new(ColorType.new(*col), 0.0, 0.0, TP.none())
    back = BoundedPlane.with_material(-Vec3.unit_z(), Vec3.new(SW * F(0.5), SH * F(0.5), SD), Vec3.unit_y(),
                                      SW, SH, F(0.001) * SD, wall((0.5, 0.75, 0.75))).to_basic_geometries()
    bottom = BoundedPlane.with_material(Vec3.unit_y(), Vec3.new(SW * F(0.5), SH, SD * F(0.5)), Vec3.unit_z(),
                                        SW, SD, F(0.001) * SD, wall((0.75, 0.5, 0.75))).to_basic_geometries()
    top = BoundedPlane.with_material(-Vec3.unit_y(), Vec3.new(SW * F(0.5), 0.0, SD * F(0.5)), Vec3.unit_z(),
                                     SW, SD, F(0.001) * SD, wall((0.75, 0.5, 0.75))).to_basic_geometries()
    left = BoundedPlane.with_material(Vec3.unit_x(), Vec3.new(0.0, SH * F(0.5), SD * F(0.5)), Vec3.unit_z(),
                                      SH, SD, F(0.001) * SD, wall((0.75, 0.75, 0.5))).to_basic_geometries()
    right = BoundedPlane.with_material(-Vec3.unit_x(), Vec3.new(SW, SH * F(0.5), SD * F(0.5)), -Vec3.unit_z(),
                                       SH, SD, F(0.001) * SD, wall((0.75, 0.75, 0.5))).to_basic_geometries()
    for tris in (back, bottom, top, left, right):
        for t in tris:
            scene.add_triangle(t)

    for pos, col, inten in (
        (Vec3.new(SW / F(2.0), SH / F(1.8), F(0.016) * SD), (0.825, 0.675, 0.5), 0.15),
        (Vec3.new(SW / F(3.5), SH / F(3.75), F(0.025) * SD), (0.825, 0.675, 0.45), 0.485),
        (Vec3.new(SW / F(1.22), SH / F(2.9), F(0.38) * SD), (0.78, 0.67, 0.45), 0.6),
        (Vec3.new(SW - F(80.0), SH / F(2.0), F(0.125) * SD), (1.0, 1.0, 1.0), 0.1),
        (Vec3.new(SW / F(2.5), SH / F(5.0), F(0.175) * SD), (0.75, 0.56, 0.65), 0.2),
        (Vec3.new(SW / F(4.0), SH / F(6.0), F(0.01) * SD), (0.01, 0.5, 0.4), 0.175),
    ):
        scene.add_light(PointLight.new(pos, ColorType.new(*col), inten).into())
    return scene
