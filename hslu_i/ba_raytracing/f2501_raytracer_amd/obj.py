"""OBJ/MTL loading with the semantics the reference gets from `tobj` 4.0.3
(`triangulate: true, single_index: true`), reference `src/scene/scene.rs:43-134` and
`src/raytracing/material.rs:96-126`.

Only what the reference consumes is parsed: `v`, `vn`, `f` (fan-triangulated), `usemtl`,
`mtllib` with `Kd`, `illum`, `Pm`, `Ps`.  Faces come out in file order, which is the order tobj
yields them (models in order of appearance, faces in order inside a model).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import numpy as np

from .f32math import F, Similarity3, Vec3
from .scene import ColorType, Material, Scene, TransmissionProperties, TriangleData


def _parse_mtl(path: str) -> Dict[str, Material]:
    mats: Dict[str, Material] = {}
    cur: Optional[str] = None
    props: Dict[str, Dict[str, str]] = {}
    with open(path, "r") as fh:
        for line in fh:
            parts = line.split()
            if not parts or parts[0].startswith("#"):
                continue
            if parts[0] == "newmtl":
                cur = parts[1]
                props[cur] = {}
            elif cur is not None:
                props[cur][parts[0]] = " ".join(parts[1:])
    for name, p in props.items():
        # material.rs:96-126
        illum = int(p["illum"]) if "illum" in p else 0
        kd = [float(v) for v in p["Kd"].split()] if "Kd" in p else [0.0, 0.0, 0.0]

        def fparam(key: str) -> float:
            try:
                return float(p.get(key, "0.0"))
            except ValueError:
                return 0.0

        metallic = fparam("Pm") if illum == 3 else 0.0
        shininess = fparam("Ps") if illum in (3, 2, 0) else 0.0
        mats[name] = Material(ColorType(*kd), F(metallic), F(shininess), TransmissionProperties.default())
    return mats


def parse_obj(path: str):
    """Returns (positions[nv,3] f32, normals[nn,3] f32, corners[nf,3,2] int (v idx, vn idx or -1),
    face_material_names list)."""
    pos: List[Tuple[float, float, float]] = []
    nrm: List[Tuple[float, float, float]] = []
    corners: List[List[Tuple[int, int]]] = []
    fmat: List[Optional[str]] = []
    mtllibs: List[str] = []
    cur_mat: Optional[str] = None
    with open(path, "r") as fh:
        for line in fh:
            if not line or line[0] == "#":
                continue
            parts = line.split()
            if not parts:
                continue
            k = parts[0]
            if k == "v":
                pos.append((float(parts[1]), float(parts[2]), float(parts[3])))
            elif k == "vn":
                nrm.append((float(parts[1]), float(parts[2]), float(parts[3])))
            elif k == "f":
                idx = []
                for tok in parts[1:]:
                    f = tok.split("/")
                    vi = int(f[0])
                    vi = vi - 1 if vi > 0 else len(pos) + vi
                    ni = -1
                    if len(f) >= 3 and f[2] != "":
                        ni = int(f[2])
                        ni = ni - 1 if ni > 0 else len(nrm) + ni
                    idx.append((vi, ni))
                for j in range(1, len(idx) - 1):  # fan triangulation (tobj triangulate)
                    corners.append([idx[0], idx[j], idx[j + 1]])
                    fmat.append(cur_mat)
            elif k == "usemtl":
                cur_mat = parts[1] if len(parts) > 1 else None
            elif k == "mtllib":
                mtllibs.append(parts[1])
    return (
        np.asarray(pos, np.float32).reshape(-1, 3),
        np.asarray(nrm, np.float32).reshape(-1, 3),
        np.asarray(corners, np.int64).reshape(-1, 3, 2),
        fmat,
        mtllibs,
    )


def pack_obj(obj_path: str, npz_path: str) -> None:
    """Packs the arrays the loader consumes (nothing else) into an .npz (tools/pack_obj.py)."""
    pos, nrm, corners, fmat, mtllibs = parse_obj(obj_path)
    mats: Dict[str, Material] = {}
    for lib in mtllibs:
        mats.update(_parse_mtl(os.path.join(os.path.dirname(obj_path), lib)))
    names = sorted(set(n for n in fmat if n is not None))
    rows = np.asarray([mats[n].row() if n in mats else Material.diffuse(ColorType(1, 1, 1)).row() for n in names],
                      np.float32).reshape(-1, 9)
    known = np.asarray([n in mats for n in names], np.bool_)
    fidx = np.asarray([names.index(n) if n is not None else -1 for n in fmat], np.int32)
    np.savez_compressed(npz_path, positions=pos, normals=nrm, corners=corners.astype(np.int32),
                        face_material=fidx, material_rows=rows, material_known=known)


def _load_packed(path: str):
    z = np.load(path)
    rows, known = z["material_rows"], z["material_known"]
    mats: Dict[str, Material] = {}
    for i in range(rows.shape[0]):
        if known[i]:
            r = rows[i]
            mats[str(i)] = Material(ColorType(r[0], r[1], r[2]), F(r[3]), F(r[4]),
                                    TransmissionProperties(F(r[5]), F(r[6]), bool(r[8] != 0), F(r[7])))
    fmat = [str(i) if i >= 0 else None for i in z["face_material"]]
    return z["positions"].astype(np.float32), z["normals"].astype(np.float32), z["corners"].astype(np.int64), fmat, mats


def load_obj_scene(path: str, transform: Optional[Similarity3], continue_on_material_failure: bool = True) -> Scene:
    if not os.path.exists(path):
        raise FileNotFoundError(path)  # tobj::LoadError::OpenFileFailed
    materials: Dict[str, Material] = {}
    if path.endswith(".npz"):
        pos, nrm, corners, fmat, materials = _load_packed(path)
    else:
        pos, nrm, corners, fmat, mtllibs = parse_obj(path)
        for lib in mtllibs:
            mp = os.path.join(os.path.dirname(path), lib)
            try:
                materials.update(_parse_mtl(mp))
            except OSError:
                if not continue_on_material_failure:
                    raise
    tr = transform if transform is not None else Similarity3.identity()
    # vectorised fp32 transform of every referenced vertex / normal (same op order as
    # Similarity3::transform_vec / Vec3::rotated_by on each element)
    P = Vec3(pos[:, 0], pos[:, 1], pos[:, 2])
    Pt = tr.transform_vec(P)
    pt = np.stack([Pt.x, Pt.y, Pt.z], axis=1).astype(np.float32)
    have_n = nrm.shape[0] > 0
    if have_n:
        N = Vec3(nrm[:, 0], nrm[:, 1], nrm[:, 2])
        Nr = tr.rotation.rotate_vec(N)
        nr = np.stack([Nr.x, Nr.y, Nr.z], axis=1).astype(np.float32)
    default_mat = Material.diffuse(ColorType(1.0, 1.0, 1.0))
    s = Scene()
    nf = corners.shape[0]
    v = pt[corners[:, :, 0]]  # [nf,3,3]
    half = np.float32(0.5)
    for i in range(nf):
        v1, v2, v3 = (Vec3(*v[i, c]) for c in range(3))
        mat = materials.get(fmat[i]) if fmat[i] is not None else None
        face_material = mat if mat is not None else default_mat
        nidx = corners[i, :, 1]
        ns = [Vec3(*nr[j]) if (have_n and j >= 0) else None for j in nidx]
        present = [n for n in ns if n is not None]
        if len(present) == 0:
            s.add_triangle(TriangleData.with_material(v1, v2, v3, face_material))
            continue
        if len(present) == 1:
            n = present[0]
        elif len(present) == 2:
            n = present[0].lerp(present[1], half)
        else:
            n = ns[0].lerp(ns[1], half).lerp(ns[2], half)
        s.add_triangle(TriangleData.with_material_and_normal(v1, v2, v3, n, face_material))
    return s
