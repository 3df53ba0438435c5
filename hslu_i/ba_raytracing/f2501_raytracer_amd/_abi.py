"""ctypes mirror of include/rt_hip.h (structures only; no library is loaded here)."""
from __future__ import annotations

import ctypes as C

import numpy as np

RT_ABI_VERSION = 4
RT_OK = 0
RT_ERR_INVALID_ARG, RT_ERR_NO_DEVICE, RT_ERR_HIP, RT_ERR_OOM, RT_ERR_UNSUPPORTED = -1, -2, -3, -4, -5
RT_FLAG_REFLECTIONS, RT_FLAG_REFRACTIONS, RT_FLAG_BACKFACE_CULLING, RT_FLAG_ANTI_ALIASING = 1, 2, 4, 8
RT_TRAVERSAL_BVH, RT_TRAVERSAL_LINEAR = 0, 1
RT_CAND_CAP_NONE = 0xFFFFFFFF
RT_TILE_ORDER_DEFAULT, RT_TILE_ORDER_ROW_MAJOR, RT_TILE_ORDER_COST = 0, 1, 2
RT_PHASES_DEFAULT, RT_PHASES_FUSED, RT_PHASES_SPLIT, RT_PHASES_FUSED_DEFER = 0, 1, 2, 3
RT_LEVELS_DEFAULT, RT_LEVELS_CHAINED, RT_LEVELS_MERGED, RT_LEVELS_PIPELINED = 0, 1, 2, 3
(RT_NOTE_RECV_FLAGS_OFF_LIGHTS, RT_NOTE_RECV_FLAGS_OFF_CULLING, RT_NOTE_RECV_FLAGS_OFF_TRAVERSAL, RT_NOTE_RECV_FLAGS_OFF_TUNING,
 RT_NOTE_RECV_FLAGS_OFF_SCENE, RT_NOTE_HARD_PAIRS_OFF, RT_NOTE_FRAME_BATCHED, RT_NOTE_CELL_LISTS_OFF, RT_NOTE_TILE_ORDER_COST_OFF,
 RT_NOTE_FRAME_DROPPED_WORK) = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512
RT_SCENE_BUDGET_DEFAULT = 128 << 20

_fp = C.POINTER(C.c_float)
_up = C.POINTER(C.c_uint32)
_ip = C.POINTER(C.c_int32)


class rt_bvh_tuning(C.Structure):
    _fields_ = [("max_leaf", C.c_uint32), ("tri_cost", C.c_float), ("split_depth", C.c_uint32), ("split_gain", C.c_float)]


class rt_tuning(C.Structure):
    _fields_ = [("shadow_candidate_cap", C.c_uint32), ("chunk_log2", C.c_uint32), ("no_aa_dedup", C.c_uint32),
                ("no_counters", C.c_uint32), ("multi_force_rccl", C.c_uint32), ("no_receiver_flags", C.c_uint32),
                ("tile_order", C.c_uint32), ("sort_bits", C.c_uint32), ("no_cell_lists", C.c_uint32),
                ("sub_frames", C.c_uint32), ("phases", C.c_uint32), ("levels", C.c_uint32)]


class rt_scene_desc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("n_spheres", C.c_uint32), ("sphere_center", _fp), ("sphere_r_sq", _fp), ("sphere_r_inv", _fp),
        ("sphere_material", _up),
        ("n_triangles", C.c_uint32), ("tri_v1", _fp), ("tri_e1", _fp), ("tri_e2", _fp), ("tri_normal", _fp),
        ("tri_material", _up),
        ("n_materials", C.c_uint32), ("materials", _fp),
        ("n_lights", C.c_uint32), ("lights", _fp),
        ("bvh", rt_bvh_tuning),
        ("device_budget_bytes", C.c_uint64),
    ]


class rt_params(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32),
        ("focus", C.c_float * 3), ("fw", C.c_float), ("fh", C.c_float), ("fd", C.c_float),
        ("eps_distance", C.c_float), ("air_ior", C.c_float), ("ambient", C.c_float),
        ("flags", C.c_uint32),
        ("aa_rays", C.c_uint32), ("aa_offsets", _fp),
        ("light_mult", C.c_uint32), ("cloud_seed", C.c_uint32), ("n_cloud_sets", C.c_uint32), ("cloud_sets", _fp),
        ("max_depth_reflection", C.c_uint32), ("max_depth_refraction", C.c_uint32),
        ("win_x0", C.c_uint32), ("win_y0", C.c_uint32), ("win_w", C.c_uint32), ("win_h", C.c_uint32),
        ("tile_size", C.c_uint32), ("n_ranks", C.c_uint32), ("rank", C.c_uint32),
        ("traversal", C.c_uint32),
        ("tuning", rt_tuning),
    ]


class rt_aux(C.Structure):
    _fields_ = [("rgb", C.c_void_p), ("hit_id", C.c_void_p), ("hit_t", C.c_void_p)]


class rt_stats(C.Structure):
    _fields_ = [
        ("rays_primary", C.c_uint64), ("rays_reflection", C.c_uint64), ("rays_refraction", C.c_uint64),
        ("rays_shadow", C.c_uint64), ("pixels_written", C.c_uint64), ("rays_traced", C.c_uint64),
        ("kernel_ms", C.c_double), ("total_ms", C.c_double), ("d2h_ms", C.c_double), ("gather_ms", C.c_double),
        ("wave_ray_passes", C.c_uint64), ("wave_ray_lanes", C.c_uint64),
        ("wave_nearest_nodes", C.c_uint64), ("wave_nearest_tris", C.c_uint64),
        ("wave_shadow_nodes", C.c_uint64), ("wave_shadow_tris", C.c_uint64), ("wave_shadow_passes", C.c_uint64),
        ("wave_nearest_tris_exact", C.c_uint64), ("wave_shadow_tris_exact", C.c_uint64),
        ("notes", C.c_uint32), ("reserved0", C.c_uint32), ("queue_bytes", C.c_uint64),
        ("setup_ms", C.c_double), ("scene_bytes", C.c_uint64),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class rt_gather_info(C.Structure):
    _fields_ = [
        ("render_ms", C.c_double), ("gather_ms", C.c_double), ("bytes_sent", C.c_uint64), ("bytes_received", C.c_uint64),
        ("n_ranks", C.c_uint32), ("rank", C.c_uint32), ("tiles_owned", C.c_uint32), ("transport", C.c_uint32),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


RT_COMM_ID_BYTES = 128
RT_TRANSPORT_NONE, RT_TRANSPORT_RCCL, RT_TRANSPORT_LOCAL = 0, 1, 2


class rt_bvh_info(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_uint32), ("n_leaves", C.c_uint32), ("max_depth", C.c_uint32), ("max_leaf_size", C.c_uint32),
        ("bytes_nodes", C.c_uint64), ("bytes_triangles", C.c_uint64), ("n_references", C.c_uint32),
    ]


class rt_scene_info(C.Structure):
    _fields_ = [
        ("bytes_geometry", C.c_uint64), ("bytes_bvh", C.c_uint64), ("bytes_flags", C.c_uint64), ("bytes_cell_lists", C.c_uint64),
        ("bytes_tables", C.c_uint64), ("bytes_workspace", C.c_uint64), ("bytes_frames", C.c_uint64), ("bytes_total", C.c_uint64),
        ("budget_bytes", C.c_uint64), ("n_receiver_cells", C.c_uint32), ("cell_lists_built", C.c_uint32),
    ]


def fptr(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_fp)


def uptr(a: np.ndarray):
    assert a.dtype == np.uint32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_up)


def make_scene_desc(flat, bvh=None, budget=0):
    """flat: FlatScene (contiguous); bvh: dict of rt_bvh_tuning fields; budget: rt_scene_desc.device_budget_bytes (0 = default).
    Returns (desc, keepalive)."""
    f = flat.contiguous()
    d = rt_scene_desc()
    d.abi_version = RT_ABI_VERSION
    d.n_spheres = f.n_spheres
    d.sphere_center, d.sphere_r_sq, d.sphere_r_inv = fptr(f.sphere_center), fptr(f.sphere_r_sq), fptr(f.sphere_r_inv)
    d.sphere_material = uptr(f.sphere_material)
    d.n_triangles = f.n_triangles
    d.tri_v1, d.tri_e1, d.tri_e2, d.tri_normal = fptr(f.tri_v1), fptr(f.tri_e1), fptr(f.tri_e2), fptr(f.tri_normal)
    d.tri_material = uptr(f.tri_material)
    d.n_materials = int(f.materials.shape[0])
    d.materials = fptr(f.materials)
    d.n_lights = int(f.lights.shape[0])
    d.lights = fptr(f.lights)
    for k, v in (bvh or {}).items():
        setattr(d.bvh, k, v)
    d.device_budget_bytes = int(budget)
    return d, f


def make_params(cfg, aa_offsets=None, cloud=None, window=None, n_ranks=1, rank=0, traversal=RT_TRAVERSAL_BVH, tuning=None):
    """cfg: RenderConfig; tuning: dict of rt_tuning fields.  Returns (params, keepalive)."""
    from . import sampling

    p = rt_params()
    p.abi_version = RT_ABI_VERSION
    p.width, p.height = cfg.width, cfg.height
    fo = cfg.focus
    p.focus[0], p.focus[1], p.focus[2] = float(fo.x), float(fo.y), float(fo.z)
    p.fw, p.fh, p.fd = float(cfg.fw), float(cfg.fh), float(cfg.fd)
    p.eps_distance = float(cfg.eps_distance)
    p.air_ior = float(cfg.air_ior)
    p.ambient = float(cfg.ambient)
    flags = 0
    if cfg.has("reflections"):
        flags |= RT_FLAG_REFLECTIONS
    if cfg.has("refractions"):
        flags |= RT_FLAG_REFRACTIONS
    if cfg.has("backface_culling"):
        flags |= RT_FLAG_BACKFACE_CULLING
    keep = []
    if cfg.has("anti_aliasing"):
        flags |= RT_FLAG_ANTI_ALIASING
        if aa_offsets is None:
            aa_offsets = sampling.aa_offsets(cfg)
        aa_offsets = np.ascontiguousarray(aa_offsets, np.float32)
        p.aa_rays = int(aa_offsets.shape[0])
        p.aa_offsets = fptr(aa_offsets)
        keep.append(aa_offsets)
    p.flags = flags
    n = cfg.point_light_multiplicator
    p.light_mult = n
    p.cloud_seed = int(cfg.cloud_seed) & 0xFFFFFFFF
    if n > 1:
        if cloud is None:
            cloud = sampling.cloud_sets(cfg)
        cloud = np.ascontiguousarray(cloud, np.float32)
        assert cloud.shape[1] == n and cloud.shape[2] == 3
        p.n_cloud_sets = int(cloud.shape[0])
        p.cloud_sets = fptr(cloud)
        keep.append(cloud)
    p.max_depth_reflection = cfg.max_depth_reflection
    p.max_depth_refraction = cfg.max_depth_refraction
    if window is not None:
        p.win_x0, p.win_y0, p.win_w, p.win_h = (int(v) for v in window)
    p.tile_size = cfg.render_stride
    p.n_ranks, p.rank = int(n_ranks), int(rank)
    p.traversal = int(traversal)
    for k, v in (tuning or {}).items():
        setattr(p.tuning, k, int(v))
    return p, keep
