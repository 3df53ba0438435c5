"""Loads librt_hip.so (the HIP path).  There is NO fallback: if the library is missing or a HIP
call fails, this raises -- the product path never routes through a CPU implementation."""
from __future__ import annotations

import ctypes as C
import os

from ._abi import rt_aux, rt_bvh_info, rt_gather_info, rt_params, rt_scene_desc, rt_scene_info, rt_stats

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_HIP_LIB selects a diagnostic build of the same library (tools/, A/B timing); default: in-tree
LIB_PATH = os.environ.get("RT_HIP_LIB") or os.path.join(_HERE, "librt_hip.so")
EXPORTS = (
    "rt_device_count", "rt_scene_create", "rt_render", "rt_render_device", "rt_render_collect_stats",
    "rt_scene_destroy", "rt_last_error", "rt_scene_bvh_info", "rt_scene_memory_info", "rt_build_id", "rt_selftest_exact_math",
    "rt_gather_layout", "rt_render_multi", "rt_render_multi_begin", "rt_render_multi_end", "rt_multi_release", "rt_comm_unique_id", "rt_comm_create", "rt_comm_destroy",
    "rt_render_gather_device", "rt_comm_last_gather", "rt_render_begin", "rt_render_poll", "rt_render_end",
)

_lib = None


class RtError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"rt_hip error {code}: {msg}")
        self.code = code


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or `make -C hslu_i/ba_raytracing/f2501_raytracer_amd/csrc`).  There is no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    lib.rt_device_count.restype = C.c_int
    lib.rt_scene_create.restype = C.c_int
    lib.rt_scene_create.argtypes = [C.POINTER(rt_scene_desc), C.c_int, C.POINTER(C.c_void_p)]
    lib.rt_render.restype = C.c_int
    lib.rt_render.argtypes = [C.c_void_p, C.POINTER(rt_params), C.c_void_p, C.POINTER(rt_aux), C.POINTER(rt_stats)]
    lib.rt_render_device.restype = C.c_int
    lib.rt_render_device.argtypes = [C.c_void_p, C.POINTER(rt_params), C.c_void_p, C.POINTER(rt_aux), C.c_void_p]
    lib.rt_render_collect_stats.restype = C.c_int
    lib.rt_render_collect_stats.argtypes = [C.c_void_p, C.POINTER(rt_stats)]
    lib.rt_scene_destroy.restype = None
    lib.rt_scene_destroy.argtypes = [C.c_void_p]
    lib.rt_last_error.restype = C.c_char_p
    lib.rt_build_id.restype = C.c_char_p
    lib.rt_selftest_exact_math.restype = C.c_int
    lib.rt_selftest_exact_math.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    lib.rt_scene_bvh_info.restype = C.c_int
    lib.rt_scene_bvh_info.argtypes = [C.c_void_p, C.POINTER(rt_bvh_info)]
    lib.rt_render_begin.restype = C.c_int
    lib.rt_render_begin.argtypes = [C.c_void_p, C.POINTER(rt_params), C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
    lib.rt_render_poll.restype = C.c_int
    lib.rt_render_poll.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
    lib.rt_render_end.restype = C.c_int
    lib.rt_render_end.argtypes = [C.c_void_p, C.POINTER(rt_stats)]
    lib.rt_scene_memory_info.restype = C.c_int
    lib.rt_scene_memory_info.argtypes = [C.c_void_p, C.POINTER(rt_scene_info)]
    u32p = C.POINTER(C.c_uint32)
    lib.rt_gather_layout.restype = C.c_int
    lib.rt_gather_layout.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, u32p, u32p]
    lib.rt_render_multi.restype = C.c_int
    lib.rt_render_multi.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(rt_params), C.c_void_p, C.POINTER(rt_stats)]
    lib.rt_render_multi_begin.restype = C.c_int
    lib.rt_render_multi_begin.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(rt_params), C.c_void_p, C.POINTER(C.c_int)]
    lib.rt_render_multi_end.restype = C.c_int
    lib.rt_render_multi_end.argtypes = [C.c_int, C.POINTER(rt_stats)]
    lib.rt_multi_release.restype = None
    lib.rt_comm_unique_id.restype = C.c_int
    lib.rt_comm_unique_id.argtypes = [C.c_void_p]
    lib.rt_comm_create.restype = C.c_int
    lib.rt_comm_create.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_void_p)]
    lib.rt_comm_destroy.restype = None
    lib.rt_comm_destroy.argtypes = [C.c_void_p]
    lib.rt_render_gather_device.restype = C.c_int
    lib.rt_render_gather_device.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(rt_params), C.c_void_p, C.c_void_p]
    lib.rt_comm_last_gather.restype = C.c_int
    lib.rt_comm_last_gather.argtypes = [C.c_void_p, C.POINTER(rt_gather_info)]
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        raise RtError(rc, load().rt_last_error().decode("utf-8", "replace"))
