"""ctypes access to the CPU oracle (oracle/rt_oracle.c).  TEST INFRASTRUCTURE: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
SO = os.path.join(ORACLE_DIR, "_build", "librt_oracle.so")

from hslu_i.ba_raytracing.f2501_raytracer_amd import _abi  # noqa: E402

_lib = None
SO_NATIVE = os.path.join(ORACLE_DIR, "_build", "librt_oracle_native.so")
_build_desc = "-O2 -march=x86-64-v3 (built where the repo was built; travels to the GPU box)"


def build(force: bool = False) -> str:
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("rt_oracle.c", "rt_simd_baseline.c", "Makefile")] + \
        [os.path.join(ROOT, "include", "rt_hip.h")]
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return SO


def use_native_build() -> str:
    """bench.py's cpu_baseline: rebuild the oracle ON THIS MACHINE with -O3 -march=native and use that build
    from now on, if a C compiler is here; otherwise keep the travelling x86-64-v3 build.  Returns a description.  Only
    timing uses this; parity tests run the default build."""
    global _lib, SO, _build_desc
    try:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "-B", "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        if os.path.exists(SO_NATIVE):
            SO, _lib = SO_NATIVE, None
            _build_desc = "-O3 -march=native, built on this machine"
    except (OSError, subprocess.CalledProcessError):
        pass
    return _build_desc


def load():
    global _lib
    if _lib is not None:
        return _lib
    if SO != SO_NATIVE:
        build()
    lib = C.CDLL(SO)
    lib.rt_cpu_render.restype = C.c_int
    lib.rt_cpu_render.argtypes = [C.POINTER(_abi.rt_scene_desc), C.POINTER(_abi.rt_params), C.c_void_p,
                                  C.POINTER(_abi.rt_aux), C.POINTER(_abi.rt_stats), C.c_int]
    lib.rt_simd_render.restype = C.c_int
    lib.rt_simd_render.argtypes = lib.rt_cpu_render.argtypes
    lib.rt_simd_render_ex.restype = C.c_int
    lib.rt_simd_render_ex.argtypes = lib.rt_cpu_render.argtypes + [C.c_uint32]
    fp = C.POINTER(C.c_float)
    lib.rt_oracle_sphere.restype = C.c_int
    lib.rt_oracle_sphere.argtypes = [C.POINTER(_abi.rt_scene_desc), C.c_uint32, fp, fp, C.c_int, fp]
    lib.rt_oracle_triangle.restype = C.c_int
    lib.rt_oracle_triangle.argtypes = [C.POINTER(_abi.rt_scene_desc), C.c_uint32, fp, fp, C.c_int, fp]
    lib.rt_oracle_fresnel.restype = None
    lib.rt_oracle_fresnel.argtypes = [fp, fp, fp, C.c_float, fp]
    lib.rt_oracle_atten.restype = C.c_float
    lib.rt_oracle_atten.argtypes = [C.c_float]
    lib.rt_oracle_pack.restype = C.c_uint32
    lib.rt_oracle_pack.argtypes = [C.c_float, C.c_float, C.c_float]
    lib.rt_oracle_refract.restype = None
    lib.rt_oracle_refract.argtypes = [fp, fp, C.c_float, fp]
    _lib = lib
    return lib


def host_cores() -> int:
    """CPUs this process may actually use: the affinity mask capped by the cgroup quota (a GPU box shows 256 CPUs in
    /proc and grants 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


LITERAL_D3, LITERAL_D4 = 1, 2  # rt_simd_baseline.c: packet-literal modes


def render(flat, cfg, window=None, n_ranks=1, rank=0, n_threads=None, aux=True, aa_offsets=None, cloud=None, impl="scalar",
           literal=0):
    """Brute-force CPU render of `window` (x0,y0,w,h) of the frame.  impl: "scalar" = the parity oracle
    (rt_oracle.c), "simd" = the 8-lane AVX2 packet baseline over 48x48 tiles (rt_simd_baseline.c).
    Returns (argb, planes, stats)."""
    lib = load()
    desc, keep = _abi.make_scene_desc(flat)
    p, keep2 = _abi.make_params(cfg, aa_offsets=aa_offsets, cloud=cloud, window=window, n_ranks=n_ranks, rank=rank)
    n = cfg.width * cfg.height
    argb = np.zeros((n,), np.uint32)
    planes = {"rgb": np.zeros((n, 3), np.float32), "hit_id": np.full((n,), -2, np.int32), "hit_t": np.zeros((n,), np.float32)}
    a = _abi.rt_aux()
    a.rgb, a.hit_id, a.hit_t = planes["rgb"].ctypes.data, planes["hit_id"].ctypes.data, planes["hit_t"].ctypes.data
    st = _abi.rt_stats()
    if n_threads is None:
        n_threads = min(host_cores(), 16)
    if literal:
        assert impl == "simd", "packet-literal modes exist only in the packet implementation"
        rc = lib.rt_simd_render_ex(C.byref(desc), C.byref(p), argb.ctypes.data, C.byref(a) if aux else None, C.byref(st),
                                   int(n_threads), int(literal))
    else:
        fn = lib.rt_simd_render if impl == "simd" else lib.rt_cpu_render
        rc = fn(C.byref(desc), C.byref(p), argb.ctypes.data, C.byref(a) if aux else None, C.byref(st), int(n_threads))
    if rc != 0:
        raise RuntimeError(f"rt_cpu_render failed: {rc}")
    return argb, planes, st.as_dict()
