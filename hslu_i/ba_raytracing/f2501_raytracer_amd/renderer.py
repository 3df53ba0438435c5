"""`ImageBuffer` and `RaytracerRenderer`: the reference's render boundary over the HIP library.

Reference: `trait Renderer<W,H,C>::render(&self, buffer: &ImageBuffer<W,H>, scene: &Scene<Vec3>)`
(`src/renderer/mod.rs:80-94`), implemented by `RaytracerRenderer<C>`
(`src/renderer/raytracer_renderer.rs:139-140,1360-1378`); `ImageBuffer<W,H>` = `[AtomicU32; W*H]`
(`src/image_buffer.rs:8-44`).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import numpy as np

from . import _abi, _lib
from .config import RenderConfig
from .scene import FlatScene, Scene


class ImageBuffer:
    """W*H packed 0xAARRGGBB pixels, row-major, zero-initialised (image_buffer.rs:27-37)."""

    def __init__(self, width: int, height: int, color: int = 0):
        self.width, self.height = int(width), int(height)
        self.buffer = np.full((self.height * self.width,), color, np.uint32)

    @staticmethod
    def new(width: int, height: int) -> "ImageBuffer":
        return ImageBuffer(width, height, 0)

    @staticmethod
    def new_with_color(width: int, height: int, color: int) -> "ImageBuffer":
        return ImageBuffer(width, height, color)

    def get_u32_slice(self) -> np.ndarray:
        return self.buffer

    def __len__(self) -> int:
        return self.buffer.shape[0]

    def as_rgb8(self) -> np.ndarray:
        """FileOutput::render_buffer's u32 -> RGB8 rows (reference src/output/file.rs:27-49)."""
        b = self.buffer.reshape(self.height, self.width)
        return np.stack([(b >> 16) & 0xFF, (b >> 8) & 0xFF, b & 0xFF], axis=-1).astype(np.uint8)


class DeviceScene:
    """Owns an `rt_scene*` (device copies + BVH)."""

    def __init__(self, flat: FlatScene, device: int = 0, bvh: Optional[Dict] = None, budget: int = 0):
        """budget: device memory for the optional acceleration tables (rt_scene_desc.device_budget_bytes; 0 = 128 MiB)."""
        lib = _lib.load()
        desc, keep = _abi.make_scene_desc(flat, bvh, budget)
        h = C.c_void_p()
        _lib.check(lib.rt_scene_create(C.byref(desc), int(device), C.byref(h)))
        self._h = h
        self.device = int(device)
        self.flat = keep

    @property
    def handle(self) -> C.c_void_p:
        if self._h is None:
            raise RuntimeError("scene destroyed")
        return self._h

    def bvh_info(self) -> Dict[str, int]:
        info = _abi.rt_bvh_info()
        _lib.check(_lib.load().rt_scene_bvh_info(self.handle, C.byref(info)))
        return {k: int(getattr(info, k)) for k, _ in info._fields_}

    def memory_info(self) -> Dict[str, int]:
        info = _abi.rt_scene_info()
        _lib.check(_lib.load().rt_scene_memory_info(self.handle, C.byref(info)))
        return {k: int(getattr(info, k)) for k, _ in info._fields_}

    def close(self):
        if self._h is not None:
            _lib.load().rt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RaytracerRenderer:
    """`RaytracerRenderer::<C>::default().render(&buffer, &scene)` on one MI355X.

    The reference renderer is a stateless ZST configured at compile time; here the configuration is
    a `RenderConfig` and the renderer caches the device scene between calls.
    """

    def __init__(self, cfg: RenderConfig, device: int = 0, traversal: int = _abi.RT_TRAVERSAL_BVH, scene_budget: int = 0):
        """scene_budget: rt_scene_desc.device_budget_bytes of the device scenes this renderer creates (0 = the library's
        default, 128 MiB for the optional acceleration tables)."""
        self.cfg = cfg
        self.device = int(device)
        self.traversal = int(traversal)
        self.scene_budget = int(scene_budget)
        self._cache: Optional[Tuple[bytes, DeviceScene]] = None
        self.last_stats: Optional[Dict] = None

    @staticmethod
    def default(cfg: Optional[RenderConfig] = None) -> "RaytracerRenderer":
        return RaytracerRenderer(cfg if cfg is not None else RenderConfig.from_features(()))

    def device_scene(self, scene) -> DeviceScene:
        """The reference's `render(&buffer, &scene)` reads the scene on every call; so does this: the scene is
        flattened and fingerprinted each time, and the cached device copy (+ BVH) is reused only when the content
        is unchanged.  Pass a `DeviceScene` to skip both."""
        if isinstance(scene, DeviceScene):
            return scene
        flat = scene.flatten() if isinstance(scene, Scene) else scene
        key = flat.fingerprint()
        if self._cache is not None and self._cache[0] == key:
            return self._cache[1]
        if self._cache is not None:
            self._cache[1].close()
        ds = DeviceScene(flat, self.device, budget=self.scene_budget)
        self._cache = (key, ds)
        return ds

    def render(self, buffer: ImageBuffer, scene, window=None, n_ranks: int = 1, rank: int = 0,
               aux: bool = False, tuning: Optional[Dict] = None):
        """Renders `scene` into `buffer` (hit pixels only).  Returns aux planes when aux=True.
        tuning: rt_tuning fields (execution knobs that never change the image)."""
        cfg = self.cfg
        if buffer.width != cfg.width or buffer.height != cfg.height:
            raise ValueError(f"buffer is {buffer.width}x{buffer.height}, config renders {cfg.width}x{cfg.height}")
        ds = self.device_scene(scene)
        p, keep = _abi.make_params(cfg, window=window, n_ranks=n_ranks, rank=rank, traversal=self.traversal, tuning=tuning)
        st = _abi.rt_stats()
        a = _abi.rt_aux()
        planes = None
        if aux:
            n = cfg.width * cfg.height
            planes = {
                "rgb": np.zeros((n, 3), np.float32),
                "hit_id": np.full((n,), -2, np.int32),
                "hit_t": np.zeros((n,), np.float32),
            }
            a.rgb, a.hit_id, a.hit_t = planes["rgb"].ctypes.data, planes["hit_id"].ctypes.data, planes["hit_t"].ctypes.data
        lib = _lib.load()
        _lib.check(lib.rt_render(ds.handle, C.byref(p), buffer.buffer.ctypes.data, C.byref(a) if aux else None, C.byref(st)))
        self.last_stats = st.as_dict()
        return planes

    def render_progressive(self, buffer: ImageBuffer, scene, on_tiles=None, rows_per_step: Optional[int] = None, poll_s: float = 0.0002,
                           tuning: Optional[Dict] = None):
        """Progressive display (reference: the render thread fills the shared u32 buffer tile by tile while the window loop
        keeps showing it, src/main.rs:327-347, renderer/mod.rs:84-209): `rt_render_begin` starts the library's render thread,
        which renders the frame in bands of `rows_per_step` tile rows (default: one row of RENDER_STRIDE tiles); this thread --
        the viewer's side -- polls, and `on_tiles(buffer, (x0, y0, w, h))` is called for every band once its rows have landed
        in `buffer` (rows below are still the caller's fill at that moment).  The final buffer equals one `render` call.
        Returns the number of bands."""
        import time

        cfg = self.cfg
        if buffer.width != cfg.width or buffer.height != cfg.height:
            raise ValueError(f"buffer is {buffer.width}x{buffer.height}, config renders {cfg.width}x{cfg.height}")
        step = cfg.render_stride * (rows_per_step if rows_per_step else 1)
        ds = self.device_scene(scene)
        p, keep = _abi.make_params(cfg, traversal=self.traversal, tuning=tuning)
        lib = _lib.load()
        h = C.c_void_p()
        _lib.check(lib.rt_render_begin(ds.handle, C.byref(p), buffer.buffer.ctypes.data, int(step), C.byref(h)))
        rows, fin, reported = C.c_uint32(0), C.c_int(0), 0
        n_bands = -(-cfg.height // step)
        try:
            while True:
                _lib.check(lib.rt_render_poll(h, C.byref(rows), C.byref(fin)))
                while reported < n_bands and min((reported + 1) * step, cfg.height) <= rows.value:
                    if on_tiles is not None:
                        on_tiles(buffer, (0, reported * step, cfg.width, min(step, cfg.height - reported * step)))
                    reported += 1
                if fin.value:
                    break
                time.sleep(poll_s)
        finally:
            st = _abi.rt_stats()
            rc = lib.rt_render_end(h, C.byref(st))
        _lib.check(rc)
        self.last_stats = st.as_dict()
        while reported < n_bands:  # (bands that landed between the last poll and the end)
            if on_tiles is not None:
                on_tiles(buffer, (0, reported * step, cfg.width, min(step, cfg.height - reported * step)))
            reported += 1
        return n_bands
