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

    def __init__(self, flat: FlatScene, device: int = 0, bvh: Optional[Dict] = None):
        lib = _lib.load()
        desc, keep = _abi.make_scene_desc(flat, bvh)
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

    def __init__(self, cfg: RenderConfig, device: int = 0, traversal: int = _abi.RT_TRAVERSAL_BVH):
        self.cfg = cfg
        self.device = int(device)
        self.traversal = int(traversal)
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
        ds = DeviceScene(flat, self.device)
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

    def render_progressive(self, buffer: ImageBuffer, scene, on_tiles=None, rows_per_step: Optional[int] = None):
        """Progressive display (reference `Renderer::render` fills the shared u32 buffer tile by tile while the
        window shows it, renderer/mod.rs:84-209, output/window.rs): the frame is rendered in horizontal bands of
        `rows_per_step` tile rows (default: one row of RENDER_STRIDE tiles) and `on_tiles(buffer, (x0, y0, w, h))`
        is called after each band has landed in `buffer`.  The result equals one `render` call (the window
        parameter of the C ABI renders exactly the pixels inside it).  Returns the number of bands."""
        cfg = self.cfg
        ts = cfg.render_stride
        step = ts * (rows_per_step if rows_per_step else 1)
        n = 0
        for y0 in range(0, cfg.height, step):
            win = (0, y0, cfg.width, min(step, cfg.height - y0))
            self.render(buffer, scene, window=win)
            n += 1
            if on_tiles is not None:
                on_tiles(buffer, win)
        return n

