"""Tile-partitioned multi-GPU rendering: one process per GPU, no data-path collective except ONE gather of the
packed pixels to rank 0.

Reference analogue: `ImageBuffer::process_chunks_parallel` hands shuffled RENDER_STRIDE tiles to rayon workers that
all write one buffer (reference `src/image_buffer.rs:48-97`, `src/renderer/mod.rs:84-90`).  Here a tile's owner is
`rt_tile_owner(tx, ty, n_ranks)` (include/rt_hip.h) -- a fixed lattice interleave instead of a shuffle, so that cost
hot-spots (glass sphere vs background) spread evenly over the ranks -- and every rank renders its tiles into a
rank-compact staging buffer whose layout `rt_gather_layout` (C ABI, host only) defines.

* `RcclGather`: the product path.  Thin wrapper over the C ABI's `rt_comm_*` / `rt_render_gather_device`
  (csrc/rt_multi.cpp): ncclGroupStart / ncclRecv x (n-1) on the root, ncclSend on the peers / ncclGroupEnd, then a
  scatter kernel.  xGMI is point to point, so every peer has its own link into rank 0: (n-1) concurrent transfers of
  <= W*H*4/n bytes, not a ring.  torch.distributed only carries the 128-byte communicator id.
* `HostGather`: the same staging layout moved through gloo on the host -- CPU tests of the N > 1 path and the
  `--backend gloo` rehearsal of bench.py on a one-GPU box.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Tuple

import numpy as np

from .config import RenderConfig


def tile_stride(n_ranks: int) -> int:
    """include/rt_hip.h rt_tile_owner: smallest odd S >= 3 coprime to n_ranks."""
    import math

    s = 3
    while math.gcd(s, n_ranks) != 1:
        s += 2
    return s


def tile_owner_map(cfg: RenderConfig, n_ranks: int, tile_size: Optional[int] = None) -> np.ndarray:
    """owner[ty, tx] = (tx + S*ty) mod n_ranks (include/rt_hip.h rt_tile_owner)."""
    ts = tile_size or cfg.render_stride
    tx, ty = -(-cfg.width // ts), -(-cfg.height // ts)
    if n_ranks <= 1:
        return np.zeros((ty, tx), np.uint32)
    s = tile_stride(n_ranks)
    yy, xx = np.meshgrid(np.arange(ty), np.arange(tx), indexing="ij")
    return ((xx + s * yy) % n_ranks).astype(np.uint32)


def owned_pixel_indices(cfg: RenderConfig, n_ranks: int, rank: int) -> np.ndarray:
    """Flat row-major pixel indices of the tiles `rank` owns (sorted)."""
    ts = cfg.render_stride
    owners = tile_owner_map(cfg, n_ranks)
    gy, gx = np.meshgrid(np.arange(cfg.height), np.arange(cfg.width), indexing="ij")
    own = owners[gy // ts, gx // ts] == rank
    return np.flatnonzero(own.ravel()).astype(np.int64)


def gather_layout(cfg: RenderConfig, n_ranks: int, tile_size: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray]:
    """(tile_slot[ty, tx], tiles_per_rank[r]) from the C ABI's rt_gather_layout (no GPU needed)."""
    from . import _lib

    ts = tile_size or cfg.render_stride
    tx, ty = -(-cfg.width // ts), -(-cfg.height // ts)
    slot = np.zeros((ty, tx), np.uint32)
    count = np.zeros((max(n_ranks, 1),), np.uint32)
    u32p = C.POINTER(C.c_uint32)
    _lib.check(_lib.load().rt_gather_layout(cfg.width, cfg.height, ts, n_ranks, slot.ctypes.data_as(u32p), count.ctypes.data_as(u32p)))
    return slot, count


def staging_index(cfg: RenderConfig, n_ranks: int, tile_size: Optional[int] = None):
    """Per frame pixel: (owner rank, position inside the owner's staging buffer) -- the host mirror of the kernel's
    `out_index` and of `rt_scatter_kernel` (csrc/rt_kernels.hip, rt_gather.hip)."""
    ts = tile_size or cfg.render_stride
    slot, count = gather_layout(cfg, n_ranks, ts)
    owners = tile_owner_map(cfg, n_ranks, ts)
    gy, gx = np.meshgrid(np.arange(cfg.height), np.arange(cfg.width), indexing="ij")
    ty, tx = gy // ts, gx // ts
    pos = slot[ty, tx].astype(np.int64) * (ts * ts) + (gy - ty * ts) * ts + (gx - tx * ts)
    return owners[ty, tx].ravel().astype(np.int64), pos.ravel(), count.astype(np.int64) * (ts * ts)


class HostGather:
    """The gather of the staging layout through gloo on the host (CPU tests, one-GPU rehearsal): every rank packs its
    tiles into its staging buffer, `dist.gather` moves them, rank 0 scatters the non-zero pixels into the frame."""

    def __init__(self, cfg: RenderConfig, world: int, rank: int, tile_size: Optional[int] = None):
        self.world, self.rank = world, rank
        owner, pos, sizes = staging_index(cfg, world, tile_size)
        self.sizes = [int(v) for v in sizes]
        self.max_len = max(self.sizes)
        mine = np.flatnonzero(owner == rank)
        self.own_pix, self.own_pos = mine, pos[mine]
        self.all_pix: List[np.ndarray] = []
        self.all_pos: List[np.ndarray] = []
        if rank == 0:
            for r in range(world):
                m = np.flatnonzero(owner == r)
                self.all_pix.append(m)
                self.all_pos.append(pos[m])
        self.bytes_sent = 0

    def pack(self, fb: np.ndarray) -> np.ndarray:
        """What the kernel writes directly on the GPU: this rank's tiles, staged 0 = "no hit"."""
        stage = np.zeros((self.max_len,), np.uint32)
        stage[self.own_pos] = fb[self.own_pix]
        return stage

    def run(self, fb: np.ndarray) -> np.ndarray:
        """fb: this rank's W*H uint32 frame (only its own tiles rendered).  Returns fb; complete on rank 0."""
        import torch
        import torch.distributed as dist

        send = torch.from_numpy(self.pack(fb).view(np.int32))
        recv = [torch.zeros(self.max_len, dtype=torch.int32) for _ in range(self.world)] if self.rank == 0 else None
        dist.gather(send, recv, dst=0)
        self.bytes_sent = 0 if self.rank == 0 else self.sizes[self.rank] * 4
        if self.rank == 0:
            for r in range(1, self.world):
                st = recv[r].numpy().view(np.uint32)
                v = st[self.all_pos[r]]
                hit = v != 0
                fb[self.all_pix[r][hit]] = v[hit]
        return fb


class RcclGather:
    """`rt_comm` of the C ABI: this rank's render + its part of the RCCL gather, all on one HIP stream."""

    def __init__(self, world: int, rank: int, device: int):
        import torch
        import torch.distributed as dist

        from . import _abi, _lib

        self.lib = _lib.load()
        ident = (C.c_uint8 * _abi.RT_COMM_ID_BYTES)()
        if rank == 0 and world > 1:
            _lib.check(self.lib.rt_comm_unique_id(ident))
        if world > 1:
            # the 128-byte id is the only thing torch.distributed carries
            box = [bytes(ident)]
            dist.broadcast_object_list(box, src=0)
            ident = (C.c_uint8 * _abi.RT_COMM_ID_BYTES).from_buffer_copy(box[0])
        h = C.c_void_p()
        _lib.check(self.lib.rt_comm_create(ident, world, rank, int(device), C.byref(h)))
        self._h = h
        self.world, self.rank = world, rank

    def render_gather(self, ds, params, fb_ptr: Optional[int], stream_ptr: Optional[int]):
        from . import _lib

        _lib.check(self.lib.rt_render_gather_device(ds.handle, self._h, C.byref(params),
                                                    C.c_void_p(fb_ptr) if fb_ptr else None,
                                                    C.c_void_p(stream_ptr) if stream_ptr else None))

    def last(self) -> dict:
        from . import _abi, _lib

        info = _abi.rt_gather_info()
        _lib.check(self.lib.rt_comm_last_gather(self._h, C.byref(info)))
        return info.as_dict()

    def close(self):
        if self._h is not None:
            self.lib.rt_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
