"""Tile-partitioned multi-GPU rendering: one process per GPU, no data-path collective except one
gather of the packed pixels to rank 0 (RCCL over xGMI on GPUs; gloo on CPU for tests).

Reference analogue: `ImageBuffer::process_chunks_parallel` hands shuffled RENDER_STRIDE tiles to
rayon workers (reference `src/image_buffer.rs:48-97`, `src/renderer/mod.rs:84-90`).  Here a tile's
owner is `rt_tile_owner(tx, ty, n_ranks)` (include/rt_hip.h) -- a fixed lattice interleave instead
of a shuffle, so that cost hot-spots (glass sphere vs background) spread evenly over the ranks.

xGMI is point-to-point, so every peer has its own link into rank 0: the gather is (n-1) concurrent
peer->root transfers of <= W*H*4/n bytes each, not a ring.
"""
from __future__ import annotations

from typing import List

import numpy as np

from .config import RenderConfig


def tile_stride(n_ranks: int) -> int:
    """include/rt_hip.h rt_tile_owner: smallest odd S >= 3 coprime to n_ranks."""
    import math

    s = 3
    while math.gcd(s, n_ranks) != 1:
        s += 2
    return s


def tile_owner_map(cfg: RenderConfig, n_ranks: int) -> np.ndarray:
    """owner[ty, tx] = (tx + S*ty) mod n_ranks (include/rt_hip.h rt_tile_owner)."""
    ts = cfg.render_stride
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


class TileGather:
    """Gathers every rank's owned pixels into rank 0's full framebuffer (int32 view of 0xAARRGGBB)."""

    def __init__(self, cfg: RenderConfig, world: int, rank: int, device, host_staging: bool = False):
        import torch

        self.world, self.rank = world, rank
        self.host_staging = host_staging  # rehearsal only: move the payload through gloo on the host
        self.use_all_gather = False      # fallback if the backend lacks gather (decided on the first call)
        self.all_buf = None
        self.calls = 0
        self.gather_error = None
        self.device = device
        comm_device = torch.device("cpu") if host_staging else device
        idx = [owned_pixel_indices(cfg, world, r) for r in range(world)]
        self.counts = [int(i.shape[0]) for i in idx]
        self.max_len = max(self.counts)
        self.own_idx = torch.from_numpy(idx[rank]).to(device)
        self.send = torch.zeros(self.max_len, dtype=torch.int32, device=device)
        self.recv: List = []
        self.all_idx: List = []
        if rank == 0:
            self.recv = [torch.zeros(self.max_len, dtype=torch.int32, device=comm_device) for _ in range(world)]
            self.all_idx = [torch.from_numpy(i).to(device) for i in idx]

    def run(self, fb, stream=None):
        import torch
        import torch.distributed as dist

        n = self.counts[self.rank]
        torch.index_select(fb, 0, self.own_idx, out=self.send[:n])
        send = self.send.cpu() if self.host_staging else self.send
        if not self.use_all_gather:
            try:
                dist.gather(send, self.recv if self.rank == 0 else None, dst=0)
            except (RuntimeError, NotImplementedError) as e:  # backend without gather: decided once, at set-up
                if self.calls > 0:
                    raise
                self.use_all_gather = True
                self.gather_error = repr(e)
        if self.use_all_gather:
            if self.all_buf is None:
                self.all_buf = torch.zeros(self.world * self.max_len, dtype=torch.int32, device=send.device)
            dist.all_gather_into_tensor(self.all_buf, send)
            if self.rank == 0:
                for r in range(1, self.world):
                    seg = self.all_buf[r * self.max_len: r * self.max_len + self.counts[r]]
                    fb.index_copy_(0, self.all_idx[r], seg.to(self.device))
            self.calls += 1
            return fb
        self.calls += 1
        if self.rank == 0:
            for r in range(1, self.world):
                fb.index_copy_(0, self.all_idx[r], self.recv[r][: self.counts[r]].to(self.device))
        return fb
