"""The N > 1 path on CPU: world_size 2 and 3 over gloo.  Each rank "renders" its own tiles (with the CPU oracle,
tile-ownership arguments exactly as the GPU ranks pass them), packs them into the rank-compact staging layout of the
C ABI (rt_gather_layout -- what the kernel writes directly on the GPU), the staging buffers are gathered to rank 0,
and rank 0 scatters them into the frame, which must equal the single-rank frame."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path, tile_size):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dataclasses

    import torch.distributed as dist

    import oracle_lib
    from hslu_i.ba_raytracing.f2501_raytracer_amd import RenderConfig, scenes
    from hslu_i.ba_raytracing.f2501_raytracer_amd.distributed import HostGather

    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = RenderConfig.from_features([], width_override=200, height_override=150)
    flat = scenes.test_scene(cfg).flatten()
    fill = 0x00123456  # the caller's fill must survive on miss pixels (alpha 0: never produced by a hit)
    part, _, _ = oracle_lib.render(flat, cfg, n_ranks=world, rank=rank, n_threads=2)
    g = HostGather(cfg, world, rank)
    if rank == 0:
        fb = np.where(part != 0, part, fill).astype(np.uint32)
    else:
        fb = part
    g.run(fb)
    if rank == 0:
        full, _, _ = oracle_lib.render(flat, cfg, n_threads=2)
        np.save(out_path, np.stack([fb, np.where(full != 0, full, fill).astype(np.uint32)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_compact_staging_gather_world_size_n(tmp_path, world):
    import torch.multiprocessing as mp

    out = str(tmp_path / "res.npy")
    mp.spawn(_worker, args=(world, _free_port(), out, None), nprocs=world, join=True)
    got, full = np.load(out)
    assert (full != 0x00123456).sum() > 1000
    assert np.array_equal(got, full)


@pytest.mark.parametrize("shape", [(200, 150, 48), (1620, 1350, 48), (3840, 2160, 48), (97, 33, 16), (48, 48, 48)])
@pytest.mark.parametrize("n_ranks", [1, 2, 3, 8])
def test_gather_layout_is_a_partition(shape, n_ranks):
    """rt_gather_layout (C ABI, host only): every tile has exactly one (owner, slot), slots of a rank are 0..count-1 in
    row-major tile order, and the per-pixel staging positions of a rank are distinct and inside its buffer."""
    from hslu_i.ba_raytracing.f2501_raytracer_amd import RenderConfig
    from hslu_i.ba_raytracing.f2501_raytracer_amd.distributed import gather_layout, staging_index, tile_owner_map

    w, h, ts = shape
    cfg = RenderConfig.from_features([], width_override=w, height_override=h)
    slot, count = gather_layout(cfg, n_ranks, ts)
    owners = tile_owner_map(cfg, n_ranks, ts)
    assert int(count.sum()) == owners.size
    for r in range(n_ranks):
        s = slot[owners == r]  # boolean indexing is row-major
        assert np.array_equal(s, np.arange(int(count[r])))
    if w * h <= 200 * 150:
        owner, pos, sizes = staging_index(cfg, n_ranks, ts)
        for r in range(n_ranks):
            p = pos[owner == r]
            assert len(np.unique(p)) == len(p) and (p.size == 0 or p.max() < sizes[r])
    # balance: the lattice interleave gives every rank the same number of tiles +- one per tile row
    assert int(count.max()) - int(count.min()) <= owners.shape[0]
