"""The N > 1 path on CPU: world_size 2 over gloo.  Each rank "renders" its own tiles (with the CPU
oracle, tile-ownership arguments exactly as the GPU ranks pass them) into a full-size buffer, then
TileGather moves the owned pixels to rank 0, which must end up with the full frame."""
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


def _worker(rank, world, port, out_path, force_all_gather=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    import oracle_lib
    from hslu_i.ba_raytracing.f2501_raytracer_amd import RenderConfig, scenes
    from hslu_i.ba_raytracing.f2501_raytracer_amd.distributed import TileGather

    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = RenderConfig.from_features([], width_override=200, height_override=150)
    flat = scenes.test_scene(cfg).flatten()
    part, _, _ = oracle_lib.render(flat, cfg, n_ranks=world, rank=rank, n_threads=2)
    fb = torch.from_numpy(part.view(np.int32).copy())
    g = TileGather(cfg, world, rank, torch.device("cpu"))
    g.use_all_gather = force_all_gather
    g.run(fb)
    if rank == 0:
        full, _, _ = oracle_lib.render(flat, cfg, n_threads=2)
        np.save(out_path, np.stack([fb.numpy().view(np.uint32), full]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_tile_gather_world_size_n(tmp_path, world):
    import torch.multiprocessing as mp

    out = str(tmp_path / "res.npy")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got, full = np.load(out)
    assert (full != 0).sum() > 1000
    assert np.array_equal(got, full)


def test_tile_gather_all_gather_fallback(tmp_path):
    import torch.multiprocessing as mp

    out = str(tmp_path / "res.npy")
    mp.spawn(_worker, args=(2, _free_port(), out, True), nprocs=2, join=True)
    got, full = np.load(out)
    assert np.array_equal(got, full)
