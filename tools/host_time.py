#!/usr/bin/env python3
"""One-GPU rehearsal of the tile-parallel split: GPU time per frame of ONE rank's share of a workload (n_ranks, rank:
the tiles rt_tile_owner gives it), frames enqueued back to back through rt_render_device --

  * on one stream (a launch cannot end before its longest wavefront does: every frame pays its own drain), and
  * on two streams used alternately (two frames in flight: the head of frame k+1 fills the CUs frame k's drain leaves idle),

in row-major and in cost order (rt_tuning.tile_order).  The per-rank figure bounds what an N-GPU run can reach before
the gather: speed-up <= whole-frame ms / slowest rank's ms.

Usage: host_time.py [workload] [--frames K] [--ranks 1,2,4,8]"""
import argparse
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from hslu_i.ba_raytracing.f2501_raytracer_amd import _abi, _lib
from hslu_i.ba_raytracing.f2501_raytracer_amd.renderer import DeviceScene

ap = argparse.ArgumentParser()
ap.add_argument("workload", nargs="?", default="c3")
ap.add_argument("--frames", type=int, default=60)
ap.add_argument("--ranks", default="1,2,4,8")
ap.add_argument("--orders", default="1,2")
ap.add_argument("--streams", default="1,2", help="numbers of streams (frames in flight) to time")
ap.add_argument("--tuning", default="", help="rt_tuning fields, k=v,k=v (A/B experiments)")
args = ap.parse_args()

lib = _lib.load()
dev = torch.device("cuda", 0)
cfg, flat, _ = bench.build_workload(args.workload)
ds = DeviceScene(flat, 0, budget=int(os.environ.get('RT_AB_BUDGET_MB', '2048')) << 20)  # (like bench.py: the per-cell lists are opted into)
NS = [int(v) for v in args.streams.split(",")]
TUNING = {k: int(v, 0) for k, v in (kv.split("=") for kv in args.tuning.split(",") if kv)}
fbs = [torch.zeros(cfg.width * cfg.height, dtype=torch.int32, device=dev) for _ in range(max(NS))]
streams = [torch.cuda.Stream(device=dev) for _ in range(max(NS))]
K = args.frames
ref = None
whole = {}
for n_ranks in (int(v) for v in args.ranks.split(",")):
    for order in (int(v) for v in args.orders.split(",")):
        rows = []
        for rank in range(n_ranks):
            p, keep = _abi.make_params(cfg, n_ranks=n_ranks, rank=rank, tuning=dict(tile_order=order, **TUNING))
            res = []
            for n_streams in NS:
                for i in range(4):  # warm-up (the first frame of a cost-ordered shape is the calibration frame)
                    _lib.check(lib.rt_render_device(ds.handle, C.byref(p), C.c_void_p(fbs[i % n_streams].data_ptr()), None,
                                                    C.c_void_p(streams[i % n_streams].cuda_stream)))
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(K):
                    _lib.check(lib.rt_render_device(ds.handle, C.byref(p), C.c_void_p(fbs[i % n_streams].data_ptr()), None,
                                                    C.c_void_p(streams[i % n_streams].cuda_stream)))
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                res.append((1e3 * (t2 - t0) / K, 1e3 * (t1 - t0) / K))
            if n_ranks == 1:
                img = fbs[0].cpu().numpy().view(np.uint32)
                if ref is None:
                    ref = img.copy()
                assert all(np.array_equal(fb.cpu().numpy().view(np.uint32), ref) for fb in fbs[:NS[-1]]), "image changed"
            rows.append((rank, res))
        tag = {1: "row-major", 2: "cost order"}[order]
        for rank, res in rows:
            print(f"ranks {n_ranks} rank {rank} {tag:10s}: " + ", ".join(f"{n} stream(s) {r[0]:.3f} ms/frame" for n, r in zip(NS, res))
                  + f" (host enqueue {res[-1][1]:.3f})")
        slow = [max(r[1][i][0] for r in rows) for i in range(len(NS))]
        if n_ranks == 1:
            whole[order] = slow
        w = whole.get(order, whole.get(1, slow))
        print(f"== {n_ranks} ranks, {tag}: slowest rank " + ", ".join(f"{sl:.3f} ms ({n} stream(s)) -> {wi / sl:.2f}x" for n, sl, wi in zip(NS, slow, w))
              + " of the whole frame's " + " / ".join(f"{wi:.3f}" for wi in w) + " ms")
ds.close()
