import ctypes as C, os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
from hslu_i.ba_raytracing.f2501_raytracer_amd import _abi, _lib
from hslu_i.ba_raytracing.f2501_raytracer_amd.renderer import DeviceScene
lib = _lib.load(); dev = torch.device('cuda', 0)
cfg, flat, _ = bench.build_workload('c3')
ds = DeviceScene(flat, 0)
fb = torch.zeros(cfg.width * cfg.height, dtype=torch.int32, device=dev)
for n_ranks, rank in ((1, 0), (8, 3), (8, 0), (4, 1), (2, 1)):
    p, keep = _abi.make_params(cfg, n_ranks=n_ranks, rank=rank)
    for _ in range(3):
        _lib.check(lib.rt_render_device(ds.handle, C.byref(p), C.c_void_p(fb.data_ptr()), None, None))
    torch.cuda.synchronize()
    K = 50
    t0 = time.perf_counter(); host = 0.0
    for _ in range(K):
        h0 = time.perf_counter()
        _lib.check(lib.rt_render_device(ds.handle, C.byref(p), C.c_void_p(fb.data_ptr()), None, None))
        host += time.perf_counter() - h0
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"ranks {n_ranks} rank {rank}: pipelined {1e3*(t2-t0)/K:.3f} ms/frame, host enqueue {1e3*host/K:.3f} ms/frame")
