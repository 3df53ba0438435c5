#!/usr/bin/env python3
"""Generates the AT-SPEC golden fixtures tests/golden/spec_*.npz: every BASELINE config exactly as
`bench.build_workload` builds it (what bench.py times), on windows of the frame that cover what the path is made
of -- the glass sphere's rim with the text behind it, text silhouettes and their penumbrae on the wall, the pile of
metallic-glass spheres (deep ray trees), a penumbra on the floor.

The brute-force scalar oracle would need hours for these windows (14 578 objects per ray, up to 140 shadow rays per hit,
ray trees of depth 8 / 21), so they are rendered OFFLINE, in the build container, with the threaded 8-lane packet
restatement (oracle/rt_simd_baseline.c), which is bit-identical to the scalar oracle (ids, t, RGB:
tests/test_simd_baseline.py; tests/test_oracle_golden.py re-renders a corner of every fixture with the scalar oracle).
Like tests/golden/*.npz these are outputs of this repo's CPU restatement, not of the reference (Rust nightly, not
buildable here: SURVEY F2/F3): they pin regressions and give the GPU a committed per-pixel expectation at spec.

    python make_spec_golden.py [case ...]        (about 40 minutes on 8 cores for all of them)
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np

# workload key of bench.WORKLOADS -> windows (x0, y0, w, h) of the frame
SPEC_CASES = {
    # BASELINE configs[2]: 1620x1350, 16 rays/px, 5 x 10 shadow rays per hit, text.obj
    "spec_c3": dict(workload="c3", windows=[(440, 500, 64, 48),    # the glass sphere's rim, "HS" behind it
                                            (380, 360, 64, 48),    # text silhouette ("KI") and its shadow on the wall
                                            (560, 1270, 64, 48),   # penumbra on the floor
                                            (200, 560, 64, 48)]),  # the text's penumbrae on the left wall
    # ... with the mesh the reference's own feature set would load (text_lowres.obj)
    "spec_c3lowres": dict(workload="c3lowres", windows=[(440, 500, 64, 48), (380, 360, 64, 48), (560, 1270, 64, 48),
                                                        (200, 560, 64, 48)]),
    # BASELINE configs[3]: + reflections / refractions, 24 rays/px, 5 x 28 shadow rays per hit, depth 8
    "spec_c4": dict(workload="c4", windows=[(440, 500, 48, 32),     # sphere rim: refraction + reflection of the text
                                            (380, 360, 48, 32),     # text
                                            (1236, 984, 16, 12),    # pile of metallic-glass spheres
                                            (560, 1270, 48, 32)]),  # floor penumbra through the glass slab
    # ... at the depth the reference's `realistic + extreme_quality` means: 21 / 21
    "spec_c4d21": dict(workload="c4d21", windows=[(440, 500, 48, 32), (380, 360, 48, 32), (1240, 988, 8, 6),
                                                  (560, 1270, 48, 32)]),
    # BASELINE configs[4]: as configs[3] at 3840x2160 (the scene changes with the aspect ratio)
    "spec_c5": dict(workload="c5", windows=[(1160, 880, 48, 32), (900, 620, 48, 32), (2952, 1580, 16, 12),
                                            (1750, 2030, 48, 32)]),
}
STAT_KEYS = ("rays_primary", "rays_reflection", "rays_refraction", "rays_shadow", "pixels_written")


def crop(cfg, win, a):
    x0, y0, w, h = win
    return a.reshape((cfg.height, cfg.width) + a.shape[1:])[y0:y0 + h, x0:x0 + w].copy()


def load(name):
    z = np.load(os.path.join(HERE, name + ".npz"))
    return json.loads(str(z["meta"])), z


if __name__ == "__main__":
    import bench
    import oracle_lib

    for name in (sys.argv[1:] or list(SPEC_CASES)):
        case = SPEC_CASES[name]
        cfg, flat, wname = bench.build_workload(case["workload"])
        arrays, stats = {}, []
        for i, win in enumerate(case["windows"]):
            t = time.time()
            argb, planes, st = oracle_lib.render(flat, cfg, window=win, impl="simd", n_threads=oracle_lib.host_cores())
            arrays[f"w{i}_argb"] = crop(cfg, win, argb)
            arrays[f"w{i}_hit_id"] = crop(cfg, win, planes["hit_id"])
            arrays[f"w{i}_hit_t"] = crop(cfg, win, planes["hit_t"])
            arrays[f"w{i}_rgb"] = crop(cfg, win, planes["rgb"])
            stats.append({k: int(st[k]) for k in STAT_KEYS})
            print(name, win, f"{time.time() - t:.0f} s", stats[-1], flush=True)
        meta = dict(workload=case["workload"], workload_name=wname, windows=[list(w) for w in case["windows"]], stats=stats,
                    width=cfg.width, height=cfg.height, generator="oracle/rt_simd_baseline.c (bit-identical to oracle/rt_oracle.c)")
        np.savez_compressed(os.path.join(HERE, name + ".npz"), meta=json.dumps(meta), **arrays)
