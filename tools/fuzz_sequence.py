#!/usr/bin/env python3
"""State fuzz of the frame scheduler: ONE scene handle, a random sequence of frames of different shapes (features, depths,
windows, tile partitions, rt_tuning) enqueued on random streams with random synchronisation points in between -- frame
slots, workspace sets, chains, queue sizing / verification, table uploads and the frames-in-flight guards all see
transitions no test scripts by hand.  Every frame buffer must equal the buffer the same parameters give on a fresh scene
handle, rendered alone.  Usage: fuzz_sequence.py first last"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import test_parity_gpu as T  # noqa: E402
from hslu_i.ba_raytracing.f2501_raytracer_amd import RenderConfig, _abi, _lib  # noqa: E402
from hslu_i.ba_raytracing.f2501_raytracer_amd.renderer import DeviceScene  # noqa: E402



class Hip:
    """Device buffers and streams straight from the HIP runtime librt_hip.so is linked against (dlsym on the library's handle
    searches its dependency tree; importing torch AFTER the library would map a second, mismatching runtime)."""

    def __init__(self, lib):
        self.h = lib

    def ok(self, rc):
        assert rc == 0, f"HIP error {rc}"

    def zeros(self, n_words):
        p = C.c_void_p()
        self.ok(self.h.hipMalloc(C.byref(p), C.c_size_t(n_words * 4)))
        self.ok(self.h.hipMemset(p, 0, C.c_size_t(n_words * 4)))
        self.ok(self.h.hipDeviceSynchronize())
        return p

    def fetch(self, p, n_words):
        out = np.zeros(n_words, np.int32)
        self.ok(self.h.hipMemcpy(C.c_void_p(out.ctypes.data), p, C.c_size_t(n_words * 4), 2))
        return out

    def free(self, p):
        self.ok(self.h.hipFree(p))

    def stream(self):
        sp = C.c_void_p()
        self.ok(self.h.hipStreamCreateWithFlags(C.byref(sp), 1))  # hipStreamNonBlocking
        return sp

    def sync(self, stream=None):
        self.ok(self.h.hipDeviceSynchronize() if stream is None else self.h.hipStreamSynchronize(stream))


def run(first, last):
    """Seeds first..last; returns the number of failures."""
    lib = _lib.load()
    hip = Hip(lib)
    FEATS = [["realistic", "anti_aliasing", "soft_shadows"], ["anti_aliasing", "high_quality"], ["soft_shadows", "reflections"], ["realistic"],
             ["anti_aliasing", "soft_shadows", "refractions"], [], ["anti_aliasing"], ["realistic", "soft_shadows"]]
    bad = 0
    for seed in range(first, last + 1):
        r = np.random.default_rng(9000 + seed)
        W, H = int(r.integers(64, 200)), int(r.integers(48, 160))
        base = RenderConfig.from_features(["realistic"], width_override=W, height_override=H, cloud_seed=seed)
        flat = T.random_scene(seed, n_spheres=int(r.integers(1, 20)), n_tris=int(r.integers(1, 600)), n_lights=int(r.integers(1, 5)), cfg=base)
        # the parameter sets of this sequence (same frame size: the buffers are W x H)
        sets = []
        for k in range(int(r.integers(3, 7))):
            feats = FEATS[int(r.integers(0, len(FEATS)))]
            secondary = any(f in feats for f in ("realistic", "reflections", "refractions"))
            cfg = RenderConfig.from_features(feats, width_override=W, height_override=H, n_cloud_sets=int(r.integers(4, 17)),
                                             depth_override=int(r.integers(1, 6)) if secondary else None, cloud_seed=seed + 17 * k)
            win = None
            if r.random() < 0.5:
                ww, wh = int(r.integers(8, W + 1)), int(r.integers(8, H + 1))
                win = (int(r.integers(0, W - ww + 1)), int(r.integers(0, H - wh + 1)), ww, wh)
            n_ranks = int(r.choice([1, 1, 2, 3]))
            rank = int(r.integers(0, n_ranks))
            tuning = {}
            if r.random() < 0.4:
                tuning["sub_frames"] = int(r.integers(1, 3))
            if secondary and r.random() < 0.25:
                tuning["chunk_log2"] = int(r.integers(10, 15))
            if r.random() < 0.2:
                tuning["tile_order"] = 2
            if r.random() < 0.2:
                tuning["no_cell_lists"] = 1
            if r.random() < 0.25:
                tuning["phases"] = _abi.RT_PHASES_SPLIT
            if secondary and r.random() < 0.5:
                tuning["levels"] = int(r.integers(1, 4))
            p, keep = _abi.make_params(cfg, window=win, n_ranks=n_ranks, rank=rank, tuning=tuning)
            # reference: a fresh scene handle, this frame alone
            ds0 = DeviceScene(flat, 0)
            fb = hip.zeros(W * H)
            _lib.check(lib.rt_render_device(ds0.handle, C.byref(p), fb, None, None))
            hip.sync()
            st = _abi.rt_stats()
            _lib.check(lib.rt_render_collect_stats(ds0.handle, C.byref(st)))
            ds0.close()
            ref = hip.fetch(fb, W * H)
            hip.free(fb)
            sets.append(dict(p=p, keep=keep, ref=ref, rays=(st.rays_primary, st.rays_reflection, st.rays_refraction, st.rays_shadow),
                             what=f"{feats} win {win} ranks {n_ranks}/{rank} {tuning}"))
        ds = DeviceScene(flat, 0)
        streams = [None] + [hip.stream() for _ in range(3)]
        pending = []  # (set index, buffer)
        n_frames = int(r.integers(20, 60))
        log = []
        for f in range(n_frames):
            i = int(r.integers(0, len(sets)))
            if r.random() < 0.5 and log:
                i = log[-1][0]  # runs of the same shape (the verified, unsynchronised path) are the common case
            j = int(r.integers(0, len(streams)))
            fb = hip.zeros(W * H)  # (synchronises: the zero fill is complete)
            _lib.check(lib.rt_render_device(ds.handle, C.byref(sets[i]["p"]), fb, None, streams[j]))
            pending.append((i, fb))
            log.append((i, j))
            u = r.random()
            if u < 0.15:
                hip.sync()
                st = _abi.rt_stats()
                _lib.check(lib.rt_render_collect_stats(ds.handle, C.byref(st)))
                got = (st.rays_primary, st.rays_reflection, st.rays_refraction, st.rays_shadow)
                if got != sets[i]["rays"]:
                    bad += 1
                    print(f"seed {seed} frame {f}: counters {got} != {sets[i]['rays']} for {sets[i]['what']}; sequence {log}")
            elif u < 0.25 and streams[j] is not None:
                hip.sync(streams[j])
        hip.sync()
        for f, (i, fb) in enumerate(pending):
            got = hip.fetch(fb, W * H)
            if not np.array_equal(got, sets[i]["ref"]):
                bad += 1
                print(f"seed {seed} frame {f}: {int((got != sets[i]['ref']).sum())} pixels differ for {sets[i]['what']}; sequence {log}")
                break
        for _, fb in pending:
            hip.free(fb)
        ds.close()
        for sp in streams[1:]:
            hip.ok(lib.hipStreamDestroy(sp))
        print(f"seed {seed}: {n_frames} frames of {len(sets)} shapes " + ("ok" if not bad else f"({bad} failures so far)"))
    print(f"{bad} failures in seeds {first}..{last}")
    return bad


if __name__ == "__main__":
    sys.stdout.reconfigure(line_buffering=True)
    sys.exit(1 if run(int(sys.argv[1]), int(sys.argv[2])) else 0)
