#!/usr/bin/env python3
"""Invariance fuzz: the image and the ray counters must not depend on any execution knob.  For random scenes (the second
generator of fuzz_sweep.py) the frame rendered with default settings is compared, packed pixel by packed pixel and float
plane by float plane (bit patterns), with the same frame rendered under random settings of rt_tuning -- chains, per-cell
lists, receiver flags, candidate cap, forced batch sizes, sort bits, launch order, AA de-duplication -- and with the union of
the tiles of 2..5 ranks at random tile sizes.  No oracle involved (fuzz_sweep.py compares with the oracle): GPU against GPU.
Usage: fuzz_knobs.py first last"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import test_parity_gpu as T  # noqa: E402
from hslu_i.ba_raytracing.f2501_raytracer_amd import RenderConfig, _abi  # noqa: E402



def run(first, last):
    """Seeds first..last; returns the number of failures."""
    COUNTERS = ("rays_primary", "rays_reflection", "rays_refraction", "rays_shadow", "pixels_written")
    bad = [0]
    for seed in range(first, last + 1):
        r = np.random.default_rng(5000 + seed)
        W, H = int(r.integers(48, 260)), int(r.integers(32, 200))
        H = max(H, W // 5 + 1)
        feats = [["realistic", "anti_aliasing", "soft_shadows"], ["anti_aliasing", "high_quality"], ["soft_shadows", "reflections"],
                 ["realistic"], ["anti_aliasing", "soft_shadows", "refractions"], ["realistic", "soft_shadows"]][seed % 6]
        secondary = any(f in feats for f in ("realistic", "reflections", "refractions"))
        cfg = RenderConfig.from_features(feats, width_override=W, height_override=H, n_cloud_sets=int(r.integers(4, 33)),
                                         depth_override=int(r.integers(1, 7)) if secondary else None, cloud_seed=seed)
        n_s = 0 if seed % 9 == 0 else int(r.integers(1, 30))
        n_t = 0 if seed % 13 == 0 else int(r.integers(1, 2500 if seed % 3 == 0 else 300))
        if n_s + n_t == 0:
            n_s = 2
        flat = T.random_scene(seed, n_spheres=n_s, n_tris=n_t, n_lights=int(r.integers(1, 6)), cfg=cfg)
        win = None
        if seed % 2:
            ww, wh = int(r.integers(8, W + 1)), int(r.integers(8, H + 1))
            win = (int(r.integers(0, W - ww + 1)), int(r.integers(0, H - wh + 1)), ww, wh)
        ref, pref, sref = T.gpu_render(cfg, flat, win)
        what = f"seed {seed} {W}x{H} {feats} spheres {n_s} tris {n_t} window {win}"

        def same(tag, a, p, s, counters=True):
            ok = np.array_equal(a, ref) and np.array_equal(p["rgb"].view(np.uint32), pref["rgb"].view(np.uint32)) and \
                np.array_equal(p["hit_id"], pref["hit_id"]) and (not counters or all(s[k] == sref[k] for k in COUNTERS))
            if not ok:
                bad[0] += 1
                print(f"{what}: DIFFERS under {tag}: {int((a != ref).sum())} packed pixels, counters "
                      f"{[(k, s[k], sref[k]) for k in COUNTERS if s[k] != sref[k]]}")
            return ok

        for trial in range(3):
            knobs = {}
            if r.random() < 0.5:
                knobs["sub_frames"] = int(r.integers(1, 3))
            if r.random() < 0.3:
                knobs["no_cell_lists"] = 1
            if r.random() < 0.25:
                knobs["no_receiver_flags"] = 1
            if r.random() < 0.35:
                knobs["shadow_candidate_cap"] = int(r.choice([1, 2, 3, 5, 8, 17, 33, 64, _abi.RT_CAND_CAP_NONE]))
            if secondary and r.random() < 0.35:
                knobs["chunk_log2"] = int(r.integers(10, 16))
            if r.random() < 0.4:
                knobs["sort_bits"] = int(r.integers(12, 25))
            if r.random() < 0.3:
                knobs["tile_order"] = int(r.integers(1, 3))
            if r.random() < 0.25:
                knobs["no_aa_dedup"] = 1
            if secondary and r.random() < 0.3:  # (the phase-split pipeline adds the same fixed-point terms: frames with secondary rays are the same bits)
                knobs["phases"] = _abi.RT_PHASES_SPLIT
            if secondary and r.random() < 0.5:  # how the levels of the ray tree are scheduled
                knobs["levels"] = int(r.integers(1, 4))
            if os.environ.get("FUZZ_VERBOSE"):
                print(f"{what}: {knobs}")
            a, p, s = T.gpu_render(cfg, flat, win, **knobs)
            same(f"tuning {knobs}", a, p, s)
        n_ranks = int(r.integers(2, 6))
        acc = np.zeros_like(ref)
        planes = {"rgb": np.zeros_like(pref["rgb"]), "hit_id": np.full_like(pref["hit_id"], 0)}
        tot = {k: 0 for k in COUNTERS}
        r_ok = True
        rk = {"sub_frames": int(r.integers(0, 3))}
        for rank in range(n_ranks):
            part, pp, sp = T.gpu_render(cfg, flat, win, n_ranks=n_ranks, rank=rank, **rk)
            if ((acc != 0) & (part != 0)).any():
                r_ok = False
            acc |= part
            for k in COUNTERS:
                tot[k] += sp[k]
        if not (r_ok and np.array_equal(acc, ref) and all(tot[k] == sref[k] for k in COUNTERS)):
            bad[0] += 1
            print(f"{what}: union of {n_ranks} ranks DIFFERS: {int((acc != ref).sum())} packed pixels, counters "
                  f"{[(k, tot[k], sref[k]) for k in COUNTERS if tot[k] != sref[k]]}")
        # the same partition through rt_render_multi (rank-compact staging written by the kernels, gather, scatter), at a random tile size
        ts = int(r.choice([16, 32, 48, 64]))
        got, _ = T.render_multi(cfg, flat, n_ranks, window=win, tile_size=ts)
        if not np.array_equal(got, ref):
            bad[0] += 1
            print(f"{what}: rt_render_multi with {n_ranks} ranks, tile size {ts} DIFFERS: {int((got != ref).sum())} packed pixels")
        print(f"seed {seed} ok" if not bad[0] else f"seed {seed} done ({bad[0]} failures so far)")
    print(f"{bad[0]} failures in seeds {first}..{last}")
    return bad[0]


if __name__ == "__main__":
    sys.stdout.reconfigure(line_buffering=True)
    sys.exit(1 if run(int(sys.argv[1]), int(sys.argv[2])) else 0)
