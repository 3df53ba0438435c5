"""The oracle must reproduce its committed golden fixtures (tests/golden/*.npz, made by
tests/golden/make_golden.py): hit ids / distances / pixel indices bit-exact, RGB to 2e-6 (libm
tanhf/powf may differ in the last ulp between machines), packed pixels within 1 LSB."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden  # noqa: E402
import oracle_lib  # noqa: E402

CASES = sorted(make_golden.CASES)


def load_fixture(name):
    z = np.load(os.path.join(HERE, "golden", name + ".npz"))
    return json.loads(str(z["meta"])), z


def check_against_fixture(name, argb, planes, stats, cfg, rgb_tol):
    meta, z = load_fixture(name)
    win = meta["window"]
    g_argb = make_golden.crop(cfg, win, argb)
    g_id = make_golden.crop(cfg, win, planes["hit_id"])
    g_t = make_golden.crop(cfg, win, planes["hit_t"])
    g_rgb = make_golden.crop(cfg, win, planes["rgb"])
    assert np.array_equal(g_id, z["hit_id"])
    assert np.array_equal(g_argb != 0, z["argb"] != 0)
    hit = z["hit_id"] >= 0
    assert np.array_equal(g_t[hit].view(np.uint32), z["hit_t"][hit].view(np.uint32))
    assert float(np.abs(g_rgb - z["rgb"]).max()) <= rgb_tol
    for sh in (24, 16, 8, 0):
        a = ((g_argb >> sh) & 0xFF).astype(np.int32)
        b = ((z["argb"] >> sh) & 0xFF).astype(np.int32)
        assert np.abs(a - b).max() <= 1
    for k, v in meta["stats"].items():
        assert stats[k] == v, (k, stats[k], v)


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden(name):
    case = make_golden.CASES[name]
    cfg, flat = make_golden.build(case)
    argb, planes, st = oracle_lib.render(flat, cfg, window=case["window"])
    check_against_fixture(name, argb, planes, st, cfg, rgb_tol=2e-6)


def test_oracle_thread_count_does_not_change_results():
    case = make_golden.CASES["c1_test_scene"]
    cfg, flat = make_golden.build(case)
    a1, p1, _ = oracle_lib.render(flat, cfg, window=case["window"], n_threads=1)
    a8, p8, _ = oracle_lib.render(flat, cfg, window=case["window"], n_threads=8)
    assert np.array_equal(a1, a8) and np.array_equal(p1["rgb"], p8["rgb"])


def test_oracle_tile_partition_union():
    """n_ranks/rank window semantics of the oracle: union of ranks == full window, disjoint."""
    case = make_golden.CASES["c1_test_scene"]
    cfg, flat = make_golden.build(case)
    full, _, sf = oracle_lib.render(flat, cfg, window=case["window"])
    acc = np.zeros_like(full)
    for r in range(4):
        part, _, _ = oracle_lib.render(flat, cfg, window=case["window"], n_ranks=4, rank=r)
        assert not ((acc != 0) & (part != 0)).any()
        acc |= part
    assert np.array_equal(acc, full)


# ---- the at-spec fixtures (tests/golden/spec_*.npz, made offline by tests/golden/make_spec_golden.py with the threaded packet
# restatement): a corner of every fixture's first window is rendered again with the SCALAR oracle and must be the same bits ----
import make_spec_golden  # noqa: E402

SPEC_CASES = sorted(n for n in make_spec_golden.SPEC_CASES if os.path.exists(os.path.join(HERE, "golden", n + ".npz")))


def check_spec_window(z, i, win, cfg, argb, planes, stats, rgb_tol, sub=None, want_stats=None):
    """`sub` = (dx, dy, w, h) inside the fixture's window `win`: compare only that part."""
    dx, dy, w, h = sub if sub else (0, 0, win[2], win[3])
    region = (win[0] + dx, win[1] + dy, w, h)
    sl = (slice(dy, dy + h), slice(dx, dx + w))
    g_id, g_t = make_spec_golden.crop(cfg, region, planes["hit_id"]), make_spec_golden.crop(cfg, region, planes["hit_t"])
    g_rgb, g_argb = make_spec_golden.crop(cfg, region, planes["rgb"]), make_spec_golden.crop(cfg, region, argb)
    f_id, f_t, f_rgb, f_argb = z[f"w{i}_hit_id"][sl], z[f"w{i}_hit_t"][sl], z[f"w{i}_rgb"][sl], z[f"w{i}_argb"][sl]
    assert np.array_equal(g_id, f_id), f"{int((g_id != f_id).sum())} hit ids differ"
    assert np.array_equal(g_argb != 0, f_argb != 0)
    hit = f_id >= 0
    assert np.array_equal(g_t[hit].view(np.uint32), f_t[hit].view(np.uint32)), "hit t not bit-exact"
    d = float(np.abs(g_rgb - f_rgb).max())
    assert d <= rgb_tol, d
    for sh in (24, 16, 8, 0):
        assert np.abs(((g_argb >> sh) & 0xFF).astype(np.int32) - ((f_argb >> sh) & 0xFF).astype(np.int32)).max() <= 1
    if want_stats is not None:
        for k, v in want_stats.items():
            assert stats[k] == v, (k, stats[k], v)
    return d


@pytest.mark.parametrize("name", SPEC_CASES)
def test_scalar_oracle_reproduces_a_corner_of_every_at_spec_fixture(name):
    import bench
    meta, z = make_spec_golden.load(name)
    cfg, flat, _ = bench.build_workload(meta["workload"])
    assert (cfg.width, cfg.height) == (meta["width"], meta["height"])
    win = meta["windows"][0]
    sub = (win[2] // 2 - 3, win[3] // 2 - 2, 6, 4)  # 24 pixels in the middle of the first window (the sphere's rim)
    region = (win[0] + sub[0], win[1] + sub[1], sub[2], sub[3])
    argb, planes, st = oracle_lib.render(flat, cfg, window=region)
    check_spec_window(z, 0, win, cfg, argb, planes, st, rgb_tol=2e-6, sub=sub)


def test_at_spec_fixtures_cover_every_baseline_config_with_3000_pixels():
    for name in ("spec_c3", "spec_c3lowres", "spec_c4", "spec_c4d21", "spec_c5"):
        meta, z = make_spec_golden.load(name)
        assert sum(w[2] * w[3] for w in meta["windows"]) >= 3000, name
        for i, w in enumerate(meta["windows"]):
            assert z[f"w{i}_hit_id"].shape == (w[3], w[2]) and z[f"w{i}_rgb"].shape == (w[3], w[2], 3)
            assert (z[f"w{i}_hit_id"] >= 0).any()
