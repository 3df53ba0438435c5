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
