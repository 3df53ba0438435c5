"""Every BASELINE config exactly as bench.py builds it, on the committed at-spec fixtures (tests/golden/spec_*.npz: windows over
the glass sphere's rim, text silhouettes, the pile of metallic-glass spheres, floor and wall penumbrae -- >= 3 000 pixels per
config, rendered offline with the threaded packet restatement of the oracle): hit ids and pixel indices exact, `t` bit-exact,
RGB within 1e-4, ray counters equal."""
import numpy as np
import pytest

import bench
from test_oracle_golden import SPEC_CASES, check_spec_window, make_spec_golden  # (test_oracle_golden puts tests/golden on the path)
from test_parity_gpu import RGB_TOL, gpu_render

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", SPEC_CASES)
def test_gpu_reproduces_at_spec_fixture(name):
    meta, z = make_spec_golden.load(name)
    cfg, flat, _ = bench.build_workload(meta["workload"])
    worst, n_px = 0.0, 0
    for i, win in enumerate(meta["windows"]):
        argb, planes, st = gpu_render(cfg, flat, tuple(win))
        worst = max(worst, check_spec_window(z, i, win, cfg, argb, planes, st, rgb_tol=RGB_TOL, want_stats=meta["stats"][i]))
        n_px += win[2] * win[3]
    print(f"{name}: {n_px} at-spec pixels, max |dRGB| vs the fixture {worst:.2e}")
