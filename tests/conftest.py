import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib

    return oracle_lib.load()


@pytest.fixture(scope="session", autouse=True)
def _native_libraries_built():
    """The .so files travel with the working tree; if a checkout lacks them, build them once
    (hipcc cross-compiles gfx950 without a GPU).  Never falls back to anything else."""
    lib = os.path.join(ROOT, "hslu_i", "ba_raytracing", "f2501_raytracer_amd", "librt_hip.so")
    if not os.path.exists(lib):
        import __graft_entry__

        __graft_entry__.build()
