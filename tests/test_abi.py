"""The C-ABI library loads and exports every symbol include/rt_hip.h declares; the ctypes mirror has
the same struct layout as the C header.  No compute calls (no GPU needed)."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

from hslu_i.ba_raytracing.f2501_raytracer_amd import _abi, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rt_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"^\s*(?:int|void|const char\*)\s+(rt_[a-z_]+)\s*\(", src, flags=re.M)
    return sorted(set(names))


def test_header_declares_the_expected_entry_points():
    assert set(declared_functions()) == set(_lib.EXPORTS)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _lib.load()
    for name in declared_functions():
        assert hasattr(lib, name), name
    assert lib.rt_last_error() is not None
    assert lib.rt_device_count() >= 0


def test_ctypes_layout_matches_header(tmp_path):
    prog = tmp_path / "sz.c"
    prog.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "rt_hip.h"\n'
        "int main(void){printf(\"%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n\", sizeof(rt_scene_desc), sizeof(rt_params), sizeof(rt_aux),"
        " sizeof(rt_stats), sizeof(rt_bvh_info), offsetof(rt_params, aa_offsets), offsetof(rt_params, cloud_sets),"
        " offsetof(rt_params, traversal), offsetof(rt_params, tuning), offsetof(rt_scene_desc, bvh), sizeof(rt_gather_info),"
        " offsetof(rt_stats, rays_traced), offsetof(rt_stats, gather_ms), offsetof(rt_stats, notes), offsetof(rt_stats, queue_bytes),"
        " offsetof(rt_params, tuning) + offsetof(rt_tuning, levels)); return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(prog)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(_abi.rt_scene_desc), C.sizeof(_abi.rt_params), C.sizeof(_abi.rt_aux), C.sizeof(_abi.rt_stats),
            C.sizeof(_abi.rt_bvh_info), _abi.rt_params.aa_offsets.offset, _abi.rt_params.cloud_sets.offset,
            _abi.rt_params.traversal.offset, _abi.rt_params.tuning.offset, _abi.rt_scene_desc.bvh.offset,
            C.sizeof(_abi.rt_gather_info), _abi.rt_stats.rays_traced.offset, _abi.rt_stats.gather_ms.offset,
            _abi.rt_stats.notes.offset, _abi.rt_stats.queue_bytes.offset, _abi.rt_params.tuning.offset + _abi.rt_tuning.levels.offset]
    assert got == want


def test_hash_spec_matches_between_c_and_python(tmp_path):
    """rt_tile_owner (ABI spec) restated in distributed.py must agree with the header."""
    from hslu_i.ba_raytracing.f2501_raytracer_amd import RenderConfig
    from hslu_i.ba_raytracing.f2501_raytracer_amd.distributed import tile_owner_map
    prog = tmp_path / "own.c"
    prog.write_text('#include <stdio.h>\n#include "rt_hip.h"\nint main(void){for(unsigned n=1;n<=9;n++)for(unsigned y=0;y<5;y++)'
                    'for(unsigned x=0;x<7;x++)printf("%u ", rt_tile_owner(x,y,n)); return 0;}\n')
    exe = tmp_path / "own"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(prog)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    cfg = RenderConfig.from_features([], width_override=7 * 48, height_override=5 * 48)
    want = []
    for n in range(1, 10):
        want.extend(tile_owner_map(cfg, n).ravel().tolist())
    assert got == want


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/librt_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def _build_c_example(tmp_path):
    lib_dir = os.path.dirname(_lib.LIB_PATH)
    exe = tmp_path / "c_abi_example"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_example.c"),
                           "-L", lib_dir, "-lrt_hip", f"-Wl,-rpath,{lib_dir}", "-o", str(exe)])
    return exe


def test_c_example_links_against_the_abi(tmp_path):
    """A plain C caller (what a Rust / cgo / JNI binding amounts to) compiles and links with nothing but
    include/rt_hip.h and librt_hip.so; without a GPU it reports that and exits 0."""
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    exe = _build_c_example(tmp_path)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "pixels written" in out.stdout or "no HIP device" in out.stdout


@pytest.mark.gpu
def test_c_example_renders_on_the_gpu(tmp_path):
    exe = _build_c_example(tmp_path)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "rays 7680," in out.stdout and "pixels written 0 " not in out.stdout
    assert "two frames in flight match rt_render" in out.stdout and "3 ranks on one GPU match" in out.stdout
    # the multi-GPU entry point with n_gpu = 1 and with 3 tile-partitioned ranks rehearsed on the one GPU
    assert "rt_render_multi: 1 GPU and 3 ranks on one GPU match rt_render" in out.stdout
    assert "progressive frame matches rt_render" in out.stdout and "scene holds" in out.stdout
