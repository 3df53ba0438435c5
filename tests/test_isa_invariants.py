"""Static checks of the compiled kernels (CPU only: hipcc cross-compiles gfx950 here).

The render kernels are written as wave-uniform control flow over per-lane data: walks, candidate loops and sample loops
branch on wave votes, never on a lane's own value.  One per-lane pointer in a walk's early-out (round 3, rt_flags_kernel)
turned the whole walk into a divergent loop -- node index in a VGPR, EXEC narrowed lane by lane -- and hung it.  The ISA
shows that directly: a divergent loop ends in `s_andn2_b64 exec, exec, ...` + `s_cbranch_execnz/z`."""
import os
import re
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "hslu_i", "ba_raytracing", "f2501_raytracer_amd", "csrc")
ASM = os.path.join(CSRC, "rt_kernels.s")

# kernel -> divergent loops it is ALLOWED to have
ALLOWED = {
    "rt_primary_kernel": 0,
    "rt_primary_stream_kernel": 0,
    "rt_shade_kernel": 0,
    "rt_trace_kernel": 0,
    "rt_hard_kernel": 1,   # the stackless per-lane walk of incoherent (hit point, light) pairs: divergent by design
    "rt_flags_kernel": 1,  # the per-lane binary search for the triangle that owns a receiver cell
    # merged levels
    "rt_hit_spawn_kernel": 0,
    "rt_trace_spawn_kernel": 0,
    "rt_primary_pre_kernel": 0,
    # the phase-split pipeline (rt_phases.h)
    "rt_hit_kernel": 0,
    "rt_classify0_kernel": 0,
    "rt_classify_kernel": 0,
    "rt_sets0_list_kernel": 0,
    "rt_sets_list_kernel": 0,
    "rt_sets0_walk_kernel": 0,
    "rt_sets_walk_kernel": 0,
}


def _asm_text():
    srcs = [os.path.join(CSRC, f) for f in ("rt_kernels.hip", "rt_phases.h", "rt_internal.h", "Makefile")]
    if not os.path.exists(ASM) or os.path.getmtime(ASM) < max(os.path.getmtime(f) for f in srcs):
        if not shutil.which("hipcc") and not os.path.exists("/opt/rocm/bin/hipcc"):
            pytest.skip("no hipcc")
        subprocess.run(["make", "-C", CSRC, "asm"], check=True, capture_output=True, timeout=900)
    return open(ASM).read()


def _kernel_bodies(text):
    out = {}
    for m in re.finditer(r"^(_ZN12_GLOBAL__N_1\d+(\w+?)E\w*):.*?\n(.*?)^\s*\.amdhsa_kernel \1", text, re.S | re.M):
        out[m.group(2)] = m.group(3)
    return out


def test_render_kernels_have_no_divergent_loops():
    bodies = _kernel_bodies(_asm_text())
    for name, allowed in ALLOWED.items():
        assert name in bodies, (name, sorted(bodies))
        n = len(re.findall(r"s_andn2_b64 exec, exec,", bodies[name]))
        assert n <= allowed, f"{name}: {n} divergent loops (allowed {allowed}): some walk or sample loop branches on a per-lane value"


def test_walks_fetch_their_nodes_through_scalar_loads():
    bodies = _kernel_bodies(_asm_text())
    for name in ("rt_primary_kernel", "rt_primary_stream_kernel", "rt_shade_kernel", "rt_trace_kernel", "rt_flags_kernel", "rt_hit_kernel",
                 "rt_classify0_kernel", "rt_classify_kernel", "rt_hit_spawn_kernel", "rt_trace_spawn_kernel", "rt_primary_pre_kernel"):
        assert "s_load_dwordx16" in bodies[name], f"{name}: no 64-byte scalar node fetch (a walk lost its uniformity)"


def test_lane_indexed_stores_ignore_exec():
    bodies = _kernel_bodies(_asm_text())
    for name in ("rt_primary_kernel", "rt_shade_kernel", "rt_trace_kernel", "rt_flags_kernel", "rt_hit_kernel", "rt_classify0_kernel", "rt_classify_kernel"):
        assert "v_writelane_b32" in bodies[name], name
