"""Host BVH builder (csrc/rt_bvh.cpp) checked on the CPU: the builder is plain C++, compiled here host-only with a small
probe.  Properties the kernels rely on: every index stays inside the tree, the tree is acyclic and references every
triangle, parents' boxes contain their children's, and ABSENT children carry NaN boxes (an inverted infinite box passes the
slab test of an all-negative direction: round 3's second hung kernel)."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "hslu_i", "ba_raytracing", "f2501_raytracer_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

PROBE = r'''
#include <hip/hip_runtime.h>
#include <cstring>
#include "rt_internal.h"
// nodes_out: [max_nodes][16] words (the 64-byte RtNode), order_out: [max_slots]; returns the node count (or -1: too many)
extern "C" int bvh_probe(const float* v1, const float* e1, const float* e2, const uint8_t* no_split, uint32_t n, uint32_t* nodes_out,
                         uint32_t max_nodes, uint32_t* order_out, uint32_t max_slots, uint32_t* info) {
  RtBvh b;
  rt_bvh_tuning t{};
  rt_build_bvh(v1, e1, e2, no_split, n, t, &b);
  if (b.nodes.size() > max_nodes || b.tri_order.size() > max_slots) return -1;
  static_assert(sizeof(RtNode) == 64, "node size");
  memcpy(nodes_out, b.nodes.data(), b.nodes.size() * sizeof(RtNode));
  memcpy(order_out, b.tri_order.data(), b.tri_order.size() * 4);
  info[0] = b.n_leaves, info[1] = b.max_depth, info[2] = b.max_leaf, info[3] = (uint32_t)b.tri_order.size();
  return (int)b.nodes.size();
}
'''
EMPTY = 0xFFFFFFFF
DUP = 0x80000000
IDX = 0x3FFFFFFF


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    d = tmp_path_factory.mktemp("bvh_probe")
    src = d / "probe.cpp"
    src.write_text(PROBE)
    so = d / "probe.so"
    subprocess.run([HIPCC, "-O2", "-std=c++17", "-fPIC", "-x", "hip", "--cuda-host-only", "-I", CSRC, "-I", os.path.join(ROOT, "include"),
                    "-shared", "-o", str(so), str(src), os.path.join(CSRC, "rt_bvh.cpp")], check=True, capture_output=True, timeout=600)
    return C.CDLL(str(so))


def build(probe, v1, e1, e2, no_split=None):
    n = len(v1)
    v1, e1, e2 = (np.ascontiguousarray(a, np.float32).reshape(-1, 3) for a in (v1, e1, e2))
    ns = np.zeros(max(n, 1), np.uint8) if no_split is None else np.ascontiguousarray(no_split, np.uint8)
    max_nodes, max_slots = 4 * n + 8, 8 * n + 8
    nodes = np.zeros((max_nodes, 16), np.uint32)
    order = np.zeros(max_slots, np.uint32)
    info = np.zeros(4, np.uint32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    k = probe.bvh_probe(p(v1), p(e1), p(e2), p(ns), C.c_uint32(n), p(nodes), C.c_uint32(max_nodes), p(order), C.c_uint32(max_slots), p(info))
    assert k >= 1
    return nodes[:k], order[: info[3]], info


def children(node):
    f = node.view(np.float32)
    # RtNode: lo0[3] c0 hi0[3] n0 lo1[3] c1 hi1[3] n1
    return [(f[0:3], f[4:7], int(node[3]), int(node[7])), (f[8:11], f[12:15], int(node[11]), int(node[15]))]


def soup(seed, n):
    r = np.random.default_rng(seed)
    v1 = r.uniform(0, 1, (n, 3))
    s = np.where(r.random(n) < 0.2, 0.3, 0.05)[:, None]
    e1, e2 = r.normal(0, 1, (n, 3)) * s, r.normal(0, 1, (n, 3)) * s
    return v1.astype(np.float32), e1.astype(np.float32), e2.astype(np.float32)


def test_empty_tree_has_two_absent_children_with_nan_boxes(probe):
    z = np.zeros((0, 3), np.float32)
    nodes, order, info = build(probe, z, z, z)
    assert len(nodes) == 1 and len(order) == 0
    for lo, hi, c, n in children(nodes[0]):
        assert c == EMPTY and n == 0 and np.isnan(lo).all() and np.isnan(hi).all()


@pytest.mark.parametrize("n", [1, 2, 3, 4])
def test_single_leaf_tree_wraps_the_leaf_and_an_absent_child(probe, n):
    v1, e1, e2 = soup(7, n)
    nodes, order, info = build(probe, v1, e1, e2)
    ch = [c for node in nodes for c in children(node)]
    absent = [c for c in ch if c[2] == EMPTY]
    for lo, hi, c, cnt in absent:
        assert cnt == 0 and np.isnan(lo).all() and np.isnan(hi).all(), "an absent child must not pass any slab test"
    assert sorted(int(o & IDX) for o in order if not (o & DUP)) == list(range(n))


@pytest.mark.parametrize("seed,n", [(1, 5), (2, 17), (3, 200), (4, 977), (5, 3000)])
def test_tree_is_acyclic_in_range_nested_and_complete(probe, seed, n):
    v1, e1, e2 = soup(seed, n)
    nodes, order, info = build(probe, v1, e1, e2)
    v = np.stack([v1, v1 + e1, v1 + e2], 1)                     # [n][3 vertices][3]
    tri_lo, tri_hi = v.min(1), v.max(1)
    seen_nodes, seen_slots = set(), set()

    def visit(i, lo_p, hi_p, depth):
        assert 0 <= i < len(nodes) and i not in seen_nodes, "node index out of range or visited twice (a cycle)"
        seen_nodes.add(i)
        assert depth <= 64, "deeper than the 64 lanes of the walk stack"
        for lo, hi, c, cnt in children(nodes[i]):
            assert c != EMPTY, "a tree of more than one leaf has no absent child"
            assert np.isfinite(lo).all() and np.isfinite(hi).all() and (lo <= hi).all()
            if lo_p is not None:
                assert (lo >= lo_p - 1e-6).all() and (hi <= hi_p + 1e-6).all(), "child box outside its parent's"
            if cnt:  # a leaf: slots c .. c + cnt - 1
                assert c + cnt <= len(order)
                for s in range(c, c + cnt):
                    assert s not in seen_slots
                    seen_slots.add(s)
                    t = int(order[s] & IDX)
                    assert t < n
                    # the (padded) leaf box holds the triangle (unsplit build: the whole triangle)
                    assert (tri_lo[t] >= lo - 1e-5).all() and (tri_hi[t] <= hi + 1e-5).all()
            else:
                visit(c, lo, hi, depth + 1)

    visit(0, None, None, 1)
    assert len(seen_nodes) == len(nodes) and len(seen_slots) == len(order)
    firsts = sorted(int(o & IDX) for o in order if not (o & DUP))
    assert firsts == list(range(n)), "every triangle is referenced exactly once as a first reference"
    assert 1 <= info[2] <= 4  # (the default cap: 4 triangles per leaf)
