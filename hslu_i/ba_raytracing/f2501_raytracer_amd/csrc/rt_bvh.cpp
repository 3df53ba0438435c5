// rt_bvh.cpp -- host-side binned-SAH BVH2 builder over the scene's triangles.
//
// No reference counterpart: the reference scans every object linearly for every ray
// (src/raytracing/raytracer.rs:48,180).  The BVH only prunes work; traversal in rt_kernels.hip
// keeps the linear scan's result (nearest hit with ties going to the later object; shadow rays
// visit every hit <= tmax).  To guarantee that, every box is padded so the fp32 slab test cannot
// cull a triangle that the literal matrix-inverse test (triangle.rs:149-212) would accept.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "rt_internal.h"

namespace {

struct Aabb {
  float lo[3], hi[3];
  void reset() {
    for (int a = 0; a < 3; a++) {
      lo[a] = INFINITY;
      hi[a] = -INFINITY;
    }
  }
  void grow(const float* p) {
    for (int a = 0; a < 3; a++) {
      lo[a] = std::min(lo[a], p[a]);
      hi[a] = std::max(hi[a], p[a]);
    }
  }
  void grow(const Aabb& b) {
    for (int a = 0; a < 3; a++) {
      lo[a] = std::min(lo[a], b.lo[a]);
      hi[a] = std::max(hi[a], b.hi[a]);
    }
  }
  float half_area() const {
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (dx < 0 || dy < 0 || dz < 0) return 0.f;
    return dx * dy + dy * dz + dz * dx;
  }
};

constexpr int kBins = 16;
// tunables (environment overrides are for experiments only: RT_BVH_MAX_LEAF, RT_BVH_TRI_COST)
uint32_t kMaxLeaf = 4;
float kTraversalCost = 1.0f;
float kTriCost = 2.0f;

struct Builder {
  const std::vector<Aabb>& tb;      // per-triangle padded bounds
  const std::vector<float>& cent;   // per-triangle centroid [n][3]
  std::vector<uint32_t>& order;     // permutation being partitioned
  std::vector<RtNode>& nodes;
  uint32_t n_leaves = 0, max_depth = 0, max_leaf = 0;

  struct Child {
    Aabb box;
    uint32_t c, n;  // n == 0: node index c
  };

  // builds the subtree over order[begin, end); returns how a parent refers to it
  Child build(uint32_t begin, uint32_t end, uint32_t depth) {
    Child me;
    me.box.reset();
    for (uint32_t i = begin; i < end; i++) me.box.grow(tb[order[i]]);
    uint32_t count = end - begin;
    max_depth = std::max(max_depth, depth);

    auto make_leaf = [&]() {
      me.c = begin;
      me.n = count;
      n_leaves++;
      max_leaf = std::max(max_leaf, count);
      return me;
    };
    if (count <= 1 || (count <= 2 && kMaxLeaf >= 2)) return make_leaf();

    // centroid bounds
    Aabb cb;
    cb.reset();
    for (uint32_t i = begin; i < end; i++) cb.grow(&cent[3 * (size_t)order[i]]);

    float best_cost = INFINITY;
    int best_axis = -1, best_split = -1;
    for (int a = 0; a < 3; a++) {
      float ext = cb.hi[a] - cb.lo[a];
      if (!(ext > 0.f)) continue;
      Aabb bb[kBins];
      uint32_t bc[kBins] = {0};
      for (int b = 0; b < kBins; b++) bb[b].reset();
      float scale = (float)kBins / ext;
      for (uint32_t i = begin; i < end; i++) {
        uint32_t t = order[i];
        int b = (int)((cent[3 * (size_t)t + a] - cb.lo[a]) * scale);
        b = std::min(std::max(b, 0), kBins - 1);
        bb[b].grow(tb[t]);
        bc[b]++;
      }
      float right_area[kBins];
      uint32_t right_cnt[kBins];
      Aabb acc;
      acc.reset();
      uint32_t cnt = 0;
      for (int b = kBins - 1; b > 0; b--) {
        acc.grow(bb[b]);
        cnt += bc[b];
        right_area[b] = acc.half_area();
        right_cnt[b] = cnt;
      }
      acc.reset();
      cnt = 0;
      for (int b = 0; b < kBins - 1; b++) {
        acc.grow(bb[b]);
        cnt += bc[b];
        if (cnt == 0 || right_cnt[b + 1] == 0) continue;
        float cost = acc.half_area() * (float)cnt + right_area[b + 1] * (float)right_cnt[b + 1];
        if (cost < best_cost) {
          best_cost = cost;
          best_axis = a;
          best_split = b;
        }
      }
    }

    float parent_area = me.box.half_area();
    float leaf_cost = kTriCost * (float)count;
    float split_cost = best_axis >= 0 && parent_area > 0.f
                           ? kTraversalCost + kTriCost * best_cost / parent_area
                           : INFINITY;
    if (count <= kMaxLeaf && leaf_cost <= split_cost) return make_leaf();

    uint32_t mid;
    if (best_axis < 0) {
      // all centroids coincide: split by index
      if (count <= kMaxLeaf) return make_leaf();
      mid = begin + count / 2;
    } else {
      float ext = cb.hi[best_axis] - cb.lo[best_axis];
      float scale = (float)kBins / ext;
      auto it = std::partition(order.begin() + begin, order.begin() + end, [&](uint32_t t) {
        int b = (int)((cent[3 * (size_t)t + best_axis] - cb.lo[best_axis]) * scale);
        b = std::min(std::max(b, 0), kBins - 1);
        return b <= best_split;
      });
      mid = (uint32_t)(it - order.begin());
      if (mid == begin || mid == end) mid = begin + count / 2;
    }

    uint32_t idx = (uint32_t)nodes.size();
    nodes.emplace_back();
    Child l = build(begin, mid, depth + 1);
    Child r = build(mid, end, depth + 1);
    RtNode& nd = nodes[idx];
    for (int a = 0; a < 3; a++) {
      nd.lo0[a] = l.box.lo[a];
      nd.hi0[a] = l.box.hi[a];
      nd.lo1[a] = r.box.lo[a];
      nd.hi1[a] = r.box.hi[a];
    }
    nd.c0 = l.c;
    nd.n0 = l.n;
    nd.c1 = r.c;
    nd.n1 = r.n;
    me.c = idx;
    me.n = 0;
    return me;
  }
};

}  // namespace

void rt_build_bvh(const float* v1, const float* e1, const float* e2, uint32_t n, RtBvh* out) {
  if (const char* e = getenv("RT_BVH_MAX_LEAF")) kMaxLeaf = (uint32_t)atoi(e) < 1 ? 1u : (uint32_t)atoi(e);
  if (const char* e = getenv("RT_BVH_TRI_COST")) kTriCost = (float)atof(e);
  out->nodes.clear();
  out->tri_order.resize(n);
  for (uint32_t i = 0; i < n; i++) out->tri_order[i] = i;
  out->n_leaves = out->max_depth = out->max_leaf = 0;

  std::vector<Aabb> tb(n);
  std::vector<float> cent(3 * (size_t)n);
  for (uint32_t i = 0; i < n; i++) {
    float p0[3], p1[3], p2[3];
    for (int a = 0; a < 3; a++) {
      p0[a] = v1[3 * (size_t)i + a];
      p1[a] = p0[a] + e1[3 * (size_t)i + a];
      p2[a] = p0[a] + e2[3 * (size_t)i + a];
    }
    Aabb b;
    b.reset();
    b.grow(p0);
    b.grow(p1);
    b.grow(p2);
    // Padding: absolute 2e-5 plus 1e-4 of the triangle's largest extent plus 4 ulp of the
    // coordinate magnitude.  The literal test accepts u,v slightly outside [0,1] through rounding,
    // and p1/p2 above are themselves rounded.
    float ext = 0.f, mag = 0.f;
    for (int a = 0; a < 3; a++) {
      ext = std::max(ext, b.hi[a] - b.lo[a]);
      mag = std::max(mag, std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a])));
    }
    float pad = 2e-5f + 1e-4f * ext + 4.0f * 1.1920929e-7f * mag;
    for (int a = 0; a < 3; a++) {
      b.lo[a] -= pad;
      b.hi[a] += pad;
      cent[3 * (size_t)i + a] = (p0[a] + p1[a] + p2[a]) * (1.0f / 3.0f);
    }
    tb[i] = b;
  }

  RtNode root;
  memset(&root, 0, sizeof(root));
  for (int a = 0; a < 3; a++) {
    root.lo0[a] = root.lo1[a] = INFINITY;
    root.hi0[a] = root.hi1[a] = -INFINITY;
  }
  root.c0 = root.c1 = RT_NODE_EMPTY;
  root.n0 = root.n1 = 0;
  if (n == 0) {
    out->nodes.push_back(root);
    return;
  }

  Builder bld{tb, cent, out->tri_order, out->nodes};
  Builder::Child top = bld.build(0, n, 1);
  if (top.n != 0) {
    // the whole scene is a single leaf: wrap it in a root whose second child is empty
    for (int a = 0; a < 3; a++) {
      root.lo0[a] = top.box.lo[a];
      root.hi0[a] = top.box.hi[a];
    }
    root.c0 = top.c;
    root.n0 = top.n;
    out->nodes.insert(out->nodes.begin(), root);
  }
  // else: nodes[0] is the root already (build() reserves its slot before recursing)
  out->n_leaves = bld.n_leaves;
  out->max_depth = bld.max_depth;
  out->max_leaf = bld.max_leaf;
}
