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
// tunables of the build in progress on this thread (rt_bvh_tuning of the scene description; thread_local: scenes for
// distinct GPUs may be created concurrently)
thread_local uint32_t kMaxLeaf = 4;
constexpr float kTraversalCost = 1.0f;
thread_local float kTriCost = 2.0f;
thread_local int kSplitDepth = 0;       // early split clipping: at most 2^depth references per triangle (0 = off: measured slower)
thread_local float kSplitGain = 0.8f;   // split only if area(left) + area(right) < gain * area(whole)

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

// ---- early split clipping -------------------------------------------------------------------------
// The text mesh is full of slivers (median AABB area = 3.8x the triangle's, p90 8.3x): a wavefront
// that enters such a box almost always misses the triangle.  Before the BVH is built, the REFERENCE
// of a triangle whose box is loose is split at the midpoint of its longest axis (the triangle is
// clipped against the plane, each half gets the box of its clipped polygon), recursively.  A
// triangle may then be referenced from several leaves; the traversal tests the whole triangle with
// the literal test whenever it meets one of its references.  That is result-neutral for nearest
// hits (same t and id) and for opaque occluders (a lane stops at its first opaque hit), but NOT for
// transmissive triangles, whose opacity/filter contributions must be counted once: those are never
// split (`no_split`).
namespace {

struct Ref {
  Aabb box;
  uint32_t tri;
};

// clips polygon `in` (n points) against the half space  sign*(p[axis] - pos) <= 0
int clip_poly(const float (*in)[3], int n, int axis, float pos, float sign, float (*out)[3]) {
  int m = 0;
  for (int i = 0; i < n; i++) {
    const float* a = in[i];
    const float* b = in[(i + 1) % n];
    float da = sign * (a[axis] - pos), db = sign * (b[axis] - pos);
    if (da <= 0.f) {
      for (int k = 0; k < 3; k++) out[m][k] = a[k];
      m++;
    }
    if ((da < 0.f && db > 0.f) || (da > 0.f && db < 0.f)) {
      float t = da / (da - db);
      for (int k = 0; k < 3; k++) out[m][k] = a[k] + t * (b[k] - a[k]);
      out[m][axis] = pos;
      m++;
    }
  }
  return m;
}

void split_refs(const float (*poly)[3], int n, const Aabb& bounds, uint32_t tri, int depth, float min_extent,
                std::vector<Ref>& out) {
  Aabb box;
  box.reset();
  for (int i = 0; i < n; i++) box.grow(poly[i]);
  for (int a = 0; a < 3; a++) {  // never grow beyond the parent's bounds
    box.lo[a] = std::max(box.lo[a], bounds.lo[a]);
    box.hi[a] = std::min(box.hi[a], bounds.hi[a]);
  }
  int axis = 0;
  float ext = 0.f;
  for (int a = 0; a < 3; a++)
    if (box.hi[a] - box.lo[a] > ext) {
      ext = box.hi[a] - box.lo[a];
      axis = a;
    }
  bool done = depth <= 0 || ext <= min_extent || n < 3;
  if (!done) {
    float pos = 0.5f * (box.lo[axis] + box.hi[axis]);
    float l[10][3], r[10][3];
    int nl = clip_poly(poly, n, axis, pos, 1.f, l);
    int nr = clip_poly(poly, n, axis, pos, -1.f, r);
    if (nl >= 3 && nr >= 3) {
      Aabb bl, brr;
      bl.reset();
      brr.reset();
      for (int i = 0; i < nl; i++) bl.grow(l[i]);
      for (int i = 0; i < nr; i++) brr.grow(r[i]);
      // split only when it pays: the two halves together are clearly smaller than the whole
      if (bl.half_area() + brr.half_area() < kSplitGain * box.half_area()) {
        Aabb bndl = box, bndr = box;
        bndl.hi[axis] = pos;
        bndr.lo[axis] = pos;
        split_refs(l, nl, bndl, tri, depth - 1, min_extent, out);
        split_refs(r, nr, bndr, tri, depth - 1, min_extent, out);
        return;
      }
    }
  }
  Ref rf;
  rf.box = box;
  rf.tri = tri;
  out.push_back(rf);
}

}  // namespace

void rt_build_bvh(const float* v1, const float* e1, const float* e2, const uint8_t* no_split, uint32_t n,
                  const rt_bvh_tuning& tuning, RtBvh* out) {
  kMaxLeaf = tuning.max_leaf ? (tuning.max_leaf > 64u ? 64u : tuning.max_leaf) : 4u;
  kTriCost = tuning.tri_cost > 0.f ? tuning.tri_cost : 2.0f;
  kSplitDepth = tuning.split_depth > 8u ? 8 : (int)tuning.split_depth;
  kSplitGain = tuning.split_gain > 0.f ? tuning.split_gain : 0.8f;
  out->nodes.clear();
  out->tri_order.clear();
  out->n_leaves = out->max_depth = out->max_leaf = 0;

  // references (one or more per triangle)
  std::vector<Ref> refs;
  refs.reserve((size_t)n * 2);
  // splitting stops at the median triangle extent: below that boxes are as tight as the typical one
  float min_extent = 0.f;
  if (n && kSplitDepth > 0) {
    std::vector<float> exts(n);
    for (uint32_t i = 0; i < n; i++) {
      float m = 0.f;
      for (int a = 0; a < 3; a++) {
        float p0 = v1[3 * (size_t)i + a], p1 = p0 + e1[3 * (size_t)i + a], p2 = p0 + e2[3 * (size_t)i + a];
        m = std::max(m, std::max(p0, std::max(p1, p2)) - std::min(p0, std::min(p1, p2)));
      }
      exts[i] = m;
    }
    std::nth_element(exts.begin(), exts.begin() + n / 2, exts.end());
    min_extent = exts[n / 2];
  }
  for (uint32_t i = 0; i < n; i++) {
    float poly[3][3];
    for (int a = 0; a < 3; a++) {
      poly[0][a] = v1[3 * (size_t)i + a];
      poly[1][a] = poly[0][a] + e1[3 * (size_t)i + a];
      poly[2][a] = poly[0][a] + e2[3 * (size_t)i + a];
    }
    Aabb b;
    b.reset();
    for (int k = 0; k < 3; k++) b.grow(poly[k]);
    bool split = kSplitDepth > 0 && !(no_split && no_split[i]);
    size_t first = refs.size();
    split_refs(poly, 3, b, i, split ? kSplitDepth : 0, min_extent, refs);
    // Padding: absolute 2e-5 plus 1e-4 of the triangle's largest extent plus 4 ulp of the
    // coordinate magnitude.  The literal test accepts u,v slightly outside [0,1] through rounding,
    // p1/p2 above are themselves rounded, and clipped boxes carry the rounding of the clip.
    float ext = 0.f, mag = 0.f;
    for (int a = 0; a < 3; a++) {
      ext = std::max(ext, b.hi[a] - b.lo[a]);
      mag = std::max(mag, std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a])));
    }
    float pad = 2e-5f + 1e-4f * ext + 4.0f * 1.1920929e-7f * mag;
    for (size_t k = first; k < refs.size(); k++)
      for (int a = 0; a < 3; a++) {
        refs[k].box.lo[a] -= pad;
        refs[k].box.hi[a] += pad;
      }
  }
  const uint32_t nr = (uint32_t)refs.size();
  std::vector<uint32_t> order(nr);
  std::vector<Aabb> tb(nr);
  std::vector<float> cent(3 * (size_t)nr);
  for (uint32_t i = 0; i < nr; i++) {
    order[i] = i;
    tb[i] = refs[i].box;
    for (int a = 0; a < 3; a++) cent[3 * (size_t)i + a] = 0.5f * (refs[i].box.lo[a] + refs[i].box.hi[a]);
  }

  RtNode root;
  memset(&root, 0, sizeof(root));
  // An absent child's box is NaN: every comparison of a slab test against it is false in every copy of the tree, whatever
  // the signs of the direction.  (An inverted infinite box -- lo = +inf, hi = -inf -- is NOT safe: a direction with all
  // components negative turns it into the slab (-inf, +inf) on every axis; the candidate walk of rt_flags_kernel then
  // followed RT_NODE_EMPTY of a scene without triangles into unmapped memory: fuzz variant 2, seed 11.)
  for (int a = 0; a < 3; a++) {
    root.lo0[a] = root.lo1[a] = NAN;
    root.hi0[a] = root.hi1[a] = NAN;
  }
  root.c0 = root.c1 = RT_NODE_EMPTY;
  root.n0 = root.n1 = 0;
  if (n == 0) {
    out->nodes.push_back(root);
    return;
  }

  Builder bld{tb, cent, order, out->nodes};
  Builder::Child top = bld.build(0, nr, 1);
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
  // leaf slot -> triangle; later references of a triangle (in slot order) carry RT_TRI_DUPLICATE
  out->tri_order.resize(nr);
  std::vector<uint8_t> seen(n, 0);
  for (uint32_t slot = 0; slot < nr; slot++) {
    uint32_t t = refs[order[slot]].tri;
    out->tri_order[slot] = t | (seen[t] ? RT_TRI_DUPLICATE : 0u);
    seen[t] = 1;
  }
}
