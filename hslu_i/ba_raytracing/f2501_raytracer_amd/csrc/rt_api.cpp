// rt_api.cpp -- the C ABI of include/rt_hip.h over the HIP runtime.
//
// rt_scene owns the device copies of the flattened Scene (reference src/scene/scene.rs:24-27), the
// BVH, and reusable device workspaces (parameter tables, counters, ray queues, accumulator), so a
// render call performs no allocation when its shape repeats (graph-capture friendly: rt_render_device
// only enqueues async work on the caller's stream).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rt_host.h"

static void trace_point(hipStream_t stream, const char* what, uint32_t a = 0, uint32_t b = 0, uint32_t c = 0);  // RT_TRACE_LAUNCHES

thread_local std::string g_err;

int rt_fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

extern "C" {

const char* rt_last_error(void) { return g_err.c_str(); }

#ifndef RT_BUILD_ID
#define RT_BUILD_ID "unknown"
#endif
const char* rt_build_id(void) { return RT_BUILD_ID; }

int rt_selftest_exact_math(int device, const float* in, float* out_sqrt, float* out_rcp, uint32_t n) {
  if (!in || !out_sqrt || !out_rcp) return fail(RT_ERR_INVALID_ARG, "null argument");
  if (device < 0 || device >= rt_device_count()) return fail(RT_ERR_INVALID_ARG, "device %d out of range", device);
  HIP_TRY(hipSetDevice(device));
  DevBuf b;
  int rc = b.ensure((size_t)n * 12 + 12);
  if (rc != RT_OK) return rc;
  float* d = (float*)b.p;
  HIP_TRY(hipMemcpy(d, in, (size_t)n * 4, hipMemcpyHostToDevice));
  hipError_t e = (hipError_t)rt_launch_selftest_math(d, d + n, d + 2 * (size_t)n, n, nullptr);
  if (e != hipSuccess) {
    b.release();
    return fail(RT_ERR_HIP, "selftest launch failed: %s", hipGetErrorString(e));
  }
  hipError_t e1 = hipMemcpy(out_sqrt, d + n, (size_t)n * 4, hipMemcpyDeviceToHost);
  hipError_t e2 = hipMemcpy(out_rcp, d + 2 * (size_t)n, (size_t)n * 4, hipMemcpyDeviceToHost);
  b.release();
  if (e1 != hipSuccess || e2 != hipSuccess) return fail(RT_ERR_HIP, "selftest copy failed");
  return RT_OK;
}

int rt_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void rt_scene_destroy(rt_scene* s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  if (s->tables_ev) (void)hipEventDestroy(s->tables_ev);
  for (hipEvent_t e : s->frame_ev)
    if (e) (void)hipEventDestroy(e);
  for (auto& w : s->ws) {
    if (w.cnt_ev) (void)hipEventDestroy(w.cnt_ev);
    if (w.cnt_host) (void)hipHostFree(w.cnt_host);
    if (w.fork_ev) (void)hipEventDestroy(w.fork_ev);
    w.acc.release();
    for (auto& l : w.lane) {
      if (l.stream) (void)hipStreamSynchronize(l.stream), (void)hipStreamDestroy(l.stream);
      if (l.done_ev) (void)hipEventDestroy(l.done_ev);
      for (int k = 0; k < 2; k++) {
        if (l.shade_stream[k]) (void)hipStreamSynchronize(l.shade_stream[k]), (void)hipStreamDestroy(l.shade_stream[k]);
        if (l.shade_done[k]) (void)hipEventDestroy(l.shade_done[k]);
      }
      for (hipEvent_t e : l.level_ev) (void)hipEventDestroy(e);
      if (l.hit_ev) (void)hipEventDestroy(l.hit_ev);
      for (DevBuf* b : {&l.queues, &l.qcount, &l.trace_ws, &l.hard, &l.hitrec, &l.sets}) b->release();
    }
  }
  for (DevBuf* b : {&s->blob, &s->aa, &s->cloud, &s->counters, &s->suplist, &s->fb, &s->aux_rgb, &s->costmap, &s->aux_id, &s->aux_t, &s->flag_geo, &s->flags, &s->cell_lists, &s->progress_fb})
    b->release();
  delete s;
}

int rt_scene_create(const rt_scene_desc* d, int device, rt_scene** out) {
  if (!d || !out) return fail(RT_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  if (d->abi_version != RT_ABI_VERSION)
    return fail(RT_ERR_INVALID_ARG, "rt_scene_desc.abi_version %u != %u", d->abi_version, RT_ABI_VERSION);
  if (d->n_spheres && (!d->sphere_center || !d->sphere_r_sq || !d->sphere_material))
    return fail(RT_ERR_INVALID_ARG, "sphere arrays missing");
  if (d->n_triangles && (!d->tri_v1 || !d->tri_e1 || !d->tri_e2 || !d->tri_normal || !d->tri_material))
    return fail(RT_ERR_INVALID_ARG, "triangle arrays missing");
  if ((d->n_spheres || d->n_triangles) && (!d->n_materials || !d->materials))
    return fail(RT_ERR_INVALID_ARG, "materials missing");
  if (d->n_lights && !d->lights) return fail(RT_ERR_INVALID_ARG, "lights missing");
  for (uint32_t i = 0; i < d->n_spheres; i++)
    if (d->sphere_material[i] >= d->n_materials) return fail(RT_ERR_INVALID_ARG, "sphere %u: material out of range", i);
  for (uint32_t i = 0; i < d->n_triangles; i++)
    if (d->tri_material[i] >= d->n_materials) return fail(RT_ERR_INVALID_ARG, "triangle %u: material out of range", i);

  int ndev = rt_device_count();
  if (ndev <= 0) return fail(RT_ERR_NO_DEVICE, "no HIP device visible");
  if (device < 0 || device >= ndev) return fail(RT_ERR_INVALID_ARG, "device %d out of range (%d visible)", device, ndev);
  HIP_TRY(hipSetDevice(device));

  rt_scene* s = new rt_scene();
  s->device = device;
  s->budget = d->device_budget_bytes ? d->device_budget_bytes : RT_SCENE_BUDGET_DEFAULT;
  auto bail = [&](int rc) {
    rt_scene_destroy(s);
    return rc;
  };
  auto upload = [&](DevBuf& b, const void* src, size_t bytes) -> int {
    int rc = b.ensure(bytes);
    if (rc != RT_OK) return rc;
    if (bytes) {
      hipError_t e = hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice);
      if (e != hipSuccess) return fail(RT_ERR_HIP, "hipMemcpy H2D failed: %s", hipGetErrorString(e));
    }
    return RT_OK;
  };

  const uint32_t ns = d->n_spheres, nt = d->n_triangles;
  int rc;
  // All scene arrays live in ONE device allocation (`blob`): the kernels address them as base + 32-bit byte
  // offset, which the scalar loads take as an SGPR offset (2 scalar instructions per address instead of 4, and
  // one base pointer instead of nine in SGPRs).
  std::vector<unsigned char> blob;
  auto put = [&](uint32_t* off, const void* src, size_t bytes) {
    blob.resize((blob.size() + 255) / 256 * 256);
    *off = (uint32_t)blob.size();
    if (bytes) blob.insert(blob.end(), (const unsigned char*)src, (const unsigned char*)src + bytes);
    blob.resize(blob.size() + 64);  // the widest scalar load may read past the last record
  };
  {
    // {cx, cy, cz, r_sq} per sphere, then one float per sphere: an upper bound of the radius (candidate culling)
    std::vector<float> sp(5 * (size_t)ns);
    for (uint32_t i = 0; i < ns; i++) {
      sp[4 * i + 0] = d->sphere_center[3 * i + 0];
      sp[4 * i + 1] = d->sphere_center[3 * i + 1];
      sp[4 * i + 2] = d->sphere_center[3 * i + 2];
      sp[4 * i + 3] = d->sphere_r_sq[i];
      sp[4 * (size_t)ns + i] = std::sqrt(std::fabs(d->sphere_r_sq[i])) * (1.0f + 4e-7f);
    }
    put(&s->dev.off_spheres, sp.data(), 16 * (size_t)ns);
    put(&s->dev.off_sphere_rad, sp.data() + 4 * (size_t)ns, 4 * (size_t)ns);
    put(&s->dev.off_sphere_mat, d->sphere_material, (size_t)ns * 4);
  }
  RtBvh bvh;
  std::vector<uint8_t> no_split(nt, 0);  // = transmissive
  {
    // transmissive triangles must be referenced exactly once (their shadow contributions add up)
    for (uint32_t i = 0; i < nt; i++) {
      const float* r = d->materials + (size_t)d->tri_material[i] * RT_MATERIAL_STRIDE;
      no_split[i] = (r[RT_MAT_HAS_OPACITY] != 0.0f && !(std::fabs(r[RT_MAT_OPACITY]) <= 1.1920929e-7f)) ? 1 : 0;
    }
    rt_build_bvh(d->tri_v1, d->tri_e1, d->tri_e2, no_split.data(), nt, d->bvh, &bvh);
  }
  {
    // bounds of everything a ray can hit (Morton keys of secondary hit points)
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    auto grow = [&](float x, float y, float z) {
      const float v[3] = {x, y, z};
      for (int a = 0; a < 3; a++)
        if (std::isfinite(v[a])) lo[a] = std::fmin(lo[a], v[a]), hi[a] = std::fmax(hi[a], v[a]);
    };
    for (uint32_t i = 0; i < ns; i++) {
      const float* c = d->sphere_center + 3 * (size_t)i;
      const float r = std::sqrt(std::fabs(d->sphere_r_sq[i]));
      grow(c[0] - r, c[1] - r, c[2] - r), grow(c[0] + r, c[1] + r, c[2] + r);
    }
    for (uint32_t i = 0; i < nt; i++) {
      const float *v = d->tri_v1 + 3 * (size_t)i, *a = d->tri_e1 + 3 * (size_t)i, *b = d->tri_e2 + 3 * (size_t)i;
      grow(v[0], v[1], v[2]), grow(v[0] + a[0], v[1] + a[1], v[2] + a[2]), grow(v[0] + b[0], v[1] + b[1], v[2] + b[2]);
    }
    for (int a = 0; a < 3; a++) {
      if (!(lo[a] <= hi[a])) lo[a] = 0.f, hi[a] = 1.f;
      s->aabb_lo[a] = lo[a], s->aabb_hi[a] = hi[a];
    }
  }
  const uint32_t n_slots = (uint32_t)bvh.tri_order.size();
  {
    // leaf-order intersection records; shading records: [0,n_slots) leaf order, then canonical order
    std::vector<float> isect(12 * (size_t)n_slots), shade(4 * ((size_t)n_slots + nt));
    auto put_shade = [&](size_t dst, uint32_t t) {
      float* sh = &shade[4 * dst];
      sh[0] = d->tri_normal[3 * (size_t)t + 0];
      sh[1] = d->tri_normal[3 * (size_t)t + 1];
      sh[2] = d->tri_normal[3 * (size_t)t + 2];
      uint32_t m = d->tri_material[t];
      memcpy(&sh[3], &m, 4);
    };
    for (uint32_t t = 0; t < nt; t++) put_shade((size_t)n_slots + t, t);
    for (uint32_t slot = 0; slot < n_slots; slot++) {
      uint32_t t = bvh.tri_order[slot] & ~RT_TRI_DUPLICATE;
      const float* v1 = d->tri_v1 + 3 * (size_t)t;
      const float* e1 = d->tri_e1 + 3 * (size_t)t;
      const float* e2 = d->tri_e2 + 3 * (size_t)t;
      // X = e1 x e2 in ultraviolet's cross form, bit-equal to cross(-e1, -e2) (triangle.rs:174-177).
      // volatile keeps the host compiler from contracting mul+add into an fma.
      volatile float a0 = e1[1] * e2[2], b0 = e1[2] * e2[1];
      volatile float a1 = e1[2] * e2[0], b1 = e1[0] * e2[2];
      volatile float a2 = e1[0] * e2[1], b2 = e1[1] * e2[0];
      float X[3] = {a0 + (-b0), a1 + (-b1), a2 + (-b2)};
      float* q = &isect[12 * (size_t)slot];
      q[0] = v1[0], q[1] = v1[1], q[2] = v1[2], q[3] = e1[0];
      q[4] = e1[1], q[5] = e1[2], q[6] = e2[0], q[7] = e2[1];
      q[8] = e2[2], q[9] = X[0], q[10] = X[1], q[11] = X[2];
      put_shade(slot, t);
    }
    put(&s->dev.off_tri_isect, isect.data(), isect.size() * 4);
    {
      // Receiver cells: every triangle carries an R x R grid over its (u, v) coordinates, cells of about 1/1024 of the
      // scene's diagonal (R = 1 for the small triangles of a mesh, up to 1024 for a wall).  The flags themselves depend on
      // the light clouds and are computed by rt_flags_kernel when a frame first needs them (prepare()).
      double diag2 = 0.0, pmax = 0.0;
      for (int a = 0; a < 3; a++) {
        diag2 += (double)(s->aabb_hi[a] - s->aabb_lo[a]) * (s->aabb_hi[a] - s->aabb_lo[a]);
        pmax = std::fmax(pmax, std::fmax(std::fabs((double)s->aabb_lo[a]), std::fabs((double)s->aabb_hi[a])));
      }
      std::vector<float> recv(12 * (size_t)nt), geo(12 * (size_t)nt);
      uint64_t total = 0;
      // (a scene of many wall-sized triangles: coarser cells until the flags stay below 2^26 cells = 128 MiB)
      double cell_used = 0.0;
      for (double cell = std::sqrt(diag2) / 1024.0;; cell *= 2.0) {
        total = 0;
        cell_used = cell;
        for (uint32_t t = 0; t < nt; t++) {
          const float *v1 = d->tri_v1 + 3 * (size_t)t, *e1 = d->tri_e1 + 3 * (size_t)t, *e2 = d->tri_e2 + 3 * (size_t)t;
          const double n[3] = {(double)e1[1] * e2[2] - (double)e1[2] * e2[1], (double)e1[2] * e2[0] - (double)e1[0] * e2[2],
                               (double)e1[0] * e2[1] - (double)e1[1] * e2[0]};
          const double nn = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
          double l1 = 0, l2 = 0;
          for (int a = 0; a < 3; a++) l1 += (double)e1[a] * e1[a], l2 += (double)e2[a] * e2[a];
          uint32_t Rr = 0;
          float* q = &recv[12 * (size_t)t];
          for (int k = 0; k < 12; k++) q[k] = 0.f;
          if (nn > 0.0 && std::isfinite(nn) && cell > 0.0) {
            Rr = (uint32_t)std::fmin(1024.0, std::fmax(1.0, std::ceil(std::sqrt(std::fmax(l1, l2)) / cell)));
            // u = (p - v1) . (e2 x n) / n.n,  v = (p - v1) . (n x e1) / n.n
            const double au[3] = {(e2[1] * n[2] - e2[2] * n[1]) / nn, (e2[2] * n[0] - e2[0] * n[2]) / nn, (e2[0] * n[1] - e2[1] * n[0]) / nn};
            const double av[3] = {(n[1] * e1[2] - n[2] * e1[1]) / nn, (n[2] * e1[0] - n[0] * e1[2]) / nn, (n[0] * e1[1] - n[1] * e1[0]) / nn};
            const double au0 = -(v1[0] * au[0] + v1[1] * au[1] + v1[2] * au[2]), av0 = -(v1[0] * av[0] + v1[1] * av[1] + v1[2] * av[2]);
            q[0] = (float)au[0], q[1] = (float)au[1], q[2] = (float)au[2], q[3] = (float)au0;
            q[4] = (float)av[0], q[5] = (float)av[1], q[6] = (float)av[2], q[7] = (float)av0;
            for (int k = 0; k < 8; k++)
              if (!std::isfinite(q[k])) Rr = 0;
            // The kernel evaluates the maps in fp32: a sliver's are ill-conditioned.  The cells are computed 5 % larger
            // than they are; keep the error of u * R, v * R below 4 % of a cell (R = 1 needs no coordinates at all).
            const double err = 4e-7 * std::fmax((std::fabs(au[0]) + std::fabs(au[1]) + std::fabs(au[2])) * pmax + std::fabs(au0),
                                                (std::fabs(av[0]) + std::fabs(av[1]) + std::fabs(av[2])) * pmax + std::fabs(av0));
            if (Rr > 1u && err * Rr > 0.04) Rr = (uint32_t)std::fmax(1.0, std::floor(0.04 / err));
          }
          const uint32_t first = (uint32_t)total;
          memcpy(&q[8], &Rr, 4), memcpy(&q[9], &first, 4);
          float* g = &geo[12 * (size_t)t];
          g[0] = v1[0], g[1] = v1[1], g[2] = v1[2], memcpy(&g[3], &Rr, 4);
          g[4] = e1[0], g[5] = e1[1], g[6] = e1[2], memcpy(&g[7], &first, 4);
          g[8] = e2[0], g[9] = e2[1], g[10] = e2[2], g[11] = 0.f;
          total += (uint64_t)Rr * Rr;
        }
        // (the flags -- 2 bytes per cell, plus this kernel input of 48 bytes per triangle -- must fit the scene's budget for optional tables)
        if (total <= (1ull << 26) && total * 2u + geo.size() * 4u <= s->budget) break;
        if (total <= nt) break;  // (one cell per triangle: coarser does not exist)
      }
      // a budget not even the coarsest flags fit: no receiver cells at all (rt_stats.notes: RT_NOTE_RECV_FLAGS_OFF_SCENE)
      const bool cells_fit = total <= (1ull << 26) && total * 2u + geo.size() * 4u <= s->budget;
      if (!cells_fit) total = 0;
      put(&s->dev.off_recv, recv.data(), recv.size() * 4);
      s->n_tri_cells = (uint32_t)total;
      {
        // sphere receivers: a cube map of directions per sphere, cells of about the same size on its surface
        std::vector<uint32_t> srecv(2 * (size_t)ns + 2, 0u);
        for (uint32_t i = 0; i < ns; i++) {
          const double r = std::sqrt(std::fabs((double)d->sphere_r_sq[i]));
          uint32_t Rs = 0;
          if (cells_fit && std::isfinite(r) && r > 0.0 && cell_used > 0.0) Rs = (uint32_t)std::fmin(256.0, std::fmax(1.0, std::ceil(1.5708 * r / cell_used)));
          if (total + 6ull * Rs * Rs > (1ull << 27) || (total + 6ull * Rs * Rs) * 2u + geo.size() * 4u > s->budget) Rs = 0;
          srecv[2 * i] = Rs, srecv[2 * i + 1] = (uint32_t)total;
          total += 6ull * Rs * Rs;
        }
        put(&s->dev.off_srecv, srecv.data(), srecv.size() * 4);
      }
      s->n_cells = (uint32_t)total;
      if (s->n_cells) {
        if ((rc = upload(s->flag_geo, geo.data(), geo.size() * 4)) != RT_OK) return bail(rc);
        if ((rc = s->flags.ensure((size_t)s->n_cells * 2 + 64)) != RT_OK) return bail(rc);
      }
    }
    put(&s->dev.off_tri_shade, shade.data(), shade.size() * 4);
    std::vector<uint32_t> ids(bvh.tri_order);
    for (uint32_t slot = 0; slot < n_slots; slot++)
      if (no_split[ids[slot] & ~RT_TRI_DUPLICATE]) ids[slot] |= RT_TRI_TRANSMISSIVE;
    put(&s->dev.off_tri_id, ids.data(), (size_t)n_slots * 4);
    put(&s->dev.off_nodes, bvh.nodes.data(), bvh.nodes.size() * sizeof(RtNode));
    {
      // per-octant copies for the soft-shadow candidate walk: planes pre-selected (lo = entry, hi = exit), the
      // child that is entered first along the octant's diagonal stored first
      const size_t nn = bvh.nodes.size();
      std::vector<RtNode> oct(8 * nn);
      for (uint32_t o = 0; o < 8; o++)
        for (size_t i = 0; i < nn; i++) {
          const RtNode& src = bvh.nodes[i];
          RtNode d0 = src;
          float key[2] = {0.f, 0.f};
          for (int a = 0; a < 3; a++) {
            const bool neg = (o >> a) & 1u;
            if (src.c0 != RT_NODE_EMPTY) {
              d0.lo0[a] = neg ? src.hi0[a] : src.lo0[a];
              d0.hi0[a] = neg ? src.lo0[a] : src.hi0[a];
              key[0] += neg ? -src.hi0[a] : src.lo0[a];
            }
            if (src.c1 != RT_NODE_EMPTY) {
              d0.lo1[a] = neg ? src.hi1[a] : src.lo1[a];
              d0.hi1[a] = neg ? src.lo1[a] : src.hi1[a];
              key[1] += neg ? -src.hi1[a] : src.lo1[a];
            }
          }
          if (src.c0 != RT_NODE_EMPTY && src.c1 != RT_NODE_EMPTY && key[1] < key[0]) {
            RtNode sw = d0;
            memcpy(sw.lo0, d0.lo1, 12), memcpy(sw.hi0, d0.hi1, 12), sw.c0 = d0.c1, sw.n0 = d0.n1;
            memcpy(sw.lo1, d0.lo0, 12), memcpy(sw.hi1, d0.hi0, 12), sw.c1 = d0.c0, sw.n1 = d0.n0;
            d0 = sw;
          }
          oct[o * nn + i] = d0;
        }
      put(&s->dev.off_nodes_oct, oct.data(), oct.size() * sizeof(RtNode));
    }
    {
      // threaded copy (depth first, skip links) for the stackless per-lane walk of incoherent wavefronts
      std::vector<RtThrNode> thr;
      struct Emit {
        const std::vector<RtNode>& nodes;
        std::vector<RtThrNode>& out;
        void child(const float* lo, const float* hi, uint32_t c, uint32_t n) {
          if (c == RT_NODE_EMPTY) return;
          const size_t idx = out.size();
          RtThrNode t;
          memcpy(t.lo, lo, 12), memcpy(t.hi, hi, 12);
          t.skip = 0;
          t.leaf = n ? ((n << 24) | c) : 0u;
          out.push_back(t);
          if (!n) node(c);
          out[idx].skip = (uint32_t)out.size();
        }
        void node(uint32_t i) {
          const RtNode nd = nodes[i];
          child(nd.lo0, nd.hi0, nd.c0, nd.n0);
          child(nd.lo1, nd.hi1, nd.c1, nd.n1);
        }
      } emit{bvh.nodes, thr};
      if (!bvh.nodes.empty()) emit.node(0);
      if (n_slots >= (1u << 24)) return bail(fail(RT_ERR_UNSUPPORTED, "more than 2^24 triangle references"));
      s->dev.n_thr = (uint32_t)thr.size();
      put(&s->dev.off_nodes_thr, thr.data(), thr.size() * sizeof(RtThrNode));
    }
  }
  {
    std::vector<float> m(12 * (size_t)d->n_materials, 0.f);
    for (uint32_t i = 0; i < d->n_materials; i++) {
      const float* r = d->materials + (size_t)i * RT_MATERIAL_STRIDE;
      float* o = &m[12 * (size_t)i];
      o[0] = r[RT_MAT_R], o[1] = r[RT_MAT_G], o[2] = r[RT_MAT_B], o[3] = r[RT_MAT_METALLIC];
      o[4] = r[RT_MAT_SHININESS], o[5] = r[RT_MAT_IOR], o[6] = r[RT_MAT_OPACITY], o[7] = r[RT_MAT_BOOST];
      o[8] = r[RT_MAT_HAS_OPACITY];
      // constants of compute_fresnel against other_ior = 1.0 (every shadow ray, raytracer.rs:64-66): the two IEEE
      // divisions of a wave-uniform material would otherwise run on the vector ALU per occluder hit per sample.
      // volatile: no host-side contraction; the same single-precision operations the kernel would execute.
      volatile float ior = r[RT_MAT_IOR], one = 1.0f;
      volatile float inv_ior = one / ior;
      volatile float q = (one - ior) / (one + ior);
      volatile float f0 = q * q;
      o[9] = inv_ior, o[10] = f0;
    }
    put(&s->dev.off_materials, m.data(), m.size() * 4);
    std::vector<float> l(8 * (size_t)d->n_lights, 0.f);
    for (uint32_t i = 0; i < d->n_lights; i++) {
      const float* r = d->lights + (size_t)i * RT_LIGHT_STRIDE;
      float* o = &l[8 * (size_t)i];
      o[0] = r[0], o[1] = r[1], o[2] = r[2], o[3] = r[6];
      o[4] = r[3], o[5] = r[4], o[6] = r[5];
    }
    put(&s->dev.off_lights, l.data(), l.size() * 4);
  }
  if ((rc = s->counters.ensure(RT_SLOTS * RT_COUNTER_REPLICAS * 16 * sizeof(unsigned long long))) != RT_OK) return bail(rc);
  s->bytes_bvh = bvh.nodes.size() * sizeof(RtNode) * 9u + (size_t)s->dev.n_thr * sizeof(RtThrNode);

  if (blob.size() >= (size_t)1 << 32) return bail(fail(RT_ERR_UNSUPPORTED, "scene data exceeds 4 GiB"));
  if ((rc = upload(s->blob, blob.data(), blob.size())) != RT_OK) return bail(rc);
  s->dev.base = (const char*)s->blob.p;
  s->dev.n_spheres = ns;
  s->dev.n_triangles = nt;
  s->dev.n_slots = n_slots;
  s->dev.n_lights = d->n_lights;
  s->dev.n_nodes = (uint32_t)bvh.nodes.size();
  s->info.n_nodes = (uint32_t)bvh.nodes.size();
  s->info.n_leaves = bvh.n_leaves;
  s->info.max_depth = bvh.max_depth;
  s->info.max_leaf_size = bvh.max_leaf;
  s->info.bytes_nodes = bvh.nodes.size() * sizeof(RtNode);
  s->info.bytes_triangles = (size_t)n_slots * (48 + 16 + 4) + (size_t)nt * 16;
  s->info.n_references = n_slots;
  if (bvh.max_depth + 2 > 64) return bail(fail(RT_ERR_UNSUPPORTED, "BVH depth %u exceeds the traversal stack", bvh.max_depth));
  *out = s;
  return RT_OK;
}

int rt_scene_bvh_info(const rt_scene* s, rt_bvh_info* out) {
  if (!s || !out) return fail(RT_ERR_INVALID_ARG, "null argument");
  *out = s->info;
  return RT_OK;
}

int rt_scene_memory_info(const rt_scene* s, rt_scene_info* out) {
  if (!s || !out) return fail(RT_ERR_INVALID_ARG, "null argument");
  memset(out, 0, sizeof(*out));
  out->bytes_bvh = s->bytes_bvh;
  out->bytes_geometry = s->blob.cap > s->bytes_bvh ? s->blob.cap - s->bytes_bvh : 0;
  out->bytes_flags = s->flags.cap + s->flag_geo.cap;
  out->bytes_cell_lists = s->cell_lists.cap;
  out->bytes_tables = s->aa.cap + s->cloud.cap + s->counters.cap + s->suplist.cap + s->costmap.cap;
  for (const auto& w : s->ws) {
    out->bytes_workspace += w.bytes();
    for (const auto& l : w.lane) out->bytes_workspace += l.qcount.cap;
  }
  out->bytes_frames = s->fb.cap + s->aux_rgb.cap + s->aux_id.cap + s->aux_t.cap + s->progress_fb.cap;
  out->bytes_total = out->bytes_geometry + out->bytes_bvh + out->bytes_flags + out->bytes_cell_lists + out->bytes_tables + out->bytes_workspace +
                     out->bytes_frames;
  out->budget_bytes = s->budget;
  out->n_receiver_cells = s->n_cells;
  out->cell_lists_built = s->cell_lists_built ? 1u : 0u;
  return RT_OK;
}

}  // extern "C"

int rt_validate_params(const rt_params* p) {
  if (!p) return fail(RT_ERR_INVALID_ARG, "null params");
  if (p->abi_version != RT_ABI_VERSION)
    return fail(RT_ERR_INVALID_ARG, "rt_params.abi_version %u != %u", p->abi_version, RT_ABI_VERSION);
  if (p->width == 0 || p->height == 0) return fail(RT_ERR_INVALID_ARG, "empty frame");
  if ((uint64_t)p->width * p->height > 0x7FFFFFFFull) return fail(RT_ERR_INVALID_ARG, "frame too large");
  if (p->win_w && (p->win_x0 + (uint64_t)p->win_w > p->width || p->win_y0 + (uint64_t)p->win_h > p->height || !p->win_h))
    return fail(RT_ERR_INVALID_ARG, "window outside the frame");
  if ((p->flags & RT_FLAG_ANTI_ALIASING) && p->aa_rays > 0 && !p->aa_offsets)
    return fail(RT_ERR_INVALID_ARG, "aa_offsets missing");
  if (p->light_mult > 1 && (!p->cloud_sets || p->n_cloud_sets == 0))
    return fail(RT_ERR_INVALID_ARG, "cloud_sets missing");
  if (p->n_ranks > 1 && p->rank >= p->n_ranks) return fail(RT_ERR_INVALID_ARG, "rank out of range");
  if (p->n_ranks > 1 && p->tile_size != 0 && p->tile_size < 16) return fail(RT_ERR_UNSUPPORTED, "tile_size < 16");
  if (p->traversal > RT_TRAVERSAL_LINEAR) return fail(RT_ERR_INVALID_ARG, "unknown traversal mode");
  if (p->max_depth_reflection > 64 || p->max_depth_refraction > 64)
    return fail(RT_ERR_UNSUPPORTED, "recursion depth > 64");
  if (p->tuning.shadow_candidate_cap > 64u && p->tuning.shadow_candidate_cap != RT_CAND_CAP_NONE)
    return fail(RT_ERR_INVALID_ARG, "tuning.shadow_candidate_cap > 64");
  if (p->tuning.chunk_log2 && (p->tuning.chunk_log2 < 10u || p->tuning.chunk_log2 > 26u))
    return fail(RT_ERR_INVALID_ARG, "tuning.chunk_log2 outside 10..26");
  if (p->tuning.sort_bits && (p->tuning.sort_bits < 12u || p->tuning.sort_bits > 24u))
    return fail(RT_ERR_INVALID_ARG, "tuning.sort_bits outside 12..24");
  if (p->tuning.sub_frames > RT_LANES) return fail(RT_ERR_INVALID_ARG, "tuning.sub_frames > %u", (unsigned)RT_LANES);
  if (p->tuning.levels > RT_LEVELS_PIPELINED) return fail(RT_ERR_INVALID_ARG, "tuning.levels > %u", (unsigned)RT_LEVELS_PIPELINED);
  if (p->tuning.phases > RT_PHASES_FUSED_DEFER) return fail(RT_ERR_INVALID_ARG, "tuning.phases > %u", (unsigned)RT_PHASES_FUSED_DEFER);
  return RT_OK;
}

extern "C" {

// fills the device parameter block, uploading tables / sizing workspaces as needed
static int prepare(rt_scene* s, const rt_params* p, uint32_t* argb_dev, const rt_aux* aux_dev, hipStream_t stream,
                   RtDevParams* P) {
  memset(P, 0, sizeof(*P));
  s->notes = 0;
  P->width = p->width;
  P->height = p->height;
  memcpy(P->focus, p->focus, sizeof(P->focus));
  P->fw = p->fw, P->fh = p->fh, P->fd = p->fd;
  P->eps_distance = p->eps_distance;
  P->air_ior = p->air_ior;
  P->ambient = p->ambient;
  P->flags = p->flags;
  const bool aa = (p->flags & RT_FLAG_ANTI_ALIASING) && p->aa_rays > 0;
  P->aa_rays = aa ? p->aa_rays : 0;
  int rc;
  bool uploaded = false;
  // a table may only be overwritten once the frames that read the old one are done, and kernels on another stream
  // may only start once the upload has landed
  auto begin_upload = [&]() -> int {
    if (!s->tables_ev) HIP_TRY(hipEventCreateWithFlags(&s->tables_ev, hipEventDisableTiming));
    // (also covers the host staging vectors below: the previous asynchronous upload has read them by now)
    if (s->tables_pending) HIP_TRY(hipStreamSynchronize(s->tables_stream));
    for (int b = 0; b < RT_SLOTS; b++)  // every frame still in flight (on whatever stream) reads the old tables
      if (s->frame_pending[b]) HIP_TRY(hipEventSynchronize(s->frame_ev[b]));
    uploaded = true;
    s->tables_version++;
    return RT_OK;
  };
  P->aa_unique = 1;
  if (aa) {
    const size_t n = p->aa_rays;
    const bool dedup = !p->tuning.no_aa_dedup;
    if (s->aa_host.size() != 2 * n || memcmp(s->aa_host.data(), p->aa_offsets, 2 * n * 4) != 0 || s->aa_dedup != dedup) {
      // Distinct offsets in first-occurrence order.  Offsets are compared as VALUES: the origin is pixel + offset,
      // and equal values (+0 / -0 included: the pixel coordinate is never -0) give bit-identical rays.
      std::vector<float> uq;
      std::vector<uint32_t> mult, src(n);
      for (size_t k = 0; k < n; k++) {
        const float x = p->aa_offsets[2 * k], y = p->aa_offsets[2 * k + 1];
        size_t j = uq.size() / 2;
        if (dedup)
          for (j = 0; j < uq.size() / 2; j++)
            if (uq[2 * j] == x && uq[2 * j + 1] == y) break;
        if (j == uq.size() / 2) uq.push_back(x), uq.push_back(y), mult.push_back(0);
        mult[j]++;
        src[k] = (uint32_t)j;
      }
      const size_t U = mult.size();
      if ((rc = begin_upload()) != RT_OK) return rc;
      s->aa_table.resize(3 * U + n);
      memcpy(s->aa_table.data(), uq.data(), 2 * U * 4);
      memcpy(s->aa_table.data() + 2 * U, mult.data(), U * 4);
      memcpy(s->aa_table.data() + 3 * U, src.data(), n * 4);
      if ((rc = s->aa.ensure(s->aa_table.size() * 4)) != RT_OK) return rc;
      HIP_TRY(hipMemcpyAsync(s->aa.p, s->aa_table.data(), s->aa_table.size() * 4, hipMemcpyHostToDevice, stream));
      s->aa_host.assign(p->aa_offsets, p->aa_offsets + 2 * n);
      s->aa_unique = (uint32_t)U;
      s->aa_dedup = dedup;
    }
    P->aa_unique = s->aa_unique;
    P->weighted = s->aa_unique != p->aa_rays;
    P->aa_offsets = (const float*)s->aa.p;
    P->aa_mult = (const uint32_t*)s->aa.p + 2 * (size_t)s->aa_unique;
    P->aa_src = (const uint32_t*)s->aa.p + 3 * (size_t)s->aa_unique;
  }
  P->light_mult = p->light_mult < 1 ? 1 : p->light_mult;
  P->cloud_seed = p->cloud_seed;
  P->n_cloud_sets = p->n_cloud_sets;
  if (P->light_mult > 1) {
    size_t n = (size_t)p->n_cloud_sets * P->light_mult * 3;
    const bool new_table = s->cloud_host.size() != n || memcmp(s->cloud_host.data(), p->cloud_sets, n * 4) != 0;
    const bool new_scale = s->cloud_ball_f[0] != p->fw || s->cloud_ball_f[1] != p->fh || s->cloud_ball_f[2] != p->fd;
    if (new_table || new_scale) {
      // The device table holds the offsets already multiplied by (fw, fh, fd) (light.rs:218: the same IEEE
      // single multiply the kernel would do, done once here), one float4 per sample position.
      if ((rc = begin_upload()) != RT_OK) return rc;
      if ((rc = s->cloud.ensure(n / 3 * 16)) != RT_OK) return rc;
      if (new_table) s->cloud_host.assign(p->cloud_sets, p->cloud_sets + n);
      s->cloud_ball[3] = -1.f;  // recompute the bounding ball
      s->cloud_ball_f[0] = p->fw, s->cloud_ball_f[1] = p->fh, s->cloud_ball_f[2] = p->fd;
      const float f[3] = {p->fw, p->fh, p->fd};
      s->cloud_scaled.assign(n / 3 * 4, 0.0f);
      for (size_t i = 0; i < n; i++) s->cloud_scaled[i / 3 * 4 + i % 3] = s->cloud_host[i] * f[i % 3];
      HIP_TRY(hipMemcpyAsync(s->cloud.p, s->cloud_scaled.data(), n / 3 * 16, hipMemcpyHostToDevice, stream));
    }
    P->cloud_sets = (const float4*)s->cloud.p;
    // bounding ball of the offsets cs * (fw, fh, fd) over all sets (cached with the table)
    if (s->cloud_ball[3] < 0.f || new_scale) {
      float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
      const float f[3] = {p->fw, p->fh, p->fd};
      for (size_t i = 0; i < n; i++) {
        float v = s->cloud_host[i] * f[i % 3];
        lo[i % 3] = std::fmin(lo[i % 3], v);
        hi[i % 3] = std::fmax(hi[i % 3], v);
      }
      float r2 = 0.f;
      for (int a = 0; a < 3; a++) {
        s->cloud_ball[a] = 0.5f * (lo[a] + hi[a]);
        float h = 0.5f * (hi[a] - lo[a]);
        r2 += h * h;
      }
      s->cloud_ball[3] = std::sqrt(r2) * 1.001f + 1e-6f;
    }
    memcpy(P->cloud_centre, s->cloud_ball, 12);
    P->cloud_delta = s->cloud_ball[3];
    {
      const float e = p->eps_distance;
      P->beam_delta = P->cloud_delta + 2.0f * e;
      P->beam_delta_e5 = P->beam_delta + 1e-5f;
      P->beam_eps_push = 0.998f * e;
      P->beam_eps_ulp = (1.3e-7f + 2.5e-6f) * e;
      P->beam_eps_o = 1.01f * e + 2.0f * P->beam_eps_ulp;
      P->beam_eps_198 = 1.98f * e;
    }
    const uint32_t cap = p->tuning.shadow_candidate_cap;
    P->cand_cap = cap == RT_CAND_CAP_NONE ? 0u : (cap ? cap : 64u);
    // receiver flags: cells no triangle / sphere can shadow for a light skip the candidate walk (rt_flags_kernel)
    P->recv_flags = nullptr;
    // (why the flags are off, when they are: rt_stats.notes)
    if (p->tuning.no_receiver_flags || P->cand_cap == 0u) s->notes |= RT_NOTE_RECV_FLAGS_OFF_TUNING;
    if (!s->n_cells || !(P->cloud_delta > 0.0f)) s->notes |= RT_NOTE_RECV_FLAGS_OFF_SCENE;
    if (s->dev.n_lights > 8u) s->notes |= RT_NOTE_RECV_FLAGS_OFF_LIGHTS;
    if (p->traversal != RT_TRAVERSAL_BVH) s->notes |= RT_NOTE_RECV_FLAGS_OFF_TRAVERSAL;
    if (p->flags & RT_FLAG_BACKFACE_CULLING) s->notes |= RT_NOTE_RECV_FLAGS_OFF_CULLING;
    if (!p->tuning.no_receiver_flags && P->cand_cap != 0u && s->n_cells && s->dev.n_lights <= 8u && p->traversal == RT_TRAVERSAL_BVH &&
        !(p->flags & RT_FLAG_BACKFACE_CULLING) && P->cloud_delta > 0.0f) {
      const float key[8] = {P->beam_delta, p->eps_distance, P->cloud_centre[0], P->cloud_centre[1], P->cloud_centre[2], 1.f, 0.f, 0.f};
      if (memcmp(key, s->flags_key, sizeof(key)) != 0) {
        if ((rc = begin_upload()) != RT_OK) return rc;  // (waits for kernels of an earlier frame that read the old flags)
        RtDevParams B = *P;
        B.flag_out = (uint16_t*)s->flags.p;
        B.flag_geo = (const float4*)s->flag_geo.p;
        B.n_cells = s->n_cells;
        B.n_tri_cells = s->n_tri_cells;
        // per-cell candidate lists, written by the same kernel: 16 bytes per cell and light (16-bit leaf slots), when the
        // scene allows it and a quarter of the free memory holds them
        s->cell_lists_built = false;
        B.cell_list_out = nullptr;
        const size_t list_bytes = (size_t)s->n_cells * s->dev.n_lights * 16u;
        size_t free_b = 0, total_b = 0;
        const size_t flag_bytes = s->flags.cap + s->flag_geo.cap;
        if (s->dev.n_slots <= 65533u && flag_bytes + list_bytes <= s->budget && hipMemGetInfo(&free_b, &total_b) == hipSuccess &&
            list_bytes <= (free_b + s->cell_lists.cap) / 4 && s->cell_lists.ensure(list_bytes + 64) == RT_OK) {
          HIP_TRY(hipMemsetAsync(s->cell_lists.p, 0xFF, list_bytes, stream));
          B.cell_list_out = (uint16_t*)s->cell_lists.p;
          s->cell_lists_built = true;
        }
        if (!s->cell_lists_built) s->cell_lists.release();  // (a table built under an earlier, larger need)
        hipError_t e = (hipError_t)rt_launch_flags(s->dev, B, stream);
        if (e != hipSuccess) return fail(RT_ERR_HIP, "rt_flags_kernel launch failed: %s", hipGetErrorString(e));
        trace_point(stream, "rt_flags_kernel: cells, lights, lists", s->n_cells, s->dev.n_lights, s->cell_lists_built ? 1u : 0u);
        memcpy(s->flags_key, key, sizeof(key));
      }
      P->recv_flags = (const uint16_t*)s->flags.p;
      if (s->cell_lists_built && !p->tuning.no_cell_lists) P->cell_lists = (const uint4*)s->cell_lists.p;
    }
    if (!P->cell_lists) s->notes |= RT_NOTE_CELL_LISTS_OFF;
  }
  P->max_depth_reflection = p->max_depth_reflection;
  P->max_depth_refraction = p->max_depth_refraction;
  s->sort_bits_wanted = p->tuning.sort_bits;
  s->lanes_wanted = p->tuning.sub_frames;
  s->phases_wanted = p->tuning.phases;
  s->levels_wanted = p->tuning.levels;
  if (p->win_w) {
    P->win_x0 = p->win_x0, P->win_y0 = p->win_y0, P->win_w = p->win_w, P->win_h = p->win_h;
  } else {
    P->win_x0 = P->win_y0 = 0;
    P->win_w = p->width;
    P->win_h = p->height;
  }
  P->tile_size = p->tile_size ? p->tile_size : 48u;
  P->n_ranks = p->n_ranks;
  P->rank = p->rank;
  P->traversal = p->traversal;
  P->argb = argb_dev;
  if (aux_dev) {
    P->aux_rgb = aux_dev->rgb;
    P->aux_hit_id = aux_dev->hit_id;
    P->aux_hit_t = aux_dev->hit_t;
  }
  {
    // this frame's slot (counter block + workspace set): one whose last frame has finished if there is one, else the one
    // used longest ago -- whose frame this stream then waits for
    // (a slot whose last frame ran on THIS stream is taken first: the stream orders the two frames anyway, and a host that
    // runs far ahead of the GPU on two streams then holds two workspace sets, not one per slot)
    int blk = -1, oldest = 0, same = -1;
    for (int b = 0; b < RT_SLOTS; b++) {
      if (!s->frame_ev[b]) HIP_TRY(hipEventCreateWithFlags(&s->frame_ev[b], hipEventDisableTiming));
      if (s->frame_pending[b] && hipEventQuery(s->frame_ev[b]) == hipSuccess) s->frame_pending[b] = false;
      if (blk < 0 && !s->frame_pending[b]) blk = b;
      if (same < 0 && s->frame_seq[b] && s->frame_stream[b] == stream) same = b;
      if (s->frame_seq[b] < s->frame_seq[oldest]) oldest = b;
    }
    if (same >= 0 && s->frame_pending[same]) blk = same;  // still running: queue up behind it on its stream
    if (blk < 0) blk = oldest;
    if (s->frame_pending[blk] && s->frame_stream[blk] != stream) HIP_TRY(hipStreamWaitEvent(stream, s->frame_ev[blk], 0));
    s->frame_stream[blk] = stream;
    s->frame_seq[blk] = ++s->frame_no;
    s->cur_block = blk;
    unsigned long long* blk_p = (unsigned long long*)s->counters.p + (size_t)blk * RT_COUNTER_REPLICAS * 16;
    P->counters = p->tuning.no_counters ? nullptr : blk_p;
    HIP_TRY(hipMemsetAsync(blk_p, 0, RT_COUNTER_REPLICAS * 16 * sizeof(unsigned long long), stream));
  }

  const bool aa_on = P->aa_rays > 0;
  if (aa_on && P->aa_rays > 256) return fail(RT_ERR_UNSUPPORTED, "aa_rays > 256");
  for (int a = 0; a < 3; a++) {
    const float ext = s->aabb_hi[a] - s->aabb_lo[a];
    P->morton_lo[a] = s->aabb_lo[a] - 0.01f * ext;
    P->morton_scale[a] = ext > 0.f ? 1024.0f / (1.02f * ext) : 0.f;
  }
  // The super-tiles (16x16 pixels) this launch renders, in launch order.  Multi-GPU: only those that hold pixels of this
  // rank's tiles.  RT_TILE_ORDER_COST: heaviest first, by the cost map measured on a calibration frame of this shape.
  P->sup_list = nullptr;
  P->n_sup = ((P->win_w + 15u) / 16u) * ((P->win_h + 15u) / 16u);
  uint32_t order = p->tuning.tile_order == RT_TILE_ORDER_COST ? RT_TILE_ORDER_COST : RT_TILE_ORDER_ROW_MAJOR;
  if (order == RT_TILE_ORDER_COST && !rt_has_cost_kernel()) order = RT_TILE_ORDER_ROW_MAJOR, s->notes |= RT_NOTE_TILE_ORDER_COST_OFF;
  s->cost_wanted = false;
  if (order == RT_TILE_ORDER_COST) {
    // (the calibration frame times the super-tiles THIS rank owns, and only its primary kernel: secondary levels are not part of the cost)
    const uint32_t ck[13] = {P->win_x0, P->win_y0, P->win_w, P->win_h, P->width, P->height, P->flags, P->aa_rays, P->light_mult,
                             p->max_depth_reflection | (p->max_depth_refraction << 8) | (p->traversal << 16), P->n_ranks, P->rank, P->tile_size};
    if (memcmp(ck, s->cost_key, sizeof(ck)) != 0) s->cost_valid = false, memcpy(s->cost_key, ck, sizeof(ck));
    s->cost_wanted = !s->cost_valid;  // the caller runs the calibration frame (calibrate_costs)
  }
  if (P->n_ranks > 1 || (order == RT_TILE_ORDER_COST && s->cost_valid)) {
    const uint32_t key[8] = {P->win_x0, P->win_y0, P->win_w, P->win_h, P->tile_size, P->n_ranks, P->rank,
                             order == RT_TILE_ORDER_COST && s->cost_valid ? 2u : 1u};
    if (memcmp(key, s->sup_key, sizeof(key)) != 0 || s->sup_host.empty()) {
      if ((rc = begin_upload()) != RT_OK) return rc;
      s->sup_host.clear();
      const uint32_t st_x = (P->win_w + 15u) / 16u, st_y = (P->win_h + 15u) / 16u;
      for (uint32_t sy = 0; sy < st_y; sy++)
        for (uint32_t sx = 0; sx < st_x; sx++) {
          // a 16x16 super-tile spans at most 2 tiles per axis (tile_size >= 16): its corners decide
          uint32_t x0 = P->win_x0 + sx * 16u, y0 = P->win_y0 + sy * 16u;
          uint32_t x1 = x0 + 15u < P->win_x0 + P->win_w - 1u ? x0 + 15u : P->win_x0 + P->win_w - 1u;
          uint32_t y1 = y0 + 15u < P->win_y0 + P->win_h - 1u ? y0 + 15u : P->win_y0 + P->win_h - 1u;
          bool own = P->n_ranks <= 1;
          for (uint32_t yy : {y0, y1})
            for (uint32_t xx : {x0, x1})
              own = own || rt_tile_owner(xx / P->tile_size, yy / P->tile_size, P->n_ranks) == P->rank;
          if (own) s->sup_host.push_back(sy * st_x + sx);
        }
      if (key[7] == 2u && s->cost_host.size() == (size_t)st_x * st_y)
        std::stable_sort(s->sup_host.begin(), s->sup_host.end(), [&](uint32_t a, uint32_t b) { return s->cost_host[a] > s->cost_host[b]; });
      memcpy(s->sup_key, key, sizeof(key));
      if ((rc = s->suplist.ensure(s->sup_host.size() * 4 + 4)) != RT_OK) return rc;
      HIP_TRY(hipMemcpyAsync(s->suplist.p, s->sup_host.data(), s->sup_host.size() * 4, hipMemcpyHostToDevice, stream));
    }
    P->sup_list = (const uint32_t*)s->suplist.p;
    P->n_sup = (uint32_t)s->sup_host.size();
  }
  if (uploaded) {
    HIP_TRY(hipEventRecord(s->tables_ev, stream));
    s->tables_stream = stream;
    s->tables_pending = true;
  } else if (s->tables_pending && s->tables_stream != stream) {
    HIP_TRY(hipStreamWaitEvent(stream, s->tables_ev, 0));
  }
  s->last_stream = stream;
  s->rendered = true;
  return RT_OK;
}

// ---- frame scheduler -------------------------------------------------------------------------------
// Without secondary rays a frame is ONE launch of the primary kernel.  With reflections / refractions every child ray
// becomes an independent work item in HBM ("ray streaming"): the reference's recursion (single_raytrace,
// raytracer_renderer.rs:147-264: a node spawns calculate_reflection :526-729 and calculate_refractions :279-524) is run
// LEVEL BY LEVEL.  Level k reads one of two ray queues and appends its children to the other:
//     primary -> [hard pairs] -> for k = 1 .. depth:  trace(k) -> sort(k) -> shade(k) -> [hard pairs]  -> resolve
// Every launch takes its size from the device (the counter the launch before it wrote) and walks it with a grid-stride
// loop, so the host never waits for a level: a steady-state frame is enqueued without one synchronisation.  What the
// host contributes is a GUESS of each grid -- the counts of the previous frame of the same shape, read back
// asynchronously -- and the queue sizes.  Queues are sized by need: the first frame of a shape runs with a generous
// estimate and is verified (one synchronisation at its end: were children or pairs dropped?); if so the queues grow to
// what the counters say was needed and the frame is rendered again.  A verified shape renders asynchronously from then
// on (the scene is static, so its ray counts repeat exactly).  Pixel sums use a fixed-point accumulator (order
// independent, hence bit-reproducible), resolved to packed pixels by a last kernel.
static const size_t RT_QUEUE_BUDGET = (size_t)160 << 30;  // hard ceiling; the real limit is half of the free HBM
#define RT_CNT_OVERFLOW 0u                       // dropped children
#define RT_CNT_LEVEL(k) (k)                      // 1 .. levels + 1: rays appended to level k (dropped ones included)
#define RT_CNT_HARD(levels) ((levels) + 2u)      // hard pairs waiting
#define RT_CNT_HARD_STAT(levels) ((levels) + 3u) // [0] dropped pairs, [1] largest batch of pairs
#define RT_CNT_HITS(levels, k) ((levels) + 5u + (k))  // rays of level k that hit something
#define RT_CNT_SETS(levels, k, c) (2u * (levels) + 8u + 3u * (k) + (c))  // phase-split pipeline: sets of class c at level k = 0 .. levels
#define RT_CNT_SEG(levels, k) (2u * (levels) + 8u + 3u * ((levels) + 1u) + (k))  // merged levels: first queue index of level k = 1 .. levels + 1
#define RT_CNT_TOTAL(levels) (2u * (levels) + 8u + 3u * ((levels) + 1u) + (levels) + 2u)

// Diagnostics: RT_TRACE_LAUNCHES=1 in the environment makes every launch of a frame wait for its kernel and report it on
// stderr (which launch of which level does not come back, with which sizes); never set in timed runs.
static bool trace_launches() {
  static const bool on = [] {
    const char* v = getenv("RT_TRACE_LAUNCHES");
    return v && *v && *v != '0';
  }();
  return on;
}
static void trace_point(hipStream_t stream, const char* what, uint32_t a, uint32_t b, uint32_t c) {
  if (!trace_launches()) return;
  fprintf(stderr, "[rt_hip] %s (%u, %u, %u) launched ...", what, a, b, c);
  fflush(stderr);
  const hipError_t e = hipStreamSynchronize(stream);
  fprintf(stderr, " %s\n", e == hipSuccess ? "done" : hipGetErrorString(e));
  fflush(stderr);
}

static uint32_t grid_for(uint64_t items, uint32_t per_wg, uint32_t cap_wgs) {
  uint64_t w = (items + items / 16u + per_wg - 1u) / per_wg + 8u;  // a little above the guess; the loop covers the rest
  if (w > cap_wgs) w = cap_wgs;
  return w ? (uint32_t)w : 1u;
}

static int render_frame_impl(rt_scene* s, RtDevParams& P, hipStream_t stream, uint32_t forced_chunk_log2, bool blocking);

// how the levels of a ray tree are run (rt_tuning.levels)
// (measured, round 4: MERGED config 4 40.6 ms alone / 39.3 with two frames in flight against 49.8 / 42.6 CHAINED, config 5 120.4 / 118.4
// against 126.4 / 122.3; PIPELINED 47.7 / 42.9 and 129.4 / 123.3: one hit-point order over all levels is worth more than the overlap)
static uint32_t rt_levels_mode(uint32_t wanted) {
  if (wanted == RT_LEVELS_DEFAULT) return RT_LEVELS_MERGED;
  return wanted;
}

// which form of the render loop a frame takes (rt_tuning.phases; RT_PHASES_DEFAULT: by frame shape)
static bool rt_use_phases(uint32_t wanted, const RtDevParams& P, bool secondary) {
  if (P.cost_map) return false;  // (the calibration frame of RT_TILE_ORDER_COST times the fused primary kernel)
  if (wanted == RT_PHASES_SPLIT) return true;
  if (wanted == RT_PHASES_FUSED) return false;
  (void)secondary;
  return false;
}

// A frame that fails half-way (HIP / launch error, out of memory) leaves partial sums in the pixel accumulator: mark
// the accumulator dirty so that the next frame clears it.
static int render_frame(rt_scene* s, RtDevParams& P, hipStream_t stream, uint32_t forced_chunk_log2, bool blocking = false) {
  const int rc = render_frame_impl(s, P, stream, forced_chunk_log2, blocking);
  if (rc != RT_OK) s->ws[s->cur_ws].acc_pixels = 0;
  // marks the end of this frame's use of its counter block (prepare() of a later frame waits for it)
  if (hipEventRecord(s->frame_ev[s->cur_block], stream) == hipSuccess) s->frame_pending[s->cur_block] = true;
  s->last_block = s->cur_block;
  return rc;
}

static int render_frame_impl(rt_scene* s, RtDevParams& P, hipStream_t stream, uint32_t forced_chunk_log2, bool blocking) {
  const bool secondary = (P.flags & (RT_FLAG_REFLECTIONS | RT_FLAG_REFRACTIONS)) != 0;
  const uint32_t total_wgs = rt_primary_total_wgs(P);
  s->queue_bytes = 0;
  // The phase-split pipeline (rt_phases.h; rt_tuning.phases): hit -> classify -> one kernel per class of (wavefront, light)
  // set -> resolve, instead of the fused kernels.  Frames without secondary rays then also sum through the accumulator.
  const bool split = rt_use_phases(s->phases_wanted, P, secondary);
  P.resolve_counts_written = split ? 1u : 0u;
  // RT_PHASES_FUSED_DEFER: the fused kernels, but a frame without secondary rays also sums through the accumulator, so that its
  // incoherent (wavefront, light) sets can be deferred to rt_hard_kernel like those of a frame with secondary rays
  const bool defer = !secondary && !split && !P.cost_map && s->phases_wanted == RT_PHASES_FUSED_DEFER && P.light_mult > 1;
  if (!secondary && !split && !defer) {
    P.acc = nullptr;
    P.q_out = nullptr;
    P.batch_first_wg = 0, P.batch_stride = 1, P.batch_group_log2 = 0;
    hipError_t e = (hipError_t)rt_launch_primary(s->dev, P, total_wgs, stream);
    if (e != hipSuccess) return fail(RT_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    trace_point(stream, "rt_primary_kernel: workgroups", total_wgs);
    return RT_OK;
  }
  int rc;
  if (total_wgs == 0) return RT_OK;  // this rank owns no tile inside the window (more ranks than tiles): nothing to trace, nothing to resolve
  const uint32_t levels = !secondary ? 0u : (P.max_depth_reflection > P.max_depth_refraction ? P.max_depth_reflection : P.max_depth_refraction);
  if (secondary && levels == 0) return fail(RT_ERR_INVALID_ARG, "secondary rays enabled with depth 0");
  const size_t npix = (size_t)P.width * P.height;
  const uint64_t items = (uint64_t)total_wgs * 256u;  // primary work items (threads) of the frame
  // Soft-shadow sets of incoherent wavefronts are deferred to rt_hard_kernel as (hit point, light) pairs
  const bool hard = (secondary || split || defer) && P.light_mult > 1 && P.light_mult <= 64 && P.traversal == RT_TRAVERSAL_BVH && s->dev.n_triangles && P.cand_cap != 0;
  if (!hard && P.light_mult > 1) s->notes |= RT_NOTE_HARD_PAIRS_OFF;
  // merged levels (rt_tuning.levels): every level traced first (the trace kernel appends the children), then ONE sort and ONE shade launch
  uint32_t levels_mode = (secondary && !split) ? rt_levels_mode(s->levels_wanted) : RT_LEVELS_CHAINED;
  if (levels_mode == RT_LEVELS_MERGED && s->levels_wanted == RT_LEVELS_DEFAULT) {
    // The merged queue holds every level of the tree at once (config 4: 5.0 GB against the 2.9 GB of two alternating queues).  Where a
    // third of the free memory (plus what this scene's workspaces already hold) does not take about four times the primary work items,
    // the library's choice is the chained schedule rather than a frame cut into batches.
    size_t free_b = 0, total_b = 0, held = 0;
    for (const auto& o : s->ws) held += o.bytes();
    const uint64_t want = (uint64_t)total_wgs * 256u * 4u * (64u + 12u);
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || want > (free_b + held) / 3u) levels_mode = RT_LEVELS_CHAINED;
  }
  const bool pipelined = levels_mode == RT_LEVELS_PIPELINED;
  const bool merged = levels_mode == RT_LEVELS_MERGED || pipelined;  // (one append-only queue per chain, children spawned by the trace kernel)
  const uint32_t n_cnt = RT_CNT_TOTAL(levels);
  // ---- chains.  The ray tree of a frame is a chain of launches, one per level, each with a drain of its own (a launch
  // cannot end before its longest wavefront does).  A frame that has the GPU to itself is therefore split into two
  // interleaved halves of its primary work-group list that run as independent chains -- own queues, own counters, own
  // stream -- and meet in the pixel accumulator: the head of one chain's launch fills the drain of the other's
  // (rt_tuning.sub_frames; a frame alone: config 4 51.8 -> 49.5 ms, at depth 21 102.7 -> 92.2).  When the host keeps
  // frames in flight itself the other FRAME is the better filler (42.3 ms against 43.3 with chains on top): sub_frames = 0
  // uses two chains only while no other frame of the scene is running (and stays with one for 8 frames after the last
  // overlap, so that a pipeline that drains now and then does not flip -- every flip re-verifies the queue sizes).
  uint32_t lanes = s->lanes_wanted ? std::min<uint32_t>(s->lanes_wanted, RT_LANES) : RT_LANES;
  if (!s->lanes_wanted) {
    bool busy = false;
    for (int b = 0; b < RT_SLOTS; b++)
      if (b != s->cur_block && s->frame_pending[b] && hipEventQuery(s->frame_ev[b]) == hipErrorNotReady) busy = true;
    s->calm_frames = busy ? 0u : std::min<uint32_t>(s->calm_frames + 1u, 1u << 30);
    if (s->calm_frames < 8u) lanes = 1;
  }
  if (forced_chunk_log2 || items < (1ull << 16) || (!secondary && !s->lanes_wanted)) lanes = 1;
  if (merged && !s->lanes_wanted) lanes = 1;  // (measured: two chains add nothing once the levels share one shade launch)  // (forced batch sizes: the batching itself is under test; tiny frames: nothing to overlap)
  // This frame's workspace set: the one of its slot -- unless that would mean ALLOCATING a second set on a device that
  // cannot spare the memory (a partitioned or shared GPU): then the frame waits for the frame that uses set 0 and takes it.
  int wsi = s->cur_block;
  if (wsi > 0 && !s->ws[wsi].lane[0].queues.p && s->ws[0].lane[0].queues.p) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < 3 * s->ws[0].bytes()) wsi = 0;
  }
  if (s->ws_last_block[wsi] != s->cur_block && s->frame_pending[s->ws_last_block[wsi]])
    HIP_TRY(hipStreamWaitEvent(stream, s->frame_ev[s->ws_last_block[wsi]], 0));
  s->ws_last_block[wsi] = s->cur_block;
  s->cur_ws = wsi;
  rt_scene::StreamWs& w = s->ws[wsi];
  for (uint32_t j = 0; j < lanes; j++) {
    if ((rc = w.lane[j].qcount.ensure((size_t)n_cnt * 4)) != RT_OK) return rc;
    if (j && !w.lane[j].stream) HIP_TRY(hipStreamCreateWithFlags(&w.lane[j].stream, hipStreamNonBlocking));
    if (j && !w.lane[j].done_ev) HIP_TRY(hipEventCreateWithFlags(&w.lane[j].done_ev, hipEventDisableTiming));
  }
  for (uint32_t j = lanes; j < RT_LANES; j++)  // memory by need: a set that runs one chain does not keep the other chain's queues
    if (w.lane[j].queues.p || w.lane[j].trace_ws.p || w.lane[j].hard.p || w.lane[j].hitrec.p || w.lane[j].sets.p) {
      if (w.lane[j].stream) HIP_TRY(hipStreamSynchronize(w.lane[j].stream));
      HIP_TRY(hipStreamSynchronize(stream));  // (behind the wait for the set's last frame enqueued above)
      w.lane[j].queues.release(), w.lane[j].trace_ws.release(), w.lane[j].hard.release(), w.lane[j].hitrec.release(), w.lane[j].sets.release();
      w.lane[j].sort_hist_clean = nullptr;
    }
  if (!w.cnt_host) HIP_TRY(hipHostMalloc((void**)&w.cnt_host, RT_LANES * RT_CNT_STRIDE * 4, hipHostMallocDefault));
  if (!w.cnt_ev) HIP_TRY(hipEventCreateWithFlags(&w.cnt_ev, hipEventDisableTiming));
  if (!w.fork_ev) HIP_TRY(hipEventCreateWithFlags(&w.fork_ev, hipEventDisableTiming));

  // ---- the shape of this frame: what its ray counts depend on.  Same key as the last verified frame = same counts.
  StreamKey key;
  memset(&key, 0, sizeof(key));
  key.width = P.width, key.height = P.height, key.flags = P.flags, key.aa_rays = P.aa_rays, key.aa_unique = P.aa_unique;
  key.light_mult = P.light_mult, key.depth_refl = P.max_depth_reflection, key.depth_refr = P.max_depth_refraction;
  key.win[0] = P.win_x0, key.win[1] = P.win_y0, key.win[2] = P.win_w, key.win[3] = P.win_h;
  key.tile_size = P.tile_size, key.n_ranks = P.n_ranks, key.rank = P.rank, key.traversal = P.traversal, key.cand_cap = P.cand_cap;
  key.cloud_seed = P.cloud_seed, key.n_cloud_sets = P.n_cloud_sets, key.forced = forced_chunk_log2;
  key.tables = s->tables_version;
  memcpy(key.f, P.focus, 12), key.f[3] = P.fw, key.f[4] = P.fh, key.f[5] = P.fd, key.f[6] = P.eps_distance, key.f[7] = P.air_ior;
  key.staged = P.stage_slot != nullptr, key.flags_on = P.recv_flags != nullptr, key.n_sup = P.n_sup, key.lanes = lanes;
  key.merged = levels_mode;
  key.split = (split ? 1u : 0u) | (defer ? 2u : 0u), key.sort_bits = s->sort_bits_wanted, key.lists_on = P.cell_lists != nullptr;
  if (memcmp(&key, &s->stream_key, sizeof(key)) != 0) {
    s->stream_key = key;
    s->key_gen++;
    s->stream_verified = false;
    s->est_valid = false;
    s->q_cap = s->hard_cap = s->batch_items = 0;
  }
  // the counts of an earlier frame of this shape, if their read-back has landed
  for (auto& o : s->ws)
    if (o.cnt_pending && hipEventQuery(o.cnt_ev) == hipSuccess) {
      o.cnt_pending = false;
      if (o.cnt_key_gen != s->key_gen) continue;  // (the counts of another frame shape say nothing about this one)
      // A verified shape renders without waiting for its counters -- but how many pairs a frame defers depends on how its rays
      // were packed into wavefronts (atomic order in the sort), so a later frame can need a little more than the verified one
      // did.  If a frame dropped anything, say so (rt_stats.notes) and verify -- i.e. size and, if needed, render again -- the next.
      bool dropped_any = false;
      for (uint32_t j = 0; j < o.cnt_host_lanes; j++) {
        const uint32_t* c = o.cnt_host + (size_t)j * RT_CNT_STRIDE;
        if (c[RT_CNT_OVERFLOW] || c[RT_CNT_HARD_STAT(o.cnt_host_levels)]) {
          dropped_any = true;
          s->hard_cap = std::max<uint32_t>(s->hard_cap, (uint32_t)std::min<uint64_t>((uint64_t)c[RT_CNT_HARD_STAT(o.cnt_host_levels) + 1u] * 5u / 4u + 256u, 0xFFFFFF00ull));
        }
      }
      if (dropped_any) {
        s->stream_verified = false, s->est_valid = false;
        s->sticky_notes |= RT_NOTE_FRAME_DROPPED_WORK;
        continue;
      }
      if (o.cnt_host_levels == levels && o.cnt_host_lanes == lanes && o.cnt_host_valid) memcpy(s->est, o.cnt_host, sizeof(s->est)), s->est_valid = true;
    }

  // chains other than the caller's must have drained before this function returns on an error path (their work refers to
  // the workspace set; only the caller's stream is guarded by the frame event)
  struct Joiner {
    rt_scene::StreamWs& w;
    uint32_t lanes;
    bool forked = false, joined = false;
    bool shading = false;  // pipelined levels: work is in flight on the chains' shade streams
    ~Joiner() {
      if (forked && !joined)
        for (uint32_t j = 1; j < lanes; j++) (void)hipStreamSynchronize(w.lane[j].stream);
      if (shading)
        for (uint32_t j = 0; j < lanes; j++)
          for (int k = 0; k < 2; k++)
            if (w.lane[j].shade_stream[k]) (void)hipStreamSynchronize(w.lane[j].shade_stream[k]);
    }
  } joiner{w, lanes};

  for (int attempt = 0;; attempt++) {
    // ---- sizes, per chain.  Unknown shape: every level fits the chain's primary work items (children usually thin out; a
    // scene where they multiply is caught by the verification below), pairs = 1/8 of that.  Budget: half of the free HBM.
    if (!s->q_cap) {
      if (forced_chunk_log2) {
        s->batch_items = 1u << forced_chunk_log2;
        s->q_cap = 2u * s->batch_items;
      } else {
        const uint64_t lane_items = ((uint64_t)total_wgs + lanes - 1u) / lanes * 256u;
        s->batch_items = (uint32_t)std::min<uint64_t>(lane_items, 1ull << 28);
        // (merged levels: ONE queue of 2 q_cap rays holds every level of the tree -- three times the primary work items as a first guess)
        s->q_cap = merged ? (uint32_t)std::min<uint64_t>((uint64_t)s->batch_items * 3u / 2u, 0x7FFFFF00ull) : s->batch_items;
      }
      if (s->q_cap < (1u << 16)) s->q_cap = 1u << 16;
      s->hard_cap = hard ? std::max<uint32_t>(s->q_cap / 8u, 1u << 16) : 0u;
    }
    size_t budget = RT_QUEUE_BUDGET, free_b = 0, total_b = 0;
    const size_t held = w.bytes() - w.acc.cap;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = std::min(budget, (size_t)((double)(free_b + held) * 0.5));
    // (phase-split pipeline: 8 bytes of hit record per primary work item of a batch + the (wavefront, light) set records, dense by
    // set id: 32-byte header, 64-dword candidate list, a slot in each of the three class queues)
    auto set_cap_for = [&](uint64_t q, uint64_t batch) { return (uint32_t)(((std::max<uint64_t>(levels ? q : 0u, batch + 64u * 256u) / 64u + 4u) * std::max<uint32_t>(s->dev.n_lights, 1u) + 15u) & ~7ull); };
    auto bytes_for = [&](uint64_t q, uint64_t h) {
      size_t b = levels ? (size_t)(q * (2u * 64u + (merged ? 24u : 12u)) + (h ? (h + 64u) * 64u : 0u)) : 0u;
      if (split) b += (size_t)(s->batch_items + 64u * 256u) * 8u + (size_t)set_cap_for(q, s->batch_items) * (32u + 256u + 12u + 1u);
      return (size_t)lanes * b;
    };
    while (bytes_for(s->q_cap, s->hard_cap) > budget && s->q_cap > (1u << 16)) {
      // does not fit: smaller primary batches, queues and pair buffer in proportion
      s->q_cap = std::max<uint32_t>(s->q_cap / 2u, 1u << 16);
      s->hard_cap = hard ? std::max<uint32_t>(s->hard_cap / 2u, 1u << 16) : 0u;
      s->batch_items = std::max<uint32_t>(s->batch_items / 2u, 1u << 10);
      s->stream_verified = false;
    }
    s->batch_items = std::max<uint32_t>((s->batch_items + 255u) / 256u * 256u, 256u);
    s->q_cap = (uint32_t)std::min<uint64_t>(((uint64_t)s->q_cap + 255u) / 256u * 256u, 0xFFFFFF00ull);  // (16-byte aligned arrays behind it)
    // (more rays per level = more rays per bucket: two more key bits for 4K-sized frames: config 5 136.2 -> 133.4 ms)
    // (merged levels sort the rays of every level at once, three times a level's: config 4 20 / 22 / 23 / 24 bits = 41.3 / 39.9 / 39.3 / 38.9 ms)
    P.sort_bits = s->sort_bits_wanted ? s->sort_bits_wanted : ((merged || items >= (32ull << 20)) ? RT_SORT_BITS_DEFAULT + 2u : RT_SORT_BITS_DEFAULT);
    const uint32_t n_buckets = 1u << P.sort_bits;
    rc = RT_OK;
    // (merged levels: the two queues of q_cap rays are ONE queue of 2 q_cap -- same memory -- and the sort workspace covers all of it)
    const uint32_t q_sort_cap = merged ? 2u * s->q_cap : s->q_cap;
    const uint32_t set_cap = split ? set_cap_for(s->q_cap, s->batch_items) : 0u;
    const size_t hitrec_items = (size_t)s->batch_items + 64u * 256u;  // (an interleaved chain's launch is rounded up to whole groups)
    for (uint32_t j = 0; j < lanes && rc == RT_OK; j++) {
      if (levels) {
        rc = w.lane[j].queues.ensure((size_t)2 * s->q_cap * RT_QUEUE_QUADS * sizeof(float4));
        if (rc == RT_OK) rc = w.lane[j].trace_ws.ensure((size_t)q_sort_cap * 12 + (size_t)n_buckets * 8 + (n_buckets / RT_SORT_TILE) * 4 + 256);
      }
      if (rc == RT_OK && hard) rc = w.lane[j].hard.ensure(((size_t)s->hard_cap + 64u) * 4u * sizeof(float4));
      if (rc == RT_OK && (split || merged)) rc = w.lane[j].hitrec.ensure(hitrec_items * 8u);
      if (rc == RT_OK && split) rc = w.lane[j].sets.ensure((size_t)set_cap * (32u + 256u + 12u + 1u) + 256u);
    }
    if (rc == RT_ERR_OOM && s->q_cap > (1u << 16) && attempt < 12) {
      s->q_cap /= 2u, s->hard_cap = hard ? std::max<uint32_t>(s->hard_cap / 2u, 1u << 16) : 0u;
      s->batch_items = std::max<uint32_t>(s->batch_items / 2u, 1u << 10);
      continue;
    }
    if (rc != RT_OK) return rc;
    if (w.acc_pixels != npix) {
      if ((rc = w.acc.ensure(npix * 4 * sizeof(long long))) != RT_OK) return rc;
      HIP_TRY(hipMemsetAsync(w.acc.p, 0, npix * 4 * sizeof(long long), stream));
      w.acc_pixels = npix;
    }
    const uint32_t n_batches = (uint32_t)((items + s->batch_items - 1) / s->batch_items);
    if (n_batches > lanes) s->notes |= RT_NOTE_FRAME_BATCHED;
    const uint32_t cap_wgs = (q_sort_cap + 255u) / 256u;
    const bool guess = s->est_valid && n_batches == lanes;  // grids from the previous frame's counts (else: whole capacity)
    const uint32_t ppw = 64u / (P.light_mult < 2u ? 2u : P.light_mult), pairs_per_wg = 4u * (ppw ? ppw : 1u);
    const uint32_t hard_cap_wgs = hard ? (s->hard_cap + pairs_per_wg - 1u) / pairs_per_wg : 1u;

    // ---- every chain's view of the frame: the caller's parameters with the chain's own queues and counters
    RtDevParams Pl[RT_LANES];
    float4* q[RT_LANES][2];
    uint32_t* counts[RT_LANES];
    for (uint32_t j = 0; j < lanes; j++) {
      rt_scene::Lane& L = w.lane[j];
      RtDevParams& Q = Pl[j];
      Q = P;
      uint32_t* ws = (uint32_t*)L.trace_ws.p;
      Q.sort_slot = (uint2*)ws;  // (8-byte aligned: first)
      Q.sh_idx = ws + (size_t)2 * q_sort_cap;
      Q.sort_hist = Q.sh_idx + q_sort_cap;
      Q.sort_offs = Q.sort_hist + n_buckets;
      Q.sort_tile = Q.sort_offs + n_buckets;
      if (split) {
        Q.hitrec = (uint2*)L.hitrec.p;
        Q.set_hdr = (uint4*)L.sets.p;                                        // [set_cap][2] uint4
        Q.set_list = (uint32_t*)L.sets.p + (size_t)set_cap * 8u;             // [set_cap][64]
        Q.set_q = (uint32_t*)L.sets.p + (size_t)set_cap * (8u + 64u);        // [3][set_cap]
        Q.set_cls = (uint8_t*)((uint32_t*)L.sets.p + (size_t)set_cap * (8u + 64u + 3u));  // [set_cap] bytes
        Q.set_cap = set_cap;
        Q.set_lights = s->dev.n_lights;
      }
      if (levels && (L.sort_hist_clean != (void*)Q.sort_hist || L.sort_hist_buckets != n_buckets)) {
        // a fresh (moved, resized) histogram: zero it once; every use leaves it zero
        HIP_TRY(hipMemsetAsync(Q.sort_hist, 0, (size_t)n_buckets * 4, stream));
        L.sort_hist_clean = (void*)Q.sort_hist;
        L.sort_hist_buckets = n_buckets;
      }
      counts[j] = (uint32_t*)L.qcount.p;
      HIP_TRY(hipMemsetAsync(counts[j], 0, (size_t)n_cnt * 4, stream));
      Q.acc = (long long*)w.acc.p;
      Q.q_capacity = s->q_cap;
      Q.q_overflow = counts[j] + RT_CNT_OVERFLOW;
      Q.hard_q = hard ? (float4*)L.hard.p : nullptr;
      Q.hard_capacity = s->hard_cap;
      Q.hard_count = counts[j] + RT_CNT_HARD(levels);
      Q.hard_stat = counts[j] + RT_CNT_HARD_STAT(levels);
      q[j][0] = (float4*)L.queues.p, q[j][1] = (float4*)L.queues.p + (size_t)s->q_cap * RT_QUEUE_QUADS;
    }
    s->queue_bytes = 0;
    for (auto& o : s->ws) s->queue_bytes += o.bytes();
    if (lanes > 1) {  // fork: the other chains start behind everything enqueued on the caller's stream so far
      HIP_TRY(hipEventRecord(w.fork_ev, stream));
      for (uint32_t j = 1; j < lanes; j++) HIP_TRY(hipStreamWaitEvent(w.lane[j].stream, w.fork_ev, 0));
      joiner.forked = true, joiner.joined = false;
    }
    auto run_hard = [&](uint32_t j, hipStream_t st) -> int {
      if (!hard) return RT_OK;
      RtDevParams& Q = Pl[j];
      const uint32_t g = guess ? grid_for(s->est[j][RT_CNT_HARD_STAT(levels) + 1u], pairs_per_wg, hard_cap_wgs) : std::min(hard_cap_wgs, 16384u);
      hipError_t e = (hipError_t)rt_launch_hard(s->dev, Q, g, st);
      if (e != hipSuccess) return fail(RT_ERR_HIP, "hard-pair launch failed: %s", hipGetErrorString(e));
      trace_point(st, "rt_hard_kernel: workgroups, pair capacity, chain", g, s->hard_cap, j);
      HIP_TRY(hipMemsetAsync(Q.hard_count, 0, 4, st));  // (stream ordered: behind the kernel that read it)
      return RT_OK;
    };
    const uint32_t batch_wgs = s->batch_items / 256u;
    // One batch per chain (the rule): the chains INTERLEAVE, groups of 64 workgroups of the list alternately, so that each
    // gets its share of the expensive regions (contiguous halves = sky and text: the sky's chain is done at once and the
    // text's runs alone).  Frames batched for memory: contiguous batches, dealt to the chains in turn.
    const uint32_t group_log2 = 6u, n_groups = (total_wgs + 63u) >> 6;
    const bool interleave = lanes > 1 && n_batches == lanes;
    uint32_t lane_batches[RT_LANES] = {0};
    for (uint32_t b = 0, w0 = 0; interleave ? b < lanes : w0 < total_wgs; b++, w0 += batch_wgs) {
      const uint32_t j = b % lanes;
      RtDevParams& Q = Pl[j];
      hipStream_t st = j ? w.lane[j].stream : stream;
      if (lane_batches[j]++) HIP_TRY(hipMemsetAsync(counts[j] + 1, 0, (size_t)(levels + 1) * 4, st));  // the chain's next batch: its level counters
      uint32_t nw = 0;
      if (interleave) {
        // groups j, j + lanes, ...; workgroups past the end of the list find no pixel and leave
        nw = ((n_groups + lanes - 1u - j) / lanes) << group_log2;
        Q.batch_first_wg = j << group_log2, Q.batch_stride = lanes, Q.batch_group_log2 = group_log2;
      } else {
        nw = std::min(total_wgs - w0, batch_wgs);
        Q.batch_first_wg = w0, Q.batch_stride = 1, Q.batch_group_log2 = 0;
      }
      Q.q_in = nullptr, Q.q_in_count = nullptr;
      Q.q_out = levels ? q[j][0] : nullptr;
      Q.q_out_count = counts[j] + RT_CNT_LEVEL(1);
      Q.q_capacity = q_sort_cap;  // (merged levels: the chain's two queues are one, and RT_CNT_LEVEL(1) is its running total)
      Q.seg_lo = Q.seg_hi = nullptr;
      hipError_t e;
      // the (wavefront, light) sets K2 queued for level k, one launch per class (grids: last frame's counts of this shape)
      auto run_sets = [&](uint32_t k, uint32_t n_sets_host) -> int {
        // class bytes -> class queues (level 0: the launch's set ids are known here; deeper levels: from the device-side hit count)
        const uint32_t n_sets_guess = k == 0 ? n_sets_host : (guess ? (s->est[j][RT_CNT_HITS(levels, k)] / 64u + 2u) * s->dev.n_lights : set_cap);
        Q.set_n = k == 0 ? n_sets_host : 0u;
        hipError_t ec = (hipError_t)rt_launch_compact(Q, std::min<uint32_t>((n_sets_guess + 2047u) / 2048u + 1u, (set_cap + 2047u) / 2048u), st);
        if (ec != hipSuccess) return fail(RT_ERR_HIP, "compaction launch failed: %s", hipGetErrorString(ec));
        for (int c = 0; c < 3; c++) {
          if (c == 0 && rt_phases_arrive_inline()) continue;  // (ARRIVE sets are finished by K2 itself in this build)
          const uint32_t cap_sets = (set_cap + 3u) / 4u;
          const uint32_t g = guess ? grid_for(s->est[j][RT_CNT_SETS(levels, k, c)], 4u, cap_sets) : cap_sets;
          hipError_t es = (hipError_t)rt_launch_sets(s->dev, Q, k == 0, c, g, st);
          if (es != hipSuccess) return fail(RT_ERR_HIP, "set-kernel launch failed: %s", hipGetErrorString(es));
          trace_point(st, "rt_sets kernel: level, class, workgroups", k, (uint32_t)c, g);
        }
        return RT_OK;
      };
      if (split) {
        if (lane_batches[j] > 1) HIP_TRY(hipMemsetAsync(counts[j] + RT_CNT_SETS(levels, 0, 0), 0, (size_t)3 * (levels + 1) * 4, st));  // next batch: its set counters
        Q.set_count = counts[j] + RT_CNT_SETS(levels, 0, 0);
        Q.set_items = nw * 256u;
        if ((size_t)nw * 256u > hitrec_items) return fail(RT_ERR_HIP, "internal: primary batch of %u workgroups exceeds the hit-record buffer", nw);
        e = (hipError_t)rt_launch_hit(s->dev, Q, nw, st);
        if (e != hipSuccess) return fail(RT_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
        trace_point(st, "rt_hit_kernel: first workgroup, workgroups", w0, nw);
        e = (hipError_t)rt_launch_classify(s->dev, Q, true, nw, st);
        if (e != hipSuccess) return fail(RT_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
        trace_point(st, "rt_classify0_kernel: first workgroup, workgroups, set capacity", w0, nw, set_cap);
        if ((rc = run_sets(0, nw * 4u * s->dev.n_lights)) != RT_OK) return rc;
      } else if (merged) {
        // The camera rays' hits and children first (rt_hit_spawn_kernel: 44 VGPRs, no scratch), so that the levels below can be traced
        // at once -- latency-bound launches -- while the hits are SHADED on a stream of the chain's (rt_primary_pre_kernel: issue bound).
        rt_scene::Lane& L = w.lane[j];
        if ((size_t)nw * 256u > hitrec_items) return fail(RT_ERR_HIP, "internal: primary batch of %u workgroups exceeds the hit-record buffer", nw);
        if (!L.shade_stream[0]) HIP_TRY(hipStreamCreateWithFlags(&L.shade_stream[0], hipStreamNonBlocking));
        if (!L.shade_done[0]) HIP_TRY(hipEventCreateWithFlags(&L.shade_done[0], hipEventDisableTiming));
        if (!L.hit_ev) HIP_TRY(hipEventCreateWithFlags(&L.hit_ev, hipEventDisableTiming));
        Q.hitrec = (uint2*)L.hitrec.p;
        Q.hit_spawns = 1u;
        e = (hipError_t)rt_launch_hit(s->dev, Q, nw, st);
        if (e != hipSuccess) return fail(RT_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
        trace_point(st, "rt_hit_spawn_kernel: first workgroup, workgroups, queue capacity", w0, nw, q_sort_cap);
        HIP_TRY(hipEventRecord(L.hit_ev, st));
        HIP_TRY(hipStreamWaitEvent(L.shade_stream[0], L.hit_ev, 0));
        joiner.shading = true;
        RtDevParams Pp = Q;
        Pp.q_out = nullptr, Pp.q_out_count = nullptr, Pp.hit_spawns = 0u;
        e = (hipError_t)rt_launch_primary(s->dev, Pp, nw, L.shade_stream[0]);
        if (e != hipSuccess) return fail(RT_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
        trace_point(L.shade_stream[0], "rt_primary_pre_kernel: first workgroup, workgroups", w0, nw);
        Q.hit_spawns = 0u;
      } else {
        e = (hipError_t)rt_launch_primary(s->dev, Q, nw, st);
        if (e != hipSuccess) return fail(RT_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
        trace_point(st, "rt_primary_stream_kernel: first workgroup, workgroups, queue capacity", w0, nw, s->q_cap);
      }
      if (merged) {
        // ---- every level traced first: rt_trace_spawn_kernel finds the hits of the slice [seg[k], seg[k + 1]) of the chain's ONE queue and
        // appends their children behind it; the slice's end is a stream-ordered snapshot of the queue's running total
        // (no rt_hard_kernel in between: the camera rays are being shaded on the chain's shade stream meanwhile and defer pairs of their
        // own; every deferred pair of the frame waits in the pair queue for the one launch behind the join)
        rt_scene::Lane& L = w.lane[j];
        uint32_t* total = counts[j] + RT_CNT_LEVEL(1);
        if (lane_batches[j] > 1) HIP_TRY(hipMemsetAsync(counts[j] + RT_CNT_SEG(levels, 0), 0, (size_t)(levels + 2u) * 4, st));
        if (pipelined) {
          for (int k = 0; k < 2; k++) {
            if (!L.shade_stream[k]) HIP_TRY(hipStreamCreateWithFlags(&L.shade_stream[k], hipStreamNonBlocking));
            if (!L.shade_done[k]) HIP_TRY(hipEventCreateWithFlags(&L.shade_done[k], hipEventDisableTiming));
          }
          while (L.level_ev.size() < levels) {
            hipEvent_t ev = nullptr;
            HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            L.level_ev.push_back(ev);
          }
        }
        Q.q_in = q[j][0];
        Q.q_out = q[j][0];
        Q.q_out_count = total;
        Q.q_in_count = total;
        for (uint32_t k = 1; k <= levels; k++) {
          HIP_TRY(hipMemcpyAsync(counts[j] + RT_CNT_SEG(levels, k + 1u), total, 4, hipMemcpyDeviceToDevice, st));
          Q.seg_lo = counts[j] + RT_CNT_SEG(levels, k);
          Q.seg_hi = counts[j] + RT_CNT_SEG(levels, k + 1u);
          Q.q_out = q[j][0], Q.q_out_count = total;
          const uint32_t n_k = guess ? s->est[j][RT_CNT_SEG(levels, k + 1u)] - s->est[j][RT_CNT_SEG(levels, k)] : 0u;
          const uint32_t g_rays = guess ? grid_for(n_k, 256u, cap_wgs) : cap_wgs;
          e = (hipError_t)rt_launch_trace(s->dev, Q, g_rays, st);
          if (e != hipSuccess) return fail(RT_ERR_HIP, "trace launch failed: %s", hipGetErrorString(e));
          trace_point(st, "rt_trace_spawn_kernel: level, workgroups, chain", k, g_rays, j);
          if (pipelined) {
            // the level's own hit-point order (its slice of sh_idx), then its shading on one of the chain's two shade streams: the
            // next levels are traced meanwhile, and the head of level k + 1's shading fills the drain of level k's
            Q.sort_hits = counts[j] + RT_CNT_HITS(levels, k);
            e = (hipError_t)rt_launch_sort(Q, g_rays, st);
            if (e != hipSuccess) return fail(RT_ERR_HIP, "sort launch failed: %s", hipGetErrorString(e));
            HIP_TRY(hipEventRecord(L.level_ev[k - 1u], st));
            hipStream_t ss = L.shade_stream[k & 1u];
            HIP_TRY(hipStreamWaitEvent(ss, L.level_ev[k - 1u], 0));
            joiner.shading = true;
            RtDevParams S = Q;
            S.q_out = nullptr, S.q_out_count = nullptr;  // (the children exist already)
            const uint32_t g_hits = guess ? grid_for(s->est[j][RT_CNT_HITS(levels, k)], 256u, cap_wgs) : cap_wgs;
            e = (hipError_t)rt_launch_shade(s->dev, S, g_hits, ss);
            if (e != hipSuccess) return fail(RT_ERR_HIP, "shade launch failed: %s", hipGetErrorString(e));
            trace_point(ss, "rt_shade_kernel: level, workgroups, chain (pipelined)", k, g_hits, j);
          }
        }
        Q.seg_lo = Q.seg_hi = nullptr;
        Q.q_out = nullptr, Q.q_out_count = nullptr;
        if (pipelined) {
          for (int k = 0; k < 2; k++) {  // join: the pairs the shade launches deferred, and the resolve, need all of them
            HIP_TRY(hipEventRecord(L.shade_done[k], L.shade_stream[k]));
            HIP_TRY(hipStreamWaitEvent(st, L.shade_done[k], 0));
          }
          joiner.shading = false;  // (joined on the device: the chain's stream now orders everything behind the shade launches)
        } else {
          Q.hitrec = nullptr;
          // ---- MERGED: one hit-point order over the rays of all levels, one shade launch (no children: they exist already)
          Q.sort_hits = counts[j] + RT_CNT_HITS(levels, 1);
          const uint32_t g_all = guess ? grid_for(s->est[j][RT_CNT_LEVEL(1)], 256u, cap_wgs) : cap_wgs;
          const uint32_t g_hits = guess ? grid_for(s->est[j][RT_CNT_HITS(levels, 1)], 256u, cap_wgs) : cap_wgs;
          e = (hipError_t)rt_launch_sort(Q, g_all, st);
          if (e != hipSuccess) return fail(RT_ERR_HIP, "sort launch failed: %s", hipGetErrorString(e));
          e = (hipError_t)rt_launch_shade(s->dev, Q, g_hits, st);
          if (e != hipSuccess) return fail(RT_ERR_HIP, "shade launch failed: %s", hipGetErrorString(e));
          trace_point(st, "rt_shade_kernel (all levels): workgroups, chain", g_hits, j);
          HIP_TRY(hipEventRecord(L.shade_done[0], L.shade_stream[0]));  // join: the camera rays' shading
          HIP_TRY(hipStreamWaitEvent(st, L.shade_done[0], 0));
          joiner.shading = false;
        }
      }
      for (uint32_t k = 1; k <= levels && !merged; k++) {
        if ((rc = run_hard(j, st)) != RT_OK) return rc;  // the pairs the launch before deferred
        Q.q_in = q[j][(k - 1u) & 1u];
        Q.q_in_count = counts[j] + RT_CNT_LEVEL(k);
        Q.sort_hits = counts[j] + RT_CNT_HITS(levels, k);
        if (k < levels) {
          Q.q_out = q[j][k & 1u];
          Q.q_out_count = counts[j] + RT_CNT_LEVEL(k + 1u);
        } else {
          Q.q_out = nullptr;  // rays of the last level have depth 1: no children possible
          Q.q_out_count = nullptr;
        }
        const uint32_t g_rays = guess ? grid_for(s->est[j][RT_CNT_LEVEL(k)], 256u, cap_wgs) : cap_wgs;
        const uint32_t g_hits = guess ? grid_for(s->est[j][RT_CNT_HITS(levels, k)], 256u, cap_wgs) : cap_wgs;
        e = (hipError_t)rt_launch_trace(s->dev, Q, g_rays, st);
        if (e != hipSuccess) return fail(RT_ERR_HIP, "trace launch failed: %s", hipGetErrorString(e));
        trace_point(st, "rt_trace_kernel: level, workgroups, chain", k, g_rays, j);
        e = (hipError_t)rt_launch_sort(Q, g_rays, st);
        if (e != hipSuccess) return fail(RT_ERR_HIP, "sort launch failed: %s", hipGetErrorString(e));
        trace_point(st, "sort kernels: level, buckets, chain", k, n_buckets, j);
        if (split) {
          Q.set_count = counts[j] + RT_CNT_SETS(levels, k, 0);
          e = (hipError_t)rt_launch_classify(s->dev, Q, false, g_hits, st);
          if (e != hipSuccess) return fail(RT_ERR_HIP, "classify launch failed: %s", hipGetErrorString(e));
          trace_point(st, "rt_classify_kernel: level, workgroups, chain", k, g_hits, j);
          if ((rc = run_sets(k, 0u)) != RT_OK) return rc;
        } else {
          e = (hipError_t)rt_launch_shade(s->dev, Q, g_hits, st);
          if (e != hipSuccess) return fail(RT_ERR_HIP, "shade launch failed: %s", hipGetErrorString(e));
          trace_point(st, "rt_shade_kernel: level, workgroups, chain", k, g_hits, j);
        }
      }
      if ((rc = run_hard(j, st)) != RT_OK) return rc;  // pairs deferred by the last level's shading
    }
    if (lanes > 1) {  // join: the resolve needs every chain's sums
      for (uint32_t j = 1; j < lanes; j++) {
        HIP_TRY(hipEventRecord(w.lane[j].done_ev, w.lane[j].stream));
        HIP_TRY(hipStreamWaitEvent(stream, w.lane[j].done_ev, 0));
      }
      joiner.joined = true;
    }
    hipError_t e = (hipError_t)rt_launch_resolve(Pl[0], stream);
    if (e != hipSuccess) return fail(RT_ERR_HIP, "resolve launch failed: %s", hipGetErrorString(e));
    trace_point(stream, "rt_resolve_kernel: attempt", (uint32_t)attempt);

    // ---- the frame's counters come back asynchronously (grids of the next frame); an unverified shape waits for them
    if (w.cnt_pending && (!s->stream_verified || blocking)) {  // (an older read-back still owns the pinned buffer)
      HIP_TRY(hipEventSynchronize(w.cnt_ev));
      w.cnt_pending = false;
    }
    if (!w.cnt_pending) {
      for (uint32_t j = 0; j < lanes; j++)
        HIP_TRY(hipMemcpyAsync(w.cnt_host + (size_t)j * RT_CNT_STRIDE, counts[j], (size_t)n_cnt * 4, hipMemcpyDeviceToHost, stream));
      HIP_TRY(hipEventRecord(w.cnt_ev, stream));
      w.cnt_pending = true;
      w.cnt_host_levels = levels;
      w.cnt_host_lanes = lanes;
      w.cnt_host_valid = n_batches == lanes;
      w.cnt_key_gen = s->key_gen;
    }
    if (s->stream_verified && !blocking) return RT_OK;
    HIP_TRY(hipEventSynchronize(w.cnt_ev));
    w.cnt_pending = false;
    uint32_t dropped = 0, dropped_pairs = 0, need = 0, need_pairs = 0;
    for (uint32_t j = 0; j < lanes; j++) {
      const uint32_t* c = w.cnt_host + (size_t)j * RT_CNT_STRIDE;
      dropped += c[RT_CNT_OVERFLOW], dropped_pairs += c[RT_CNT_HARD_STAT(levels)];
      // (merged levels: RT_CNT_LEVEL(1) is the running total of ONE queue of 2 q_cap rays)
      for (uint32_t k = 1; k <= levels + 1u; k++) need = std::max(need, merged ? (c[RT_CNT_LEVEL(k)] + 1u) / 2u : c[RT_CNT_LEVEL(k)]);
      need_pairs = std::max(need_pairs, c[RT_CNT_HARD_STAT(levels) + 1u]);
    }
    if (!dropped && !dropped_pairs) {
      if (w.cnt_host_valid) memcpy(s->est, w.cnt_host, sizeof(s->est)), s->est_valid = true;
      s->stream_verified = true;
      // headroom for the frames that now run unverified: a quarter more pairs than this frame deferred (takes effect with the next
      // frame's allocation; nothing was dropped, so this frame stands)
      if (hard && n_batches == lanes && (uint64_t)need_pairs * 5u / 4u > s->hard_cap)
        s->hard_cap = (uint32_t)std::min<uint64_t>((uint64_t)need_pairs * 5u / 4u + 256u, 0xFFFFFF00ull);
      return RT_OK;
    }
    // children or pairs were dropped: the counters say what the frame needed; render it again with that
    if (attempt >= (merged ? 12 : 6)) return fail(RT_ERR_HIP, "%u child rays / %u pair batches were dropped (queues could not be sized)", dropped, dropped_pairs);
    if (n_batches == lanes && !forced_chunk_log2) {
      // (one queue for all levels: the rays that were dropped would have had children of their own, so the count is a lower bound)
      if (merged && dropped) need = std::max<uint32_t>(need + need / 4u, s->q_cap + s->q_cap / 2u);
      if (need > s->q_cap) s->q_cap = (uint32_t)std::min<uint64_t>((uint64_t)need + need / 16u + 256u, 0xFFFFFF00ull);
      if (need_pairs > s->hard_cap) s->hard_cap = (uint32_t)std::min<uint64_t>((uint64_t)need_pairs + need_pairs / 8u + 256u, 0xFFFFFF00ull);
    } else {
      // (batched or forced: the counters are those of the last batch only -- grow geometrically)
      if (dropped) s->q_cap = (uint32_t)std::min<uint64_t>((uint64_t)s->q_cap * 2u, 0xFFFFFF00ull);
      if (dropped_pairs) s->hard_cap = (uint32_t)std::min<uint64_t>((uint64_t)s->hard_cap * 4u, 0xFFFFFF00ull);
    }
    s->est_valid = false;
    w.acc_pixels = 0;  // partial sums: clear the accumulator
    // ... and the ray counters of the abandoned attempt (rt_stats counts what the reference casts, once)
    if (P.counters) HIP_TRY(hipMemsetAsync(P.counters, 0, RT_COUNTER_REPLICAS * 16 * sizeof(unsigned long long), stream));
  }
}

// prepare() + the calibration frame of RT_TILE_ORDER_COST when this frame shape has no cost map yet: the frame is rendered
// once in row-major order with every wavefront of the primary kernel adding its run time to its super-tile's entry (a
// complete, valid frame into the caller's buffer), the map is read back (one synchronisation, once per scene and frame
// shape -- like the receiver flags) and prepare() runs again, now sorting the list.
static int prepare_ordered(rt_scene* s, const rt_params* p, uint32_t* argb_dev, const rt_aux* aux_dev, hipStream_t stream, RtDevParams* P,
                           const uint32_t* stage_slot, uint32_t stage_tiles_x) {
  int rc = prepare(s, p, argb_dev, aux_dev, stream, P);
  if (rc != RT_OK || !s->cost_wanted) return rc;
  const size_t n_sup_all = (size_t)((P->win_w + 15u) / 16u) * ((P->win_h + 15u) / 16u);
  if ((rc = s->costmap.ensure(n_sup_all * 4 + 4)) != RT_OK) return rc;
  HIP_TRY(hipMemsetAsync(s->costmap.p, 0, n_sup_all * 4, stream));
  P->cost_map = (uint32_t*)s->costmap.p;
  P->stage_slot = stage_slot;
  P->stage_tiles_x = stage_tiles_x;
  if ((rc = render_frame(s, *P, stream, p->tuning.chunk_log2)) != RT_OK) return rc;
  s->cost_host.assign(n_sup_all, 0u);
  HIP_TRY(hipMemcpyAsync(s->cost_host.data(), s->costmap.p, n_sup_all * 4, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  s->cost_valid = true;
  return prepare(s, p, argb_dev, aux_dev, stream, P);
}

int rt_render_device(rt_scene* s, const rt_params* p, uint32_t* argb_dev, const rt_aux* aux_dev, void* hip_stream) {
  if (!s || !argb_dev) return fail(RT_ERR_INVALID_ARG, "null argument");
  int rc = rt_validate_params(p);
  if (rc != RT_OK) return rc;
  HIP_TRY(hipSetDevice(s->device));
  RtDevParams P;
  if ((rc = prepare_ordered(s, p, argb_dev, aux_dev, (hipStream_t)hip_stream, &P, nullptr, 0)) != RT_OK) return rc;
  return render_frame(s, P, (hipStream_t)hip_stream, p->tuning.chunk_log2);
}

int rt_render_collect_stats(rt_scene* s, rt_stats* st) {
  if (!s || !st) return fail(RT_ERR_INVALID_ARG, "null argument");
  return rt_collect_stats_slot(s, s->last_block, st);
}

}  // extern "C"

// the ray counters of the frame that used `slot` last (the caller has waited for that frame)
int rt_collect_stats_slot(rt_scene* s, int slot, rt_stats* st) {
  if (!s || !st || slot < 0 || slot >= RT_SLOTS) return fail(RT_ERR_INVALID_ARG, "bad argument");
  HIP_TRY(hipSetDevice(s->device));
  unsigned long long all[RT_COUNTER_REPLICAS * 16];
  HIP_TRY(hipMemcpy(all, (unsigned long long*)s->counters.p + (size_t)slot * RT_COUNTER_REPLICAS * 16, sizeof(all), hipMemcpyDeviceToHost));
  unsigned long long c[16] = {0};
  for (unsigned r = 0; r < RT_COUNTER_REPLICAS; r++)
    for (unsigned i = 0; i < 16; i++) c[i] += all[r * 16 + i];
  st->rays_primary = c[0];
  st->rays_reflection = c[1];
  st->rays_refraction = c[2];
  st->rays_shadow = c[3];
  st->pixels_written = c[4];
  st->wave_ray_passes = c[5];
  st->wave_ray_lanes = c[6];
  st->wave_nearest_nodes = c[7];
  st->wave_nearest_tris = c[8];
  st->wave_shadow_nodes = c[9];
  st->wave_shadow_tris = c[10];
  st->wave_shadow_passes = c[11];
  st->wave_nearest_tris_exact = c[12];
  st->wave_shadow_tris_exact = c[13];
  st->rays_traced = c[14];
  st->notes = s->notes | s->sticky_notes;
  s->sticky_notes = 0;
  st->queue_bytes = s->queue_bytes;
  rt_scene_info mi;
  if (rt_scene_memory_info(s, &mi) == RT_OK) st->scene_bytes = mi.bytes_total;
  return RT_OK;
}

extern "C" {


int rt_render(rt_scene* s, const rt_params* p, uint32_t* argb, const rt_aux* aux, rt_stats* stats) {
  if (!s || !argb) return fail(RT_ERR_INVALID_ARG, "null argument");
  if (s->progress_active) return fail(RT_ERR_INVALID_ARG, "a progressive render owns this scene until rt_render_end");
  int rc = rt_validate_params(p);
  if (rc != RT_OK) return rc;
  HIP_TRY(hipSetDevice(s->device));
  auto t_begin = std::chrono::steady_clock::now();
  const size_t npix = (size_t)p->width * p->height;
  // Only the window travels (ChunkView, image_buffer.rs:178-251): the caller's fill of its rows goes up so that
  // miss pixels keep it (image_buffer.rs:27-37), the rendered rows come back.  A progressive band of a 1620-wide
  // frame moves 48 rows, not the frame.
  const uint32_t wx = p->win_w ? p->win_x0 : 0u, wy = p->win_w ? p->win_y0 : 0u;
  const uint32_t ww = p->win_w ? p->win_w : p->width, wh = p->win_w ? p->win_h : p->height;
  const size_t first = (size_t)wy * p->width + wx;
  auto copy_window = [&](void* dst, const void* src, size_t bytes_per_px, hipMemcpyKind kind) -> hipError_t {
    if (ww == p->width)  // whole rows: one contiguous block
      return hipMemcpy((char*)dst + first * bytes_per_px, (const char*)src + first * bytes_per_px,
                       (size_t)wh * p->width * bytes_per_px, kind);
    return hipMemcpy2D((char*)dst + first * bytes_per_px, (size_t)p->width * bytes_per_px,
                       (const char*)src + first * bytes_per_px, (size_t)p->width * bytes_per_px, (size_t)ww * bytes_per_px, wh, kind);
  };
  if ((rc = s->fb.ensure(npix * 4)) != RT_OK) return rc;
  HIP_TRY(copy_window(s->fb.p, argb, 4, hipMemcpyHostToDevice));
  rt_aux ad{};
  if (aux) {
    if (aux->rgb) {
      if ((rc = s->aux_rgb.ensure(npix * 12)) != RT_OK) return rc;
      HIP_TRY(copy_window(s->aux_rgb.p, aux->rgb, 12, hipMemcpyHostToDevice));
      ad.rgb = (float*)s->aux_rgb.p;
    }
    if (aux->hit_id) {
      if ((rc = s->aux_id.ensure(npix * 4)) != RT_OK) return rc;
      HIP_TRY(copy_window(s->aux_id.p, aux->hit_id, 4, hipMemcpyHostToDevice));
      ad.hit_id = (int32_t*)s->aux_id.p;
    }
    if (aux->hit_t) {
      if ((rc = s->aux_t.ensure(npix * 4)) != RT_OK) return rc;
      HIP_TRY(copy_window(s->aux_t.p, aux->hit_t, 4, hipMemcpyHostToDevice));
      ad.hit_t = (float*)s->aux_t.p;
    }
  }
  EventPair ev, ev_setup;
  HIP_TRY(hipEventCreate(&ev.e0));
  HIP_TRY(hipEventCreate(&ev.e1));
  HIP_TRY(hipEventCreate(&ev_setup.e0));
  HIP_TRY(hipEventRecord(ev_setup.e0, nullptr));  // what prepare enqueues (table uploads, rt_flags_kernel) is timed as setup_ms
  RtDevParams P;
  if ((rc = prepare_ordered(s, p, (uint32_t*)s->fb.p, aux ? &ad : nullptr, nullptr, &P, nullptr, 0)) != RT_OK) return rc;
  HIP_TRY(hipEventRecord(ev.e0, nullptr));
  if ((rc = render_frame(s, P, nullptr, p->tuning.chunk_log2, true)) != RT_OK) return rc;
  HIP_TRY(hipEventRecord(ev.e1, nullptr));
  HIP_TRY(hipEventSynchronize(ev.e1));
  float ms = 0.f, setup_ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, ev.e0, ev.e1));
  HIP_TRY(hipEventElapsedTime(&setup_ms, ev_setup.e0, ev.e0));
  auto t_copy = std::chrono::steady_clock::now();
  HIP_TRY(copy_window(argb, s->fb.p, 4, hipMemcpyDeviceToHost));
  const double d2h_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_copy).count();
  if (aux) {
    if (aux->rgb) HIP_TRY(copy_window(aux->rgb, s->aux_rgb.p, 12, hipMemcpyDeviceToHost));
    if (aux->hit_id) HIP_TRY(copy_window(aux->hit_id, s->aux_id.p, 4, hipMemcpyDeviceToHost));
    if (aux->hit_t) HIP_TRY(copy_window(aux->hit_t, s->aux_t.p, 4, hipMemcpyDeviceToHost));
  }
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    if ((rc = rt_render_collect_stats(s, stats)) != RT_OK) return rc;
    stats->kernel_ms = ms;
    stats->setup_ms = setup_ms;
    stats->d2h_ms = d2h_ms;
    stats->total_ms =
        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  }
  return RT_OK;
}

}  // extern "C"

int rt_render_device_staged(rt_scene* s, const rt_params* p, uint32_t* out_dev, const uint32_t* stage_slot,
                            uint32_t tiles_x, hipStream_t stream) {
  if (!s || !out_dev) return fail(RT_ERR_INVALID_ARG, "null argument");
  int rc = rt_validate_params(p);
  if (rc != RT_OK) return rc;
  HIP_TRY(hipSetDevice(s->device));
  RtDevParams P;
  if ((rc = prepare_ordered(s, p, out_dev, nullptr, stream, &P, stage_slot, tiles_x)) != RT_OK) return rc;
  P.stage_slot = stage_slot;
  P.stage_tiles_x = tiles_x;
  return render_frame(s, P, stream, p->tuning.chunk_log2);
}

