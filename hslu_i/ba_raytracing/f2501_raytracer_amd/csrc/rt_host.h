// rt_host.h -- host-side internals shared by rt_api.cpp (single-GPU entry points) and rt_multi.cpp (multi-GPU
// gather).  Not part of the public ABI.
#pragma once

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "rt_internal.h"

// records the thread-local message of rt_last_error() and returns `code`
int rt_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
#define fail rt_fail

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return fail(e_ == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_HIP, "%s failed: %s", #expr, \
                  hipGetErrorString(e_));                                                      \
  } while (0)

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return RT_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes < 256 ? 256 : bytes;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
      p = nullptr;
      return fail(RT_ERR_OOM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    cap = want;
    return RT_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

struct EventPair {  // destroyed on every return path
  hipEvent_t e0 = nullptr, e1 = nullptr;
  ~EventPair() {
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
  }
};

struct rt_scene {
  int device = 0;
  RtDevScene dev{};
  rt_bvh_info info{};
  DevBuf blob;  // spheres, triangles, materials, lights, BVH (RtDevScene offsets)
  // per-render workspaces
  DevBuf aa, cloud, counters, queues, qcount, acc, fb, aux_rgb, aux_id, aux_t, suplist, trace_ws, sort_tmp, hard;
  std::vector<uint32_t> sup_host;
  uint32_t sup_key[7] = {0, 0, 0, 0, 0, 0, 0};  // window, tile size, n_ranks, rank the list was built for
  size_t acc_pixels = 0;  // pixels the (zeroed) accumulator currently covers; 0 = must be cleared before use
  uint32_t chunk = 1u << 16;  // rays per secondary launch / primary batch of the current frame
  float aabb_lo[3] = {0.f, 0.f, 0.f}, aabb_hi[3] = {1.f, 1.f, 1.f};  // bounds of all objects (Morton keys)
  // host copies of the parameter tables last uploaded (skip re-upload when unchanged)
  std::vector<float> aa_host, cloud_host, cloud_scaled;
  std::vector<uint32_t> aa_table;  // device image: [2U] offsets (float bits), [U] multiplicities, [n] sample -> thread
  uint32_t aa_unique = 0;
  bool aa_dedup = true;
  // The tables are uploaded on the stream of the call that changed them; a later call on another stream waits for
  // that upload (tables_ev) before its kernels read them.
  hipEvent_t tables_ev = nullptr;
  hipStream_t tables_stream = nullptr, last_stream = nullptr;
  bool tables_pending = false, rendered = false;
  float cloud_ball[4] = {0.f, 0.f, 0.f, -1.f};  // centre offset (scene units) + radius of all cloud offsets
  // receiver flags (rt_flags_kernel): cell tables built with the scene, flags rebuilt when the beam constants change
  DevBuf flag_geo, flags;
  uint32_t n_cells = 0, n_tri_cells = 0;
  float flags_key[8] = {-1.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // beam_delta, eps, cloud centre: what the flags were built for
  float cloud_ball_f[3] = {0.f, 0.f, 0.f};
};


// rt_api.cpp internals used by the multi-GPU path
// Multi-GPU staging: when `stage_slot` (device, [tiles_x * tiles_y]) is given, packed pixels are stored into the
// rank-compact staging buffer `out_dev` (tile slot * tile_size^2 + offset inside the tile) instead of a W x H frame.
int rt_render_device_staged(rt_scene* s, const rt_params* p, uint32_t* out_dev, const uint32_t* stage_slot,
                            uint32_t tiles_x, hipStream_t stream);
int rt_validate_params(const rt_params* p);
