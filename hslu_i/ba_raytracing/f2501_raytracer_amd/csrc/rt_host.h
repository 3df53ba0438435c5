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

// frames of one scene that can be in flight at once (slots: counter block + workspace set)
#define RT_SLOTS 4
// chains a frame with secondary rays is split into (rt_tuning.sub_frames)
#define RT_LANES 2
// dwords of a chain's device-side counter block (levels <= 64: 2 * 64 + 8 level counters + 3 * 65 set-class counters)
#define RT_CNT_STRIDE 384

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

// what the ray counts of a frame with secondary rays depend on (besides the scene): a frame with the key of the last
// verified frame renders without a synchronisation
struct StreamKey {
  uint32_t width, height, flags, aa_rays, aa_unique, light_mult, depth_refl, depth_refr, win[4], tile_size, n_ranks, rank, traversal,
      cand_cap, cloud_seed, n_cloud_sets, forced, tables, staged, flags_on, n_sup, lanes, split, sort_bits, lists_on, merged;
  float f[8];
};

struct rt_scene {
  int device = 0;
  RtDevScene dev{};
  rt_bvh_info info{};
  DevBuf blob;  // spheres, triangles, materials, lights, BVH (RtDevScene offsets)
  // per-render workspaces
  DevBuf aa, cloud, counters, fb, aux_rgb, aux_id, aux_t, suplist;
  DevBuf progress_fb;   // rt_render_begin: the device frame its bands are rendered into
  bool progress_active = false;  // a progressive render (rt_render_begin .. rt_render_end) owns the scene
  uint64_t budget = 0;  // rt_scene_desc.device_budget_bytes as applied: bounds flags + per-cell lists
  size_t bytes_bvh = 0; // of `blob`: nodes + octant copies + threaded copy
  // What ONE frame with secondary rays owns while it is in flight: ray queues, sort workspace, hard-pair queue, level
  // counters (+ their pinned read-back), pixel accumulator.  Two sets, so that two such frames can be in flight (the
  // second set is only allocated when a frame is enqueued while the one before it is still running).
  // A frame's ray tree runs as RT_LANES independent CHAINS (halves of its primary work-group list), each with its own
  // queues, sort workspace, pair queue and level counters; chain 0 on the caller's stream, the others on streams of the
  // set (forked / joined with events).  The chains meet in the pixel accumulator.
  struct Lane {
    DevBuf queues, trace_ws, hard, qcount;
    DevBuf hitrec, sets;              // phase-split pipeline: hit records of the primary launch, (wavefront, light) set records
    void* sort_hist_clean = nullptr;  // the histogram (address, size) that is known to be zero
    uint32_t sort_hist_buckets = 0;
    hipStream_t stream = nullptr;     // chains 1..: their own stream
    hipEvent_t done_ev = nullptr;     // ... and the event the caller's stream waits for before the resolve
    // pipelined levels (rt_tuning.levels): the two streams levels are shaded on alternately, their join events, one event per traced level
    hipStream_t shade_stream[2] = {nullptr, nullptr};
    hipEvent_t shade_done[2] = {nullptr, nullptr};
    std::vector<hipEvent_t> level_ev;
    hipEvent_t hit_ev = nullptr;      // merged levels: the camera rays' hits (and children) exist -- their shading may start on shade_stream[0]
  };
  struct StreamWs {
    Lane lane[RT_LANES];
    DevBuf acc;
    size_t acc_pixels = 0;            // pixels the (zeroed) accumulator currently covers; 0 = must be cleared before use
    uint32_t* cnt_host = nullptr;     // pinned, [RT_LANES][RT_CNT_STRIDE]: asynchronous read-back of the chains' counters
    hipEvent_t cnt_ev = nullptr, fork_ev = nullptr;
    bool cnt_pending = false, cnt_host_valid = false;
    uint32_t cnt_host_levels = 0, cnt_host_lanes = 0, cnt_key_gen = 0;
    size_t bytes() const {
      size_t b = acc.cap;
      for (const Lane& l : lane) b += l.queues.cap + l.trace_ws.cap + l.hard.cap + l.hitrec.cap + l.sets.cap;
      return b;
    }
  } ws[RT_SLOTS];
  std::vector<uint32_t> sup_host;
  uint32_t sup_key[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // window, tile size, n_ranks, rank, order the list was built for
  // RT_TILE_ORDER_COST: measured cost per super-tile (window-relative index) for cost_key = frame shape + what a ray costs
  DevBuf costmap;
  std::vector<uint32_t> cost_host;
  uint32_t cost_key[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  bool cost_valid = false, cost_wanted = false;
  // ray streaming (frames with secondary rays): sizes shared by both workspace sets, see render_frame_impl
  uint32_t q_cap = 0, hard_cap = 0, batch_items = 0;  // per chain: rays per queue, pairs; primary work items per batch
  StreamKey stream_key{};
  uint32_t key_gen = 0;          // bumped whenever stream_key changes (counter read-backs are stamped with it)
  uint32_t sticky_notes = 0;     // notes about frames already delivered (reported by the next rt_render_collect_stats)
  bool stream_verified = false;  // a frame of this key ran without dropping a ray or a pair
  uint32_t est[RT_LANES][RT_CNT_STRIDE] = {{0}};  // the counters of the last complete frame of this key, per chain (grids of the next one)
  bool est_valid = false;
  uint32_t sort_bits_wanted = 0;    // rt_tuning.sort_bits of the current frame (0 = default)
  uint32_t lanes_wanted = 0;        // rt_tuning.sub_frames of the current frame (0 = default)
  uint32_t phases_wanted = 0;       // rt_tuning.phases of the current frame (0 = default)
  uint32_t levels_wanted = 0;       // rt_tuning.levels of the current frame (0 = default)
  uint32_t calm_frames = 1u << 30;  // frames enqueued in a row while no other frame of the scene was running (sub_frames = 0: auto)
  uint32_t tables_version = 0;      // bumped whenever a parameter table (AA samples, light clouds, flags, tile list) is uploaded
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
  DevBuf cell_lists;     // per (receiver cell, light): 8 x uint16 leaf slots surviving the cell's fat beam (rt_flags_kernel)
  bool cell_lists_built = false;
  uint32_t n_cells = 0, n_tri_cells = 0;
  float flags_key[8] = {-1.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // beam_delta, eps, cloud centre: what the flags were built for
  float cloud_ball_f[3] = {0.f, 0.f, 0.f};
  // Frames in flight.  Consecutive frames may be enqueued on different streams so that the head of one overlaps the
  // drain of the other (a launch cannot end before its longest wavefront does).  A frame owns one of RT_SLOTS SLOTS -- a
  // counter block and, with secondary rays, a workspace set (ws[]) -- guarded by the event recorded behind the frame that
  // used the slot last.  A frame takes a slot whose last frame has finished if there is one (a host that renders frame by
  // frame only ever uses slot 0), else the one used longest ago.
  hipEvent_t frame_ev[RT_SLOTS] = {};
  bool frame_pending[RT_SLOTS] = {};
  uint32_t frame_seq[RT_SLOTS] = {};
  hipStream_t frame_stream[RT_SLOTS] = {};  // the stream the slot's last frame was enqueued on
  uint32_t frame_no = 0;
  int cur_block = 0, last_block = 0;
  int cur_ws = 0;                  // workspace set of the frame being enqueued (normally its slot)
  int ws_last_block[RT_SLOTS] = {0, 1, 2, 3};  // the slot of the frame that used each workspace set last
  // which fast paths the last frame did NOT take (rt_stats.notes)
  uint32_t notes = 0;
  size_t queue_bytes = 0;  // ray queues + hard-pair queue + sort workspace of the last frame with secondary rays
};


// rt_api.cpp internals used by the multi-GPU path
// Multi-GPU staging: when `stage_slot` (device, [tiles_x * tiles_y]) is given, packed pixels are stored into the
// rank-compact staging buffer `out_dev` (tile slot * tile_size^2 + offset inside the tile) instead of a W x H frame.
int rt_render_device_staged(rt_scene* s, const rt_params* p, uint32_t* out_dev, const uint32_t* stage_slot,
                            uint32_t tiles_x, hipStream_t stream);
int rt_validate_params(const rt_params* p);
// the ray counters of the frame that used frame slot `slot` of the scene last (rt_scene::cur_block right after a frame
// was enqueued); the caller has waited for that frame
int rt_collect_stats_slot(rt_scene* s, int slot, rt_stats* st);
