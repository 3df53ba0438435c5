// rt_multi.cpp -- the multi-GPU side of `Renderer::render` (reference src/renderer/mod.rs:80-94,146-209: one frame,
// RENDER_STRIDE tiles handed to parallel workers, all writing one ImageBuffer, src/image_buffer.rs:48-97).
//
// Every GPU renders the tiles it owns (rt_tile_owner, include/rt_hip.h) straight into a rank-compact staging buffer;
// the only communication is ONE gather of those buffers to the root:
//     ncclGroupStart;  root: ncclRecv x (n-1);  others: ncclSend;  ncclGroupEnd;  root: scatter kernel.
// xGMI is point to point: each peer has its own link into the root, so the n-1 transfers run side by side (no ring),
// <= W*H*4/n bytes each (4.1 MB per peer for a 4K frame over 8 GPUs).
//
// Two host shapes are served: one process driving all GPUs (rt_render_multi -- the reference's single process) and
// one process per GPU (rt_comm_* + rt_render_gather_device -- bench.py under torch.distributed.run).
#include <rccl/rccl.h>

#include <chrono>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

#include "rt_host.h"

static_assert(RT_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "rt_comm id is an ncclUniqueId");

#define NCCL_TRY(expr)                                                                              \
  do {                                                                                              \
    ncclResult_t r_ = (expr);                                                                       \
    if (r_ != ncclSuccess) return fail(RT_ERR_HIP, "%s failed: %s", #expr, ncclGetErrorString(r_)); \
  } while (0)

namespace {

// Staging layout of one frame shape (host + device copies); see rt_gather_layout
struct Layout {
  uint32_t key[4] = {0, 0, 0, 0};  // width, height, tile_size, n_ranks
  uint32_t tiles_x = 0, tiles_y = 0, tile_size = 48;
  std::vector<uint32_t> slot, count, rank_off;  // rank_off[r]: pixel offset of rank r inside the root's receive buffer
  uint64_t recv_pixels = 0;
  bool matches(uint32_t w, uint32_t h, uint32_t ts, uint32_t n) const { return key[0] == w && key[1] == h && key[2] == ts && key[3] == n; }
  void build(uint32_t w, uint32_t h, uint32_t ts, uint32_t n) {
    key[0] = w, key[1] = h, key[2] = ts, key[3] = n;
    tile_size = ts;
    tiles_x = (w + ts - 1) / ts, tiles_y = (h + ts - 1) / ts;
    slot.assign((size_t)tiles_x * tiles_y, 0);
    count.assign(n, 0);
    rt_gather_layout(w, h, ts, n, slot.data(), count.data());
    rank_off.assign(n, 0);
    recv_pixels = 0;
    for (uint32_t r = 1; r < n; r++) {  // the root's own tiles never travel
      rank_off[r] = (uint32_t)recv_pixels;
      recv_pixels += (uint64_t)count[r] * ts * ts;
    }
  }
};

}  // namespace

struct rt_comm {
  uint32_t n_ranks = 1, rank = 0;
  int device = 0;
  ncclComm_t nccl = nullptr;
  bool owns_nccl = false;
  uint32_t transport = RT_TRANSPORT_NONE;
  Layout lay;
  DevBuf slot_dev, off_dev;  // tile -> slot, rank -> receive offset (device copies of the layout)
  DevBuf stage;              // rank != 0: this rank's tiles
  DevBuf recv;               // rank 0: the other ranks' tiles, rank r at rank_off[r]
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};  // render start, render end, gather end
  bool timed = false;
  rt_gather_info info{};

  int ensure_layout(const rt_params* p, hipStream_t stream) {
    const uint32_t ts = p->tile_size ? p->tile_size : 48u;
    if (lay.matches(p->width, p->height, ts, n_ranks)) return RT_OK;
    // (the previous frame's kernels may still read the old tables)
    HIP_TRY(hipStreamSynchronize(stream));
    lay.build(p->width, p->height, ts, n_ranks);
    int rc;
    if ((rc = slot_dev.ensure(lay.slot.size() * 4)) != RT_OK) return rc;
    if ((rc = off_dev.ensure(lay.rank_off.size() * 4)) != RT_OK) return rc;
    HIP_TRY(hipMemcpy(slot_dev.p, lay.slot.data(), lay.slot.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(off_dev.p, lay.rank_off.data(), lay.rank_off.size() * 4, hipMemcpyHostToDevice));
    if (rank == 0) {
      if ((rc = recv.ensure(lay.recv_pixels * 4 + 4)) != RT_OK) return rc;
    } else {
      if ((rc = stage.ensure(stage_bytes() + 4)) != RT_OK) return rc;
    }
    return RT_OK;
  }
  size_t stage_bytes() const { return (size_t)lay.count[rank] * lay.tile_size * lay.tile_size * 4; }
  int ensure_events() {
    for (auto& e : ev)
      if (!e) HIP_TRY(hipEventCreate(&e));
    return RT_OK;
  }
  // this rank's tiles -> staging (rank != 0) or straight into the frame (rank 0); enqueued on `stream`
  int render(rt_scene* s, const rt_params* params, uint32_t* argb_dev, hipStream_t stream) {
    if (s->device != device) return fail(RT_ERR_INVALID_ARG, "scene lives on device %d, communicator on %d", s->device, device);
    rt_params p = *params;
    p.n_ranks = n_ranks;
    p.rank = rank;
    int rc = rt_validate_params(&p);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipSetDevice(device));
    if ((rc = ensure_layout(&p, stream)) != RT_OK) return rc;
    if ((rc = ensure_events()) != RT_OK) return rc;
    HIP_TRY(hipEventRecord(ev[0], stream));
    if (rank == 0) {
      if (!argb_dev) return fail(RT_ERR_INVALID_ARG, "rank 0 needs the frame buffer");
      rc = rt_render_device(s, &p, argb_dev, nullptr, stream);
    } else {
      HIP_TRY(hipMemsetAsync(stage.p, 0, stage_bytes(), stream));  // 0 = "no hit"
      rc = rt_render_device_staged(s, &p, (uint32_t*)stage.p, (const uint32_t*)slot_dev.p, lay.tiles_x, stream);
    }
    if (rc != RT_OK) return rc;
    HIP_TRY(hipEventRecord(ev[1], stream));
    info.tiles_owned = lay.count[rank];
    info.bytes_sent = rank ? stage_bytes() : 0;
    info.bytes_received = rank ? 0 : lay.recv_pixels * 4;
    return RT_OK;
  }
  // this rank's send / receives; the caller brackets the ranks of one process with ncclGroupStart / ncclGroupEnd
  int exchange(hipStream_t stream) {
    if (rank == 0) {
      for (uint32_t r = 1; r < n_ranks; r++)
        NCCL_TRY(ncclRecv((uint32_t*)recv.p + lay.rank_off[r], (size_t)lay.count[r] * lay.tile_size * lay.tile_size, ncclUint32, (int)r,
                          nccl, stream));
    } else {
      NCCL_TRY(ncclSend(stage.p, stage_bytes() / 4, ncclUint32, 0, nccl, stream));
    }
    return RT_OK;
  }
  int scatter(uint32_t* argb_dev, uint32_t width, uint32_t height, hipStream_t stream) {
    if (n_ranks > 1) {
      hipError_t e = (hipError_t)rt_launch_scatter(argb_dev, (const uint32_t*)recv.p, (const uint32_t*)off_dev.p, (const uint32_t*)slot_dev.p,
                                                   width, height, lay.tile_size, lay.tiles_x, n_ranks, stream);
      if (e != hipSuccess) return fail(RT_ERR_HIP, "scatter launch failed: %s", hipGetErrorString(e));
    }
    return RT_OK;
  }
  int finish(hipStream_t stream) {
    HIP_TRY(hipEventRecord(ev[2], stream));
    timed = true;
    return RT_OK;
  }
  void release() {
    (void)hipSetDevice(device);
    for (auto& e : ev)
      if (e) (void)hipEventDestroy(e), e = nullptr;
    for (DevBuf* b : {&slot_dev, &off_dev, &stage, &recv}) b->release();
    if (nccl && owns_nccl) (void)ncclCommDestroy(nccl);
    nccl = nullptr;
  }
};

extern "C" {

int rt_gather_layout(uint32_t width, uint32_t height, uint32_t tile_size, uint32_t n_ranks, uint32_t* tile_slot,
                     uint32_t* tiles_per_rank) {
  if (width == 0 || height == 0) return fail(RT_ERR_INVALID_ARG, "empty frame");
  const uint32_t ts = tile_size ? tile_size : 48u, n = n_ranks ? n_ranks : 1u;
  const uint32_t tiles_x = (width + ts - 1) / ts, tiles_y = (height + ts - 1) / ts;
  std::vector<uint32_t> count(n, 0);
  for (uint32_t ty = 0; ty < tiles_y; ty++)
    for (uint32_t tx = 0; tx < tiles_x; tx++) {
      const uint32_t o = rt_tile_owner(tx, ty, n);
      if (tile_slot) tile_slot[(size_t)ty * tiles_x + tx] = count[o];
      count[o]++;
    }
  if (tiles_per_rank) memcpy(tiles_per_rank, count.data(), n * sizeof(uint32_t));
  return RT_OK;
}

int rt_comm_unique_id(uint8_t id[RT_COMM_ID_BYTES]) {
  if (!id) return fail(RT_ERR_INVALID_ARG, "null argument");
  ncclUniqueId u;
  NCCL_TRY(ncclGetUniqueId(&u));
  memcpy(id, u.internal, RT_COMM_ID_BYTES);
  return RT_OK;
}

int rt_comm_create(const uint8_t* id, uint32_t n_ranks, uint32_t rank, int device, rt_comm** out) {
  if (!out) return fail(RT_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  if (n_ranks == 0 || rank >= n_ranks) return fail(RT_ERR_INVALID_ARG, "rank %u out of range (%u ranks)", rank, n_ranks);
  if (n_ranks > 1 && !id) return fail(RT_ERR_INVALID_ARG, "communicator id missing");
  const int ndev = rt_device_count();
  if (device < 0 || device >= ndev) return fail(RT_ERR_INVALID_ARG, "device %d out of range (%d visible)", device, ndev);
  HIP_TRY(hipSetDevice(device));
  std::unique_ptr<rt_comm> c(new rt_comm());
  c->n_ranks = n_ranks, c->rank = rank, c->device = device;
  c->info.n_ranks = n_ranks, c->info.rank = rank;
  if (n_ranks > 1) {
    ncclUniqueId u;
    memcpy(u.internal, id, RT_COMM_ID_BYTES);
    NCCL_TRY(ncclCommInitRank(&c->nccl, (int)n_ranks, u, (int)rank));
    c->owns_nccl = true;
    c->transport = RT_TRANSPORT_RCCL;
    int cnt = 0, ur = -1;
    NCCL_TRY(ncclCommCount(c->nccl, &cnt));
    NCCL_TRY(ncclCommUserRank(c->nccl, &ur));
    c->info.n_ranks = (uint32_t)cnt, c->info.rank = (uint32_t)ur;
  }
  c->info.transport = c->transport;
  *out = c.release();
  return RT_OK;
}

void rt_comm_destroy(rt_comm* c) {
  if (!c) return;
  c->release();
  delete c;
}

int rt_render_gather_device(rt_scene* s, rt_comm* c, const rt_params* params, uint32_t* argb_dev, void* hip_stream) {
  if (!s || !c || !params) return fail(RT_ERR_INVALID_ARG, "null argument");
  hipStream_t stream = (hipStream_t)hip_stream;
  int rc = c->render(s, params, argb_dev, stream);
  if (rc != RT_OK) return rc;
  if (c->n_ranks > 1) {
    NCCL_TRY(ncclGroupStart());
    rc = c->exchange(stream);
    ncclResult_t ge = ncclGroupEnd();
    if (rc != RT_OK) return rc;
    if (ge != ncclSuccess) return fail(RT_ERR_HIP, "ncclGroupEnd failed: %s", ncclGetErrorString(ge));
    if (c->rank == 0 && (rc = c->scatter(argb_dev, params->width, params->height, stream)) != RT_OK) return rc;
  }
  return c->finish(stream);
}

int rt_comm_last_gather(rt_comm* c, rt_gather_info* out) {
  if (!c || !out) return fail(RT_ERR_INVALID_ARG, "null argument");
  if (!c->timed) return fail(RT_ERR_INVALID_ARG, "no gather has been enqueued on this communicator");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipEventSynchronize(c->ev[2]));
  float a = 0.f, b = 0.f;
  HIP_TRY(hipEventElapsedTime(&a, c->ev[0], c->ev[1]));
  HIP_TRY(hipEventElapsedTime(&b, c->ev[1], c->ev[2]));
  c->info.render_ms = a;
  c->info.gather_ms = b;
  *out = c->info;
  return RT_OK;
}

}  // extern "C"

// ---- one process, all GPUs ---------------------------------------------------------------------------------------
namespace {

struct MultiCtx {
  std::vector<int> devices;
  std::vector<std::unique_ptr<rt_comm>> comm;
  std::vector<hipStream_t> stream;
  std::vector<ncclComm_t> nccl;  // ncclCommInitAll
  uint32_t transport = RT_TRANSPORT_NONE;
  DevBuf fb;  // root: W x H frame
  hipEvent_t peer_done = nullptr;
  ~MultiCtx() {
    for (size_t i = 0; i < comm.size(); i++) {
      comm[i]->release();
      (void)hipSetDevice(devices[i]);
      if (stream[i]) (void)hipStreamDestroy(stream[i]);
    }
    for (ncclComm_t c : nccl)
      if (c) (void)ncclCommDestroy(c);
    if (!devices.empty()) {
      (void)hipSetDevice(devices[0]);
      fb.release();
      if (peer_done) (void)hipEventDestroy(peer_done);
    }
  }
};

std::mutex g_multi_mutex;
std::map<std::vector<int>, std::unique_ptr<MultiCtx>> g_multi;  // key: devices, then the force-RCCL flag

int get_ctx(const std::vector<int>& devices, bool force_rccl, MultiCtx** out) {
  std::vector<int> key(devices);
  key.push_back(force_rccl ? 1 : 0);
  auto it = g_multi.find(key);
  if (it != g_multi.end()) {
    *out = it->second.get();
    return RT_OK;
  }
  std::unique_ptr<MultiCtx> m(new MultiCtx());
  m->devices = devices;
  const size_t n = devices.size();
  bool distinct = true, same = true;
  for (size_t i = 0; i < n; i++) {
    same = same && devices[i] == devices[0];
    for (size_t j = 0; j < i; j++) distinct = distinct && devices[i] != devices[j];
  }
  if (n > 1 && !distinct && !same)
    return fail(RT_ERR_UNSUPPORTED, "per_gpu must name distinct GPUs (RCCL), or one GPU for all ranks (rehearsal)");
  m->transport = n == 1 ? RT_TRANSPORT_NONE : ((distinct || force_rccl) ? RT_TRANSPORT_RCCL : RT_TRANSPORT_LOCAL);
  if (m->transport == RT_TRANSPORT_RCCL) {
    m->nccl.assign(n, nullptr);
    NCCL_TRY(ncclCommInitAll(m->nccl.data(), (int)n, devices.data()));
  }
  for (size_t i = 0; i < n; i++) {
    std::unique_ptr<rt_comm> c(new rt_comm());
    c->n_ranks = (uint32_t)n, c->rank = (uint32_t)i, c->device = devices[i];
    c->transport = m->transport;
    c->info.n_ranks = (uint32_t)n, c->info.rank = (uint32_t)i, c->info.transport = m->transport;
    if (m->transport == RT_TRANSPORT_RCCL) c->nccl = m->nccl[i];  // owned by the context
    m->comm.push_back(std::move(c));
    m->stream.push_back(nullptr);
    HIP_TRY(hipSetDevice(devices[i]));
    HIP_TRY(hipStreamCreateWithFlags(&m->stream[i], hipStreamNonBlocking));
  }
  HIP_TRY(hipSetDevice(devices[0]));
  HIP_TRY(hipEventCreateWithFlags(&m->peer_done, hipEventDisableTiming));
  *out = m.get();
  g_multi[key] = std::move(m);
  return RT_OK;
}

}  // namespace

extern "C" {

void rt_multi_release(void) {
  std::lock_guard<std::mutex> lock(g_multi_mutex);
  g_multi.clear();
}

int rt_render_multi(rt_scene* const* per_gpu, int n_gpu, const rt_params* params, uint32_t* argb, rt_stats* stats) {
  if (!per_gpu || n_gpu <= 0 || !argb) return fail(RT_ERR_INVALID_ARG, "null argument");
  if (n_gpu > 64) return fail(RT_ERR_UNSUPPORTED, "more than 64 GPUs");
  int rc = rt_validate_params(params);
  if (rc != RT_OK) return rc;
  std::vector<int> devices;
  for (int i = 0; i < n_gpu; i++) {
    if (!per_gpu[i]) return fail(RT_ERR_INVALID_ARG, "per_gpu[%d] is null", i);
    for (int j = 0; j < i; j++)
      if (per_gpu[j] == per_gpu[i]) return fail(RT_ERR_INVALID_ARG, "per_gpu[%d] and per_gpu[%d] are the same scene object", j, i);
    devices.push_back(per_gpu[i]->device);
  }
  std::lock_guard<std::mutex> lock(g_multi_mutex);  // one multi-GPU frame at a time per process
  auto t_begin = std::chrono::steady_clock::now();
  MultiCtx* m = nullptr;
  if ((rc = get_ctx(devices, params->tuning.multi_force_rccl != 0, &m)) != RT_OK) return rc;
  const size_t n = (size_t)n_gpu;
  const size_t npix = (size_t)params->width * params->height;

  // root frame: the caller's fill of the window goes up (miss pixels keep it), the finished window comes back
  HIP_TRY(hipSetDevice(devices[0]));
  if ((rc = m->fb.ensure(npix * 4)) != RT_OK) return rc;
  const uint32_t wx = params->win_w ? params->win_x0 : 0u, wy = params->win_w ? params->win_y0 : 0u;
  const uint32_t ww = params->win_w ? params->win_w : params->width, wh = params->win_w ? params->win_h : params->height;
  const size_t first = (size_t)wy * params->width + wx;
  auto copy_window = [&](void* dst, const void* src, hipMemcpyKind kind) -> hipError_t {
    if (ww == params->width) return hipMemcpy((char*)dst + first * 4, (const char*)src + first * 4, (size_t)wh * params->width * 4, kind);
    return hipMemcpy2D((char*)dst + first * 4, (size_t)params->width * 4, (const char*)src + first * 4, (size_t)params->width * 4,
                       (size_t)ww * 4, wh, kind);
  };
  HIP_TRY(copy_window(m->fb.p, argb, hipMemcpyHostToDevice));

  // render: one host thread per GPU (a frame with reflections / refractions blocks on its ray-queue levels)
  std::vector<int> rcs(n, RT_OK);
  std::vector<std::string> msgs(n);
  auto work = [&](size_t i) {
    rcs[i] = m->comm[i]->render(per_gpu[i], params, i == 0 ? (uint32_t*)m->fb.p : nullptr, m->stream[i]);
    if (rcs[i] != RT_OK) msgs[i] = rt_last_error();
  };
  if (n == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (size_t i = 0; i < n; i++) th.emplace_back(work, i);
    for (auto& t : th) t.join();
  }
  for (size_t i = 0; i < n; i++)
    if (rcs[i] != RT_OK) {
      for (size_t j = 0; j < n; j++) {  // drain what the other GPUs enqueued before reporting
        (void)hipSetDevice(devices[j]);
        (void)hipStreamSynchronize(m->stream[j]);
      }
      return fail(rcs[i], "GPU %zu (device %d): %s", i, devices[i], msgs[i].c_str());
    }

  // gather
  if (m->transport == RT_TRANSPORT_RCCL) {
    NCCL_TRY(ncclGroupStart());
    for (size_t i = 0; i < n && rc == RT_OK; i++) {
      (void)hipSetDevice(devices[i]);
      rc = m->comm[i]->exchange(m->stream[i]);
    }
    ncclResult_t ge = ncclGroupEnd();
    if (rc != RT_OK) return rc;
    if (ge != ncclSuccess) return fail(RT_ERR_HIP, "ncclGroupEnd failed: %s", ncclGetErrorString(ge));
  } else if (m->transport == RT_TRANSPORT_LOCAL) {
    // several ranks on one GPU (rehearsal): their staging buffers are copied device to device on the root's stream
    rt_comm* root = m->comm[0].get();
    for (size_t i = 1; i < n; i++) {
      HIP_TRY(hipSetDevice(devices[i]));
      HIP_TRY(hipEventRecord(m->peer_done, m->stream[i]));
      HIP_TRY(hipSetDevice(devices[0]));
      HIP_TRY(hipStreamWaitEvent(m->stream[0], m->peer_done, 0));
      HIP_TRY(hipEventSynchronize(m->peer_done));  // the single event is reused for the next peer
      HIP_TRY(hipMemcpyAsync((uint32_t*)root->recv.p + root->lay.rank_off[i], m->comm[i]->stage.p, m->comm[i]->stage_bytes(),
                             hipMemcpyDeviceToDevice, m->stream[0]));
    }
  }
  HIP_TRY(hipSetDevice(devices[0]));
  if ((rc = m->comm[0]->scatter((uint32_t*)m->fb.p, params->width, params->height, m->stream[0])) != RT_OK) return rc;
  for (size_t i = 0; i < n; i++) {
    HIP_TRY(hipSetDevice(devices[i]));
    if ((rc = m->comm[i]->finish(m->stream[i])) != RT_OK) return rc;
  }
  for (size_t i = 0; i < n; i++) {
    HIP_TRY(hipSetDevice(devices[i]));
    HIP_TRY(hipStreamSynchronize(m->stream[i]));
  }
  HIP_TRY(hipSetDevice(devices[0]));
  auto t_copy = std::chrono::steady_clock::now();
  HIP_TRY(copy_window(argb, m->fb.p, hipMemcpyDeviceToHost));
  const double d2h_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_copy).count();

  if (stats) {
    memset(stats, 0, sizeof(*stats));
    for (size_t i = 0; i < n; i++) {
      rt_stats st;
      memset(&st, 0, sizeof(st));
      if ((rc = rt_render_collect_stats(per_gpu[i], &st)) != RT_OK) return rc;
      stats->rays_primary += st.rays_primary;
      stats->rays_reflection += st.rays_reflection;
      stats->rays_refraction += st.rays_refraction;
      stats->rays_shadow += st.rays_shadow;
      stats->pixels_written += st.pixels_written;
      stats->rays_traced += st.rays_traced;
      rt_gather_info gi;
      if ((rc = rt_comm_last_gather(m->comm[i].get(), &gi)) != RT_OK) return rc;
      if (gi.render_ms > stats->kernel_ms) stats->kernel_ms = gi.render_ms;
      if (i == 0) stats->gather_ms = gi.gather_ms;
    }
    stats->d2h_ms = d2h_ms;
    stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  }
  return RT_OK;
}

}  // extern "C"
