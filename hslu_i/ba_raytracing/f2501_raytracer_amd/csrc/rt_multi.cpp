// rt_multi.cpp -- the multi-GPU side of `Renderer::render` (reference src/renderer/mod.rs:80-94,146-209: one frame,
// RENDER_STRIDE tiles handed to parallel workers, all writing one ImageBuffer, src/image_buffer.rs:48-97).
//
// Every GPU renders the tiles it owns (rt_tile_owner, include/rt_hip.h) straight into a rank-compact staging buffer;
// the only communication is ONE gather of those buffers to the root:
//     ncclGroupStart;  root: ncclRecv x (n-1);  others: ncclSend;  ncclGroupEnd;  root: scatter kernel.
// xGMI is point to point: each peer has its own link into the root, so the n-1 transfers run side by side (no ring),
// <= W*H*4/n bytes each (4.1 MB per peer for a 4K frame over 8 GPUs).
//
// TWO FRAMES IN FLIGHT.  A rank's share of a frame is a short launch (0.7 ms for config 3 over 8 GPUs) that cannot end
// before its longest wavefront does (up to 1 ms), so a rank that renders frame after frame on one stream idles through
// every drain.  Staging and receive buffers are therefore double buffered and the gather runs on a stream of its own:
// the caller may enqueue frame k+1 (on another stream) while frame k drains and travels; the gathers themselves stay in
// call order on the communicator's stream (RCCL wants one order on all ranks).
//
// Two host shapes are served: one process driving all GPUs (rt_render_multi[_begin/_end] -- the reference's single
// process) and one process per GPU (rt_comm_* + rt_render_gather_device -- bench.py under torch.distributed.run).
//
// RCCL is loaded on first use (dlopen): a single-GPU consumer of librt_hip.so never maps it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

#include "rt_host.h"

static_assert(RT_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "rt_comm id is an ncclUniqueId");

namespace {

// ---- RCCL entry points, resolved on first use ----------------------------------------------------------------------
// Symbols already in the process win (a launcher that loaded RCCL itself, or tests/mock_rccl under LD_PRELOAD);
// otherwise librccl.so.1 is opened -- by soname, so a copy another library of the process brought along is shared.
struct Rccl {
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclCommUserRank) CommUserRank = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string error;  // empty = usable
};

const Rccl& rccl() {
  static Rccl* r = [] {
    Rccl* q = new Rccl();  // (never destroyed: must outlive every static destructor that could still call it)
    void* lib = nullptr;
    auto sym = [&](const char* name) -> void* {
      if (void* p = dlsym(RTLD_DEFAULT, name)) return p;
      if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
      if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
      if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
      return lib ? dlsym(lib, name) : nullptr;
    };
#define RT_RCCL_SYM(field, name)                                       \
  q->field = reinterpret_cast<decltype(q->field)>(sym(#name));         \
  if (!q->field && q->error.empty()) q->error = std::string("RCCL entry point ") + #name + " not found (librccl.so.1 missing?)"
    RT_RCCL_SYM(GetUniqueId, ncclGetUniqueId);
    RT_RCCL_SYM(CommInitRank, ncclCommInitRank);
    RT_RCCL_SYM(CommInitAll, ncclCommInitAll);
    RT_RCCL_SYM(CommDestroy, ncclCommDestroy);
    RT_RCCL_SYM(CommAbort, ncclCommAbort);
    RT_RCCL_SYM(CommCount, ncclCommCount);
    RT_RCCL_SYM(CommUserRank, ncclCommUserRank);
    RT_RCCL_SYM(Send, ncclSend);
    RT_RCCL_SYM(Recv, ncclRecv);
    RT_RCCL_SYM(GroupStart, ncclGroupStart);
    RT_RCCL_SYM(GroupEnd, ncclGroupEnd);
    RT_RCCL_SYM(GetErrorString, ncclGetErrorString);
#undef RT_RCCL_SYM
    return q;
  }();
  return *r;
}

#define RCCL_READY()                                                               \
  do {                                                                             \
    if (!rccl().error.empty()) return fail(RT_ERR_UNSUPPORTED, "%s", rccl().error.c_str()); \
  } while (0)

#define NCCL_TRY(expr)                                                                                         \
  do {                                                                                                         \
    ncclResult_t r_ = (expr);                                                                                  \
    if (r_ != ncclSuccess) return fail(RT_ERR_HIP, "%s failed: %s", #expr, rccl().GetErrorString(r_));         \
  } while (0)

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

// one of the two frames a communicator can have in flight
struct FrameSlot {
  DevBuf stage;  // rank != 0: this rank's tiles
  DevBuf recv;   // rank 0: the other ranks' tiles, rank r at rank_off[r]
  hipEvent_t t0 = nullptr, rendered = nullptr, done = nullptr;  // render start / render end / gather end (timed)
  bool pending = false;  // `done` has been recorded and nobody has waited for it on the host yet
};

}  // namespace

struct rt_comm {
  uint32_t n_ranks = 1, rank = 0;
  int device = 0;
  ncclComm_t nccl = nullptr;
  bool owns_nccl = false, dead = false;
  uint32_t transport = RT_TRANSPORT_NONE;
  Layout lay;
  DevBuf slot_dev, off_dev;  // tile -> slot, rank -> receive offset (device copies of the layout)
  FrameSlot fs[2];
  uint32_t next = 0;
  int cur = 0;  // slot of the frame being enqueued / enqueued last
  hipStream_t comm_stream = nullptr;  // the gathers run here, in call order
  bool timed = false;
  rt_gather_info info{};

  size_t stage_bytes() const { return (size_t)lay.count[rank] * lay.tile_size * lay.tile_size * 4; }

  int ensure_layout(const rt_params* p) {
    const uint32_t ts = p->tile_size ? p->tile_size : 48u;
    if (!lay.matches(p->width, p->height, ts, n_ranks)) {
      // (frames in flight still read the old tables and buffers)
      for (FrameSlot& f : fs)
        if (f.pending) {
          HIP_TRY(hipEventSynchronize(f.done));
          f.pending = false;
        }
      lay.build(p->width, p->height, ts, n_ranks);
      int rc;
      if ((rc = slot_dev.ensure(lay.slot.size() * 4)) != RT_OK) return rc;
      if ((rc = off_dev.ensure(lay.rank_off.size() * 4)) != RT_OK) return rc;
      HIP_TRY(hipMemcpy(slot_dev.p, lay.slot.data(), lay.slot.size() * 4, hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(off_dev.p, lay.rank_off.data(), lay.rank_off.size() * 4, hipMemcpyHostToDevice));
    }
    FrameSlot& f = fs[cur];
    if (rank == 0) return f.recv.ensure(lay.recv_pixels * 4 + 4);
    return f.stage.ensure(stage_bytes() + 4);
  }
  int ensure_objects() {
    for (FrameSlot& f : fs)
      for (hipEvent_t* e : {&f.t0, &f.rendered, &f.done})
        if (!*e) HIP_TRY(hipEventCreate(e));
    if (!comm_stream && n_ranks > 1) HIP_TRY(hipStreamCreateWithFlags(&comm_stream, hipStreamNonBlocking));
    return RT_OK;
  }
  // Everything that can fail BEFORE this rank has anything to render: a failure here leaves the peers of a
  // process-per-GPU run without a partner for their send / receive, so the caller aborts the communicator.
  // want_slot: which of the two frame slots to use (rt_render_multi keeps its ranks in step with its own frame slots);
  // -1: alternate.
  int frame_open(const rt_scene* s, const rt_params* params, rt_params* p, const uint32_t* argb_dev, hipStream_t stream, int want_slot = -1) {
    if (dead) return fail(RT_ERR_HIP, "the communicator was aborted by an earlier failure");
    if (s->device != device) return fail(RT_ERR_INVALID_ARG, "scene lives on device %d, communicator on %d", s->device, device);
    *p = *params;
    p->n_ranks = n_ranks;  // (the caller's n_ranks / rank are ignored, stale values included)
    p->rank = rank;
    int rc = rt_validate_params(p);
    if (rc != RT_OK) return rc;
    if (rank == 0 && !argb_dev) return fail(RT_ERR_INVALID_ARG, "rank 0 needs the frame buffer");
    HIP_TRY(hipSetDevice(device));
    if ((rc = ensure_objects()) != RT_OK) return rc;
    cur = want_slot >= 0 ? (want_slot & 1) : (int)(next++ & 1u);
    FrameSlot& f = fs[cur];
    // the frame that used this slot last must have left it (its gather is behind `done`)
    if (f.pending) HIP_TRY(hipStreamWaitEvent(stream, f.done, 0));
    if ((rc = ensure_layout(p)) != RT_OK) return rc;
    return RT_OK;
  }
  // this rank's tiles -> staging (rank != 0) or straight into the frame (rank 0); enqueued on `stream`.  A failure
  // leaves a zeroed staging buffer behind ("no hit" everywhere): the rank can still take part in the gather.
  int frame_render(rt_scene* s, const rt_params* p, uint32_t* argb_dev, hipStream_t stream) {
    FrameSlot& f = fs[cur];
    HIP_TRY(hipEventRecord(f.t0, stream));
    int rc;
    if (rank == 0) {
      rc = rt_render_device(s, p, argb_dev, nullptr, stream);
    } else {
      HIP_TRY(hipMemsetAsync(f.stage.p, 0, stage_bytes(), stream));  // 0 = "no hit"
      rc = rt_render_device_staged(s, p, (uint32_t*)f.stage.p, (const uint32_t*)slot_dev.p, lay.tiles_x, stream);
      if (rc != RT_OK) {
        const std::string msg = rt_last_error();
        (void)hipMemsetAsync(f.stage.p, 0, stage_bytes(), stream);
        (void)fail(rc, "%s", msg.c_str());
      }
    }
    hipError_t e = hipEventRecord(f.rendered, stream);
    if (e == hipSuccess && comm_stream) e = hipStreamWaitEvent(comm_stream, f.rendered, 0);
    if (rc != RT_OK) return rc;
    if (e != hipSuccess) return fail(RT_ERR_HIP, "event after the render failed: %s", hipGetErrorString(e));
    info.tiles_owned = lay.count[rank];
    info.bytes_sent = rank ? stage_bytes() : 0;
    info.bytes_received = rank ? 0 : lay.recv_pixels * 4;
    return RT_OK;
  }
  // this rank's send / receives on its gather stream; the caller brackets the ranks of one process with
  // ncclGroupStart / ncclGroupEnd
  int exchange() {
    FrameSlot& f = fs[cur];
    if (rank == 0) {
      for (uint32_t r = 1; r < n_ranks; r++)
        NCCL_TRY(rccl().Recv((uint32_t*)f.recv.p + lay.rank_off[r], (size_t)lay.count[r] * lay.tile_size * lay.tile_size, ncclUint32, (int)r,
                             nccl, comm_stream));
    } else {
      NCCL_TRY(rccl().Send(f.stage.p, stage_bytes() / 4, ncclUint32, 0, nccl, comm_stream));
    }
    return RT_OK;
  }
  int scatter(uint32_t* argb_dev, uint32_t width, uint32_t height) {
    if (n_ranks > 1) {
      hipError_t e = (hipError_t)rt_launch_scatter(argb_dev, (const uint32_t*)fs[cur].recv.p, (const uint32_t*)off_dev.p,
                                                   (const uint32_t*)slot_dev.p, width, height, lay.tile_size, lay.tiles_x, n_ranks, comm_stream);
      if (e != hipSuccess) return fail(RT_ERR_HIP, "scatter launch failed: %s", hipGetErrorString(e));
    }
    return RT_OK;
  }
  // end of the frame: `done` behind the gather; the caller's stream continues only after it (stream semantics of the
  // call: whatever is enqueued on `stream` next sees the gathered frame)
  int frame_close(hipStream_t stream) {
    FrameSlot& f = fs[cur];
    HIP_TRY(hipEventRecord(f.done, comm_stream ? comm_stream : stream));
    if (comm_stream) HIP_TRY(hipStreamWaitEvent(stream, f.done, 0));
    f.pending = true;
    timed = true;
    return RT_OK;
  }
  void abort_comm() {
    if (nccl && owns_nccl && rccl().CommAbort) (void)rccl().CommAbort(nccl);
    nccl = nullptr;
    dead = true;
  }
  void release() {
    (void)hipSetDevice(device);
    for (FrameSlot& f : fs) {
      if (f.pending) (void)hipEventSynchronize(f.done);
      for (hipEvent_t* e : {&f.t0, &f.rendered, &f.done})
        if (*e) (void)hipEventDestroy(*e), *e = nullptr;
      f.stage.release(), f.recv.release();
      f.pending = false;
    }
    if (comm_stream) (void)hipStreamDestroy(comm_stream), comm_stream = nullptr;
    for (DevBuf* b : {&slot_dev, &off_dev}) b->release();
    if (nccl && owns_nccl && rccl().CommDestroy) (void)rccl().CommDestroy(nccl);
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
  RCCL_READY();
  ncclUniqueId u;
  NCCL_TRY(rccl().GetUniqueId(&u));
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
    RCCL_READY();
    ncclUniqueId u;
    memcpy(u.internal, id, RT_COMM_ID_BYTES);
    NCCL_TRY(rccl().CommInitRank(&c->nccl, (int)n_ranks, u, (int)rank));
    c->owns_nccl = true;
    c->transport = RT_TRANSPORT_RCCL;
    int cnt = 0, ur = -1;
    NCCL_TRY(rccl().CommCount(c->nccl, &cnt));
    NCCL_TRY(rccl().CommUserRank(c->nccl, &ur));
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
  rt_params p;
  int rc = c->frame_open(s, params, &p, argb_dev, stream);
  if (rc != RT_OK) {
    // Nothing of this frame was enqueued and this rank cannot take part in its gather: the peers' sends / receives would
    // wait for it forever.  Aborting the communicator fails them instead (the communicator is unusable afterwards).
    if (c->n_ranks > 1 && !c->dead) {
      const std::string msg = rt_last_error();
      c->abort_comm();
      return fail(rc, "%s (communicator aborted: this rank could not take part in the gather)", msg.c_str());
    }
    return rc;
  }
  // A render that fails (out of memory in the ray queues, a launch error) must not leave the other ranks alone in the
  // gather either: the rank still sends its (zeroed) staging buffer / posts its receives, and reports the error afterwards.
  const int rc_render = c->frame_render(s, &p, argb_dev, stream);
  const std::string msg_render = rc_render != RT_OK ? rt_last_error() : "";
  if (c->n_ranks > 1) {
    NCCL_TRY(rccl().GroupStart());
    rc = c->exchange();
    ncclResult_t ge = rccl().GroupEnd();
    if (rc != RT_OK) return rc;
    if (ge != ncclSuccess) return fail(RT_ERR_HIP, "ncclGroupEnd failed: %s", rccl().GetErrorString(ge));
    if (c->rank == 0 && (rc = c->scatter(argb_dev, p.width, p.height)) != RT_OK) return rc;
  }
  if ((rc = c->frame_close(stream)) != RT_OK) return rc;
  if (rc_render != RT_OK) return fail(rc_render, "%s", msg_render.c_str());
  return RT_OK;
}

int rt_comm_last_gather(rt_comm* c, rt_gather_info* out) {
  if (!c || !out) return fail(RT_ERR_INVALID_ARG, "null argument");
  if (!c->timed) return fail(RT_ERR_INVALID_ARG, "no gather has been enqueued on this communicator");
  HIP_TRY(hipSetDevice(c->device));
  FrameSlot& f = c->fs[c->cur];
  HIP_TRY(hipEventSynchronize(f.done));
  f.pending = false;
  float a = 0.f, b = 0.f;
  HIP_TRY(hipEventElapsedTime(&a, f.t0, f.rendered));
  HIP_TRY(hipEventElapsedTime(&b, f.rendered, f.done));
  c->info.render_ms = a;
  c->info.gather_ms = b;
  *out = c->info;
  return RT_OK;
}

}  // extern "C"

// ---- one process, all GPUs ---------------------------------------------------------------------------------------
namespace {

struct MultiFrame {  // one of the two frames of a context that can be in flight
  DevBuf fb;         // root: W x H frame
  bool open = false;
  uint32_t* argb = nullptr;  // the caller's host buffer
  rt_params params{};
  std::vector<rt_scene*> scenes;
  std::vector<int> slots;  // the frame slot each scene rendered this frame in (its counters)
  std::chrono::steady_clock::time_point t_begin;
  uint32_t generation = 0;
};

struct MultiCtx {
  std::vector<int> devices;
  std::vector<std::unique_ptr<rt_comm>> comm;
  std::vector<hipStream_t> stream[2];  // render streams, one per GPU and frame slot
  std::vector<ncclComm_t> nccl;        // ncclCommInitAll
  uint32_t transport = RT_TRANSPORT_NONE;
  MultiFrame frame[2];
  hipEvent_t peer_done = nullptr;
  int id = 0;  // part of a frame's ticket
};

// The cache of contexts is never destroyed: at process exit the HIP and RCCL runtimes may already be gone when static
// destructors run, and tearing streams / communicators down then is a known way to crash or hang.  rt_multi_release()
// frees everything explicitly while the runtimes are alive.
std::mutex& multi_mutex() {
  static std::mutex* m = new std::mutex();
  return *m;
}
std::map<std::vector<int>, MultiCtx*>& multi_cache() {  // key: devices, then the force-RCCL flag
  static auto* m = new std::map<std::vector<int>, MultiCtx*>();
  return *m;
}

void destroy_ctx(MultiCtx* m) {
  for (size_t i = 0; i < m->comm.size(); i++) {
    m->comm[i]->release();
    (void)hipSetDevice(m->devices[i]);
    for (auto& sv : m->stream)
      if (i < sv.size() && sv[i]) (void)hipStreamDestroy(sv[i]);
  }
  for (ncclComm_t c : m->nccl)
    if (c && rccl().CommDestroy) (void)rccl().CommDestroy(c);
  if (!m->devices.empty()) {
    (void)hipSetDevice(m->devices[0]);
    for (MultiFrame& f : m->frame) f.fb.release();
    if (m->peer_done) (void)hipEventDestroy(m->peer_done);
  }
  delete m;
}

int get_ctx(const std::vector<int>& devices, bool force_rccl, MultiCtx** out) {
  std::vector<int> key(devices);
  key.push_back(force_rccl ? 1 : 0);
  auto& cache = multi_cache();
  auto it = cache.find(key);
  if (it != cache.end()) {
    *out = it->second;
    return RT_OK;
  }
  std::unique_ptr<MultiCtx, void (*)(MultiCtx*)> m(new MultiCtx(), destroy_ctx);
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
    RCCL_READY();
    m->nccl.assign(n, nullptr);
    NCCL_TRY(rccl().CommInitAll(m->nccl.data(), (int)n, devices.data()));
  }
  for (size_t i = 0; i < n; i++) {
    std::unique_ptr<rt_comm> c(new rt_comm());
    c->n_ranks = (uint32_t)n, c->rank = (uint32_t)i, c->device = devices[i];
    c->transport = m->transport;
    c->info.n_ranks = (uint32_t)n, c->info.rank = (uint32_t)i, c->info.transport = m->transport;
    if (m->transport == RT_TRANSPORT_RCCL) c->nccl = m->nccl[i];  // owned by the context
    m->comm.push_back(std::move(c));
    HIP_TRY(hipSetDevice(devices[i]));
    for (auto& sv : m->stream) {
      sv.push_back(nullptr);
      HIP_TRY(hipStreamCreateWithFlags(&sv[i], hipStreamNonBlocking));
    }
  }
  HIP_TRY(hipSetDevice(devices[0]));
  HIP_TRY(hipEventCreateWithFlags(&m->peer_done, hipEventDisableTiming));
  static int next_id = 1;
  m->id = next_id++ & 0x7FFFFF;
  *out = m.get();
  cache[key] = m.release();
  return RT_OK;
}

int multi_begin_locked(rt_scene* const* per_gpu, int n_gpu, const rt_params* params, uint32_t* argb, int* ticket) {
  if (!per_gpu || n_gpu <= 0 || !argb || !params || !ticket) return fail(RT_ERR_INVALID_ARG, "null argument");
  if (n_gpu > 64) return fail(RT_ERR_UNSUPPORTED, "more than 64 GPUs");
  {
    rt_params v = *params;  // (n_ranks / rank are ignored: validate what will be rendered, not stale values)
    v.n_ranks = (uint32_t)n_gpu, v.rank = 0;
    int rc = rt_validate_params(&v);
    if (rc != RT_OK) return rc;
  }
  std::vector<int> devices;
  for (int i = 0; i < n_gpu; i++) {
    if (!per_gpu[i]) return fail(RT_ERR_INVALID_ARG, "per_gpu[%d] is null", i);
    for (int j = 0; j < i; j++)
      if (per_gpu[j] == per_gpu[i]) return fail(RT_ERR_INVALID_ARG, "per_gpu[%d] and per_gpu[%d] are the same scene object", j, i);
    devices.push_back(per_gpu[i]->device);
  }
  MultiCtx* m = nullptr;
  int rc = get_ctx(devices, params->tuning.multi_force_rccl != 0, &m);
  if (rc != RT_OK) return rc;
  const int fi = !m->frame[0].open ? 0 : 1;  // a free frame slot
  MultiFrame& F = m->frame[fi];
  if (F.open) return fail(RT_ERR_INVALID_ARG, "two frames are already in flight on these GPUs: call rt_render_multi_end first");
  const size_t n = (size_t)n_gpu;
  const size_t npix = (size_t)params->width * params->height;
  F.t_begin = std::chrono::steady_clock::now();

  // root frame: the caller's fill of the window goes up (miss pixels keep it), the finished window comes back
  HIP_TRY(hipSetDevice(devices[0]));
  if ((rc = F.fb.ensure(npix * 4)) != RT_OK) return rc;
  const uint32_t wx = params->win_w ? params->win_x0 : 0u, wy = params->win_w ? params->win_y0 : 0u;
  const uint32_t ww = params->win_w ? params->win_w : params->width, wh = params->win_w ? params->win_h : params->height;
  const size_t first = (size_t)wy * params->width + wx;
  {
    hipStream_t s0 = m->stream[fi][0];
    hipError_t e = ww == params->width
                       ? hipMemcpyAsync((char*)F.fb.p + first * 4, (const char*)argb + first * 4, (size_t)wh * params->width * 4, hipMemcpyHostToDevice, s0)
                       : hipMemcpy2DAsync((char*)F.fb.p + first * 4, (size_t)params->width * 4, (const char*)argb + first * 4,
                                          (size_t)params->width * 4, (size_t)ww * 4, wh, hipMemcpyHostToDevice, s0);
    if (e != hipSuccess) return fail(RT_ERR_HIP, "upload of the frame's fill failed: %s", hipGetErrorString(e));
  }

  // render.  Frames with reflections / refractions block their host thread on the ray-queue levels: one thread per GPU.
  std::vector<rt_params> ps(n);
  for (size_t i = 0; i < n; i++)
    if ((rc = m->comm[i]->frame_open(per_gpu[i], params, &ps[i], i == 0 ? (uint32_t*)F.fb.p : (uint32_t*)nullptr, m->stream[fi][i], fi)) != RT_OK) return rc;
  std::vector<int> rcs(n, RT_OK);
  std::vector<std::string> msgs(n);
  F.slots.assign(n, 0);
  auto work = [&](size_t i) {
    (void)hipSetDevice(devices[i]);
    rcs[i] = m->comm[i]->frame_render(per_gpu[i], &ps[i], i == 0 ? (uint32_t*)F.fb.p : nullptr, m->stream[fi][i]);
    F.slots[i] = per_gpu[i]->cur_block;
    if (rcs[i] != RT_OK) msgs[i] = rt_last_error();
  };
  const bool blocking = (params->flags & (RT_FLAG_REFLECTIONS | RT_FLAG_REFRACTIONS)) != 0;
  if (n == 1 || !blocking) {
    for (size_t i = 0; i < n; i++) work(i);
  } else {
    std::vector<std::thread> th;
    for (size_t i = 0; i < n; i++) th.emplace_back(work, i);
    for (auto& t : th) t.join();
  }
  for (size_t i = 0; i < n; i++)
    if (rcs[i] != RT_OK) {
      for (size_t j = 0; j < n; j++) {  // drain what the other GPUs enqueued before reporting
        (void)hipSetDevice(devices[j]);
        (void)hipStreamSynchronize(m->stream[fi][j]);
      }
      return fail(rcs[i], "GPU %zu (device %d): %s", i, devices[i], msgs[i].c_str());
    }

  // gather (on the communicators' own streams, behind each rank's render)
  if (m->transport == RT_TRANSPORT_RCCL) {
    NCCL_TRY(rccl().GroupStart());
    for (size_t i = 0; i < n && rc == RT_OK; i++) {
      (void)hipSetDevice(devices[i]);
      rc = m->comm[i]->exchange();
    }
    ncclResult_t ge = rccl().GroupEnd();
    if (rc != RT_OK) return rc;
    if (ge != ncclSuccess) return fail(RT_ERR_HIP, "ncclGroupEnd failed: %s", rccl().GetErrorString(ge));
  } else if (m->transport == RT_TRANSPORT_LOCAL) {
    // several ranks on one GPU (rehearsal): their staging buffers are copied device to device on the root's gather stream
    rt_comm* root = m->comm[0].get();
    HIP_TRY(hipSetDevice(devices[0]));
    for (size_t i = 1; i < n; i++) {
      rt_comm* c = m->comm[i].get();
      HIP_TRY(hipStreamWaitEvent(root->comm_stream, c->fs[c->cur].rendered, 0));
      HIP_TRY(hipMemcpyAsync((uint32_t*)root->fs[root->cur].recv.p + root->lay.rank_off[i], c->fs[c->cur].stage.p, c->stage_bytes(),
                             hipMemcpyDeviceToDevice, root->comm_stream));
    }
    // a peer's staging buffer is free again once the root has copied it
    HIP_TRY(hipEventRecord(m->peer_done, root->comm_stream));
    for (size_t i = 1; i < n; i++) HIP_TRY(hipStreamWaitEvent(m->comm[i]->comm_stream, m->peer_done, 0));
  }
  HIP_TRY(hipSetDevice(devices[0]));
  if ((rc = m->comm[0]->scatter((uint32_t*)F.fb.p, params->width, params->height)) != RT_OK) return rc;
  for (size_t i = 0; i < n; i++) {
    HIP_TRY(hipSetDevice(devices[i]));
    if ((rc = m->comm[i]->frame_close(m->stream[fi][i])) != RT_OK) return rc;
  }
  F.open = true;
  F.argb = argb;
  F.params = *params;
  F.scenes.assign(per_gpu, per_gpu + n);
  F.generation++;
  // ticket: which context, which frame slot, which use of it
  static_assert(sizeof(int) >= 4, "ticket");
  *ticket = (m->id << 8) | (int)((F.generation & 0x7Fu) << 1) | fi;
  return RT_OK;
}

int multi_end_locked(int ticket, rt_stats* stats) {
  const int ctx_id = ticket >> 8, fi = ticket & 1;
  MultiCtx* m = nullptr;
  for (auto& kv : multi_cache())
    if (kv.second->id == ctx_id) m = kv.second;
  if (ticket < 0 || !m) return fail(RT_ERR_INVALID_ARG, "unknown ticket");
  MultiFrame& F = m->frame[fi];
  if (!F.open || (int)((F.generation & 0x7Fu) << 1) != (ticket & 0xFE)) return fail(RT_ERR_INVALID_ARG, "ticket is not in flight");
  F.open = false;
  const size_t n = m->devices.size();
  for (size_t i = 0; i < n; i++) {
    HIP_TRY(hipSetDevice(m->devices[i]));
    HIP_TRY(hipStreamSynchronize(m->stream[fi][i]));  // (waits for the gather too: frame_close made the stream wait for it)
  }
  const rt_params* params = &F.params;
  HIP_TRY(hipSetDevice(m->devices[0]));
  const uint32_t wx = params->win_w ? params->win_x0 : 0u, wy = params->win_w ? params->win_y0 : 0u;
  const uint32_t ww = params->win_w ? params->win_w : params->width, wh = params->win_w ? params->win_h : params->height;
  const size_t first = (size_t)wy * params->width + wx;
  auto t_copy = std::chrono::steady_clock::now();
  if (ww == params->width)
    HIP_TRY(hipMemcpy((char*)F.argb + first * 4, (const char*)F.fb.p + first * 4, (size_t)wh * params->width * 4, hipMemcpyDeviceToHost));
  else
    HIP_TRY(hipMemcpy2D((char*)F.argb + first * 4, (size_t)params->width * 4, (const char*)F.fb.p + first * 4, (size_t)params->width * 4,
                        (size_t)ww * 4, wh, hipMemcpyDeviceToHost));
  const double d2h_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_copy).count();
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    for (size_t i = 0; i < n; i++) {
      rt_stats st;
      memset(&st, 0, sizeof(st));
      int rc;
      if ((rc = rt_collect_stats_slot(F.scenes[i], F.slots[i], &st)) != RT_OK) return rc;  // (this frame's own counter block)
      stats->rays_primary += st.rays_primary;
      stats->rays_reflection += st.rays_reflection;
      stats->rays_refraction += st.rays_refraction;
      stats->rays_shadow += st.rays_shadow;
      stats->pixels_written += st.pixels_written;
      stats->rays_traced += st.rays_traced;
      stats->notes |= st.notes;
      stats->queue_bytes += st.queue_bytes;
      FrameSlot& f = m->comm[i]->fs[fi];
      f.pending = false;  // (its stream has drained)
      float a = 0.f, b = 0.f;
      HIP_TRY(hipSetDevice(m->devices[i]));
      if (hipEventElapsedTime(&a, f.t0, f.rendered) == hipSuccess && a > stats->kernel_ms) stats->kernel_ms = a;
      if (i == 0 && hipEventElapsedTime(&b, f.rendered, f.done) == hipSuccess) stats->gather_ms = b;
    }
    stats->d2h_ms = d2h_ms;
    stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - F.t_begin).count();
  }
  return RT_OK;
}

}  // namespace

extern "C" {

void rt_multi_release(void) {
  std::lock_guard<std::mutex> lock(multi_mutex());
  for (auto& kv : multi_cache()) destroy_ctx(kv.second);
  multi_cache().clear();
}

int rt_render_multi_begin(rt_scene* const* per_gpu, int n_gpu, const rt_params* params, uint32_t* argb, int* ticket) {
  std::lock_guard<std::mutex> lock(multi_mutex());
  return multi_begin_locked(per_gpu, n_gpu, params, argb, ticket);
}

int rt_render_multi_end(int ticket, rt_stats* stats) {
  std::lock_guard<std::mutex> lock(multi_mutex());
  return multi_end_locked(ticket, stats);
}

int rt_render_multi(rt_scene* const* per_gpu, int n_gpu, const rt_params* params, uint32_t* argb, rt_stats* stats) {
  std::lock_guard<std::mutex> lock(multi_mutex());  // one multi-GPU frame at a time through this entry point
  int ticket = -1;
  int rc = multi_begin_locked(per_gpu, n_gpu, params, argb, &ticket);
  if (rc != RT_OK) return rc;
  return multi_end_locked(ticket, stats);
}

}  // extern "C"
