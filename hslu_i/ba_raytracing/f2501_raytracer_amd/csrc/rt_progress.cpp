// rt_progress.cpp -- progressive read-back behind the C ABI: rt_render_begin / rt_render_poll / rt_render_end.
//
// Reference: the render thread fills the shared ImageBuffer tile by tile while the UI thread keeps blitting the same
// buffer (src/main.rs:327-347: thread::spawn(render) + WindowOutput::render_loop reading get_u32_slice,
// src/image_buffer.rs:39-44,243-250: relaxed atomic stores / loads).  Here `begin` starts a render thread of the library's own:
// it renders the frame in bands of tile rows (the window parameter of the ABI: ChunkView, image_buffer.rs:178-251) on a stream
// of its own, each band followed by the copy of its rows into a pinned host buffer, and publishes how many rows have landed;
// `poll` -- the UI thread's side, never blocking -- copies the rows that are new into the caller's buffer; `end` joins.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "rt_host.h"

struct rt_progress {
  rt_scene* scene = nullptr;
  rt_params params{};
  std::vector<float> aa, cloud;  // the caller's tables (its arrays may go away after `begin`)
  uint32_t* argb = nullptr;      // the caller's frame (host, W*H): filled by poll / end
  uint32_t* pinned = nullptr;    // host staging the bands are copied into
  uint32_t wx = 0, wy = 0, ww = 0, wh = 0, band_rows = 0;
  hipStream_t stream = nullptr;
  std::thread worker;
  std::atomic<uint32_t> rows_done{0};  // rows of the window (from its top) that are final in `pinned`
  std::atomic<int> finished{0};
  uint32_t rows_copied = 0;            // ... that poll has handed to the caller
  int rc = RT_OK;
  std::string err;
  rt_stats stats{};
  std::chrono::steady_clock::time_point t0;
};

static void progress_worker(rt_progress* pr) {
  rt_scene* s = pr->scene;
  auto bail = [&](int rc) {
    pr->rc = rc;
    pr->err = rt_last_error();
    pr->finished.store(1, std::memory_order_release);
  };
  if (hipSetDevice(s->device) != hipSuccess) return bail(fail(RT_ERR_HIP, "hipSetDevice failed on the render thread"));
  uint32_t* fb = (uint32_t*)s->progress_fb.p;
  const size_t W = pr->params.width;
  for (uint32_t y = 0; y < pr->wh; y += pr->band_rows) {
    const uint32_t h = std::min(pr->band_rows, pr->wh - y);
    rt_params pb = pr->params;
    pb.win_x0 = pr->wx, pb.win_y0 = pr->wy + y, pb.win_w = pr->ww, pb.win_h = h;
    int rc = rt_render_device(s, &pb, fb, nullptr, pr->stream);
    if (rc != RT_OK) return bail(rc);
    const size_t first = (size_t)(pr->wy + y) * W + pr->wx;
    hipError_t e = pr->ww == W ? hipMemcpyAsync(pr->pinned + first, fb + first, (size_t)h * W * 4, hipMemcpyDeviceToHost, pr->stream)
                               : hipMemcpy2DAsync(pr->pinned + first, W * 4, fb + first, W * 4, (size_t)pr->ww * 4, h, hipMemcpyDeviceToHost, pr->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(pr->stream);
    if (e != hipSuccess) return bail(fail(RT_ERR_HIP, "progressive band copy failed: %s", hipGetErrorString(e)));
    rt_stats st{};
    if ((rc = rt_render_collect_stats(s, &st)) != RT_OK) return bail(rc);
    pr->stats.rays_primary += st.rays_primary, pr->stats.rays_reflection += st.rays_reflection, pr->stats.rays_refraction += st.rays_refraction;
    pr->stats.rays_shadow += st.rays_shadow, pr->stats.pixels_written += st.pixels_written, pr->stats.rays_traced += st.rays_traced;
    pr->stats.notes |= st.notes;
    pr->stats.queue_bytes = std::max(pr->stats.queue_bytes, st.queue_bytes);
    pr->stats.scene_bytes = st.scene_bytes;
    pr->rows_done.store(y + h, std::memory_order_release);  // (the band's rows are in `pinned`: the copy has completed)
  }
  pr->finished.store(1, std::memory_order_release);
}

// copies the rows [rows_copied, upto) of the window from the pinned staging into the caller's frame
static void hand_over(rt_progress* pr, uint32_t upto) {
  const size_t W = pr->params.width;
  for (uint32_t y = pr->rows_copied; y < upto; y++) {
    const size_t at = (size_t)(pr->wy + y) * W + pr->wx;
    memcpy(pr->argb + at, pr->pinned + at, (size_t)pr->ww * 4);
  }
  pr->rows_copied = std::max(pr->rows_copied, upto);
}

extern "C" {

int rt_render_begin(rt_scene* s, const rt_params* p, uint32_t* argb, uint32_t band_rows, rt_progress** out) {
  if (!s || !argb || !out) return fail(RT_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  int rc = rt_validate_params(p);
  if (rc != RT_OK) return rc;
  if (s->progress_active) return fail(RT_ERR_INVALID_ARG, "a progressive render of this scene is already running (rt_render_end it first)");
  HIP_TRY(hipSetDevice(s->device));
  rt_progress* pr = new rt_progress();
  pr->scene = s;
  pr->params = *p;
  // the tables the frame reads for as long as it runs: private copies
  if ((p->flags & RT_FLAG_ANTI_ALIASING) && p->aa_rays && p->aa_offsets) {
    pr->aa.assign(p->aa_offsets, p->aa_offsets + 2 * (size_t)p->aa_rays);
    pr->params.aa_offsets = pr->aa.data();
  }
  if (p->light_mult > 1 && p->cloud_sets) {
    pr->cloud.assign(p->cloud_sets, p->cloud_sets + (size_t)p->n_cloud_sets * p->light_mult * 3);
    pr->params.cloud_sets = pr->cloud.data();
  }
  pr->argb = argb;
  pr->wx = p->win_w ? p->win_x0 : 0u, pr->wy = p->win_w ? p->win_y0 : 0u;
  pr->ww = p->win_w ? p->win_w : p->width, pr->wh = p->win_w ? p->win_h : p->height;
  const uint32_t ts = p->tile_size ? p->tile_size : 48u;
  pr->band_rows = band_rows ? band_rows : ts;  // RENDER_STRIDE rows: one row of the reference's tiles (renderer/mod.rs:84-90)
  pr->t0 = std::chrono::steady_clock::now();
  const size_t npix = (size_t)p->width * p->height;
  auto cleanup = [&](int code) {
    if (pr->pinned) (void)hipHostFree(pr->pinned);
    if (pr->stream) (void)hipStreamDestroy(pr->stream);
    delete pr;
    return code;
  };
  if ((rc = s->progress_fb.ensure(npix * 4)) != RT_OK) return cleanup(rc);
  hipError_t e = hipHostMalloc((void**)&pr->pinned, npix * 4, hipHostMallocDefault);
  if (e != hipSuccess) return cleanup(fail(RT_ERR_OOM, "hipHostMalloc(%zu) failed: %s", npix * 4, hipGetErrorString(e)));
  if ((e = hipStreamCreateWithFlags(&pr->stream, hipStreamNonBlocking)) != hipSuccess)
    return cleanup(fail(RT_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e)));
  // the caller's fill of the window's rows goes up (miss pixels keep it, image_buffer.rs:27-37) -- and into the staging, so that
  // rows handed over early are the caller's own pixels where nothing was hit
  const size_t first = (size_t)pr->wy * p->width + pr->wx, span = (size_t)(pr->wh - 1) * p->width + pr->ww;
  memcpy(pr->pinned + first, argb + first, span * 4);
  if ((e = hipMemcpyAsync((uint32_t*)s->progress_fb.p + first, pr->pinned + first, span * 4, hipMemcpyHostToDevice, pr->stream)) != hipSuccess)
    return cleanup(fail(RT_ERR_HIP, "hipMemcpyAsync H2D failed: %s", hipGetErrorString(e)));
  s->progress_active = true;
  pr->worker = std::thread(progress_worker, pr);
  *out = pr;
  return RT_OK;
}

int rt_render_poll(rt_progress* pr, uint32_t* rows_done, int* finished) {
  if (!pr) return fail(RT_ERR_INVALID_ARG, "null argument");
  const int fin = pr->finished.load(std::memory_order_acquire);
  const uint32_t r = pr->rows_done.load(std::memory_order_acquire);
  hand_over(pr, r);
  if (rows_done) *rows_done = r;
  if (finished) *finished = fin;
  return RT_OK;
}

int rt_render_end(rt_progress* pr, rt_stats* stats) {
  if (!pr) return fail(RT_ERR_INVALID_ARG, "null argument");
  if (pr->worker.joinable()) pr->worker.join();
  hand_over(pr, pr->rows_done.load(std::memory_order_acquire));
  const int rc = pr->rc;
  const std::string err = pr->err;
  if (stats) {
    *stats = pr->stats;
    stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - pr->t0).count();
  }
  (void)hipSetDevice(pr->scene->device);
  (void)hipStreamSynchronize(pr->stream);
  (void)hipStreamDestroy(pr->stream);
  (void)hipHostFree(pr->pinned);
  pr->scene->progress_active = false;
  delete pr;
  if (rc != RT_OK) return fail(rc, "%s", err.c_str());
  return RT_OK;
}

}  // extern "C"
