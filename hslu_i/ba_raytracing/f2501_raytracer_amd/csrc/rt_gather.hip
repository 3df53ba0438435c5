// rt_gather.hip -- root side of the multi-GPU tile gather: the staged tiles of the other ranks, received into one
// buffer (rank r at recv + rank_off[r]), are copied into the W x H frame.  Reference analogue: every rayon worker
// writes its tiles straight into the shared ImageBuffer (src/image_buffer.rs:48-97,219-251); with one GPU per rank
// the tiles travel over xGMI first (rt_multi.cpp).  HBM-bound byte copy, one thread per pixel of a frame row: a
// wavefront reads up to two runs of a staged tile row and writes 256 contiguous bytes (<= 33 MB per frame at 4K).
#include <hip/hip_runtime.h>

#include "rt_internal.h"

namespace {

__global__ __launch_bounds__(256) void rt_scatter_kernel(uint32_t* __restrict__ argb, const uint32_t* __restrict__ recv,
                                                         const uint32_t* __restrict__ rank_off,
                                                         const uint32_t* __restrict__ tile_slot, uint32_t width, uint32_t height,
                                                         uint32_t tile_size, uint32_t tiles_x, uint32_t n_ranks) {
  const uint32_t gx = blockIdx.x * 256u + threadIdx.x, gy = blockIdx.y;
  if (gx >= width || gy >= height) return;
  const uint32_t tx = gx / tile_size, ty = gy / tile_size;
  const uint32_t owner = rt_tile_owner(tx, ty, n_ranks);
  if (owner == 0u) return;  // the root rendered its own tiles straight into the frame
  const uint32_t src = rank_off[owner] + tile_slot[ty * tiles_x + tx] * tile_size * tile_size + (gy - ty * tile_size) * tile_size +
                       (gx - tx * tile_size);
  const uint32_t v = recv[src];
  // a staged 0 means "no hit": the pixel keeps the caller's fill (image_buffer.rs:27-37); hit pixels carry alpha 0xFF
  if (v != 0u) argb[(size_t)gy * width + gx] = v;
}

}  // namespace

int rt_launch_scatter(uint32_t* argb, const uint32_t* recv, const uint32_t* rank_off, const uint32_t* tile_slot, uint32_t width,
                      uint32_t height, uint32_t tile_size, uint32_t tiles_x, uint32_t n_ranks, void* stream) {
  if (width == 0 || height == 0) return 0;
  hipLaunchKernelGGL(rt_scatter_kernel, dim3((width + 255u) / 256u, height), dim3(256), 0, (hipStream_t)stream, argb, recv, rank_off,
                     tile_slot, width, height, tile_size, tiles_x, n_ranks);
  return (int)hipGetLastError();
}
