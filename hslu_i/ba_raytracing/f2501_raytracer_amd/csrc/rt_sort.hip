// rt_sort.hip -- hit-point ordering of a ray-tree level: the last two steps of a counting sort whose sizes never leave
// the device.  Reference analogue: none (the reference recurses per ray, src/renderer/raytracer_renderer.rs:147-264,
// 279-729; here a level of the recursion is a queue of rays shaded in the order of their hit points, because 64 rays
// that hit neighbouring surface points share one soft-shadow candidate walk).
//
//   rt_trace_kernel        (rt_kernels.hip) bucket = top bits of the Morton key of the hit point; rank inside the bucket from
//                          the histogram (one atomic per wavefront and distinct bucket); both stored in sort_slot[ray]
//   rt_sort_tiles_kernel   one workgroup per RT_SORT_TILE buckets: exclusive prefix inside the tile, tile total, clears the
//                          histogram for the next level (coalesced 16-byte accesses, 1 MiB for 2^18 buckets)
//   rt_sort_bases_kernel   one workgroup: exclusive prefix over the tile totals; the grand total = the number of rays that
//                          hit something = what rt_shade_kernel reads as its size
//   rt_sort_place_kernel   ray i -> sorted position tile base + bucket offset + rank: sh_idx[position] = i
//
// HBM-bound integer work: per ray 8 B written (trace), 8 B read and 4 B written (place), 4 B read by the shade kernel; per
// bucket 12 B.  (Round 3 first kept bucket and rank in the ray's 64-byte record: the place kernel then fetched 2.6 GB per
// frame of config 4 for 0.3 GB of information.)  rocPRIM's 4-pass radix sort of (key, index) pairs moved 64 B per ray and needed its size on the host.
#include <hip/hip_runtime.h>

#include "rt_internal.h"

namespace {

// exclusive prefix of the 256 per-thread sums of a workgroup; returns this thread's offset, total in *total
__device__ __forceinline__ uint32_t block_exclusive(uint32_t v, uint32_t* lds /* [4] */, uint32_t* total) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)inc, off, 64);
    if (lane >= (uint32_t)off) inc += o;
  }
  if (lane == 63u) lds[wave] = inc;
  __syncthreads();
  uint32_t before = 0, all = 0;
#pragma unroll
  for (uint32_t w = 0; w < 4u; w++) {
    const uint32_t t = lds[w];
    if (w < wave) before += t;
    all += t;
  }
  *total = all;
  return before + inc - v;
}

__global__ __launch_bounds__(256) void rt_sort_tiles_kernel(uint32_t* __restrict__ hist, uint32_t* __restrict__ offs,
                                                            uint32_t* __restrict__ tile_total) {
  __shared__ uint32_t lds[4];
  // RT_SORT_TILE = 4096 buckets per workgroup: 16 consecutive buckets per thread, 4 x uint4
  uint4* h4 = (uint4*)(hist + (size_t)blockIdx.x * RT_SORT_TILE) + threadIdx.x * 4u;
  uint4 c[4];
  uint32_t sum = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    c[k] = h4[k];
    sum += c[k].x + c[k].y + c[k].z + c[k].w;
  }
  uint32_t total;
  uint32_t run = block_exclusive(sum, lds, &total);
  uint4* o4 = (uint4*)(offs + (size_t)blockIdx.x * RT_SORT_TILE) + threadIdx.x * 4u;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    uint4 o;
    o.x = run, run += c[k].x;
    o.y = run, run += c[k].y;
    o.z = run, run += c[k].z;
    o.w = run, run += c[k].w;
    o4[k] = o;
    h4[k] = make_uint4(0u, 0u, 0u, 0u);  // the histogram is ready for the next level
  }
  if (threadIdx.x == 0) tile_total[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void rt_sort_bases_kernel(uint32_t* __restrict__ tile, uint32_t n_tiles, uint32_t* __restrict__ n_hits) {
  __shared__ uint32_t lds[4];
  // n_tiles <= 4096 (2^24 buckets): up to 16 consecutive tiles per thread
  const uint32_t per = (n_tiles + 255u) / 256u;
  uint32_t c[16], sum = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) c[k] = 0u;
  for (uint32_t k = 0; k < per && k < 16u; k++) {
    const uint32_t t = threadIdx.x * per + k;
    c[k] = t < n_tiles ? tile[t] : 0u;
    sum += c[k];
  }
  uint32_t total;
  uint32_t run = block_exclusive(sum, lds, &total);
  for (uint32_t k = 0; k < per && k < 16u; k++) {
    const uint32_t t = threadIdx.x * per + k;
    if (t < n_tiles) tile[t] = run;
    run += c[k];
  }
  if (threadIdx.x == 0) *n_hits = total;
}

__global__ __launch_bounds__(256) void rt_sort_place_kernel(RtDevParams P) {
  // the level's rays: the whole queue, or (pipelined levels) its slice [*seg_lo, *seg_hi), whose sorted positions start at seg_lo too
  uint32_t n = P.seg_hi ? *P.seg_hi : *(const uint32_t*)P.q_in_count;
  n = n < P.q_capacity ? n : P.q_capacity;
  uint32_t first = P.seg_lo ? *P.seg_lo : 0u;
  first = first < n ? first : n;
  for (uint32_t i = first + blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
    const uint2 br = P.sort_slot[i];  // {bucket (all ones: a miss, not shaded), rank inside the bucket}
    if (br.x == 0xFFFFFFFFu) continue;
    P.sh_idx[first + P.sort_tile[br.x / RT_SORT_TILE] + P.sort_offs[br.x] + br.y] = i;
  }
}

}  // namespace

int rt_launch_sort(const RtDevParams& p, uint32_t n_wgs_place, void* stream) {
  const uint32_t n_tiles = (1u << p.sort_bits) / RT_SORT_TILE;
  hipLaunchKernelGGL(rt_sort_tiles_kernel, dim3(n_tiles), dim3(256), 0, (hipStream_t)stream, p.sort_hist, p.sort_offs, p.sort_tile);
  hipLaunchKernelGGL(rt_sort_bases_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, p.sort_tile, n_tiles, p.sort_hits);
  hipLaunchKernelGGL(rt_sort_place_kernel, dim3(n_wgs_place ? n_wgs_place : 1u), dim3(256), 0, (hipStream_t)stream, p);
  return (int)hipGetLastError();
}
