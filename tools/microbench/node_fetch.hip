// Micro-benchmark: what one DEPENDENT step of a wave-cooperative BVH walk costs as a function of where the 64-byte node
// comes from.  A step = fetch node[idx] (64 B, wave-uniform index), slab-test its two boxes against the lanes' rays
// (12 fma + min/max, as csrc/rt_kernels.hip box_planes), vote, pick the next index from the node -- the chain
// s_load -> v_fma -> v_cmp -> s_cbranch of the walks that SQ_WAIT_ANY (38 % of wave-cycles, profiles/r02j) points at.
//
//   mode 0  s_load_dwordx16 through the scalar cache, table of 64 nodes (4 KiB: always a scalar-cache hit)
//   mode 1  s_load_dwordx16, table of 128 Ki nodes (8 MiB: scalar-cache miss, L2 hit), random chain
//   mode 2  LDS: the 64-node table staged in LDS once per workgroup, 4 x ds_read_b128 (same address in all lanes =
//           broadcast) + v_readfirstlane of the two child indices; box planes stay in VGPRs
//   mode 3  as 0 with a BVH4-shaped step: ONE 128-byte fetch (2 x s_load_dwordx16), 4 boxes tested, i.e. half as many
//           dependent steps for the same number of boxes
//   mode 4  vector loads with a wave-uniform address (4 x global_load_dwordx4 through the vector L1, node lands in VGPRs)
//   mode 5  BVH4-shaped step from LDS (8 x ds_read_b128)
//   mode 6  BVH4-shaped step through vector loads (8 x global_load_dwordx4, uniform address)
//   mode 7  a 32-byte node: boxes as 12 x uint16 on a global grid (lo rounded down, hi up), children in 2 dwords:
//           ONE s_load_dwordx8 per step, 12 v_cvt_f32 to unpack, the grid scale folded into the lanes' fma constants
//   mode 8  s_load_dwordx8 only (no unpacking: what the fetch alone costs at half the bytes)
//   mode 9  s_load_dwordx4 only
// Reported: nanoseconds per step of ONE wave (latency, 1 wave per SIMD) and per step per SIMD with 6 resident waves
// (what the other waves can hide).
// Build: hipcc --offload-arch=gfx950 -O3 -o node_fetch node_fetch.hip ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

struct Node {  // csrc/rt_internal.h RtNode
  float lo0[3];
  uint32_t c0;
  float hi0[3];
  uint32_t n0;
  float lo1[3];
  uint32_t c1;
  float hi1[3];
  uint32_t n1;
};

template <class T>
__device__ __forceinline__ T uload(const T* p) {
  typedef const uint32_t __attribute__((address_space(4))) * CP;
  CP q = (CP)(uintptr_t)p;
  uint32_t w[sizeof(T) / 4];
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 4; i++) w[i] = q[i];
  T v;
  __builtin_memcpy(&v, w, sizeof(T));
  return v;
}

__device__ __forceinline__ unsigned long long slab(const float* lo, const float* hi, float ix, float iy, float iz, float nx, float ny, float nz) {
  const float tn = fmaxf(fmaxf(__builtin_fmaf(lo[0], ix, nx), __builtin_fmaf(lo[1], iy, ny)), __builtin_fmaf(lo[2], iz, nz));
  const float tm = fminf(fminf(__builtin_fmaf(hi[0], ix, nx), __builtin_fmaf(hi[1], iy, ny)), __builtin_fmaf(hi[2], iz, nz));
  return __builtin_amdgcn_ballot_w64(tn <= tm);
}

template <int MODE>
__global__ __launch_bounds__(256) void walk(const Node* __restrict__ nodes, uint32_t mask, int steps, uint32_t* out) {
  __shared__ Node lds[64];
  if (MODE == 2 || MODE == 5) {
    if (threadIdx.x < 64) lds[threadIdx.x] = nodes[threadIdx.x];
    __syncthreads();
  }
  const float ix = 1.0f + threadIdx.x * 1e-3f, iy = 0.9f, iz = 1.1f, nx = -0.2f, ny = -0.1f, nz = -0.3f;
  uint32_t idx = (blockIdx.x * 4u + (threadIdx.x >> 6)) & mask;
  unsigned long long acc = 0;
  for (int s = 0; s < steps; s++) {
    // the index must be KNOWN uniform for the compiler to issue scalar loads (round 3's first table had this missing: its
    // "scalar" rows were per-lane vector loads and are withdrawn, profiles/r03_ab_experiments.md 4a)
    if (MODE == 0 || MODE == 1 || MODE == 3 || MODE >= 7) idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx);
    if (MODE == 0 || MODE == 1) {
      const Node nd = uload(&nodes[idx]);
      const unsigned long long h0 = slab(nd.lo0, nd.hi0, ix, iy, iz, nx, ny, nz), h1 = slab(nd.lo1, nd.hi1, ix, iy, iz, nx, ny, nz);
      acc += h0 ^ h1;
      idx = (h0 ? nd.c0 : nd.c1) & mask;  // (the vote decides which child index is followed: a dependent scalar chain)
    } else if (MODE == 2) {
      const float4* p = (const float4*)&lds[__builtin_amdgcn_readfirstlane(idx)];
      const float4 a = p[0], b = p[1], c = p[2], d = p[3];  // ds_read_b128 x 4, every lane the same address
      const float lo0[3] = {a.x, a.y, a.z}, hi0[3] = {b.x, b.y, b.z}, lo1[3] = {c.x, c.y, c.z}, hi1[3] = {d.x, d.y, d.z};
      const unsigned long long h0 = slab(lo0, hi0, ix, iy, iz, nx, ny, nz), h1 = slab(lo1, hi1, ix, iy, iz, nx, ny, nz);
      acc += h0 ^ h1;
      const uint32_t c0 = __builtin_amdgcn_readfirstlane(__float_as_uint(a.w)), c1 = __builtin_amdgcn_readfirstlane(__float_as_uint(c.w));
      idx = (h0 ? c0 : c1) & mask;
    } else if (MODE == 4) {
      const float4* p = (const float4*)&nodes[__builtin_amdgcn_readfirstlane(idx)];
      float4 a = p[0], b = p[1], c = p[2], d = p[3];  // global_load_dwordx4 x 4, every lane the same address
      asm volatile("" : "+v"(a.x), "+v"(b.x), "+v"(c.x), "+v"(d.x));  // (keep them vector loads)
      const float lo0[3] = {a.x, a.y, a.z}, hi0[3] = {b.x, b.y, b.z}, lo1[3] = {c.x, c.y, c.z}, hi1[3] = {d.x, d.y, d.z};
      const unsigned long long h0 = slab(lo0, hi0, ix, iy, iz, nx, ny, nz), h1 = slab(lo1, hi1, ix, iy, iz, nx, ny, nz);
      acc += h0 ^ h1;
      const uint32_t c0 = __builtin_amdgcn_readfirstlane(__float_as_uint(a.w)), c1 = __builtin_amdgcn_readfirstlane(__float_as_uint(c.w));
      idx = (h0 ? c0 : c1) & mask;
    } else if (MODE == 7) {
      struct Q { uint32_t w[6]; uint32_t c0, c1; };
      const Q q = uload((const Q*)nodes + idx);
      float lo0[3], hi0[3], lo1[3], hi1[3];
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const uint32_t w0 = q.w[a], w1 = q.w[3 + a];
        lo0[a] = (float)(w0 & 0xFFFFu), hi0[a] = (float)(w0 >> 16), lo1[a] = (float)(w1 & 0xFFFFu), hi1[a] = (float)(w1 >> 16);
      }
      const unsigned long long h0 = slab(lo0, hi0, ix, iy, iz, nx, ny, nz), h1 = slab(lo1, hi1, ix, iy, iz, nx, ny, nz);
      acc += h0 ^ h1;
      idx = (h0 ? q.c0 : q.c1) & mask;
    } else if (MODE == 8) {
      struct Q { float lo0[3]; uint32_t c0; float hi0[3]; uint32_t c1; };
      const Q q = uload((const Q*)nodes + idx);
      const unsigned long long h0 = slab(q.lo0, q.hi0, ix, iy, iz, nx, ny, nz), h1 = slab(q.hi0, q.lo0, ix, iy, iz, nx, ny, nz);
      acc += h0 ^ h1;
      idx = (h0 ? q.c0 : q.c1) & mask;
    } else if (MODE == 9) {
      struct Q { float lo0[3]; uint32_t c0; };
      const Q q = uload((const Q*)nodes + idx);
      const unsigned long long h0 = slab(q.lo0, q.lo0, ix, iy, iz, nx, ny, nz), h1 = slab(q.lo0, q.lo0, iy, ix, iz, nx, ny, nz);
      acc += h0 ^ h1;
      idx = (h0 ? q.c0 : q.c0 * 7u + 1u) & mask;
    } else if (MODE == 5 || MODE == 6) {
      const uint32_t i0 = __builtin_amdgcn_readfirstlane(idx), i1 = (i0 + 1u) & mask;
      const float4* p = MODE == 5 ? (const float4*)&lds[i0] : (const float4*)&nodes[i0];
      const float4* q = MODE == 5 ? (const float4*)&lds[i1] : (const float4*)&nodes[i1];
      float4 a = p[0], b = p[1], c = p[2], d = p[3], e = q[0], f = q[1], g = q[2], h = q[3];
      if (MODE == 6) asm volatile("" : "+v"(a.x), "+v"(b.x), "+v"(c.x), "+v"(d.x), "+v"(e.x), "+v"(f.x), "+v"(g.x), "+v"(h.x));
      const float l0[3] = {a.x, a.y, a.z}, u0[3] = {b.x, b.y, b.z}, l1[3] = {c.x, c.y, c.z}, u1[3] = {d.x, d.y, d.z};
      const float l2[3] = {e.x, e.y, e.z}, u2[3] = {f.x, f.y, f.z}, l3[3] = {g.x, g.y, g.z}, u3[3] = {h.x, h.y, h.z};
      const unsigned long long h0 = slab(l0, u0, ix, iy, iz, nx, ny, nz), h1 = slab(l1, u1, ix, iy, iz, nx, ny, nz);
      const unsigned long long h2 = slab(l2, u2, ix, iy, iz, nx, ny, nz), h3 = slab(l3, u3, ix, iy, iz, nx, ny, nz);
      acc += h0 ^ h1 ^ h2 ^ h3;
      const uint32_t c0 = __builtin_amdgcn_readfirstlane(__float_as_uint(a.w)), c1 = __builtin_amdgcn_readfirstlane(__float_as_uint(c.w));
      const uint32_t c2 = __builtin_amdgcn_readfirstlane(__float_as_uint(e.w)), c3 = __builtin_amdgcn_readfirstlane(__float_as_uint(g.w));
      idx = (h0 ? c0 : (h1 ? c1 : (h2 ? c2 : c3))) & mask;
    } else {
      const Node na = uload(&nodes[idx]), nb = uload(&nodes[(idx + 1u) & mask]);  // one 128-byte BVH4 node
      const unsigned long long h0 = slab(na.lo0, na.hi0, ix, iy, iz, nx, ny, nz), h1 = slab(na.lo1, na.hi1, ix, iy, iz, nx, ny, nz);
      const unsigned long long h2 = slab(nb.lo0, nb.hi0, ix, iy, iz, nx, ny, nz), h3 = slab(nb.lo1, nb.hi1, ix, iy, iz, nx, ny, nz);
      acc += h0 ^ h1 ^ h2 ^ h3;
      idx = (h0 ? na.c0 : (h1 ? na.c1 : (h2 ? nb.c0 : nb.c1))) & mask;
    }
  }
  if ((threadIdx.x & 63u) == 0) out[blockIdx.x * 4u + (threadIdx.x >> 6)] = (uint32_t)acc + idx;
}

template <int MODE>
void run(const char* name, const Node* nodes, uint32_t mask, int waves_per_simd) {
  uint32_t* out;
  const int blocks = 256 * waves_per_simd, steps = 20000;
  hipMalloc(&out, (size_t)blocks * 4 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  walk<MODE><<<blocks, 256>>>(nodes, mask, 100, out);
  hipEventRecord(e0);
  walk<MODE><<<blocks, 256>>>(nodes, mask, steps, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s waves/SIMD %d: %8.1f ns per step of a wave, %7.1f ns per step per SIMD\n", name, waves_per_simd, ms * 1e6 / steps,
         ms * 1e6 / steps / waves_per_simd);
  hipFree(out);
}

int main() {
  const uint32_t n_small = 64, n_big = 128 * 1024;
  std::vector<Node> h(n_big);
  srand(7);
  for (uint32_t i = 0; i < n_big; i++) {
    Node& n = h[i];
    for (int a = 0; a < 3; a++) n.lo0[a] = 0.1f * a, n.hi0[a] = 1.0f + a, n.lo1[a] = 0.2f * a, n.hi1[a] = 2.0f + a;
    n.c0 = (uint32_t)rand() * 2654435761u, n.c1 = (uint32_t)rand() * 40503u + 1u;  // random chain (masked by the kernel)
    n.n0 = n.n1 = 0;
  }
  Node* d;
  hipMalloc(&d, sizeof(Node) * n_big);
  hipMemcpy(d, h.data(), sizeof(Node) * n_big, hipMemcpyHostToDevice);
  for (int w : {1, 3, 6}) {
    run<0>("s_load x16, 4 KiB table (scalar-cache hit)", d, n_small - 1, w);
    run<1>("s_load x16, 8 MiB table (scalar miss, L2 hit)", d, n_big - 1, w);
    run<2>("LDS ds_read_b128 x4 + readfirstlane", d, n_small - 1, w);
    run<3>("BVH4 step: 2 x s_load x16, 4 boxes (hit)", d, n_small - 1, w);
    run<4>("vector loads, uniform address, 4 KiB table", d, n_small - 1, w);
    run<4>("vector loads, uniform address, 8 MiB table", d, n_big - 1, w);
    run<5>("BVH4 step from LDS (8 x ds_read_b128)", d, n_small - 1, w);
    run<6>("BVH4 step, vector loads, 4 KiB table", d, n_small - 1, w);
    run<7>("32-byte node: s_load x8 + 12 cvt (hit)", d, n_small - 1, w);
    run<7>("32-byte node: s_load x8 + 12 cvt, 4 MiB table", d, n_big - 1, w);
    run<8>("s_load x8 only (hit)", d, n_small - 1, w);
    run<9>("s_load x4 only (hit)", d, n_small - 1, w);
  }
  hipFree(d);
  return 0;
}
