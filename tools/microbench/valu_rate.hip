// Micro-benchmark: VALU issue rate of scalar vs packed fp32 ops on gfx950, N waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP 64
template <int MODE>
__global__ void k(float* out, int iters, const float* __restrict__ uni) {
  const float su0 = uni[0], su1 = uni[1];
  unsigned long long acc = 0;
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  const float m = 1.0000001f, c = 1e-7f;
  const f2 pm = {m, m}, pc = {c, c};
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < REP; r++) {
      if (MODE == 0) {  // 8 independent scalar fma
        a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
        a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
      } else if (MODE == 1) {  // 4 independent packed fma (same flops as mode 0)
        p0 = __builtin_elementwise_fma(p0, pm, pc); p1 = __builtin_elementwise_fma(p1, pm, pc);
        p2 = __builtin_elementwise_fma(p2, pm, pc); p3 = __builtin_elementwise_fma(p3, pm, pc);
      } else if (MODE == 2) {  // 8 scalar mul
        a0 *= m; a1 *= m; a2 *= m; a3 *= m; a4 *= m; a5 *= m; a6 *= m; a7 *= m;
      } else if (MODE == 3) {  // 4 packed mul
        p0 *= pm; p1 *= pm; p2 *= pm; p3 *= pm;
      } else if (MODE == 5) {  // 8 compares + selects (v_cmp -> vcc/sgpr, v_cndmask)
        a0 = a0 > a1 ? a0 * m : a1; a2 = a2 > a3 ? a2 * m : a3; a4 = a4 > a5 ? a4 * m : a5; a6 = a6 > a7 ? a6 * m : a7;
        a1 = a1 > a2 ? a1 * m : a2; a3 = a3 > a4 ? a3 * m : a4; a5 = a5 > a6 ? a5 * m : a6; a7 = a7 > a0 ? a7 * m : a0;
      } else if (MODE == 6) {  // 4 IEEE divisions
        a0 = m / a0; a1 = m / a1; a2 = m / a2; a3 = m / a3;
      } else if (MODE == 7) {  // 4 IEEE sqrt
        a0 = __builtin_sqrtf(a0 + 2.0f); a1 = __builtin_sqrtf(a1 + 2.0f); a2 = __builtin_sqrtf(a2 + 2.0f); a3 = __builtin_sqrtf(a3 + 2.0f);
      } else if (MODE == 8) {  // 8 min3/max3
        a0 = fmaxf(fmaxf(a0, a1), a2); a1 = fminf(fminf(a1, a2), a3); a2 = fmaxf(fmaxf(a2, a3), a4); a3 = fminf(fminf(a3, a4), a5);
        a4 = fmaxf(fmaxf(a4, a5), a6); a5 = fminf(fminf(a5, a6), a7); a6 = fmaxf(fmaxf(a6, a7), a0); a7 = fminf(fminf(a7, a0), a1);
        a0 += c; a1 += c; a2 += c; a3 += c; a4 += c; a5 += c; a6 += c; a7 += c;
      } else if (MODE == 9) {  // 4 ballots feeding scalar state
        unsigned long long b = __ballot(a0 > a1); a0 += (float)__popcll(b) * c; a1 *= m;
        b = __ballot(a2 > a3); a2 += (float)__popcll(b) * c; a3 *= m;
        b = __ballot(a4 > a5); a4 += (float)__popcll(b) * c; a5 *= m;
        b = __ballot(a6 > a7); a6 += (float)__popcll(b) * c; a7 *= m;
      } else if (MODE == 10) {  // 4 fast rcp (v_rcp_f32 only)
        a0 = __builtin_amdgcn_rcpf(a0) + m; a1 = __builtin_amdgcn_rcpf(a1) + m; a2 = __builtin_amdgcn_rcpf(a2) + m; a3 = __builtin_amdgcn_rcpf(a3) + m;
      } else if (MODE == 11) {  // 8 fma with an SGPR operand (uniform value from a scalar load)
        a0 = __builtin_fmaf(a0, su0, c); a1 = __builtin_fmaf(a1, su1, c); a2 = __builtin_fmaf(a2, su0, c); a3 = __builtin_fmaf(a3, su1, c);
        a4 = __builtin_fmaf(a4, su0, c); a5 = __builtin_fmaf(a5, su1, c); a6 = __builtin_fmaf(a6, su0, c); a7 = __builtin_fmaf(a7, su1, c);
      } else if (MODE == 12) {  // 8 mul+add pairs not contracted, SGPR operand
        a0 = a0 * su0 + c; a1 = a1 * su1 + c; a2 = a2 * su0 + c; a3 = a3 * su1 + c;
        a4 = a4 * su0 + c; a5 = a5 * su1 + c; a6 = a6 * su0 + c; a7 = a7 * su1 + c;
      } else if (MODE == 13) {  // 8 compares into masks + ballots (v_cmp only)
        unsigned long long m0 = __ballot(a0 > su0), m1 = __ballot(a1 > su1), m2 = __ballot(a2 < su0), m3 = __ballot(a3 < su1);
        unsigned long long m4 = __ballot(a4 > su0), m5 = __ballot(a5 > su1), m6 = __ballot(a6 < su0), m7 = __ballot(a7 < su1);
        acc += m0 ^ m1 ^ m2 ^ m3 ^ m4 ^ m5 ^ m6 ^ m7;
        a0 += c;
      } else if (MODE == 4) {  // 8 scalar max
        a0 = fmaxf(a0, c); a1 = fmaxf(a1, m); a2 = fmaxf(a2, c); a3 = fmaxf(a3, m);
        a4 = fmaxf(a4, c); a5 = fmaxf(a5, m); a6 = fmaxf(a6, c); a7 = fmaxf(a7, m);
        a0 += c; a1 += c; a2 += c; a3 += c; a4 += c; a5 += c; a6 += c; a7 += c;
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + (float)(acc & 1);
}

template <int MODE>
void run(const char* name, int waves_per_simd, int n_instr_per_rep) {
  float* out;
  int threads = 256;  // 4 waves per block -> 1 per SIMD
  int blocks = 256 * waves_per_simd;
  hipMalloc(&out, (size_t)blocks * threads * 4);
  int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float* uni; hipMalloc(&uni, 64); float hu[2] = {1.0000001f, 0.9999999f}; hipMemcpy(uni, hu, 8, hipMemcpyHostToDevice);
  k<MODE><<<blocks, threads>>>(out, 10, uni);
  hipEventRecord(e0);
  k<MODE><<<blocks, threads>>>(out, iters, uni);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double instr = (double)iters * REP * n_instr_per_rep;  // per wave
  double cyc = ms * 1e-3 * 2.4e9;                        // nominal cycles
  printf("%-14s waves/SIMD %d: %.3f ms  -> %.2f cycles per wave-instruction per SIMD (nominal 2.4 GHz)\n", name, waves_per_simd, ms,
         cyc / (instr * waves_per_simd));
  hipFree(out);
}

int main() {
  for (int w : {4}) {
    run<0>("fma scalar x8", w, 8);
    run<1>("pk_fma x4", w, 4);
    run<2>("mul scalar x8", w, 8);
    run<3>("pk_mul x4", w, 4);
    run<4>("max+add x16", w, 16);
    run<5>("cmp+sel+mul x8", w, 8);
    run<6>("ieee div x4", w, 4);
    run<7>("ieee sqrt x4", w, 4);
    run<8>("minmax3+add x8", w, 8);
    run<9>("ballot grp x4", w, 4);
    run<10>("rcp+add x4", w, 4);
    run<11>("fma sgpr x8", w, 8);
    run<12>("mul,add sgpr x16", w, 16);
    run<13>("cmp->mask x8(+1)", w, 9);
  }
  return 0;
}
