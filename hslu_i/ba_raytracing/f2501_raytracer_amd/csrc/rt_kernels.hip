// rt_kernels.hip -- the per-pixel render loop as hand-written gfx950 (CDNA4) kernels.
//
// Path implemented (reference file:line, all fp32):
//   primary rays            src/renderer/raytracer_renderer.rs:1190-1357, src/renderer/mod.rs:146-209
//   AA sample accumulation  raytracer_renderer.rs:918-1016 (sample table comes in through rt_params)
//   nearest hit             src/raytracing/raytracer.rs:162-220
//   shadow / transmittance  raytracer.rs:24-106
//   sphere / triangle test  src/geometry/basic/sphere.rs:78-162, triangle.rs:149-212
//   Whitted shading         raytracer_renderer.rs:147-264 (node), :731-874 (lighting),
//                           :526-729 (reflection), :279-524 (refraction), :266-277 (attenuation)
//   Fresnel / absorption    src/raytracing/material.rs:468-525, :213-231
//   point light             src/scene/lighting/light.rs:261-299, cloud :183-225
//   pixel pack              src/output/window.rs:105-109
//
// Execution model (MI355X-first, nothing like the reference's rayon + 8-lane packets):
//   * every ray is an independent work item.  rt_primary[_stream]_kernel: one thread per (pixel, DISTINCT AA sample) --
//     repeats of the reference's sample table are traced once and weighted by their multiplicity -- a wavefront = the
//     samples of a few adjacent pixels; rt_trace_kernel / rt_shade_kernel: one thread per queued reflection /
//     refraction ray (nearest hit, radix sort by hit point, shading in hit-point order).  Children are appended to
//     per-level ray queues in HBM (SoA float4 planes, one atomic per wavefront) that the host drains deepest level
//     first in frame-sized chunks; pixel sums of secondary rays use 64-bit fixed-point atomics (order independent,
//     bit-reproducible);
//   * soft shadows walk the BVH once per (wavefront, light): the N jittered shadow rays of a hit point share a
//     candidate triangle list (beam-level conservative culling), kept in the lanes of a VGPR.  Wavefronts whose hit
//     points are unrelated (the list overflows) defer their (hit point, light) pairs to rt_hard_kernel: N samples on
//     N lanes, every lane walking a threaded (stackless) copy of the tree for its own ray;
//   * BVH traversal is WAVE-COOPERATIVE: the 64 rays of a wavefront walk the tree together.  The
//     current node index is wave-uniform, node / triangle / sphere / light records are fetched with
//     scalar loads (constant address space -> s_load_dwordx4/x16), and the traversal stack is ONE
//     stack per wavefront kept in the 64 lanes of a VGPR, driven by __ballot votes; lanes whose ray
//     misses a box are masked for that subtree;
//   * a conservative, staged triangle pre-filter keeps the IEEE division of the literal test for
//     the few triangles some lane can actually hit;
//   * latency bound (dependent scalar-load -> vote -> branch chains): compiled for 6 waves per SIMD;
//   * no MFMA: this is branchy fp32 intersection math, not a contraction.
//
// Numerics: compiled with -ffp-contract=off; fused multiply-adds appear exactly where the
// reference calls mul_add (written __builtin_fmaf).  Division and sqrt are the correctly rounded
// forms (hipcc default) wherever a hit / occlusion / spawn decision depends on them, so every such
// decision and every t is bit-identical to the CPU restatement in oracle/; colour-only factors use
// v_rcp/v_rsq (1 ulp) and exp2/log2-based tanh/pow.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "rt_internal.h"

#define RT_EPS 1.1920929e-7f

// keeps a value opaque to the optimiser (no forwarding / CSE across this point)
#define RT_OPAQUE(v) asm volatile("" : "+v"(v))
// same for a wave-uniform value: gives it an SGPR of its own (kernel arguments arrive in 8/16-dword tuples, and
// hipcc spills and reloads a tuple as a whole -- 8 v_readlane to get at one of its fields inside a loop)
#define RT_OPAQUE_S(v) asm volatile("" : "+s"(v))
// Lane mask of a predicate.  HIP's __ballot(int) first widens the bool to an int in a VGPR and compares it back
// (v_cndmask + v_cmp per vote); the builtin takes the i1 as it is (the compare result already is the mask).
#define wave_ballot(pred) __builtin_amdgcn_ballot_w64((bool)(pred))
// hipcc only folds a vote into its compare when the predicate IS one compare; a combined predicate costs a
// v_cndmask + v_cmp round trip.  So lane sets are kept as 64-bit masks (SGPR pairs) and combined with scalar
// and/or/andn2; lane_of() turns a mask back into a per-lane predicate where a select needs one (free).
typedef unsigned long long lanemask;
#define lane_of(mask) __builtin_amdgcn_inverse_ballot_w64(mask)

// Streaming accesses of the ray queues (each record is written once and read twice, gigabytes per frame): non-temporal, so
// that they do not push the wavefronts' scratch lines out of L2 (make NT=0 for the A/B)
#ifndef RT_NT
#define RT_NT 1
#endif
typedef float rt_f4v __attribute__((ext_vector_type(4)));
typedef uint32_t rt_u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 stream_load4(const float4* p) {
  if (!RT_NT) return *p;
  const rt_f4v v = __builtin_nontemporal_load((const rt_f4v*)p);
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void stream_store4(float4* p, float4 v) {
  if (!RT_NT) {
    *p = v;
    return;
  }
  rt_f4v w;
  w.x = v.x, w.y = v.y, w.z = v.z, w.w = v.w;
  __builtin_nontemporal_store(w, (rt_f4v*)p);
}

// the workgroup of the frame's list that launched workgroup b of a batch stands for (RtDevParams::batch_stride)
__device__ __forceinline__ uint32_t rt_batch_wg(const RtDevParams& P, uint32_t b) {
  const uint32_t gl = P.batch_group_log2;
  return P.batch_first_wg + (((b >> gl) * P.batch_stride) << gl) + (b & ((1u << gl) - 1u));
}

// Wave-uniform values kept in the LANES of one VGPR (walk stacks, candidate lists): entry `lane` := value.  v_writelane_b32
// ignores EXEC, as v_readlane_b32 does: the entry is stored whether or not lane `lane` is enabled where the compiler placed
// the code.  (A select on the lane id is not: rt_flags_kernel's last wavefront ran its walks with EXEC = the lanes that
// own a cell -- one lane for a scene of 64 k + 1 cells -- and lost every push to a higher lane: a walk that popped stale
// entries for ever, found by the fuzz sweep, seed 61.  The cause was a per-lane pointer in the walk's early-out condition,
// which made the whole walk a divergent loop; that condition is uniform again, and this store no longer depends on it.)
#ifndef RT_LANE_PUT_SELECT
#define RT_LANE_PUT_SELECT 0
#endif
// (clang has no __builtin_amdgcn_writelane; the intrinsic is declared the way the HIP device headers declare theirs)
extern "C" __device__ int rt_llvm_writelane(int value, int lane, int old) __asm("llvm.amdgcn.writelane");
__device__ __forceinline__ void lane_put(uint32_t& reg, uint32_t lane, uint32_t value, uint32_t lane_id) {
  if (RT_LANE_PUT_SELECT)
    reg = (lane_id == lane) ? value : reg;
  else
    reg = (uint32_t)rt_llvm_writelane((int)value, (int)lane, (int)reg);
}
// Diagnostic build (make PROFILE=1 -> librt_hip_prof.so): wave-level shader-clock timers (s_memtime) around
// the regions of the light loop; their sums replace the work counters in rt_stats (tools/perf_ab.py --prof).
#ifndef RT_PROFILE
#define RT_PROFILE 0
#endif
#ifndef RT_SKIP
#define RT_SKIP 0
#endif
// Work statistics (rt_stats.wave_*: node visits, triangle tests, wave passes) are compiled in only by
// `make STATS=1` (librt_hip_stats.so, tools/perf_ab.py): 9 live SGPRs and an s_add per node / triangle
// that the shipped kernels do not pay for.  The ray counters (rays_*, pixels_written) are always on.
#ifndef RT_WORK_STATS
#define RT_WORK_STATS 0
#endif
#if RT_WORK_STATS || RT_PROFILE
#define WSTAT(stmt) stmt
#else
#define WSTAT(stmt) ((void)0)
#endif
#if RT_PROFILE == 1
#define PROF_T() __builtin_readcyclecounter()
#define PROF_ADD(W, i, t0) ((W).prof[i] += __builtin_readcyclecounter() - (t0))
#else
#define PROF_T() 0ull
#define PROF_ADD(W, i, t0) ((void)(t0))
#endif

namespace {

struct V3 {
  float x, y, z;
};
__device__ __forceinline__ V3 mk(float x, float y, float z) {
  V3 r;
  r.x = x;
  r.y = y;
  r.z = z;
  return r;
}
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator/(V3 a, V3 b) { return mk(a.x / b.x, a.y / b.y, a.z / b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
// ultraviolet Vec3::dot = x.mul_add(ox, y.mul_add(oy, z*oz))
__device__ __forceinline__ float dot(V3 a, V3 b) {
  return __builtin_fmaf(a.x, b.x, __builtin_fmaf(a.y, b.y, a.z * b.z));
}
// Correctly rounded sqrt and reciprocal for operands in the NORMAL range: hipcc's own correction sequences (the AMDGPU
// lowering of fsqrt / fdiv: a 1-ulp v_sqrt / v_rcp and fused residual steps) without the scaling, class tests and
// v_div_scale pair that only serve denormal or huge operands -- 9 instead of 16 and 8 instead of 11 vector instructions,
// bit-identical to `sqrtf(x)` / `1.0f / d` wherever a ray can be (lengths, determinants, discriminants; 0, inf and NaN
// behave as IEEE does).  Every hit / occlusion decision keeps going through these.
__device__ __forceinline__ float exact_sqrt(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
  const float rd = __builtin_fmaf(-sd, s, x), ru = __builtin_fmaf(-su, s, x);
  float r = rd <= 0.0f ? sd : s;  // s one ulp too large
  r = ru > 0.0f ? su : r;         // s one ulp too small
  return r;
}
__device__ __forceinline__ float exact_rcp(float d) {
  float r = __builtin_amdgcn_rcpf(d);
  r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
  float q = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
  q = __builtin_fmaf(__builtin_fmaf(-d, q, 1.0f), r, q);
  return __builtin_amdgcn_div_fixupf(q, d, 1.0f);  // d = 0, inf, NaN
}
__device__ __forceinline__ float mag(V3 a) { return exact_sqrt(dot(a, a)); }
__device__ __forceinline__ V3 normalize(V3 a) {
  float r = exact_rcp(mag(a));
  return a * r;
}
// normalize() of a vector that is ALREADY unit up to rounding: a shadow ray's direction ld = ltp * (1 / |ltp|) is
// normalised once more by Ray::new_with_mask (ray.rs:52-57).  s = |a|^2 then lies within a few ulp of 1, and there the
// correctly rounded 1 / sqrt(s) -- an IEEE sqrt and an IEEE division, 57 + 44 SIMD cycles -- is a function of the
// integer distance k of s from 1.0f:
//   s = 1 + k ulp (k >= 0):   sqrt = 1 + floor(k/2) ulp  (kx/2 - k^2 x^2/8 lies just below the tie)  ->  r = 1 - floor(k/2) ulp
//   s = 1 - j ulp/2 (j > 0):  sqrt = 1 - ceil(j/2) ulp/2,  n = ceil(j/2)                              ->  r = 1 + ceil(n/2) ulp
// (exact for |k| < 2898; checked exhaustively on the host, tests/test_host_logic.py).  Lanes outside |k| <= 1024 (never a
// unit vector; NaN / zero input) send the wavefront through the generic sequence.
__device__ __forceinline__ V3 normalize_unit(V3 a, lanemask lanes) {
  const float s = dot(a, a);
  const int k = (int)__float_as_uint(s) - 0x3f800000;
  if (wave_ballot((uint32_t)(k + 1024) > 2048u) & lanes) return normalize(a);
  const int q = ((((-k) + 1) >> 1) + 1) >> 1;
  const uint32_t rb = k >= 0 ? 0x3f800000u - (uint32_t)((k >> 1) * 2) : 0x3f800000u + (uint32_t)q;
  return a * __uint_as_float(rb);
}
__device__ __forceinline__ V3 fma_s(V3 d, float t, V3 o) {
  return mk(__builtin_fmaf(d.x, t, o.x), __builtin_fmaf(d.y, t, o.y), __builtin_fmaf(d.z, t, o.z));
}
__device__ __forceinline__ V3 reflected(V3 v, V3 n) {
  float k = 2.0f * dot(v, n);
  return v - n * k;
}
__device__ __forceinline__ float clampf(float x, float lo, float hi) {
  return fminf(fmaxf(x, lo), hi);
}
__device__ __forceinline__ bool has_nan(V3 a) { return (a.x != a.x) || (a.y != a.y) || (a.z != a.z); }

// Wave-uniform load: the address is the same in every lane, so read it through the constant address
// space -> s_load_dwordx4/x8/x16 into SGPRs (scalar cache -> L2) instead of 64 identical vector loads.
// All scene arrays are read-only for the whole launch, which the scalar cache requires.
template <class T>
__device__ __forceinline__ T uload(const T* p) {
  static_assert(sizeof(T) % 4 == 0, "dword records only");
  typedef const uint32_t __attribute__((address_space(4))) * CP;
  CP q = (CP)(uintptr_t)p;
  uint32_t w[sizeof(T) / 4];
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 4; i++) w[i] = q[i];
  T v;
  __builtin_memcpy(&v, w, sizeof(T));
  return v;
}

// Scene arrays: one allocation, base + 32-bit byte offset (rt_internal.h).  sload: wave-uniform offset ->
// `s_load_dwordx* sdst, s[base], soffset` (the offset is an SGPR operand of the load: no 64-bit address
// arithmetic); vload: per-lane offset.
template <class T>
__device__ __forceinline__ T sload(const RtDevScene& sc, uint32_t byte_off) {
  return uload((const T*)(sc.base + byte_off));
}
template <class T>
__device__ __forceinline__ T vload(const RtDevScene& sc, uint32_t byte_off) {
  return *(const T*)(sc.base + byte_off);
}

// ---- colour-only arithmetic ----------------------------------------------------------------------
// Quantities that feed ONLY the RGB value (never a hit / occlusion / spawn decision) use the
// hardware's 1-ulp reciprocal / rsqrt instead of the IEEE division / sqrt sequences (44 and 57
// SIMD cycles each on gfx950, measured): relative error ~1e-7, four orders below the 1e-4 RGB bar.
__device__ __forceinline__ float fast_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ V3 fast_normalize(V3 a) {
  float r = __builtin_amdgcn_rsqf(dot(a, a));
  return a * r;
}

// The light sigmoid (tanh(x) + 1) / 2 = 1 / (1 + exp(-2x)) and pow(b, e) for b in [0, 1], e >= 1 use the hardware
// exp2 / log2 / rcp instead of ocml's tanhf / powf (colour-only factors).  Measured: -7.5 % kernel time on
// config 3 when introduced; max |dRGB| vs the oracle over the whole parity suite < 4e-6 (7e-7 with ocml; bar
// 1e-4).  The reference itself evaluates both with `wide`'s polynomial approximations.
// -DRT_FAST_TRANS=0 selects ocml.
__device__ __forceinline__ float fast_pow01(float b, float e) {
  return b > 0.0f ? __builtin_amdgcn_exp2f(e * __builtin_amdgcn_logf(b)) : 0.0f;
}
#ifndef RT_FAST_TRANS
#define RT_FAST_TRANS 1
#endif

struct Mat {
  V3 color;
  float metallic, shininess, ior, opacity, boost;
  float inv_ior, f0_air;  // host: 1 / ior and ((1 - ior) / (1 + ior))^2, compute_fresnel's constants against other_ior = 1
  bool transmissive;  // TransmissionProperties::mask, material.rs:44-50
};

__device__ __forceinline__ Mat load_mat(const RtDevScene& sc, uint32_t idx) {
  const float4 a = vload<float4>(sc, sc.off_materials + idx * 48u);
  const float4 b = vload<float4>(sc, sc.off_materials + idx * 48u + 16u);
  const float4 c = vload<float4>(sc, sc.off_materials + idx * 48u + 32u);
  Mat m;
  m.color = mk(a.x, a.y, a.z);
  m.metallic = a.w;
  m.shininess = b.x;
  m.ior = b.y;
  m.opacity = b.z;
  m.boost = b.w;
  m.inv_ior = c.y;
  m.f0_air = c.z;
  m.transmissive = (c.x != 0.0f) && !(fabsf(m.opacity - 0.0f) <= RT_EPS);
  return m;
}

// same, material index wave-uniform
__device__ __forceinline__ Mat load_mat_u(const RtDevScene& sc, uint32_t idx) {
  const float4 a = sload<float4>(sc, sc.off_materials + idx * 48u);
  const float4 b = sload<float4>(sc, sc.off_materials + idx * 48u + 16u);
  const float4 c = sload<float4>(sc, sc.off_materials + idx * 48u + 32u);
  Mat m;
  m.color = mk(a.x, a.y, a.z);
  m.metallic = a.w;
  m.shininess = b.x;
  m.ior = b.y;
  m.opacity = b.z;
  m.boost = b.w;
  m.inv_ior = c.y;
  m.f0_air = c.z;
  m.transmissive = (c.x != 0.0f) && !(fabsf(m.opacity - 0.0f) <= RT_EPS);
  return m;
}

// Material::compute_fresnel (reflectance), material.rs:468-525
__device__ __forceinline__ V3 fresnel_reflectance(const Mat& m, V3 normal, V3 view, float other_ior) {
  if (!m.transmissive) return mk(m.metallic, m.metallic, m.metallic);
  float ior = m.ior;
  float n_dot_v = dot(normal, view);
  float cos_theta = fabsf(n_dot_v);
  bool inside = n_dot_v < 0.0f;
  float eta_t = inside ? (ior / other_ior) : (other_ior / ior);
  float sin2_t = eta_t * eta_t * (1.0f - cos_theta * cos_theta);
  bool reflective = m.metallic > 0.0f;
  bool tir = (inside && sin2_t > 1.0f) || reflective;
  float q = (other_ior - ior) / (other_ior + ior);
  float f0 = q * q;
  float omt = 1.0f - m.metallic;
  V3 f0v = mk(f0 * omt + m.color.x * m.metallic, f0 * omt + m.color.y * m.metallic,
              f0 * omt + m.color.z * m.metallic);
  float c1 = 1.0f - cos_theta;
  float c2 = c1 * c1;
  float c5 = c1 * (c2 * c2);
  V3 fres = mk(f0v.x + (1.0f - f0v.x) * c5, f0v.y + (1.0f - f0v.y) * c5, f0v.z + (1.0f - f0v.z) * c5);
  float ra = reflective ? m.metallic : 1.0f;
  return tir ? mk(ra, ra, ra) : fres;
}

// The same against other_ior = 1.0 -- what every shadow ray asks (raytracer.rs:64-66) -- with the two divisions of
// the material's constants taken from the host (bit-identical: ior / 1.0f == ior, 1.0f / ior and q * q are the same
// single-precision operations).  Red channel only: the shadow opacity uses transmittance.red.
__device__ __forceinline__ float fresnel_reflectance_air_red(const Mat& m, V3 normal, V3 view) {
  if (!m.transmissive) return m.metallic;
  float n_dot_v = dot(normal, view);
  float cos_theta = fabsf(n_dot_v);
  bool inside = n_dot_v < 0.0f;
  float eta_t = inside ? m.ior : m.inv_ior;
  float sin2_t = eta_t * eta_t * (1.0f - cos_theta * cos_theta);
  bool reflective = m.metallic > 0.0f;
  bool tir = (inside && sin2_t > 1.0f) || reflective;
  float omt = 1.0f - m.metallic;
  float f0x = m.f0_air * omt + m.color.x * m.metallic;
  float c1 = 1.0f - cos_theta;
  float c2 = c1 * c1;
  float c5 = c1 * (c2 * c2);
  float fres = f0x + (1.0f - f0x) * c5;
  float ra = reflective ? m.metallic : 1.0f;
  return tir ? ra : fres;
}

// Material::absorption, material.rs:213-231
__device__ __forceinline__ V3 absorption(const Mat& m) {
  float op = m.transmissive ? m.opacity : 1.0f;
  op = clampf(op, 0.0f, 1.0f - RT_EPS);
  return m.color * (1.0f - op);
}

// attenuation_factor_based_on_distance, raytracer_renderer.rs:266-277
__device__ __forceinline__ float atten(float t) {
  float d = fabsf(t);
  float a = 1.0f / (1.0f + d + 0.1f * d * d);
  return clampf(a, 0.0f, 1.0f);
}

// SphereData::intersect, sphere.rs:78-162.  `s` = {cx, cy, cz, r_sq} (wave-uniform).
__device__ __forceinline__ bool sphere_hit(float4 s, V3 o, V3 d, float& t_out) {
  V3 v = o - mk(s.x, s.y, s.z);
  float b = 2.0f * dot(d, v);
  float cc = dot(v, v) - s.w;
  float disc = __builtin_fmaf(b, b, (2.0f * -2.0f) * cc);
  if (!(disc >= 0.0f)) return false;
  float sq = exact_sqrt(disc);
  float mba = (-b) * 0.5f;
  float sa = sq * 0.5f;
  float t0 = mba - sa;
  float t1 = mba + sa;
  bool t0v = t0 >= 0.0f, t1v = t1 >= 0.0f;
  bool use0 = t0v && (!t1v || t0 < t1);
  bool use1 = t1v && !use0;
  t_out = use0 ? t0 : t1;
  return use0 || use1;
}

// TriangleData::intersect, triangle.rs:149-212 with ultraviolet Mat3::inversed/determinant.
// q0 = {v1.xyz, e1.x}, q1 = {e1.yz, e2.xy}, q2 = {e2.z, X.xyz} with X = e1 x e2 (host, bit-equal
// to cross(-e1, -e2)).
//
// Two phases, both wave-level:
//   phase 1  a CONSERVATIVE barycentric pre-filter on the un-divided numerators
//            u ~ (y.b)/det, v ~ (z.b)/det.  It may only say "certainly invalid": a lane is dropped
//            when u < 0, v < 0 or u + v >= 1 holds with a margin G*S that dominates every rounding
//            error of the literal sequence (S = sum |y_i b_i|; the literal u differs from
//            (y.b)/det by < 8 ulp-units of S/|det|, G = 2e-6 is > 30 of them).  Most leaf
//            triangles are missed by all 64 coherent rays of a wavefront, and for those the IEEE
//            division and everything after it is skipped.
//   phase 2  the literal sequence (same op order as the oracle), only if some lane survived.
// y, z, b and det_i are shared by both phases and computed exactly as the literal code does.
#define RT_TRI_G 2e-6f
__device__ __forceinline__ lanemask tri_hit(float4 q0, float4 q1, float4 q2, V3 o, V3 d, lanemask on, float tlimit,
                                            float& t_out, uint32_t& n_exact) {
  V3 v1 = mk(q0.x, q0.y, q0.z);
  V3 c1 = mk(-q0.w, -q1.x, -q1.y);  // -e1
  V3 c2 = mk(-q1.z, -q1.w, -q2.x);  // -e2
  V3 x = mk(q2.y, q2.z, q2.w);
  V3 b = v1 - o;
  // y = c2 x c0, z = c0 x c1 (ultraviolet cross: (a.y*b.z) + (-a.z*b.y), ...)
  V3 y = mk((c2.y * d.z) + (-c2.z * d.y), (c2.z * d.x) + (-c2.x * d.z), (c2.x * d.y) + (-c2.y * d.x));
  float det_i = dot(d, x);
  V3 z;
  {
    // ---- phase 1 (staged: u, then v and u+v, then t; each stage can end the test for the wave) ----
    const uint32_t sgn = __float_as_uint(det_i) & 0x80000000u;
    V3 ab = mk(fabsf(b.x), fabsf(b.y), fabsf(b.z));
    float yb = __builtin_fmaf(y.x, b.x, __builtin_fmaf(y.y, b.y, y.z * b.z));
    float su = __builtin_fmaf(fabsf(y.x), ab.x, __builtin_fmaf(fabsf(y.y), ab.y, fabsf(y.z) * ab.z));
    float ybs = __uint_as_float(__float_as_uint(yb) ^ sgn);  // (y.b) * sign(det)
    on &= ~wave_ballot(ybs < -RT_TRI_G * su);  // NaN anywhere -> comparison false -> not rejected
    if (!on) return 0ull;
    z = mk((d.y * c1.z) + (-d.z * c1.y), (d.z * c1.x) + (-d.x * c1.z), (d.x * c1.y) + (-d.y * c1.x));
    float zb = __builtin_fmaf(z.x, b.x, __builtin_fmaf(z.y, b.y, z.z * b.z));
    float sv = __builtin_fmaf(fabsf(z.x), ab.x, __builtin_fmaf(fabsf(z.y), ab.y, fabsf(z.z) * ab.z));
    float zbs = __uint_as_float(__float_as_uint(zb) ^ sgn);
    float ad = fabsf(det_i);
    on &= ~(wave_ballot(zbs < -RT_TRI_G * sv) | wave_ballot((ybs + zbs) - ad > RT_TRI_G * ((su + sv) + ad)));
    if (!on) return 0ull;
    // t ~ (x.b)/det: certainly <= 0 (the literal needs t > eps), or certainly beyond the caller's limit
    // (nearest hit: current best t; shadow ray: distance to the light) -- such a hit cannot count
    float xb = __builtin_fmaf(x.x, b.x, __builtin_fmaf(x.y, b.y, x.z * b.z));
    float st = __builtin_fmaf(fabsf(x.x), ab.x, __builtin_fmaf(fabsf(x.y), ab.y, fabsf(x.z) * ab.z));
    float xbs = __uint_as_float(__float_as_uint(xb) ^ sgn);
    float tl_ad = tlimit * ad;  // inf * 0 = NaN -> comparison false -> no rejection
    on &= ~(wave_ballot(xbs < -RT_TRI_G * st) | wave_ballot(xbs - tl_ad > RT_TRI_G * (st + tl_ad)));
    if (!on) return 0ull;
  }
  WSTAT(n_exact++);
  // ---- phase 2: literal -----------------------------------------------------------------------------
  float inv_det = exact_rcp(det_i);
  V3 r0 = x * inv_det, r1 = y * inv_det, r2 = z * inv_det;
  float t = r0.x * b.x + r0.y * b.y + r0.z * b.z;
  float u = r1.x * b.x + r1.y * b.y + r1.z * b.z;
  float v = r2.x * b.x + r2.y * b.y + r2.z * b.z;
  // determinant(): first cofactor c1.y*c2.z - c2.y*c1.z is bit-equal to X.x (same two products)
  float det = d.x * x.x - c1.x * (d.y * c2.z - c2.y * d.z) + c2.x * (d.y * c1.z - c1.y * d.z);
  const lanemask t_invalid = wave_ballot(t <= RT_EPS);
  const lanemask uv_invalid = wave_ballot(u < 0.0f) | wave_ballot(v < 0.0f) | wave_ballot((u + v) >= 1.0f);
  const lanemask det_invalid = wave_ballot(fabsf(det - 0.0f) <= RT_EPS);
  t_out = t;
  return on & ~(t_invalid | uv_invalid | det_invalid);
}

// per-ray constants of the slab test: reciprocal direction (magnitude clamped so that no inf and
// hence no inf-inf NaN can appear) and -o*inv, so that one box plane costs one fma
struct BoxRay {
  V3 inv, noi;
};
__device__ __forceinline__ BoxRay box_ray(V3 o, V3 d) {
  BoxRay r;
  // v_rcp_f32 (1 ulp) is enough: the slab test is conservative by construction
  r.inv = mk(clampf(__builtin_amdgcn_rcpf(d.x), -1e30f, 1e30f), clampf(__builtin_amdgcn_rcpf(d.y), -1e30f, 1e30f),
             clampf(__builtin_amdgcn_rcpf(d.z), -1e30f, 1e30f));
  r.noi = mk(-(o.x * r.inv.x), -(o.y * r.inv.y), -(o.z * r.inv.z));
  return r;
}

// Conservative slab test of BOTH child boxes of a node (boxes are padded by rt_bvh.cpp; the slack
// below covers the rounding of this test itself).  tlimit_s = t limit with its slack already added.
// Straight-line code, no short-circuit: predicates stay in SGPR lane masks.  (Picking near/far planes
// by the packet's direction signs was tried: hipcc turns the uniform selects into v_mov + v_cndmask
// triples, slower than the per-lane min/max form below.)
__device__ __forceinline__ void box_one(const float* lo, const float* hi, const BoxRay& r, float& tmin, float& tmax) {
  float tx1 = __builtin_fmaf(lo[0], r.inv.x, r.noi.x), tx2 = __builtin_fmaf(hi[0], r.inv.x, r.noi.x);
  float ty1 = __builtin_fmaf(lo[1], r.inv.y, r.noi.y), ty2 = __builtin_fmaf(hi[1], r.inv.y, r.noi.y);
  float tz1 = __builtin_fmaf(lo[2], r.inv.z, r.noi.z), tz2 = __builtin_fmaf(hi[2], r.inv.z, r.noi.z);
  tmin = fmaxf(fmaxf(fminf(tx1, tx2), fminf(ty1, ty2)), fminf(tz1, tz2));
  tmax = fminf(fminf(fmaxf(tx1, tx2), fmaxf(ty1, ty2)), fmaxf(tz1, tz2));
}
__device__ __forceinline__ void box_pair(const RtNode& nd, const BoxRay& r, float tlimit_s, lanemask& h0, lanemask& h1,
                                         float& tn0, float& tn1) {
  float tm0, tm1;
  box_one(nd.lo0, nd.hi0, r, tn0, tm0);
  box_one(nd.lo1, nd.hi1, r, tn1, tm1);
  float s0 = __builtin_fmaf(fabsf(tm0), 4e-6f, tm0 + 1e-5f);
  float s1 = __builtin_fmaf(fabsf(tm1), 4e-6f, tm1 + 1e-5f);
  h0 = wave_ballot(tn0 <= fminf(s0, tlimit_s)) & wave_ballot(s0 >= 0.0f);
  h1 = wave_ballot(tn1 <= fminf(s1, tlimit_s)) & wave_ballot(s1 >= 0.0f);
}
__device__ __forceinline__ float t_limit_slack(float tlimit) {
  return __builtin_fmaf(fabsf(tlimit), 4e-6f, tlimit + 1e-5f);
}
struct Hit {
  float t;
  int id;  // canonical object index, -1 = none
};

// State of one shadow / transmittance ray (IntersectionTest, raytracer.rs:17-106).  The reference folds every hit <= tmax
// into `opacity = clamp(opacity - (1 - io), 0, 1)` and `filter -= absorption` in ITS object order; every decrement is
// >= 0, so the results are max(0, 1 - sum(1 - io_i)) and 1 - sum(absorption_i) whatever the order -- up to float
// rounding.  A BVH visits the hits in an order that depends on the other rays of the wavefront, so the sums are kept as
// INTEGERS (2^-28 / 2^-24 fixed point): integer addition is associative, and the value a lane ends with depends on
// its own ray alone, not on the wavefront it happened to be packed into.  Both only scale the colour (an opaque hit
// occludes whatever the opacity was: clamp(op - 1, 0, 1) = 0 for every op <= 1).
struct Shadow {
  lanemask occ;        // lanes that are completely occluded (wave-uniform mask)
  uint32_t dec;        // sum of the hits' opacity decrements 1 - io, 2^-28 units, saturated at 1.0
  uint32_t fr, fg, fb; // sum of the hits' absorption per channel, 2^-24 units
};
#define RT_SH_ONE (1u << 28)
#define RT_SH_SCALE 268435456.0f
#define RT_SH_INV_SCALE (1.0f / 268435456.0f)
#define RT_FILT_SCALE 16777216.0f
#define RT_FILT_INV_SCALE (1.0f / 16777216.0f)
__device__ __forceinline__ void shadow_init(Shadow& S) {
  S.occ = 0ull;
  S.dec = 0u;
  S.fr = S.fg = S.fb = 0u;
}
__device__ __forceinline__ float shadow_opacity(const Shadow& S) { return (float)(RT_SH_ONE - S.dec) * RT_SH_INV_SCALE; }
__device__ __forceinline__ V3 shadow_filter(const Shadow& S) {
  return mk(1.0f - (float)S.fr * RT_FILT_INV_SCALE, 1.0f - (float)S.fg * RT_FILT_INV_SCALE, 1.0f - (float)S.fb * RT_FILT_INV_SCALE);
}

// one transmissive / opaque occluder on a shadow ray, raytracer.rs:53-92, applied to the lanes in `h`
// (predicated, outside divergent control flow, so that S.occ stays a wave-uniform mask; m is wave-uniform)
__device__ __forceinline__ void shadow_accumulate(Shadow& S, const Mat& m, V3 n, V3 d, lanemask h) {
  if (!m.transmissive) {
    S.occ |= h;  // io = 0: the decrement is 1
    return;
  }
  const float io = m.opacity * (1.0f - fresnel_reflectance_air_red(m, n, -d));
  const bool on = lane_of(h);
  const uint32_t di = (uint32_t)__float2uint_rn(clampf(1.0f - io, 0.0f, 1.0f) * RT_SH_SCALE);
  const uint32_t nd = min(S.dec + di, RT_SH_ONE);
  S.dec = on ? nd : S.dec;
  const V3 ab = absorption(m);  // wave-uniform
  const uint32_t ar = (uint32_t)__float2uint_rn(ab.x * RT_FILT_SCALE), ag = (uint32_t)__float2uint_rn(ab.y * RT_FILT_SCALE),
                 abl = (uint32_t)__float2uint_rn(ab.z * RT_FILT_SCALE);
  S.fr = on ? S.fr + ar : S.fr;
  S.fg = on ? S.fg + ag : S.fg;
  S.fb = on ? S.fb + abl : S.fb;
}

// same for an occluder whose material differs from lane to lane (per-lane walk, rt_hard_kernel)
template <bool CULL>
__device__ __forceinline__ void shadow_accumulate_lane(Shadow& S, const Mat& m, V3 n, V3 d, lanemask h) {
  if (CULL) h &= wave_ballot(m.transmissive || dot(d, n) < 0.75f);  // sphere.rs:137-151, triangle.rs:154-168
  S.occ |= h & wave_ballot(!m.transmissive);
  const float io = m.transmissive ? m.opacity * (1.0f - fresnel_reflectance_air_red(m, n, -d)) : 0.0f;
  const bool on = lane_of(h);
  const uint32_t di = (uint32_t)__float2uint_rn(clampf(1.0f - io, 0.0f, 1.0f) * RT_SH_SCALE);
  const uint32_t nd = min(S.dec + di, RT_SH_ONE);
  S.dec = on ? nd : S.dec;
  const V3 ab = absorption(m);
  S.fr = on ? S.fr + (uint32_t)__float2uint_rn(ab.x * RT_FILT_SCALE) : S.fr;
  S.fg = on ? S.fg + (uint32_t)__float2uint_rn(ab.y * RT_FILT_SCALE) : S.fg;
  S.fb = on ? S.fb + (uint32_t)__float2uint_rn(ab.z * RT_FILT_SCALE) : S.fb;
}

struct WaveCtx {
  // wave-level work counters (uniform)
  uint32_t n_nodes, n_tris, s_nodes, s_tris, s_passes, n_exact, s_exact;  // one ray per wavefront: 32 bits are plenty
#if RT_PROFILE
  unsigned long long t_mark;
  unsigned long long prof[7];  // 0 nearest hit, 1 candidate collection, 2 sample set-up, 3 spheres, 4 triangles, 5 lighting, 6 whole ray
#endif
};

// ------------------------------------------------------------------------------------------------
// nearest hit: spheres linearly (wave-uniform loop), triangles through the BVH or linearly
// ------------------------------------------------------------------------------------------------
template <bool CULL>
__device__ __forceinline__ Hit nearest_hit(const RtDevScene& sc, const RtDevParams& P, WaveCtx& W,
                                           bool alive, V3 o, V3 d) {
  Hit best;
  best.t = INFINITY;
  best.id = -1;
  const lanemask grp = wave_ballot(alive);
  // Sphere pre-selection across lanes (coherent packets: camera rays, sorted secondary rays): lane i looks at
  // sphere i and the ray of the wavefront's first lane.  A point of ray l at parameter t is within
  // so + t * sd of ray 0's point at t (so, sd = spread of the origins / unit directions over the wavefront), and a
  // hit lies at t <= |c - o_l| + r, so sphere i can only be hit by some lane if its centre is within
  // r + so + (|c - o_0| + so + r) * sd of ray 0.  One test for all spheres; the survivors are tested per lane.
  uint32_t pre = 0xFFFFFFFFu;
  if (sc.n_spheres > 2u && grp) {
    const uint32_t ns = sc.n_spheres < 32u ? sc.n_spheres : 32u;
    const int fl = __ffsll((long long)grp) - 1;
    auto first = [&](float v) { return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), fl)); };
    const V3 o0 = mk(first(o.x), first(o.y), first(o.z)), d0 = mk(first(d.x), first(d.y), first(d.z));
    const V3 eo = o - o0, ed = d - d0;
    float so = alive ? fabsf(eo.x) + fabsf(eo.y) + fabsf(eo.z) : 0.0f, sd = alive ? fabsf(ed.x) + fabsf(ed.y) + fabsf(ed.z) : 0.0f;
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) {
      so = fmaxf(so, __shfl_xor(so, k, 64));
      sd = fmaxf(sd, __shfl_xor(sd, k, 64));
    }
    const uint32_t lane = threadIdx.x & 63u, si = lane < ns ? lane : ns - 1u;
    const float4 sp = vload<float4>(sc, sc.off_spheres + si * 16u);
    const float rad = vload<float>(sc, sc.off_sphere_rad + si * 4u);
    const V3 w = mk(sp.x, sp.y, sp.z) - o0;
    const float tc = fmaxf(dot(w, d0), 0.0f);
    const V3 q = w - d0 * tc;
    const float reach = rad + so + (mag(w) + so + rad) * sd * 1.0001f + 1e-6f;
    pre = (uint32_t)(wave_ballot(dot(q, q) <= reach * reach * 1.0002f) & ((1ull << ns) - 1ull));
  }
  auto test_sphere = [&](uint32_t i) {
    float4 s = sload<float4>(sc, sc.off_spheres + i * 16u);
    float t;
    bool h = alive && sphere_hit(s, o, d, t);
    if (CULL && h) {  // sphere.rs:137-151
      V3 p = fma_s(d, t, o);
      V3 n = normalize(p - mk(s.x, s.y, s.z));
      Mat m = load_mat_u(sc, sload<uint32_t>(sc, sc.off_sphere_mat + i * 4u));
      h = (dot(d, n) < 0.75f) || m.transmissive;
    }
    if (h && t <= best.t) {
      best.t = t;
      best.id = (int)i;
    }
  };
  // in index order (ties go to the later object, raytracer.rs:193-213)
  for (uint32_t bits = pre & (sc.n_spheres >= 32u ? 0xFFFFFFFFu : ((1u << sc.n_spheres) - 1u)); bits; bits &= bits - 1u)
    test_sphere((uint32_t)__builtin_ctz(bits));
  for (uint32_t i = 32u; i < sc.n_spheres; i++) test_sphere(i);
  if (sc.n_triangles == 0) return best;
  const int tri_base = (int)sc.n_spheres;

  auto test_tri = [&](uint32_t slot, lanemask lanes) {
    float4 q0 = sload<float4>(sc, sc.off_tri_isect + slot * 48u);
    float4 q1 = sload<float4>(sc, sc.off_tri_isect + slot * 48u + 16u);
    float4 q2 = sload<float4>(sc, sc.off_tri_isect + slot * 48u + 32u);
    float t;
    lanemask h = tri_hit(q0, q1, q2, o, d, lanes, best.t, t, W.n_exact);
    if (CULL) {  // triangle.rs:154-168
      if (h) {
        float4 sh = sload<float4>(sc, sc.off_tri_shade + slot * 16u);
        Mat m = load_mat_u(sc, __float_as_uint(sh.w));
        if (!m.transmissive) h &= wave_ballot(dot(d, mk(sh.x, sh.y, sh.z)) < 0.75f);
      }
    }
    if (h) {
      int id = tri_base + (int)(sload<uint32_t>(sc, sc.off_tri_id + slot * 4u) & RT_TRI_INDEX_MASK);
      if (lane_of(h) && (t < best.t || (t == best.t && id > best.id))) {
        best.t = t;
        best.id = id;
      }
    }
  };

  if (P.traversal == RT_TRAVERSAL_LINEAR) {
    // the literal scan visits every triangle once: skip the extra references of split triangles
    for (uint32_t s = 0; s < sc.n_slots; s++)
      if (!(sload<uint32_t>(sc, sc.off_tri_id + s * 4u) & RT_TRI_DUPLICATE)) test_tri(s, grp);
    return best;
  }

  const BoxRay br = box_ray(o, d);
  const uint32_t lane_id = threadIdx.x & 63u;
  // The traversal stack is ONE stack per wavefront, held in the 64 lanes of a single VGPR
  // (push = select on lane id, pop = v_readlane with a scalar lane index): no LDS round trip.
  // Coherent wavefronts (all camera packets, most sorted secondary packets) point into one direction
  // octant: they walk that octant's copy of the tree (entry / exit planes pre-selected, children stored
  // near-first): 12 fma + max3/min3 per node and no ordering vote.
  // (sign of the clamped reciprocal, so that -0 components agree with the slab arithmetic)
  const lanemask ox = wave_ballot(br.inv.x < 0.0f) & grp, oy = wave_ballot(br.inv.y < 0.0f) & grp, oz = wave_ballot(br.inv.z < 0.0f) & grp;
  const bool octant_uniform = (ox == 0 || ox == grp) && (oy == 0 || oy == grp) && (oz == 0 || oz == grp);
  auto walk = [&](auto uni_tag) {
    constexpr bool UNI = decltype(uni_tag)::value;
    uint32_t nodes_off = sc.off_nodes;
    if (UNI) {
      const uint32_t oct = (ox ? 1u : 0u) | (oy ? 2u : 0u) | (oz ? 4u : 0u);
      nodes_off = sc.off_nodes_oct + __builtin_amdgcn_readfirstlane(oct) * sc.n_nodes * 64u;
    }
    uint32_t stk = 0;
    uint32_t sp = 0;
    uint32_t node = 0;
    for (;;) {
      const RtNode nd = sload<RtNode>(sc, nodes_off + node * 64u);
      WSTAT(W.n_nodes++);
      float tn0, tn1;
      lanemask h0, h1;
      if (UNI) {
        const float tls = t_limit_slack(best.t);
        tn0 = fmaxf(fmaxf(__builtin_fmaf(nd.lo0[0], br.inv.x, br.noi.x), __builtin_fmaf(nd.lo0[1], br.inv.y, br.noi.y)),
                    __builtin_fmaf(nd.lo0[2], br.inv.z, br.noi.z));
        tn1 = fmaxf(fmaxf(__builtin_fmaf(nd.lo1[0], br.inv.x, br.noi.x), __builtin_fmaf(nd.lo1[1], br.inv.y, br.noi.y)),
                    __builtin_fmaf(nd.lo1[2], br.inv.z, br.noi.z));
        const float tm0 = fminf(fminf(__builtin_fmaf(nd.hi0[0], br.inv.x, br.noi.x), __builtin_fmaf(nd.hi0[1], br.inv.y, br.noi.y)),
                                __builtin_fmaf(nd.hi0[2], br.inv.z, br.noi.z));
        const float tm1 = fminf(fminf(__builtin_fmaf(nd.hi1[0], br.inv.x, br.noi.x), __builtin_fmaf(nd.hi1[1], br.inv.y, br.noi.y)),
                                __builtin_fmaf(nd.hi1[2], br.inv.z, br.noi.z));
        const float s0 = __builtin_fmaf(fabsf(tm0), 4e-6f, tm0 + 1e-5f), s1 = __builtin_fmaf(fabsf(tm1), 4e-6f, tm1 + 1e-5f);
        h0 = wave_ballot(tn0 <= fminf(s0, tls)) & wave_ballot(s0 >= 0.0f);
        h1 = wave_ballot(tn1 <= fminf(s1, tls)) & wave_ballot(s1 >= 0.0f);
      } else {
        box_pair(nd, br, t_limit_slack(best.t), h0, h1, tn0, tn1);
      }
      // (an absent child has a NaN box in every copy: no comparison passes, never hit)
      const lanemask b0 = h0 & grp;
      const lanemask b1 = h1 & grp;
      uint32_t next = RT_NODE_EMPTY;
      bool in0 = false, in1 = false;  // internal children to descend into
      if (b0) {
        if (nd.n0) {
          WSTAT(W.n_tris += nd.n0);
          for (uint32_t k = 0; k < nd.n0; k++) test_tri(nd.c0 + k, b0);
        } else {
          in0 = true;
        }
      }
      if (b1) {
        if (nd.n1) {
          WSTAT(W.n_tris += nd.n1);
          for (uint32_t k = 0; k < nd.n1; k++) test_tri(nd.c1 + k, b1);
        } else {
          in1 = true;
        }
      }
      if (in0 && in1) {
        bool first1 = false;
        if (!UNI) {
          // near-first by wave vote among lanes that hit both children
          const lanemask both = b0 & b1;
          const lanemask pref1 = wave_ballot(tn1 < tn0) & both;
          first1 = 2 * __popcll(pref1) > __popcll(both);
        }
        lane_put(stk, sp, first1 ? nd.c0 : nd.c1, lane_id);  // push: lane `sp` of the stack register
        sp++;
        next = first1 ? nd.c1 : nd.c0;
      } else if (in0) {
        next = nd.c0;
      } else if (in1) {
        next = nd.c1;
      }
      if (next == RT_NODE_EMPTY) {
        if (sp == 0) break;
        sp--;
        next = (uint32_t)__builtin_amdgcn_readlane((int)stk, (int)sp);
      }
      node = next;
    }
  };
  if (octant_uniform)
    walk(std::true_type{});
  else
    walk(std::false_type{});
  return best;
}

// ------------------------------------------------------------------------------------------------
// shadow / transmittance ray: every hit <= tmax counts (raytracer.rs:24-106); a lane stops as soon
// as it is completely occluded (its result is discarded by the caller, :800-802).
//
// Triangles are found in one of three ways:
//   * RT_TRAVERSAL_LINEAR: the literal scan;
//   * a BVH walk for this ray (light_mult == 1, or candidate overflow);
//   * soft shadows (light_mult = N > 1): the N rays of one hit point towards the N jittered positions
//     of one light differ by < cloud_delta everywhere along their length, so the BVH is walked ONCE
//     per (wavefront, light) with the segment hit point -> cloud centre against boxes inflated by
//     that delta (collect_light_candidates), and each of the N samples only tests the collected
//     triangle slots.  The candidate set is a superset of what each sample's own walk would reach,
//     every test is still the literal one, so results are unchanged; N-fold fewer node visits.
// ------------------------------------------------------------------------------------------------
#define RT_MAX_CANDIDATES 64u  /* one VGPR's worth of lanes */
#define RT_CAND_OVERFLOW 0xFFFFFFFFu

struct CandList {
  uint32_t reg;     // lane i of this VGPR = i-th candidate triangle slot (wave-level list)
  uint32_t count;   // uniform; RT_CAND_OVERFLOW = not usable, walk the BVH per sample instead
  uint32_t spheres; // uniform bit mask: sphere i (< 32) may be touched by some sample ray of some lane
  unsigned long long umbra;  // uniform lane mask: every sample ray of the lane certainly hits one opaque triangle
#if RT_PROFILE == 5
  uint32_t own;  // per lane: candidates of the list that survive THIS lane's own beam test (the others are only another lane's)
#endif
};

// Receiver flags (rt_flags_kernel, RtDevParams::recv_flags): the collection also runs ONCE PER SCENE over the cells of
// every triangle as a receiver -- one lane = one cell (a parallelogram of the triangle's (u, v) grid), its beam the
// tube from the cell's centre to the light cloud fattened by the cell's half diagonal r (origins within eps_o + r,
// directions within delta + r), the t-rejection ("the surface the point lies on, and everything behind it") taken at the
// cell's four EXACT corners (its left side is convex in the point, so the corners bound the cell).  A cell no triangle
// survives for is flagged clear for that light: at render time a (wavefront, light) whose lanes all sit in clear cells
// has no candidate triangle and skips the walk.
struct FatBeam {
  float r;       // every point of the cell lies within r (1-norm) of the beam origin
  V3 pt[8];      // the cell's corners: a triangle's cell twice, the inner and outer corners of a sphere's
  uint16_t* list;  // this cell's candidate list for the current light (RT_CELL_LIST_SLOTS entries, preset to 0xFFFF), or nullptr
};
// Per-cell candidate lists (RtDevParams::cell_lists): what survives the fat beam of a receiver cell for a light is not
// only counted (receiver flags) but LISTED, up to RT_CELL_LIST_SLOTS leaf slots; entry 0 == RT_CELL_LIST_OVERFLOW: more.
#define RT_CELL_LIST_SLOTS 8u
#define RT_CELL_LIST_END 0xFFFFu
#define RT_CELL_LIST_OVERFLOW 0xFFFEu
enum { COLLECT_OWN = 0, COLLECT_FLAGS = 1 };

// MODE: COLLECT_OWN -- a wavefront's own collection at render time; COLLECT_FLAGS -- fat beams of receiver cells (fb): no
// list and no umbra, the result is L.umbra = lanes some triangle survives for, L.spheres = lanes (low / high word of the
// mask in count / spheres) some sphere is near.
template <bool CULL, int MODE = COLLECT_OWN>
__device__ __forceinline__ CandList collect_light_candidates(const RtDevScene& sc, WaveCtx& W, bool alive, V3 p, V3 c,
                                                              const RtDevParams& P, V3 p_first, float p_spread, uint32_t cand_cap,
                                                              const FatBeam* fb = nullptr, bool walk_tris = true,
                                                              bool have_pre = false, uint32_t pre_reg = 0, uint32_t pre_count = 0) {
  RT_OPAQUE_S(cand_cap);
  const float delta = MODE == COLLECT_FLAGS ? P.beam_delta + fb->r : P.beam_delta;
  const float delta_e5 = MODE == COLLECT_FLAGS ? P.beam_delta_e5 + fb->r : P.beam_delta_e5;
  CandList L;
  L.reg = 0;
  L.count = 0;
  L.spheres = 0xFFFFFFFFu;
  L.umbra = 0ull;
#if RT_PROFILE == 5
  L.own = 0;
#endif
  const uint32_t lane_id = threadIdx.x & 63u;
  // parametric segment x(s) = p + s*(c - p), s in [0, 1]
  V3 dseg = c - p;
  float len = fmaxf(mag(dseg), 1e-20f);
  V3 inv = mk(clampf(__builtin_amdgcn_rcpf(dseg.x), -1e30f, 1e30f), clampf(__builtin_amdgcn_rcpf(dseg.y), -1e30f, 1e30f),
              clampf(__builtin_amdgcn_rcpf(dseg.z), -1e30f, 1e30f));
  V3 noi = mk(-(p.x * inv.x), -(p.y * inv.y), -(p.z * inv.z));
  V3 dl = mk(delta * fabsf(inv.x), delta * fabsf(inv.y), delta * fabsf(inv.z));  // box inflation in s units
  float send = 1.0f + delta_e5 * __builtin_amdgcn_rcpf(len) + 1e-5f;  // past the cloud centre
  float sbeg = -(delta_e5 * __builtin_amdgcn_rcpf(len) + 1e-5f);
  // entry / exit parameters with the inflation folded into the fma constants: cn = noi - dl, cf = noi + dl
  const V3 cn = noi - dl, cf = noi + dl;
  auto box = [&](const float* lo, const float* hi, float& smin) {
    smin = fmaxf(fmaxf(fminf(__builtin_fmaf(lo[0], inv.x, cn.x), __builtin_fmaf(hi[0], inv.x, cn.x)),
                       fminf(__builtin_fmaf(lo[1], inv.y, cn.y), __builtin_fmaf(hi[1], inv.y, cn.y))),
                 fminf(__builtin_fmaf(lo[2], inv.z, cn.z), __builtin_fmaf(hi[2], inv.z, cn.z)));
    float smax = fminf(fminf(fmaxf(__builtin_fmaf(lo[0], inv.x, cf.x), __builtin_fmaf(hi[0], inv.x, cf.x)),
                             fmaxf(__builtin_fmaf(lo[1], inv.y, cf.y), __builtin_fmaf(hi[1], inv.y, cf.y))),
                       fmaxf(__builtin_fmaf(lo[2], inv.z, cf.z), __builtin_fmaf(hi[2], inv.z, cf.z)));
    float smax_s = __builtin_fmaf(fabsf(smax), 8e-6f, smax + 1e-5f);
    return wave_ballot(smin <= fminf(smax_s, send)) & wave_ballot(smax_s >= sbeg);
  };
  // The 64 segments of a wavefront (a few neighbouring pixels, one light) nearly always point into the same
  // octant.  Then the entry / exit plane of every axis is the same for all lanes -- the octant's copy of the
  // tree (RtDevScene::nodes_oct) holds them pre-selected -- and the inflation folds into the per-lane fma
  // constants: 6 fma + max3 + min3 per box instead of 12 fma + 6 min/max + 4 min/max.
  auto box_planes = [&](const float* near, const float* far, float& smin) {  // planes already selected for the octant
    smin = fmaxf(fmaxf(__builtin_fmaf(near[0], inv.x, cn.x), __builtin_fmaf(near[1], inv.y, cn.y)), __builtin_fmaf(near[2], inv.z, cn.z));
    float smax = fminf(fminf(__builtin_fmaf(far[0], inv.x, cf.x), __builtin_fmaf(far[1], inv.y, cf.y)), __builtin_fmaf(far[2], inv.z, cf.z));
    float smax_s = __builtin_fmaf(fabsf(smax), 8e-6f, smax + 1e-5f);
    return wave_ballot(smin <= fminf(smax_s, send)) & wave_ballot(smax_s >= sbeg);
  };
  // Beam-level barycentric rejection.  Every sample ray j of this lane is (o_j, D_j) = (p + do, dseg + dd)
  // with |do| <= eps_o, |dd| <= delta (un-normalised direction; u and v do not depend on its length).
  // The literal numerators are linear in (o, D):
  //   u_j * det_j = (c2 x D_j).(v1 - o_j) = (c2 x dseg).b + (c2 x dd).b - (c2 x D_j).do,   b = v1 - p
  //   det_j = D_j.X = dseg.X + dd.X
  // so with E2 = |e2|_1, B = |b|_1 + eps_o, Lp = len + delta:
  //   |u_j det_j - (c2 x dseg).b| <= E2 * (delta*B + Lp*eps_o),   |det_j - dseg.X| <= delta*|X|_1
  // A triangle is dropped for the whole beam when, with these slacks plus the rounding margin of the
  // per-sample test (2e-6 * Lp * B * E), u < 0, v < 0 or u + v >= 1 holds for EVERY sample and the sign
  // of det cannot change inside the beam.  Otherwise it stays a candidate (the per-sample test decides).
  //
  // Beam-level t rejection (the surface the hit point lies on, and everything behind it).  Sample j starts at
  // so_j = fl(p + ld_j*eps_d), so with b_j = v1 - so_j and unit ld_j || D_j:
  //   (X.b_j) sgn(det) = (X.b0) sgn(det) - eps_d |D_j.X| / |D_j| - (X.rho) sgn,   |rho_i| <= ulp(|p_i| + eps_d)
  // and |D_j.X| / |D_j| >= (|dseg.X| - dslack) / Lp.  The literal test needs t > EPS; its pre-filter already
  // treats (X.b_j) sgn < -G * sum|x_i b_j,i| as "certainly t <= 0" (tri_hit stage 3), so a triangle whose upper
  // bound of the left side stays below that for every sample can never be accepted by any of them.
  const float lenp = len + delta;
  const float rcp_lenp = __builtin_amdgcn_rcpf(lenp);
  const float t_push = P.beam_eps_push * rcp_lenp;
  // (receiver cell: the largest coordinate of any point of it; a hit point the cell stands for lies off the triangle's
  // plane by the rounding of p = o + t d, hence the doubled ulp term; origins lie within eps_o + r of the cell's centre)
  const float p_max = fmaxf(fmaxf(fabsf(p.x), fabsf(p.y)), fabsf(p.z)) + (MODE == COLLECT_FLAGS ? fb->r : 0.0f);
  const float p_ulp = __builtin_fmaf(MODE == COLLECT_FLAGS ? 4e-7f : 1.3e-7f, p_max, P.beam_eps_ulp);  // 1.3e-7 (|p| + eps) + 2.5e-6 eps
  // (p_ulp also carries G*eps_d for sum|x b_j| vs sum|x b0| and the rounding of the unit direction)
  const float eps_o = __builtin_fmaf(2.6e-7f, p_max, P.beam_eps_o) + (MODE == COLLECT_FLAGS ? fb->r + 4e-7f * p_max : 0.0f);  // 1.01 eps + 2 p_ulp >= |so_j - p|
  // returns true when NO lane in `lanes` can be hit by any of its samples (wave-uniform result); staged so
  // that a triangle every lane rejects by its first barycentric alone costs a third of the arithmetic
  auto beam_rejects_all = [&](uint32_t slot, lanemask lanes) -> bool {
    float4 q0 = sload<float4>(sc, sc.off_tri_isect + slot * 48u);
    float4 q1 = sload<float4>(sc, sc.off_tri_isect + slot * 48u + 16u);
    float4 q2 = sload<float4>(sc, sc.off_tri_isect + slot * 48u + 32u);
    V3 c1 = mk(-q0.w, -q1.x, -q1.y), c2 = mk(-q1.z, -q1.w, -q2.x), x = mk(q2.y, q2.z, q2.w);
    V3 b = mk(q0.x, q0.y, q0.z) - p;
    float det = dot(dseg, x);
    float X1 = fabsf(x.x) + fabsf(x.y) + fabsf(x.z);
    float B = fabsf(b.x) + fabsf(b.y) + fabsf(b.z) + eps_o;
    float geo = __builtin_fmaf(delta, B, lenp * eps_o) + 2e-6f * lenp * B;  // per unit edge length
    float ad = fabsf(det);
    float dslack = __builtin_fmaf(delta, X1, 2e-6f * lenp * X1);
    const lanemask open = lanes & ~wave_ballot(ad > dslack * 1.001f);  // sign of det unknown inside the beam: cannot reject
    const uint32_t sgn = __float_as_uint(det) & 0x80000000u;
    // t: the surface the hit point lies on, and everything behind it
    float xbs = __uint_as_float(__float_as_uint(dot(x, b)) ^ sgn);
    float st0 = __builtin_fmaf(fabsf(x.x), fabsf(b.x), __builtin_fmaf(fabsf(x.y), fabsf(b.y), fabsf(x.z) * fabsf(b.z)));
    lanemask rej;
    if (MODE == COLLECT_FLAGS) {
      // every corner of the cell on its own (the cell's centre alone proves nothing about its rim); ad - dslack and lenp
      // bound |D_j.X| from below and |D_j| from above over the whole fat beam, so the inequality holds per point
      const float rhs = t_push * (ad - dslack);
      rej = ~0ull;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const V3 bk = mk(q0.x, q0.y, q0.z) - fb->pt[k];
        const float xk = __uint_as_float(__float_as_uint(dot(x, bk)) ^ sgn);
        const float sk = __builtin_fmaf(fabsf(x.x), fabsf(bk.x), __builtin_fmaf(fabsf(x.y), fabsf(bk.y), fabsf(x.z) * fabsf(bk.z)));
        rej &= wave_ballot(__builtin_fmaf(4e-6f, sk, __builtin_fmaf(X1, p_ulp, xk)) < rhs);
      }
    } else {
      rej = wave_ballot(__builtin_fmaf(4e-6f, st0, __builtin_fmaf(X1, p_ulp, xbs)) < t_push * (ad - dslack));
    }
    if (!(lanes & (open | ~rej))) return true;
    // u
    V3 y = mk((c2.y * dseg.z) + (-c2.z * dseg.y), (c2.z * dseg.x) + (-c2.x * dseg.z), (c2.x * dseg.y) + (-c2.y * dseg.x));
    float E2 = fabsf(c2.x) + fabsf(c2.y) + fabsf(c2.z);
    float ybs = __uint_as_float(__float_as_uint(dot(y, b)) ^ sgn);
    float su = E2 * geo;
    rej |= wave_ballot(ybs + su < 0.0f);
    if (!(lanes & (open | ~rej))) return true;
    // v, u + v
    V3 z = mk((dseg.y * c1.z) + (-dseg.z * c1.y), (dseg.z * c1.x) + (-dseg.x * c1.z), (dseg.x * c1.y) + (-dseg.y * c1.x));
    float E1 = fabsf(c1.x) + fabsf(c1.y) + fabsf(c1.z);
    float zbs = __uint_as_float(__float_as_uint(dot(z, b)) ^ sgn);
    float sv = E1 * geo;
    rej |= wave_ballot(zbs + sv < 0.0f) | wave_ballot((ybs + zbs) - ad - (su + sv) - dslack - 2e-6f * ad > 0.0f);
    if (!(lanes & (open | ~rej))) return true;
    if (MODE == COLLECT_FLAGS) {
      const lanemask surv = lanes & (open | ~rej);
      L.umbra |= surv;  // cells this triangle survives for
      if (fb->list && lane_of(surv)) {  // ... and each such cell's own list (L.reg = the lane's count)
        if (L.reg < RT_CELL_LIST_SLOTS) fb->list[L.reg] = (uint16_t)slot;
        L.reg++;
      }
      return true;  // (nothing is listed for the wavefront)
    }
    // Umbra: the triangle is opaque and EVERY sample ray of the lane hits it between origin and light -- the
    // literal test would accept it for each j (u_j, v_j >= 0, u_j + v_j < 1, EPS < t_j <= tmax_j, |det_j| > EPS,
    // all by margins that cover the rounding of the literal sequence), so the lane is occluded for this light
    // whatever else lies on the way: none of its samples needs to be traced.
    if (!CULL && !(sload<uint32_t>(sc, sc.off_tri_id + slot * 4u) & RT_TRI_TRANSMISSIVE)) {
      const float dlo = ad - dslack;
      const float r_hi = lenp * __builtin_amdgcn_rcpf(dlo) * 1.00001f;                 // >= |D_j| / |D_j.X|
      const float r_lo = (len - delta) * __builtin_amdgcn_rcpf(ad + dslack) * 0.99999f;  // <= |D_j| / |D_j.X|
      const float ex = __builtin_fmaf(4e-6f, st0, X1 * p_ulp);
      const float g_t = 2e-6f * (st0 + X1 * eps_o) * r_hi;  // rounding of the literal t
      const float t_lo = (xbs - ex) * r_lo - 1.001f * eps_o - g_t;
      const float t_hi = (xbs + ex) * r_hi + g_t;
      const lanemask inside = wave_ballot(ybs - su > 0.0f) & wave_ballot(zbs - sv > 0.0f) &
                              wave_ballot((ybs + zbs) + (su + sv) < dlo * 0.99999f) & wave_ballot(xbs > ex) & wave_ballot(t_lo > 2e-7f) &
                              wave_ballot(t_hi < (len - delta) - 1.01f * eps_o - 3e-7f * lenp) &
                              wave_ballot(dlo * rcp_lenp > __builtin_fmaf(2e-6f, X1, 2e-7f));
      L.umbra |= lanes & ~open & inside;
    }
#if RT_PROFILE == 5
    if (lane_of(lanes & (open | ~rej))) L.own++;
#endif
    return false;
  };
  const unsigned long long grp = wave_ballot(alive);
  if (!grp) return L;
  lanemask flags_near = 0ull;  // COLLECT_FLAGS: cells some sphere is near
  // spheres: every point of every sample ray lies within delta of the centre segment, so a sphere
  // whose centre is farther than r + delta from that segment cannot be touched by any of them
  if (MODE == COLLECT_FLAGS) {
    lanemask near_m = 0ull;
    const float inv_len2 = __builtin_amdgcn_rcpf(fmaxf(dot(dseg, dseg), 1e-30f));
    for (uint32_t i = 0; i < sc.n_spheres; i++) {
      const float4 sp4 = sload<float4>(sc, sc.off_spheres + i * 16u);
      const V3 w = mk(sp4.x, sp4.y, sp4.z) - p;
      const float sp = clampf(dot(w, dseg) * inv_len2, 0.0f, 1.0f);
      const V3 q = w - dseg * sp;
      const float reach = sload<float>(sc, sc.off_sphere_rad + i * 4u) + delta;
      near_m |= wave_ballot(alive && dot(q, q) <= reach * reach * 1.0002f + 1e-12f);
    }
    flags_near = near_m;
  } else {
    uint32_t mask = 0;
    const float inv_len2 = __builtin_amdgcn_rcpf(fmaxf(dot(dseg, dseg), 1e-30f));
    const uint32_t ns = sc.n_spheres < 32u ? sc.n_spheres : 32u;
    // Pre-selection across lanes: lane i looks at sphere i and asks whether its centre is within reach of the
    // segment of the wavefront's first hit point -- with the wavefront's spread of hit points (p_spread) added to the
    // reach, since segment l stays within |p_l - p_first| of it.  One test for all spheres instead of one per
    // sphere; only the spheres that pass (mostly none or one) get the per-lane test below.
    uint32_t pre = 0;
    if (ns) {
      const uint32_t si = lane_id < ns ? lane_id : ns - 1u;
      const float4 sp4 = vload<float4>(sc, sc.off_spheres + si * 16u);
      const float rad = vload<float>(sc, sc.off_sphere_rad + si * 4u);
      const V3 ds0 = c - p_first;
      const V3 w = mk(sp4.x, sp4.y, sp4.z) - p_first;
      const float sp = clampf(dot(w, ds0) * __builtin_amdgcn_rcpf(fmaxf(dot(ds0, ds0), 1e-30f)), 0.0f, 1.0f);
      const V3 q = w - ds0 * sp;
      const float reach = rad + delta + p_spread;
      pre = (uint32_t)(wave_ballot(dot(q, q) <= reach * reach * 1.0002f + 1e-12f) & ((1ull << ns) - 1ull));
    }
    for (; pre; pre &= pre - 1u) {
      const uint32_t i = (uint32_t)__builtin_ctz(pre);
      float4 sp4 = sload<float4>(sc, sc.off_spheres + i * 16u);
      V3 w = mk(sp4.x, sp4.y, sp4.z) - p;
      const float wd = dot(w, dseg);
      float sp = clampf(wd * inv_len2, 0.0f, 1.0f);
      V3 q = w - dseg * sp;
      float reach = sload<float>(sc, sc.off_sphere_rad + i * 4u) + delta;
      bool near = dot(q, q) <= reach * reach * 1.0002f + 1e-12f;
      // Leaving rays: with v_j = so_j - centre and unit d_j, sphere_hit has no root >= 0 when cc_j = |v_j|^2 - r^2
      // is positive by more than the rounding of disc = b^2 - 4 cc (b^2 <= 4 |v|^2) and d_j.v_j > 0.  Over the beam
      // d_j.v0 >= (-w.dseg - delta |w|) / Lp =: amin and cc_j >= cc0 + 2 eps_d amin - rounding.  This is the sphere
      // the hit point lies on (lit side) and every sphere behind the hit point.
      if (!wave_ballot(alive && near)) continue;
      const float wl2 = dot(w, w), w1 = fabsf(w.x) + fabsf(w.y) + fabsf(w.z);
      const float amin = (-wd - delta * w1) * rcp_lenp;
      const float cc_lo = (wl2 - sp4.w) + P.beam_eps_198 * amin - (6e-6f * wl2 + 4.0f * p_ulp * w1);
      const bool leaving = (amin > 1e-6f * w1) && (cc_lo > 0.0f);
      if (wave_ballot(alive && near && !leaving)) mask |= 1u << i;
    }
    L.spheres = mask | (sc.n_spheres > 32u ? 0xFFFFFFFFu : 0u);
  }
  if (!walk_tris) return L;  // (wave-uniform: every lane sits in a receiver cell no triangle can shadow for this light)
  // direction octant of the wavefront (sign of inv = sign of dseg, -0 included)
  const unsigned long long mx = wave_ballot(__float_as_uint(inv.x) >> 31) & grp, my = wave_ballot(__float_as_uint(inv.y) >> 31) & grp,
                           mz = wave_ballot(__float_as_uint(inv.z) >> 31) & grp;
  const bool octant_uniform = (mx == 0 || mx == grp) && (my == 0 || my == grp) && (mz == 0 || mz == grp);
  // One walk, two instantiations.  Uniform octant (the rule): the octant's copy of the tree already holds the
  // entry / exit planes in lo / hi and its children in near-first order, so a node costs 12 fma + 2 max3/min3 and
  // no ordering vote.  Mixed octants: the generic slab test and a vote, on the original nodes.
  auto walk = [&](auto uni_tag) {
    constexpr bool UNI = decltype(uni_tag)::value;
    uint32_t nodes_off = sc.off_nodes;
    if (UNI) {
      const uint32_t oct = (mx ? 1u : 0u) | (my ? 2u : 0u) | (mz ? 4u : 0u);
      nodes_off = sc.off_nodes_oct + __builtin_amdgcn_readfirstlane(oct) * sc.n_nodes * 64u;
    }
    uint32_t stk = 0, sp = 0, node = 0;
    for (;;) {
      const RtNode nd = sload<RtNode>(sc, nodes_off + node * 64u);
      WSTAT(W.s_nodes++);
      float tn0, tn1;
      lanemask h0, h1;
      if (UNI) {
        h0 = box_planes(nd.lo0, nd.hi0, tn0);
        h1 = box_planes(nd.lo1, nd.hi1, tn1);
      } else {
        h0 = box(nd.lo0, nd.hi0, tn0);
        h1 = box(nd.lo1, nd.hi1, tn1);
      }
      // (an absent child has a NaN box in every copy: no comparison passes, never hit)
      const lanemask b0 = h0 & grp;
      const lanemask b1 = h1 & grp;
      // near-first order so that early occluders are tested first by every sample
      bool first1 = false;
      if (!UNI) {
        const unsigned long long both = b0 & b1;
        first1 = both && (2 * __popcll(wave_ballot(tn1 < tn0) & both) > __popcll(both));
      }
      uint32_t next = RT_NODE_EMPTY;
      bool in0 = false, in1 = false;
      // leaves are appended in visiting order (nearer child first)
      for (int pass = 0; pass < 2; pass++) {
        const bool second = (pass == 1) != first1;  // which child this pass handles
        const unsigned long long b = second ? b1 : b0;
        const uint32_t cc = second ? nd.c1 : nd.c0, nn = second ? nd.n1 : nd.n0;
        if (!b) continue;
        if (nn) {
          if (L.count + nn > cand_cap) {
            L.count = RT_CAND_OVERFLOW;
            return;
          }
          for (uint32_t k = 0; k < nn; k++) {
            if (beam_rejects_all(cc + k, b)) continue;  // no sample of any lane can hit it
            lane_put(L.reg, L.count, cc + k, lane_id);
            L.count++;
          }
          // every lane is in full shadow: nothing left to find (receiver cells: every cell has a survivor -- enough for
          // the flags, not for the cells' lists)
          if ((L.umbra & grp) == grp && !(MODE == COLLECT_FLAGS && P.cell_list_out)) return;  // (a UNIFORM condition: fb->list is per lane)
        } else if (second) {
          in1 = true;
        } else {
          in0 = true;
        }
      }
      if (in0 && in1) {
        lane_put(stk, sp, first1 ? nd.c0 : nd.c1, lane_id);
        sp++;
        next = first1 ? nd.c1 : nd.c0;
      } else if (in0) {
        next = nd.c0;
      } else if (in1) {
        next = nd.c1;
      }
      if (next == RT_NODE_EMPTY) {
        if (sp == 0) break;
        sp--;
        next = (uint32_t)__builtin_amdgcn_readlane((int)stk, (int)sp);
      }
      node = next;
    }
  };
  if (MODE == COLLECT_OWN && have_pre) {
    // The union of the lanes' per-cell lists stands for the walk: every triangle a sample ray of any lane can hit is in it
    // (a cell's list = what survives the cell's fat beam, a superset of what survives the beam of any point of the cell).
    // The lanes' own beams then thin it out exactly as they thin out the leaves of a walk.
    for (uint32_t i = 0; i < pre_count; i++) {
      const uint32_t slot = (uint32_t)__builtin_amdgcn_readlane((int)pre_reg, (int)i);
      if (beam_rejects_all(slot, grp)) continue;
      if (L.count >= cand_cap) {
        L.count = RT_CAND_OVERFLOW;
        return L;
      }
      lane_put(L.reg, L.count, slot, lane_id);
      L.count++;
      if ((L.umbra & grp) == grp) return L;
    }
    return L;
  }
  if (octant_uniform)
    walk(std::true_type{});
  else
    walk(std::false_type{});
  if (MODE == COLLECT_FLAGS) {
    L.count = (uint32_t)flags_near;
    L.spheres = (uint32_t)(flags_near >> 32);
  }
  return L;
}

// LIST: the caller has checked (once per light, not once per sample) that a shared candidate list exists
template <bool CULL, bool LIST>
__device__ __forceinline__ Shadow shadow_ray(const RtDevScene& sc, const RtDevParams& P, WaveCtx& W,
                                             lanemask grp, V3 o, V3 d_raw, float tmax, const CandList& cand) {
  Shadow S;
  shadow_init(S);
  V3 d = normalize_unit(d_raw, grp);  // Ray::new_with_mask re-normalises, ray.rs:52-57 (d_raw is unit up to rounding)
#if RT_PROFILE
  RT_OPAQUE(d.x);
  PROF_ADD(W, 2, W.t_mark);
#endif
  const unsigned long long t_sph = PROF_T();
  auto test_sphere = [&](uint32_t i) {
    float4 s = sload<float4>(sc, sc.off_spheres + i * 16u);
    float t = 0.0f;
    lanemask h = grp & ~S.occ & wave_ballot(sphere_hit(s, o, d, t));
    h &= wave_ballot(t <= tmax);
    if (h) {
      // (computed for every lane, applied to the lanes in h: keeps S.occ a wave-uniform mask)
      // The occluder's normal only feeds the Fresnel factor of a TRANSMISSIVE sphere, i.e. the light's colour (opacity
      // reaches the occlusion decision only for opaque occluders, where io = 0 whatever the normal): 1-ulp rsq instead
      // of IEEE sqrt + division.  With backface culling the normal decides a hit and stays exact.
      V3 p = fma_s(d, t, o);
      V3 n = CULL ? normalize(p - mk(s.x, s.y, s.z)) : fast_normalize(p - mk(s.x, s.y, s.z));
      Mat m = load_mat_u(sc, sload<uint32_t>(sc, sc.off_sphere_mat + i * 4u));
      if (CULL && !m.transmissive) h &= wave_ballot(dot(d, n) < 0.75f);
      shadow_accumulate(S, m, n, d, h);
    }
  };
  // spheres 0..31 through the bits of the (wavefront, light) mask, in index order; any further ones unconditionally
  for (uint32_t bits = cand.spheres & (sc.n_spheres >= 32u ? 0xFFFFFFFFu : ((1u << sc.n_spheres) - 1u)); bits; bits &= bits - 1u)
    test_sphere((uint32_t)__builtin_ctz(bits));
  for (uint32_t i = 32u; i < sc.n_spheres; i++) test_sphere(i);
  if (!LIST && sc.n_triangles == 0) return S;
#if RT_PROFILE
  RT_OPAQUE(S.dec);
#endif
  PROF_ADD(W, 3, t_sph);
  const unsigned long long t_tri = PROF_T();
  (void)t_tri;

  auto test_tri = [&](uint32_t slot, lanemask lanes) {
    float4 q0 = sload<float4>(sc, sc.off_tri_isect + slot * 48u);
    float4 q1 = sload<float4>(sc, sc.off_tri_isect + slot * 48u + 16u);
    float4 q2 = sload<float4>(sc, sc.off_tri_isect + slot * 48u + 32u);
    float t;
    lanemask h = tri_hit(q0, q1, q2, o, d, lanes & ~S.occ, tmax, t, W.s_exact);
    h &= wave_ballot(t <= tmax);
    if (h) {
      float4 sh = sload<float4>(sc, sc.off_tri_shade + slot * 16u);
      Mat m = load_mat_u(sc, __float_as_uint(sh.w));
      V3 n = mk(sh.x, sh.y, sh.z);
      if (CULL && !m.transmissive) h &= wave_ballot(dot(d, n) < 0.75f);
      shadow_accumulate(S, m, n, d, h);
    }
  };

  if (!LIST && P.traversal == RT_TRAVERSAL_LINEAR) {
    for (uint32_t s = 0; s < sc.n_slots; s++)
      if (!(sload<uint32_t>(sc, sc.off_tri_id + s * 4u) & RT_TRI_DUPLICATE)) test_tri(s, grp);
    return S;
  }
  WSTAT(W.s_passes++);

  if (LIST) {
    // soft shadows: test the triangle slots collected once for this (wavefront, light)
    WSTAT(W.s_tris += cand.count);
    for (uint32_t c = 0; c < cand.count; c++) {
      if (!(grp & ~S.occ)) break;
      uint32_t slot = (uint32_t)__builtin_amdgcn_readlane((int)cand.reg, (int)c);
      test_tri(slot, grp);
    }
#if RT_PROFILE
    RT_OPAQUE(S.dec);
#endif
    PROF_ADD(W, 4, t_tri);
    return S;
  }

  const BoxRay br = box_ray(o, d);
  const float tl = t_limit_slack(tmax);
  const uint32_t lane_id = threadIdx.x & 63u;
  {
    uint32_t stk = 0;
    uint32_t sp = 0;
    uint32_t node = 0;
    for (;;) {
      const lanemask live = grp & ~S.occ;
      if (!live) break;
      const RtNode nd = sload<RtNode>(sc, sc.off_nodes + node * 64u);
      WSTAT(W.s_nodes++);
      float tn0, tn1;
      lanemask h0, h1;
      box_pair(nd, br, tl, h0, h1, tn0, tn1);
      const lanemask b0 = nd.c0 != RT_NODE_EMPTY ? (h0 & live) : 0ull;
      const lanemask b1 = nd.c1 != RT_NODE_EMPTY ? (h1 & live) : 0ull;
      uint32_t next = RT_NODE_EMPTY;
      bool in0 = false, in1 = false;
      if (b0) {
        if (nd.n0) {
          WSTAT(W.s_tris += nd.n0);
          for (uint32_t k = 0; k < nd.n0; k++) test_tri(nd.c0 + k, b0);
        } else {
          in0 = true;
        }
      }
      if (b1) {
        if (nd.n1) {
          WSTAT(W.s_tris += nd.n1);
          for (uint32_t k = 0; k < nd.n1; k++) test_tri(nd.c1 + k, b1);
        } else {
          in1 = true;
        }
      }
      if (in0 && in1) {
        // any-hit: visit the child that is nearer for most lanes first -- an early occluder ends the
        // traversal for the whole wavefront
        const unsigned long long both = b0 & b1;
        const unsigned long long pref1 = wave_ballot(tn1 < tn0) & both;
        const bool first1 = 2 * __popcll(pref1) > __popcll(both);
        lane_put(stk, sp, first1 ? nd.c0 : nd.c1, lane_id);
        sp++;
        next = first1 ? nd.c1 : nd.c0;
      } else if (in0) {
        next = nd.c0;
      } else if (in1) {
        next = nd.c1;
      }
      if (next == RT_NODE_EMPTY) {
        if (sp == 0) break;
        sp--;
        next = (uint32_t)__builtin_amdgcn_readlane((int)stk, (int)sp);
      }
      node = next;
    }
  }
  return S;
}

// surface data of a hit (SurfaceInteraction, surface_interaction.rs)
struct Surf {
  V3 p, n;
  uint32_t mat;
};
__device__ __forceinline__ Surf surface_of(const RtDevScene& sc, Hit h, V3 o, V3 d) {
  Surf s;
  s.p = fma_s(d, h.t, o);
  if (h.id < (int)sc.n_spheres) {
    float4 sp = vload<float4>(sc, sc.off_spheres + (uint32_t)h.id * 16u);
    s.n = normalize(s.p - mk(sp.x, sp.y, sp.z));
    s.mat = vload<uint32_t>(sc, sc.off_sphere_mat + (uint32_t)h.id * 4u);
  } else {
    // tri_shade holds a canonical-order copy behind the leaf-order one for this lookup (rt_api.cpp)
    float4 sh = vload<float4>(sc, sc.off_tri_shade + (sc.n_slots + (uint32_t)(h.id - (int)sc.n_spheres)) * 16u);
    s.n = mk(sh.x, sh.y, sh.z);
    s.mat = __float_as_uint(sh.w);
  }
  return s;
}

__device__ __forceinline__ uint32_t to_u8(float x) {
  float cx = fminf(fmaxf(x, 0.0f), 1.0f);
  return (uint32_t)__float2uint_rn(cx * 255.0f);
}

// ---- one light sample's colour terms ---------------------------------------------------------------------------
// PointLight::calculate_contribution_at (light.rs:261-299) and the sample's share of calculate_lighting
// (raytracer_renderer.rs:800-851).  ONE function for every route a (hit point, light) pair can take -- the arrival
// loop of sets with nothing to test, the traced loops, rt_hard_kernel -- so that a sample's terms are the same bits
// whatever the wavefront around it made the kernel do (the image does not depend on how lanes are packed into
// wavefronts).  Everything here only scales the colour (tolerance 1e-4, measured < 4e-6): the direction and distance
// to the light come from one v_rsq of |ltp|^2 (the shadow ray itself is set up with the exact sequences by its caller).
// Two exact identities of the reference are used to drop work: `cosi = (ltp . n) / (|ltp| + EPS)` is `diff = n . ld` up
// to a factor 1 + EPS/|ltp| (ld = ltp / |ltp|), and its `cosi > 0` selects are implied by the `diff > 0` gate of the sum.
struct LightTerms {
  V3 mLc;    // surface colour^2 x light colour / filter (the reference multiplies the surface colour in twice)
  float lf;  // diff * intensity * opacity: the sample's share of `direct`
  float sf;  // intensity * opacity * specular lobe: its share of `specular`
  bool lit;  // diff > 0
};
// FILTERED = false: the shadow ray is known to arrive untouched (opacity 1, filter 1); an untouched FILTERED sample
// gives the same bits (x * 1.0f and x * v_rcp(1.0f) are x).
template <bool FILTERED>
__device__ __forceinline__ LightTerms light_sample_terms(V3 n, V3 view, V3 mmc_lc, float lI, float mshin, bool has_spec, V3 ltp,
                                                         const Shadow& S) {
  LightTerms T;
  const float l2 = dot(ltp, ltp);
  const float rs = __builtin_amdgcn_rsqf(l2);
  const V3 ld = ltp * rs;
  const float dist = l2 * rs;        // |ltp| (+ EPS in the reference: below the colour tolerance by three orders)
  const float diff = dot(n, ld);     // = cosi
  // att = 0.95 (EPS + dist + dist^2); (tanh(att) + 1) / 2 = 1 / (1 + exp(-2 att)): one exp2 and one rcp, already inside
  // [0, 1]; the 0.95 and the -2 log2(e) of the exponent are one constant
  const float sig = RT_FAST_TRANS ? __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(dist, dist, dist) * -2.7411205779f))
                                  : clampf((tanhf(0.95f * (RT_EPS + dist + dist * dist)) + 1.0f) / 2.0f, 0.0f, 1.0f);
  const float cint = diff * lI * sig;
  T.mLc = mmc_lc;
  float opac = 1.0f;
  if (FILTERED) {
    const V3 f = shadow_filter(S);
    T.mLc = mk(mmc_lc.x * __builtin_amdgcn_rcpf(f.x), mmc_lc.y * __builtin_amdgcn_rcpf(f.y), mmc_lc.z * __builtin_amdgcn_rcpf(f.z));
    opac = shadow_opacity(S);
  }
  float specf = 0.0f;
  if (has_spec) {
    // reflected(ld, n) = ld - 2 (ld.n) n with ld.n = diff; a reflection keeps the length, and the shading normal
    // enters only through n / |n|^2 -- the reference normalises the result, here |n| = 1 is not assumed:
    // normalize(v) . d = (v . d) / |v|
    const V3 rv = fma_s(n, -2.0f * diff, ld);
    const float base = fmaxf(dot(rv, view) * __builtin_amdgcn_rsqf(dot(rv, rv)), 0.0f);
    specf = RT_FAST_TRANS ? fast_pow01(base, fmaxf(mshin * 512.0f, 1.0f)) : powf(base, fmaxf(mshin * 512.0f, 1.0f));
  }
  T.lf = FILTERED ? diff * cint * opac : diff * cint;
  T.sf = FILTERED ? cint * opac * specf : cint * specf;
  T.lit = diff > 0.0f;
  return T;
}

enum { KIND_PRIMARY = 0, KIND_REFL = 1, KIND_REFR = 2 };

// fixed-point pixel accumulator used when secondary rays are streamed: integer adds are
// associative, so the per-pixel sum does not depend on the order in which rays retire
#define RT_ACC_SCALE 68719476736.0f /* 2^36 */
#define RT_ACC_INV_SCALE (1.0f / 68719476736.0f)

struct Wave {
  WaveCtx ctx;
  uint32_t cnt_kind[3], cnt_shadow, cnt_pass, cnt_lanes, cnt_traced;
};

__device__ __forceinline__ void wave_init(Wave& w) {
  w.ctx.n_nodes = w.ctx.n_tris = w.ctx.s_nodes = w.ctx.s_tris = w.ctx.s_passes = w.ctx.n_exact = w.ctx.s_exact = 0;
  w.cnt_kind[0] = w.cnt_kind[1] = w.cnt_kind[2] = 0;
  w.cnt_shadow = w.cnt_pass = w.cnt_lanes = w.cnt_traced = 0;
#if RT_PROFILE
  for (int i = 0; i < 7; i++) w.ctx.prof[i] = 0;
  w.ctx.t_mark = 0;
#endif
}

// Statistics: the wavefronts of a workgroup add their counters in LDS, then 15 threads issue one global
// atomic each (RT_COUNTER_REPLICAS copies on separate 128-B lines keep the per-line atomic rate off the
// critical path; the host sums them).
#define RT_N_COUNTERS 15u
// Every thread of the workgroup calls wave_flush_init once at the START of the kernel (one barrier, where nobody waits for
// anybody); wave_flush itself has no barrier: a wavefront adds its counters to the workgroup's sums in LDS and counts
// itself done, and the wavefront that arrives LAST hands the sums to the global replicas.  A wavefront that is done
// therefore ends at once -- it does not sit on its registers until the slowest of its workgroup has finished.
__device__ __forceinline__ void wave_flush_init(const RtDevParams& P, unsigned long long* lds_cnt) {
  if (!P.counters) return;  // wave-uniform (kernel argument)
  if (threadIdx.x < 16u) lds_cnt[threadIdx.x] = 0ull;
  __syncthreads();
}
__device__ __forceinline__ void wave_flush(const Wave& w, const RtDevParams& P, unsigned long long written,
                                           unsigned long long* lds_cnt) {
  if (!P.counters) return;  // wave-uniform (kernel argument)
  uint32_t before = 0u;
  if ((threadIdx.x & 63u) == 0) {
#if RT_PROFILE
    const unsigned long long v[RT_N_COUNTERS] = {w.cnt_kind[0], w.cnt_kind[1], w.cnt_kind[2], w.cnt_shadow, written,
                                                 w.cnt_pass,    w.ctx.prof[6], w.ctx.prof[0],  w.ctx.prof[1], w.ctx.prof[2],
                                                 w.ctx.prof[3], w.ctx.s_passes, w.ctx.prof[4], w.ctx.prof[5], w.cnt_traced};
#else
    const unsigned long long v[RT_N_COUNTERS] = {w.cnt_kind[0], w.cnt_kind[1], w.cnt_kind[2], w.cnt_shadow, written,
                                                 w.cnt_pass,    w.cnt_lanes,   w.ctx.n_nodes,  w.ctx.n_tris, w.ctx.s_nodes,
                                                 w.ctx.s_tris,  w.ctx.s_passes, w.ctx.n_exact, w.ctx.s_exact, w.cnt_traced};
#endif
#pragma unroll
    for (unsigned i = 0; i < RT_N_COUNTERS; i++)
      if (v[i]) atomicAdd(&lds_cnt[i], v[i]);
    // the hand-off to the wavefront that arrives last: release our sums, acquire everybody else's
    before = (uint32_t)__hip_atomic_fetch_add(&lds_cnt[15], 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  before = __builtin_amdgcn_readfirstlane(before);
  if (before + 1u == (blockDim.x >> 6) && (threadIdx.x & 63u) < RT_N_COUNTERS) {
    const uint32_t i = threadIdx.x & 63u;
    unsigned long long v = lds_cnt[i];
    if (v) atomicAdd(&P.counters[(size_t)(blockIdx.x % RT_COUNTER_REPLICAS) * 16u + i], v);
  }
}

// One ray in flight: a node of the Whitted tree (single_raytrace, raytracer_renderer.rs:147-264)
struct RayIn {
  V3 o, d_raw;    // origin, un-normalised direction (Ray::new_with_mask normalises)
  float n_start;  // refraction index of the medium the ray travels in
  V3 Wt;          // RGB weight of this node's colour in the pixel sum
  int depth;      // Option<usize>: -1 = None
  int kind;
  uint32_t pix;
  uint32_t mult;  // how many identical rays of the reference this one stands for (repeated AA samples)
};

// depth (-1 = None .. 64), kind and multiplicity of a ray in one dword (queue plane 1 .w, LDS stash)
__device__ __forceinline__ int pack_dkm(int depth, int kind, uint32_t mult) {
  return (int)((mult << 10) | ((uint32_t)(depth + 1) << 2) | (uint32_t)kind);
}
__device__ __forceinline__ void unpack_dkm(int v, int& depth, int& kind, uint32_t& mult) {
  kind = v & 3;
  depth = (int)(((uint32_t)v >> 2) & 0xFFu) - 1;
  mult = (uint32_t)v >> 10;
}

// sum of a per-lane value over the wavefront (uniform result)
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int k = 32; k >= 1; k >>= 1) v += (uint32_t)__shfl_xor((int)v, k, 64);
  return __builtin_amdgcn_readfirstlane(v);
}

struct RayOut {
  bool hit;
  float t;
  int id;
  V3 contrib;  // Wt * own terms of this node (zero on miss)
  uint32_t pix, mult;  // of the ray (valid when hit; read back from the LDS stash)
};

// appends the lanes with `on` to the ray queue (wave-level compaction: one atomic per wavefront)
__device__ __forceinline__ void queue_push(const RtDevParams& P, bool on, V3 o, V3 d, float n_start, V3 Wt,
                                           int depth, int kind, uint32_t pix, uint32_t mult) {
  unsigned long long m = wave_ballot(on);
  if (!m) return;
  uint32_t n = (uint32_t)__popcll(m);
  uint32_t base = 0;
  if ((threadIdx.x & 63u) == 0) base = atomicAdd(P.q_out_count, n);
  base = __builtin_amdgcn_readfirstlane(base);
  uint32_t rankl = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
  if (on) {
    uint32_t i = base + rankl;
    if (i < P.q_capacity) {
      float4* rec = P.q_out + (size_t)i * RT_QUEUE_QUADS;  // one 64-byte record per ray (quad 3: rt_trace_kernel)
      stream_store4(rec + 0, make_float4(o.x, o.y, o.z, n_start));
      stream_store4(rec + 1, make_float4(d.x, d.y, d.z, __int_as_float(pack_dkm(depth, kind, mult))));
      stream_store4(rec + 2, make_float4(Wt.x, Wt.y, Wt.z, __uint_as_float(pix)));
    } else {
      atomicAdd(P.q_overflow, 1u);  // the host sizes the queues from the frame before; a drop makes it render the frame again
    }
  }
}

// appends the (hit point, light) pairs of the lanes in `m` to the hard queue (see RtDevParams::hard_q): one atomic per
// wavefront, no divergent branch (idle lanes store into dump slots behind the queue)
__device__ __forceinline__ void hard_push(const RtDevParams& P, lanemask m, V3 p, V3 n, V3 view, uint32_t mat_row, uint32_t light,
                                          uint32_t pix, V3 Wa, uint32_t mult) {
  const uint32_t cnt = (uint32_t)__popcll(m);
  uint32_t base = 0;
  if ((threadIdx.x & 63u) == 0) {
    base = atomicAdd(P.hard_count, cnt);
    if (base + cnt > P.hard_capacity) atomicAdd(&P.hard_stat[0], 1u);  // (a drop makes the host render the frame again, larger)
  }
  base = __builtin_amdgcn_readfirstlane(base);
  const uint32_t rankl = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
  const uint32_t i = (lane_of(m) && base + rankl < P.hard_capacity) ? base + rankl : P.hard_capacity + (threadIdx.x & 63u);
  const size_t stride = (size_t)P.hard_capacity + 64u;
  P.hard_q[0 * stride + i] = make_float4(p.x, p.y, p.z, __uint_as_float(mat_row));
  P.hard_q[1 * stride + i] = make_float4(n.x, n.y, n.z, __uint_as_float(light));
  P.hard_q[2 * stride + i] = make_float4(view.x, view.y, view.z, __uint_as_float(pix));
  P.hard_q[3 * stride + i] = make_float4(Wa.x, Wa.y, Wa.z, __uint_as_float(mult));
}

// ------------------------------------------------------------------------------------------------
// trace + shade one ray per lane (wave-cooperative traversal inside); children go to the queue
// ------------------------------------------------------------------------------------------------
// LDS stash: [field][256 threads], one dword per lane per field.  Fields 0-11: ray state parked across the light loop;
// STREAM kernels: 12-17 = the lane's pixel contribution as three 64-bit fixed-point sums (RT_ACC_SCALE units).
#define RT_STASH_FIELDS 18u
#define RT_STASH_CELL 11u  /* receiver cell of the hit point (RT_NO_CELL: none) */
#define RT_STASH_FIX 12u
#define RT_NO_CELL 0xFFFFFFFFu
#define RT_MAT_TRANS_BIT 0x80000000u  /* stash field 6: material row | this bit when the material is transmissive */
__device__ __forceinline__ long long* stash_fix(float* stash) { return (long long*)(stash + RT_STASH_FIX * 256u); }
// PRE: the nearest hit was found by rt_trace_kernel and is passed in (`pre`); otherwise it is traced here.
// STREAM: secondary rays exist (reflections / refractions): children are queued, pixel sums go through the fixed-point
// accumulator, hard soft-shadow pairs are deferred.  The frames without them (configs 1-3) run a kernel that does not
// contain any of that code.
template <bool CULL, bool PRE, bool STREAM>
__device__ __forceinline__ RayOut process_ray(const RtDevScene& sc, const RtDevParams& P, Wave& wv, bool have,
                                              const RayIn& r, float* stash /* LDS: [RT_STASH_FIELDS][256] */,
                                              Hit pre) {
  RayOut out;
  out.hit = false;
  out.t = 0.0f;
  out.id = -1;
  out.contrib = mk(0.0f, 0.0f, 0.0f);
  out.pix = 0;
  out.mult = 1;
  WaveCtx& W = wv.ctx;
  const V3 epsv = mk(P.eps_distance, P.eps_distance, P.eps_distance);
  const uint32_t N = P.light_mult < 1u ? 1u : P.light_mult;
  const bool stream = STREAM && P.q_out != nullptr;
  if (STREAM) {
    long long* fx = stash_fix(stash) + threadIdx.x;
    fx[0] = fx[256] = fx[512] = 0;
  }

  const unsigned long long t_all = PROF_T();
  V3 d = normalize(r.d_raw);  // Ray::new_with_mask, ray.rs:52-57
  bool alive = have && !has_nan(d);
  unsigned long long bal = wave_ballot(alive);
  if (!bal) return out;
  Hit h = pre;
  if (!PRE) {
    // ray accounting: lanes entering cast_ray (camera rays here; children are counted by rt_trace_kernel), each
    // standing for `mult` rays of the reference
    wv.cnt_kind[0] += P.weighted ? wave_sum(alive ? r.mult : 0u) : (uint32_t)__popcll(bal);
    wv.cnt_traced += (uint32_t)__popcll(bal);
    WSTAT(wv.cnt_pass += 1);
    WSTAT(wv.cnt_lanes += (uint32_t)__popcll(bal));
    const unsigned long long t_n = PROF_T();
    h = nearest_hit<CULL>(sc, P, W, alive, r.o, d);
#if RT_PROFILE
    RT_OPAQUE(h.t);
#endif
    PROF_ADD(W, 0, t_n);
  }
  bool hit = alive && h.id >= 0;
  out.hit = hit;
  out.t = h.t;
  out.id = hit ? h.id : -1;
  const lanemask hit_m = wave_ballot(hit);
  if (!hit_m) return out;

  {
    // Park what the light loop does not need in LDS (SoA, one dword per lane per field: conflict
    // free), before the surface and material are fetched.  The loop below runs lights x N shadow
    // traversals; these values would otherwise sit in VGPRs (or worse, in scratch = HBM traffic)
    // for all of them.
    // a reflection child's weight carries atten(child.t), known only now (:722-726)
    float a0 = atten(h.t);
    V3 W0 = r.Wt;
    if (r.kind == KIND_REFL) W0 = W0 * a0;
    float* st = stash + threadIdx.x;
    st[0 * 256] = W0.x;
    st[1 * 256] = W0.y;
    st[2 * 256] = W0.z;
    st[3 * 256] = a0;
    st[4 * 256] = r.n_start;
    st[5 * 256] = __int_as_float(pack_dkm(r.depth, r.kind, r.mult));
    st[7 * 256] = h.t;
    st[8 * 256] = __int_as_float(out.id);
    st[9 * 256] = __uint_as_float(r.pix);
  }
  Surf sf;
  sf.p = mk(0, 0, 0);
  sf.n = mk(0, 0, 1);
  sf.mat = 0;
  if (hit) sf = surface_of(sc, h, r.o, d);
  if (N > 1) {
    // receiver flags of the cell the hit point lies in (bit l: no triangle can shadow it for light l, bit 8 + l: no sphere)
    uint32_t rflags = 0u, cell = RT_NO_CELL;
    if (P.recv_flags && hit && h.id >= (int)sc.n_spheres) {
      const uint32_t ro = sc.off_recv + (uint32_t)(h.id - (int)sc.n_spheres) * 48u;
      const float4 ru = vload<float4>(sc, ro), rv = vload<float4>(sc, ro + 16u);
      const uint2 rr = vload<uint2>(sc, ro + 32u);  // {R, first cell}
      const float Rf = (float)rr.x;
      const float cu = (__builtin_fmaf(ru.x, sf.p.x, __builtin_fmaf(ru.y, sf.p.y, ru.z * sf.p.z)) + ru.w) * Rf;
      const float cv = (__builtin_fmaf(rv.x, sf.p.x, __builtin_fmaf(rv.y, sf.p.y, rv.z * sf.p.z)) + rv.w) * Rf;
      const uint32_t ci = (uint32_t)fminf(fmaxf(cu, 0.0f), Rf - 1.0f), cj = (uint32_t)fminf(fmaxf(cv, 0.0f), Rf - 1.0f);
      // (cells beyond the hypotenuse carry no flags; the cells are computed 5 % larger than they are, which covers the
      // rounding of u and v)
      if (rr.x != 0u && ci + cj < rr.x) cell = rr.y + ci + rr.x * cj, rflags = P.recv_flags[cell];
    } else if (P.recv_flags && hit && h.id >= 0) {
      // a sphere: the cell of the direction centre -> p in the sphere's cube map (face = largest component)
      const uint2 sr = vload<uint2>(sc, sc.off_srecv + (uint32_t)h.id * 8u);  // {Rs, first cell}
      const float4 sp = vload<float4>(sc, sc.off_spheres + (uint32_t)h.id * 16u);
      const V3 dd = sf.p - mk(sp.x, sp.y, sp.z);
      const float ax = fabsf(dd.x), ay = fabsf(dd.y), az = fabsf(dd.z);
      const bool mx = ax >= ay && ax >= az, my = !mx && ay >= az;
      const float dm = mx ? dd.x : (my ? dd.y : dd.z), du = mx ? dd.y : (my ? dd.z : dd.x), dv = mx ? dd.z : (my ? dd.x : dd.y);
      const uint32_t face = (mx ? 0u : (my ? 2u : 4u)) + (dm < 0.0f ? 1u : 0u);
      const float inv = __builtin_amdgcn_rcpf(fmaxf(fabsf(dm), 1e-30f)), Rf = (float)sr.x;
      const uint32_t ci = (uint32_t)fminf(fmaxf((__builtin_fmaf(du, inv, 1.0f)) * 0.5f * Rf, 0.0f), Rf - 1.0f);
      const uint32_t cj = (uint32_t)fminf(fmaxf((__builtin_fmaf(dv, inv, 1.0f)) * 0.5f * Rf, 0.0f), Rf - 1.0f);
      if (sr.x != 0u) cell = sr.y + (face * sr.x + cj) * sr.x + ci, rflags = P.recv_flags[cell];
    }
    stash[threadIdx.x + 10 * 256] = __uint_as_float(rflags);
    stash[threadIdx.x + RT_STASH_CELL * 256] = __uint_as_float(cell);
  }
  const Mat m_lit = load_mat(sc, sf.mat);
  // the material row is simply read again after the loop (L2)
  stash[threadIdx.x + 6 * 256] = __uint_as_float(sf.mat | (m_lit.transmissive ? RT_MAT_TRANS_BIT : 0u));
  const V3 mcolor = m_lit.color;
  const float mshin = m_lit.shininess;

  // ---- calculate_lighting, raytracer_renderer.rs:731-874 ----------------------------------------
  V3 light_color = mk(0, 0, 0), spec_color = mk(0, 0, 0);
  const bool has_spec = mshin > 0.0f;
  // first hit point of the wavefront and how far the others are from it (sphere pre-selection of the candidate collection)
  V3 p_first = mk(0, 0, 0);
  float p_spread = 0.0f;
  if (N > 1 && sc.n_spheres) {
    const int fl = __ffsll((long long)hit_m) - 1;
    p_first = mk(__uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(sf.p.x), fl)),
                 __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(sf.p.y), fl)),
                 __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(sf.p.z), fl)));
    const V3 dp = sf.p - p_first;
    float m = hit ? (fabsf(dp.x) + fabsf(dp.y) + fabsf(dp.z)) : 0.0f;  // 1-norm >= distance
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    p_spread = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(m))) * 1.000001f;
  }
  // every hit point casts lights x N shadow rays in the reference (raytracer.rs:24); counted here once, whatever
  // part of them the classifications below resolve without a traversal
  wv.cnt_shadow += sc.n_lights * N * (P.weighted ? wave_sum(hit ? r.mult : 0u) : (uint32_t)__popcll(hit_m));
  for (uint32_t l = 0; l < sc.n_lights; l++) {
    const float4 L0 = sload<float4>(sc, sc.off_lights + l * 32u);
    const float4 L1 = sload<float4>(sc, sc.off_lights + l * 32u + 16u);
    const V3 lc = mk(L1.x, L1.y, L1.z);
    const float4* cs = nullptr;
    float lI = L0.w;
    if (N > 1) {
      uint32_t tix = threadIdx.x;
      RT_OPAQUE(tix);  // a real LDS read per light instead of a register held through the loop
      const uint32_t pixel = __float_as_uint(stash[9u * 256u + tix]);
      const uint32_t hsh = rt_cloud_hash(P.cloud_seed, pixel, l);
      // (a power-of-two table -- the default 1024 -- needs no integer division: ~25 vector instructions per light)
      const uint32_t set = (P.n_cloud_sets & (P.n_cloud_sets - 1u)) == 0u ? (hsh & (P.n_cloud_sets - 1u)) : (hsh % P.n_cloud_sets);
      cs = P.cloud_sets + (size_t)set * N;
      lI = (1.0f / (float)N) * L0.w;
    }
    // Lights below the horizon of the hit point: the contribution is gated by diff = max(n.ld, 0) > 0
    // (light.rs / :800-802), so a lane whose every sample direction has n.ld_j <= 0 adds nothing for this
    // light whatever its shadow rays would find -- they are not traced (they still count in rays_shadow).
    // n.D_j <= n.dc + delta |n|_1 over the cloud; the margin covers the fp32 rounding of n.ld_j.
    bool use = hit;
    {
      const float dl = (N > 1) ? P.cloud_delta : 0.0f;
      V3 dc = mk(L0.x, L0.y, L0.z) - sf.p;
      if (N > 1) dc = mk(L0.x + P.cloud_centre[0], L0.y + P.cloud_centre[1], L0.z + P.cloud_centre[2]) - sf.p;
      const float n1 = fabsf(sf.n.x) + fabsf(sf.n.y) + fabsf(sf.n.z);
      const float scale = fabsf(dc.x) + fabsf(dc.y) + fabsf(dc.z) + fabsf(sf.p.x) + fabsf(sf.p.y) + fabsf(sf.p.z) + 1.0f;
      use = hit && (dot(sf.n, dc) + n1 * __builtin_fmaf(4e-6f, scale, dl) > 0.0f);
    }
    lanemask use_m = wave_ballot(use);
    if (!use_m) {
      continue;
    }
    CandList cand;
    cand.reg = 0;
    cand.count = RT_CAND_OVERFLOW;
    cand.spheres = 0xFFFFFFFFu;
    cand.umbra = 0ull;
#if RT_PROFILE == 4
    uint32_t prof_walked = 0, prof_listed = 0, prof_pre = 0;
#endif
    if ((RT_SKIP & 4) && N > 1) {  // removal ablation: no candidate collection either
      cand.count = 0;
      cand.spheres = 0;
    } else if (N > 1 && P.cloud_delta > 0.0f && P.traversal == RT_TRAVERSAL_BVH && sc.n_triangles) {
      V3 centre = mk(L0.x + P.cloud_centre[0], L0.y + P.cloud_centre[1], L0.z + P.cloud_centre[2]);
      const unsigned long long t_c = PROF_T();
      bool walk_tris = true, test_spheres = true;
      if (!CULL && P.recv_flags) {
        uint32_t tix = threadIdx.x;
        RT_OPAQUE(tix);
        const uint32_t rf = __float_as_uint(stash[10u * 256u + tix]) >> l;
        walk_tris = (use_m & ~wave_ballot(rf & 1u)) != 0ull;
        test_spheres = walk_tris || (use_m & ~wave_ballot((rf >> 8) & 1u)) != 0ull;
      }
      // Per-cell candidate lists: when every lane in use sits in a receiver cell whose list for this light is complete,
      // their union replaces the BVH walk (typically one to three distinct cells per wavefront: lanes of one cell share a list).
      uint32_t pre_reg = 0, pre_count = 0;  // lane i of pre_reg = i-th slot of the union
      bool have_pre = false;
      if (!CULL && P.cell_lists && walk_tris) {
        uint32_t tix = threadIdx.x;
        RT_OPAQUE(tix);
        const uint32_t cidx = __float_as_uint(stash[RT_STASH_CELL * 256u + tix]);
        uint4 lst = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        if (use && cidx != RT_NO_CELL) lst = P.cell_lists[(size_t)cidx * sc.n_lights + l];
        const lanemask unusable = use_m & (wave_ballot(cidx == RT_NO_CELL) | wave_ballot((lst.x & 0xFFFFu) == RT_CELL_LIST_OVERFLOW));
        if (!unusable) {
          have_pre = true;
          const uint32_t lane_id = threadIdx.x & 63u;
          for (lanemask todo = use_m & ~wave_ballot((lst.x & 0xFFFFu) == RT_CELL_LIST_END); todo;) {
            const int fl = __ffsll((long long)todo) - 1;
            const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cidx, fl);
            auto add_slot = [&](uint32_t slot) {  // (uniform)
              if (slot >= RT_CELL_LIST_OVERFLOW) return;  // end marker
              const lanemask filled = pre_count >= 64u ? ~0ull : ((1ull << pre_count) - 1ull);
              if (wave_ballot(pre_reg == slot) & filled) return;  // another cell listed it already
              lane_put(pre_reg, pre_count, slot, lane_id);
              pre_count++;
            };
            auto add_pair = [&](uint32_t v) {
              const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)v, fl);
              add_slot(w & 0xFFFFu);
              add_slot(w >> 16);
            };
            add_pair(lst.x), add_pair(lst.y), add_pair(lst.z), add_pair(lst.w);
            todo &= ~wave_ballot(cidx == c);
          }
          if (pre_count > 64u) have_pre = false;  // (more than a VGPR's worth of lanes: walk)
        }
      }
#if RT_PROFILE == 4
      prof_walked = (walk_tris && !have_pre) ? 1u : 0u, prof_listed = have_pre ? 1u : 0u, prof_pre = have_pre ? pre_count : 0u;
#endif
      if (test_spheres) {
        cand = collect_light_candidates<CULL>(sc, W, use, sf.p, centre, P, p_first, p_spread, P.cand_cap, nullptr, walk_tris,
                                              have_pre, pre_reg, pre_count);
      } else {
        cand.count = 0;  // every lane's cell is clear of triangles and spheres for this light: nothing to test
        cand.spheres = 0;
      }
#if RT_PROFILE
      RT_OPAQUE(cand.reg);
#endif
      PROF_ADD(W, 1, t_c);
#if RT_SKIP  // removal ablation (timing only, wrong image): 1 no shadow sphere tests, 2 no shadow triangle tests,
            // 8 sets whose shared list overflows are treated as having nothing to test
      if (RT_SKIP & 1) cand.spheres = 0;
      if (RT_SKIP & 2) cand.count = 0;
      if ((RT_SKIP & 8) && cand.count == RT_CAND_OVERFLOW) cand.count = 0, cand.spheres = 0;
#endif
#if RT_PROFILE == 2  // histogram of (wavefront, light) candidate sets instead of timers
      {
        const uint32_t nsph = (uint32_t)__popc(cand.spheres & ((1u << (sc.n_spheres < 32u ? sc.n_spheres : 31u)) - 1u));
        const bool ov = cand.count == RT_CAND_OVERFLOW;
        W.prof[0] += (!ov && cand.count == 0 && nsph == 0) ? 1 : 0;   // nothing to test at all
        W.prof[1] += (!ov && cand.count == 0 && nsph > 0) ? 1 : 0;    // spheres only
        W.prof[2] += (!ov && cand.count >= 1 && cand.count <= 4) ? 1 : 0;
        W.prof[3] += (!ov && cand.count >= 5 && cand.count <= 16) ? 1 : 0;
        W.prof[4] += (!ov && cand.count > 16) ? 1 : 0;
        W.prof[5] += ov ? 1 : 0;
        W.prof[6] += nsph;
      }
#endif
    }
    if (cand.umbra) {
      // lanes in the full shadow of an opaque triangle are done with this light
      use_m &= ~cand.umbra;
      use = lane_of(use_m);
      if (!use_m) {
          continue;
      }
    }
    if (STREAM && cand.count == RT_CAND_OVERFLOW && P.hard_q && N > 1 && N <= 64u && P.traversal == RT_TRAVERSAL_BVH && sc.n_triangles &&
        P.cand_cap != 0u) {
      // The shared list overflowed: 64 unrelated hit points (an incoherent wavefront).  One wave-cooperative walk per
      // sample over the union of what 64 unrelated rays reach costs ~1e5 instructions (0.3 % of config 4's sets were
      // 26 % of its frame time).  The lanes' (hit point, light) pairs go to rt_hard_kernel instead, which spreads
      // the N samples of a pair over N lanes; their contribution reaches the pixel through the accumulator.
      uint32_t tix = threadIdx.x;
      RT_OPAQUE(tix);
      const float* st = stash + tix;
      const float a0 = st[3 * 256];
      int hd, hk;
      uint32_t hmult;
      unpack_dkm(__float_as_int(st[5 * 256]), hd, hk, hmult);
      hard_push(P, use_m, sf.p, sf.n, d, __float_as_uint(st[6 * 256]) & ~RT_MAT_TRANS_BIT, l, __float_as_uint(st[9 * 256]),
                mk(st[0 * 256] * a0, st[1 * 256] * a0, st[2 * 256] * a0), hmult);
      continue;
    }
    const bool nothing = cand.count == 0 && cand.spheres == 0;  // wave-uniform
#if RT_PROFILE == 5  // what lane regrouping could save in the sets that are traced with a shared list
    if (!nothing && cand.count != RT_CAND_OVERFLOW && N > 1) {
      const uint32_t nl = (uint32_t)__popcll(use_m);
      W.prof[0] += 1;                                   // LIST sets
      W.prof[1] += nl;                                  // lanes in them
      W.prof[2] += nl * cand.count;                     // (lane, candidate) pairs the sample loops test: every lane against the whole list
      W.prof[3] += wave_sum(use ? cand.own : 0u);       // ... pairs that survive the lane's OWN beam test
      W.prof[4] += (uint32_t)__popcll(use_m & wave_ballot(cand.own != 0u));   // lanes with a candidate of their own
      W.prof[5] += (uint32_t)__popc(cand.spheres & ((1u << (sc.n_spheres < 32u ? sc.n_spheres : 31u)) - 1u)) * nl;  // (lane, sphere) pairs
      W.prof[6] += cand.count;                          // list lengths
    }
#endif
#if RT_PROFILE == 4  // how full the wavefronts are in each class of (wavefront, light) set (what lane compaction could gain)
    {
      const uint32_t nl = (uint32_t)__popcll(use_m);
      uint32_t needy = nl;
      if (!CULL && P.recv_flags && N > 1) {
        uint32_t tix = threadIdx.x;
        RT_OPAQUE(tix);
        const uint32_t rf = __float_as_uint(stash[10u * 256u + tix]) >> l;
        needy = (uint32_t)__popcll(use_m & ~wave_ballot((rf & 1u) && ((rf >> 8) & 1u)));
      }
      if (nothing) {
        W.prof[0] += 1, W.prof[1] += nl;
      } else {
        W.prof[2] += 1, W.prof[3] += nl;
        (void)needy;
      }
      W.prof[4] += prof_walked;   // sets whose candidates came from a BVH walk
      W.prof[5] += prof_listed;   // ... from the union of per-cell lists
      W.prof[6] += prof_pre;      // slots in those unions (before the lanes' own beam tests)
    }
#endif
#if RT_PROFILE == 3
    uint32_t set_occ = 0, set_tot = 0, set_filt = 0;
#endif
    // PointLight::calculate_contribution_at, light.rs:261-299, for the lanes in reach_m.  FILTERED = false: the
    // shadow ray is known to arrive untouched (opacity 1, filter 1): no filter divisions.
    // Everything here only scales the colour (tolerance 1e-4, measured < 4e-6).  Two exact identities of the
    // reference are used to drop work: `cosi = (ltp . n) / (|ltp| + EPS)` is `diff = n . ld` up to a factor
    // 1 + EPS/|ltp| (ld = ltp / |ltp|), and its `cosi > 0` selects are implied by the `diff > 0` gate of the sum.
    const V3 mc_lc = mcolor * lc;  // per light
    const V3 mmc_lc = mcolor * mc_lc;  // (the reference multiplies the surface colour in twice: contribution colour x surface colour)
    // this light's share of `direct` and `specular`, summed over its samples in sample order from zero (the same chain
    // rt_hard_kernel runs for a deferred pair)
    // (without secondary rays nothing needs the per-light split: the chain simply runs on through all lights, in the
    // running sums themselves -- six registers less in the sample loops of the kernel that has no hard route)
    V3 dl_own = mk(0.0f, 0.0f, 0.0f), ds_own = mk(0.0f, 0.0f, 0.0f);
    V3& dl = STREAM ? dl_own : light_color;
    V3& ds = STREAM ? ds_own : spec_color;
    auto add_light = [&](auto filtered_tag, V3 ltp, const Shadow& S, lanemask reach_m) {
      constexpr bool FILTERED = decltype(filtered_tag)::value;
      const unsigned long long t_l = PROF_T();
      const LightTerms T = light_sample_terms<FILTERED>(sf.n, d, mmc_lc, lI, mshin, has_spec, ltp, S);
      if (lane_of(reach_m) && T.lit) {
        // (fused accumulation: colour-only, one rounding less than the reference's multiply + add)
        dl = fma_s(T.mLc, T.lf, dl);
        if (has_spec) ds = fma_s(lc, T.sf, ds);
      }
#if RT_PROFILE
      RT_OPAQUE(dl.x);
#endif
      PROF_ADD(W, 5, t_l);
    };
    // the cloud offsets of sample j+1 are fetched (per-lane gather, L2) before sample j is traced, so
    // the load latency hides under a whole shadow traversal
    // (every lane holds a valid set -- the pixel it hashes comes from the stash, written for idle lanes too -- so the loads
    // are unconditional: no exec-mask bracket and no register shuffle between the current and the next offsets)
    float4 cnext = make_float4(0, 0, 0, 0);
    if (N > 1 && (!STREAM || use)) cnext = cs[0];
    auto light_position = [&](uint32_t j) {
      V3 lp = mk(L0.x, L0.y, L0.z);
      if (N > 1) {
        lp.x = L0.x + cnext.x;  // light.rs:218; the table holds offset * (fw, fh, fd)
        lp.y = L0.y + cnext.y;
        lp.z = L0.z + cnext.z;
        if (!STREAM)
          cnext = cs[j + 1 < N ? j + 1 : j];
        else if (use && j + 1 < N)  // (the streaming kernels keep the guarded form: the unconditional one costs them scratch)
          cnext = cs[j + 1];
      }
      return lp;
    };
    if (nothing) {
      // no triangle and no sphere can touch any sample ray of this (wavefront, light): the shadow rays are
      // known to arrive, so their origin / length / re-normalised direction (3 IEEE sqrt, 2 IEEE div per
      // sample) are not needed; what is left of ld and |ltp| only scales the colour
      Shadow S;
      shadow_init(S);
      for (uint32_t j = 0; j < N; j++) {
        const V3 ltp = light_position(j) - sf.p;
        WSTAT(W.s_passes++);
        add_light(std::false_type{}, ltp, S, use_m);
      }
    } else {
      auto traced_samples = [&](auto list_tag) {
        for (uint32_t j = 0; j < N; j++) {
#if RT_PROFILE
          W.t_mark = PROF_T();
#endif
          const V3 lp = light_position(j);
          const V3 ltp = lp - sf.p;
          const V3 ld = ltp * exact_rcp(mag(ltp));  // normalize(ltp): the shadow ray's geometry is exact
          const V3 so = sf.p + ld * epsv;
          const float tmax = mag(lp - so);
          const Shadow S = shadow_ray<CULL, decltype(list_tag)::value>(sc, P, W, use_m, so, ld, tmax, cand);
          const lanemask reach_m = use_m & ~S.occ;
#if RT_PROFILE == 3
          set_occ += (uint32_t)__popcll(use_m & S.occ);
          set_tot += (uint32_t)__popcll(use_m);
          set_filt += (uint32_t)__popcll(reach_m & wave_ballot(S.dec != 0u));
#endif
          if (!reach_m) continue;
          add_light(std::true_type{}, ltp, S, reach_m);
        }
      };
      // (decided once per light: per-sample uniform branches cost issue slots and mask registers)
      if (cand.count != RT_CAND_OVERFLOW && P.traversal != RT_TRAVERSAL_LINEAR && sc.n_triangles)
        traced_samples(std::true_type{});
      else
        traced_samples(std::false_type{});
    }
    // this light's share of the node's own terms (:251-257: transmissive ? specular : direct + specular)
    if (STREAM) {
      // Streaming kernels: every (hit point, light) share reaches the pixel as its own fixed-point term W * atten * own_l --
      // the term rt_hard_kernel adds for a pair that was deferred to it -- so the pixel sum is the same integer
      // whichever route the pair took.
      uint32_t tix = threadIdx.x;
      RT_OPAQUE(tix);
      float* st = stash + tix;
      const float a0 = st[3 * 256];
      const V3 Wa = mk(st[0 * 256] * a0, st[1 * 256] * a0, st[2 * 256] * a0);
      const bool Tm = (__float_as_uint(st[6 * 256]) & RT_MAT_TRANS_BIT) != 0u;
      const V3 c = Wa * (Tm ? ds : (dl + ds));
      long long* fx = stash_fix(stash) + tix;
      fx[0] += __float2ll_rn(c.x * RT_ACC_SCALE);
      fx[256] += __float2ll_rn(c.y * RT_ACC_SCALE);
      fx[512] += __float2ll_rn(c.z * RT_ACC_SCALE);
    }
#if RT_PROFILE == 3  // outcome of the (wavefront, light) sets that had something to test
    if (!nothing && set_tot) {
      const bool tri = cand.count != 0;
      const int cls = set_occ == 0 ? (set_filt ? 2 : 0) : (set_occ == set_tot ? 1 : 2);  // lit / umbra / mixed
      W.prof[(tri ? 0 : 3) + cls] += 1;
      W.prof[6] += 1;
    }
#endif
  }
  PROF_ADD(W, 6, t_all);
  // ---- back from LDS -------------------------------------------------------------------------------
  V3 Wt;
  float a, n_start;
  int depth, kind;
  uint32_t pixel;
  Mat m;
  {
    // (opaque index: otherwise hipcc forwards the stored values to these loads, i.e. keeps them in
    // registers or scratch through the whole light loop, which is what the stash is there to avoid)
    uint32_t tix = threadIdx.x;
    RT_OPAQUE(tix);
    const float* st = stash + tix;
    Wt = mk(st[0 * 256], st[1 * 256], st[2 * 256]);
    a = st[3 * 256];
    n_start = st[4 * 256];
    unpack_dkm(__float_as_int(st[5 * 256]), depth, kind, out.mult);
    uint32_t mat_row = __float_as_uint(st[6 * 256]) & ~RT_MAT_TRANS_BIT;
    RT_OPAQUE(mat_row);  // keeps hipcc from carrying the row's addresses through the loop (in scratch)
    m = load_mat(sc, mat_row);
    out.t = st[7 * 256];
    out.id = __float_as_int(st[8 * 256]);
    pixel = __float_as_uint(st[9 * 256]);
    out.pix = pixel;
  }
  (void)kind;
  V3 ambient = (m.color * mk(1.0f, 1.0f, 1.0f)) * P.ambient;
  // own terms of this node (:251-257): transmissive ? spec : direct + spec
  const bool T = m.transmissive;
  if (STREAM) {
    // the lights' shares are already in the fixed-point sums; the ambient term (part of `direct`) joins them
    if (hit && !T) {
      const V3 c = mk(Wt.x * a, Wt.y * a, Wt.z * a) * ambient;
      uint32_t tix = threadIdx.x;
      RT_OPAQUE(tix);
      long long* fx = stash_fix(stash) + tix;
      fx[0] += __float2ll_rn(c.x * RT_ACC_SCALE);
      fx[256] += __float2ll_rn(c.y * RT_ACC_SCALE);
      fx[512] += __float2ll_rn(c.z * RT_ACC_SCALE);
    }
  } else if (hit) {
    V3 direct = (ambient + light_color) * a;  // :206-209
    V3 spec = spec_color * a;
    V3 own = T ? spec : (direct + spec);
    out.contrib = Wt * own;
  }
  if (!stream) return out;

  // ---- calculate_reflection, :526-729 -------------------------------------------------------------
  {
    bool spawn = false;
    V3 co = mk(0, 0, 0), cd = mk(0, 0, 1), cW = mk(0, 0, 0);
    int cdepth = 0;
    const bool R = (m.metallic > 0.0f) || T;
    if (hit && (P.flags & RT_FLAG_REFLECTIONS) && R) {
      float cos_theta = dot(d, sf.n);
      bool inside = cos_theta < 0.0f;
      V3 inormal = inside ? -sf.n : sf.n;
      float n2 = inside ? m.ior : P.air_ior;
      float eta = inside ? (n2 / n_start) : (n_start / n2);
      float cos_i = fabsf(cos_theta);
      float sin2 = eta * eta * (1.0f - cos_i * cos_i);
      bool tir = sin2 >= 1.0f;
      bool reflective = (m.metallic > 0.0f) || (T && tir);
      cdepth = depth < 0 ? (int)P.max_depth_reflection : (depth > 0 ? depth - 1 : 0);
      if (reflective && cdepth > 0) {
        V3 rr = normalize(reflected(d, sf.n));
        V3 Rf = fresnel_reflectance(m, inormal, -d, n_start);
        spawn = true;
        co = sf.p + rr * epsv;
        cd = rr;
        cW = Wt * Rf;
      }
    }
    queue_push(P, spawn, co, cd, n_start, cW, cdepth, KIND_REFL, pixel, out.mult);
  }
  // ---- calculate_refractions, :279-524 --------------------------------------------------------------
  {
    bool spawn = false;
    V3 co = mk(0, 0, 0), cd = mk(0, 0, 1), cW = mk(0, 0, 0);
    int cdepth = 0;
    float cior = 0.0f;
    if (hit && (P.flags & RT_FLAG_REFRACTIONS) && T) {
      float cos_theta = dot(d, sf.n);
      bool inside = cos_theta <= 0.0f;
      V3 inormal = inside ? -sf.n : sf.n;
      float n2 = inside ? m.ior : P.air_ior;
      float eta = inside ? (n2 / n_start) : (n_start / n2);
      float inv_eta = 1.0f / eta;
      V3 Rf = fresnel_reflectance(m, inormal, d, inv_eta);
      V3 Tr = mk(1.0f - Rf.x, 1.0f - Rf.y, 1.0f - Rf.z);
      // ultraviolet refracted(n = -inormal, eta = 1/eta)
      V3 nn = -inormal;
      float ndi = dot(nn, d);
      float kk = 1.0f - inv_eta * inv_eta * (1.0f - ndi * ndi);
      float op = m.opacity;
      int step = (op < 0.5f) ? 2 : 1;
      int fac = (op <= 0.3f) ? 3 : ((op < 0.5f) ? 2 : 1);
      cdepth = depth < 0 ? (int)P.max_depth_refraction / fac : (depth > step ? depth - step : 0);
      if (!(kk < 0.0f) && cdepth > 0) {  // kk < 0: zero vector -> NaN direction -> miss (deviation D2)
        float s = inv_eta * ndi + __builtin_sqrtf(kk);
        V3 q = normalize(d * inv_eta - nn * s);
        spawn = true;
        co = sf.p + q * epsv;
        cd = q;
        cW = (Wt * (m.boost + 1.0f)) * Tr;
        cior = n2;
      }
    }
    queue_push(P, spawn, co, cd, cior, cW, cdepth, KIND_REFR, pixel, out.mult);
  }
  return out;
}

// where the packed pixel (gx, gy) is stored: the W x H frame, or this rank's compact tile staging (multi-GPU gather)
__device__ __forceinline__ uint32_t out_index(const RtDevParams& P, uint32_t gx, uint32_t gy, uint32_t pix) {
  if (!P.stage_slot) return pix;  // wave-uniform (kernel argument)
  const uint32_t ts = P.tile_size, tx = gx / ts, ty = gy / ts;
  return P.stage_slot[ty * P.stage_tiles_x + tx] * ts * ts + (gy - ty * ts) * ts + (gx - tx * ts);
}

__device__ __forceinline__ uint32_t pack_argb(V3 c) {
  return 0xFF000000u | (to_u8(c.x) << 16) | (to_u8(c.y) << 8) | to_u8(c.z);
}

// `mult` identical rays of the reference add `mult` identical fixed-point terms: exact in the integer domain
__device__ __forceinline__ void acc_add(const RtDevParams& P, uint32_t pix, V3 c, uint32_t mult) {
  long long* a = P.acc + 4 * (size_t)pix;
  atomicAdd((unsigned long long*)&a[0], (unsigned long long)(__float2ll_rn(c.x * RT_ACC_SCALE) * (long long)mult));
  atomicAdd((unsigned long long*)&a[1], (unsigned long long)(__float2ll_rn(c.y * RT_ACC_SCALE) * (long long)mult));
  atomicAdd((unsigned long long*)&a[2], (unsigned long long)(__float2ll_rn(c.z * RT_ACC_SCALE) * (long long)mult));
}

__device__ __forceinline__ void acc_add_fixed(const RtDevParams& P, uint32_t pix, long long x, long long y, long long z, uint32_t mult) {
  long long* a = P.acc + 4 * (size_t)pix;
  if (x) atomicAdd((unsigned long long*)&a[0], (unsigned long long)(x * (long long)mult));
  if (y) atomicAdd((unsigned long long*)&a[1], (unsigned long long)(y * (long long)mult));
  if (z) atomicAdd((unsigned long long*)&a[2], (unsigned long long)(z * (long long)mult));
}

// ------------------------------------------------------------------------------------------------
// primary kernel: one thread per (pixel, DISTINCT AA sample).  The reference's sample table repeats itself
// (raytracer_renderer.rs:107-122: [0,0], then [1,1]s; the 8 directions restart in every 8-lane chunk, :1111-1116):
// 9 distinct origins among 16 samples, 9 among 24 with extreme_quality.  Repeats give bit-identical rays and
// colours (the light cloud is chosen per pixel), so every distinct sample is traced once (n_thr threads per
// pixel) and its colour is read n times by the per-pixel sum; its children carry the multiplicity.
// A 256-thread workgroup owns ppw = 256 / n_thr consecutive pixels in "tile order" (16x16 super-tiles of
// 4x4 tiles; a workgroup may straddle two super-tiles), so a wavefront's 64 camera rays are the samples of a
// few adjacent pixels -- maximally coherent for the wave-cooperative traversal.  Sample colours meet in LDS and
// are summed per pixel in the reference's lane/packet order (antialiased_raytrace, raytracer_renderer.rs:918-1016).
// ------------------------------------------------------------------------------------------------
// COST: the calibration variant of RT_TILE_ORDER_COST (wavefront run times per super-tile); a kernel of its own so that
// the shipped kernels carry none of it.
// PRE (merged levels): the camera rays' hits were found -- and their children appended -- by rt_hit_spawn_kernel; this launch only shades
// them, concurrently with the trace launches of the levels below on another stream.
template <bool CULL, bool STREAM, bool COST = false, bool PRE = false>
__device__ __forceinline__ void primary_body(const RtDevScene& sc, const RtDevParams& P, float4* lds_rgbh,
                                             float* lds_stash, unsigned long long* lds_cnt) {
  Wave wv;
  wave_init(wv);
  wave_flush_init(P, lds_cnt);
  const bool aa = (P.flags & RT_FLAG_ANTI_ALIASING) && P.aa_rays > 0;
  const uint32_t n_samples = aa ? P.aa_rays : 1u;  // samples of the reference's per-pixel sum
  const uint32_t n_thr = aa ? P.aa_unique : 1u;    // threads per pixel (distinct samples)
  // thread -> (pixel, AA sample): workgroup -> 16x16 super-tile (through the list of super-tiles this rank
  // owns, if any) -> 4x4 tile -> pixel.  Evaluated twice (before the ray and again for the accumulation) so
  // that nothing of it has to survive the light loops in registers.
  struct PixelMap {
    uint32_t slot, k, gx, gy, pix;
    bool on;
  };
  // The samples of a pixel sit in ONE wavefront whenever that wastes at most 4 lanes (9 distinct samples: 7 pixels on 63
  // lanes): the per-pixel sum then needs no workgroup barrier, and a wavefront that is done leaves the CU without waiting
  // for the slowest of its workgroup (rt_primary_wave_local is the host's copy of this rule).
  const bool wave_local = rt_primary_wave_local(n_thr);
  const uint32_t ppwave = wave_local ? 64u / n_thr : 0u;
  auto map_thread = [&](uint32_t tid) {
    PixelMap m;
    const uint32_t ppw = wave_local ? 4u * ppwave : 256u / n_thr;  // pixels per workgroup (host guarantees n_thr <= 256)
    bool slot_used;
    if (wave_local) {
      const uint32_t ln = tid & 63u, in_wave = ln / n_thr;
      m.k = ln - in_wave * n_thr;
      m.slot = (tid >> 6) * ppwave + in_wave;
      slot_used = in_wave < ppwave;
    } else {
      m.slot = tid / n_thr;
      m.k = tid - m.slot * n_thr;
      slot_used = m.slot < ppw;
    }
    const uint32_t st_x = (P.win_w + 15u) / 16u;
    const uint32_t wg = rt_batch_wg(P, blockIdx.x);
    const uint32_t g = wg * ppw + m.slot;  // pixel ordinal in super-tile order
    const uint32_t sup_slot = g >> 8, in_sup = g & 255u;
    const bool lane_used = slot_used && (sup_slot < P.n_sup);
    uint32_t sup = sup_slot;
    if (P.sup_list) sup = lane_used ? P.sup_list[sup_slot] : 0u;
    const uint32_t t4 = (in_sup >> 4) & 15u, p4 = in_sup & 15u;
    const uint32_t lx = (sup % st_x) * 16u + (t4 & 3u) * 4u + (p4 & 3u);
    const uint32_t ly = (sup / st_x) * 16u + (t4 >> 2) * 4u + (p4 >> 2);
    m.gx = P.win_x0 + lx, m.gy = P.win_y0 + ly;
    m.on = lane_used && (lx < P.win_w) && (ly < P.win_h);
    if (m.on && P.n_ranks > 1) m.on = rt_tile_owner(m.gx / P.tile_size, m.gy / P.tile_size, P.n_ranks) == P.rank;
    m.pix = m.gy * P.width + m.gx;
    return m;
  };
  const PixelMap pm = map_thread(threadIdx.x);
  const bool pix_on = pm.on;
  // cost calibration (RT_TILE_ORDER_COST): the wavefront's start time waits in LDS, not in SGPRs
  if (COST && P.cost_map && (threadIdx.x & 63u) == 0) lds_cnt[16 + (threadIdx.x >> 6)] = __builtin_readcyclecounter();
  const uint32_t gx = pm.gx, gy = pm.gy, k = pm.k;
  const float x = (float)gx * P.fw;  // renderer/mod.rs:176-180
  const float y = (float)gy * P.fh;
  const V3 coords = mk(x, y, 0.0f);
  RayIn r;
  r.o = coords;
  if (aa && pix_on) {
    r.o.x = coords.x + P.aa_offsets[2 * k];
    r.o.y = coords.y + P.aa_offsets[2 * k + 1];
  }
  r.d_raw = coords - mk(P.focus[0], P.focus[1], P.focus[2]);  // un-jittered for every sample (:1204)
  r.n_start = P.air_ior;
  const float scale = aa ? 1.0f / (float)(((n_samples + 7u) / 8u) * 8u) : 1.0f;  // :936-937
  r.Wt = mk(scale, scale, scale);  // the sample's 1/total_rays weight rides along the whole tree
  r.depth = -1;
  r.kind = KIND_PRIMARY;
  r.pix = pm.pix;
  r.mult = (P.weighted && pix_on) ? P.aa_mult[k] : 1u;

  Hit none;
  none.t = INFINITY;
  none.id = -1;
  if (PRE) {
    const uint2 hr = P.hitrec[(size_t)blockIdx.x * 256u + threadIdx.x];
    none.t = __uint_as_float(hr.x);
    none.id = (int)hr.y;
  }
  RayOut out = process_ray<CULL, PRE, STREAM>(sc, P, wv, pix_on, r, lds_stash, none);

  // ---- per-pixel accumulation of the samples ----------------------------------------------------------
  V3 cs = out.contrib;  // = own * scale (c * scale, :974,:992)
  lds_rgbh[threadIdx.x] = make_float4(cs.x, cs.y, cs.z, out.hit ? 1.0f : 0.0f);
  uint32_t tid2 = threadIdx.x;
  RT_OPAQUE(tid2);  // keeps hipcc from carrying the first mapping through process_ray
  const PixelMap pm2 = map_thread(tid2);
  const uint32_t pix = pm2.pix;
  if (pm2.k == 0 && pm2.on) {
    if (P.aux_hit_id) P.aux_hit_id[pix] = out.id;
    if (P.aux_hit_t && out.id >= 0) P.aux_hit_t[pix] = out.t;
  }
  // the samples of a pixel are the threads tid2 .. tid2 + n_thr - 1 (tid2 = the pixel's thread with k == 0)
  if (wave_local) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
  bool wrote = false;
  if (STREAM && pm2.k == 0 && pm2.on) {
    // Secondary rays are streaming: the pixel is resolved by rt_resolve_kernel from the fixed-point accumulator.  The
    // samples' own terms are already integers (process_ray); a sample the reference casts m times adds m x its sum --
    // exact, whatever the order, so tracing each distinct sample once changes no bit of the frame.
    const float4* s = lds_rgbh + tid2;
    const long long* fx = stash_fix(lds_stash) + tid2;
    long long sx = 0, sy = 0, sz = 0;
    bool any = false;
    for (uint32_t u = 0; u < n_thr; u++) {
      if (s[u].w != 0.0f) {
        any = true;
        const long long mu = P.weighted ? (long long)uload(&P.aa_mult[u]) : 1ll;
        sx += fx[u] * mu, sy += fx[256u + u] * mu, sz += fx[512u + u] * mu;
      }
    }
    if (any) {
      wrote = true;
      acc_add_fixed(P, pix, sx, sy, sz, 1u);
      P.acc[4 * (size_t)pix + 3] = 1;
    }
  }
  if (!STREAM && pm2.k == 0 && pm2.on) {
    const float4* s = lds_rgbh + tid2;
    // sample q of the reference's sum -> the thread that traced it (wave-uniform q: scalar load)
    auto src = [&](uint32_t q) { return P.weighted ? uload(&P.aa_src[q]) : q; };
    V3 color;
    bool any = false;
    if (!aa) {
      color = mk(s[0].x, s[0].y, s[0].z);
      any = s[0].w != 0.0f;
    } else {
      V3 first[8], rest[8];
#pragma unroll
      for (int l = 0; l < 8; l++) first[l] = rest[l] = mk(0, 0, 0);
#pragma unroll
      for (int l = 0; l < 8; l++) {
        if ((uint32_t)l < n_samples) {
          const float4 c = s[src((uint32_t)l)];
          if (c.w != 0.0f) {
            first[l] = mk(c.x, c.y, c.z);
            any = true;
          }
        }
      }
      for (uint32_t base = 8; base < n_samples; base += 8) {
#pragma unroll
        for (int l = 0; l < 8; l++) {
          uint32_t q = base + (uint32_t)l;
          if (q < n_samples) {
            const float4 c = s[src(q)];
            if (c.w != 0.0f) {
              rest[l] = mk(c.x, c.y, c.z) + rest[l];
              any = true;
            }
          }
        }
      }
      V3 lane8[8];
#pragma unroll
      for (int l = 0; l < 8; l++) lane8[l] = rest[l] + first[l];
      color = ((lane8[0] + lane8[4]) + (lane8[2] + lane8[6])) + ((lane8[1] + lane8[5]) + (lane8[3] + lane8[7]));
    }
    if (any) {
      wrote = true;
      P.argb[out_index(P, pm2.gx, pm2.gy, pix)] = pack_argb(color);
      if (P.aux_rgb) {
        P.aux_rgb[3 * (size_t)pix + 0] = color.x;
        P.aux_rgb[3 * (size_t)pix + 1] = color.y;
        P.aux_rgb[3 * (size_t)pix + 2] = color.z;
      }
    }
  }
  if (COST && P.cost_map && (threadIdx.x & 63u) == 0) {
    const unsigned long long dt = __builtin_readcyclecounter() - lds_cnt[16 + (threadIdx.x >> 6)];
    const uint32_t ppw = wave_local ? 4u * ppwave : 256u / n_thr;
    const uint32_t g = rt_batch_wg(P, blockIdx.x) * ppw + (wave_local ? (threadIdx.x >> 6) * ppwave : threadIdx.x / n_thr), sup_slot = g >> 8;
    if (sup_slot < P.n_sup) atomicAdd(&P.cost_map[P.sup_list ? P.sup_list[sup_slot] : sup_slot], (uint32_t)(dt >> 6));
  }
  wave_flush(wv, P, (uint32_t)__popcll(wave_ballot(wrote)), lds_cnt);
}

// Occupancy.  The kernel is LATENCY bound: a wavefront issues one instruction every ~12 cycles (dependent scalar-load ->
// vote -> branch chains of the wave-cooperative walks), so the SIMDs are fed by the number of resident waves.
// Round 1 (much more work per wave) preferred 4 waves / 108 VGPRs without scratch; measured on MI355X on this build:
//   waves/SIMD (VGPRs, scratch B/lane):  4 (108, 0)   5 (96, 60)   6 (80, 132)   7 (72, 168)   8 (64, 216)
//   config 3, ms                          7.66         6.91         6.42          7.40          8.82
//   config 4, ms                          80.2         75.9         74.0          77.9          88.1
// The 6-wave spills sit outside the innermost loops; scratch traffic stays in L2 / Infinity Cache.
#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 6
#endif
__global__ __launch_bounds__(256, RT_MIN_WAVES) void rt_primary_kernel(RtDevScene sc, RtDevParams P) {
  __shared__ float4 lds_rgbh[256];
  __shared__ __attribute__((aligned(16))) float lds_stash[RT_STASH_FIX * 256];  // (no fixed-point sums without secondary rays)
  __shared__ unsigned long long lds_cnt[20];  // 15 counters (wave_flush), [16..19]: wave start times (cost calibration)
  if (P.flags & RT_FLAG_BACKFACE_CULLING)
    primary_body<true, false>(sc, P, lds_rgbh, lds_stash, lds_cnt);
  else
    primary_body<false, false>(sc, P, lds_rgbh, lds_stash, lds_cnt);
}

// the same with secondary rays (children queued, accumulator, hard pairs): launched when reflections / refractions are on
__global__ __launch_bounds__(256, RT_MIN_WAVES) void rt_primary_stream_kernel(RtDevScene sc, RtDevParams P) {
  __shared__ float4 lds_rgbh[256];
  __shared__ __attribute__((aligned(16))) float lds_stash[RT_STASH_FIELDS * 256];
  __shared__ unsigned long long lds_cnt[20];
  if (P.flags & RT_FLAG_BACKFACE_CULLING)
    primary_body<true, true>(sc, P, lds_rgbh, lds_stash, lds_cnt);
  else
    primary_body<false, true>(sc, P, lds_rgbh, lds_stash, lds_cnt);
}

// merged levels: shading of the camera rays whose hits rt_hit_spawn_kernel found (no children here: they exist already)
__global__ __launch_bounds__(256, RT_MIN_WAVES) void rt_primary_pre_kernel(RtDevScene sc, RtDevParams P) {
  __shared__ float4 lds_rgbh[256];
  __shared__ __attribute__((aligned(16))) float lds_stash[RT_STASH_FIELDS * 256];
  __shared__ unsigned long long lds_cnt[20];
  if (P.flags & RT_FLAG_BACKFACE_CULLING)
    primary_body<true, true, false, true>(sc, P, lds_rgbh, lds_stash, lds_cnt);
  else
    primary_body<false, true, false, true>(sc, P, lds_rgbh, lds_stash, lds_cnt);
}

// calibration frames of RT_TILE_ORDER_COST (once per scene and frame shape): the same kernels + the per-super-tile timers.
// Only in builds made with `make COST=1`: the launch order it serves gains nothing once two frames are in flight,
// and its 354 spilled SGPRs do not belong in the shipped library.  Without it RT_TILE_ORDER_COST renders in row-major order
// (rt_stats.notes: RT_NOTE_TILE_ORDER_COST_OFF).
#ifndef RT_COST_KERNEL
#define RT_COST_KERNEL 0
#endif
#if RT_COST_KERNEL
__global__ __launch_bounds__(256, RT_MIN_WAVES) void rt_primary_cost_kernel(RtDevScene sc, RtDevParams P) {
  __shared__ float4 lds_rgbh[256];
  __shared__ __attribute__((aligned(16))) float lds_stash[RT_STASH_FIELDS * 256];
  __shared__ unsigned long long lds_cnt[20];
  const bool cull = (P.flags & RT_FLAG_BACKFACE_CULLING) != 0;
  if (P.acc) {
    if (cull) primary_body<true, true, true>(sc, P, lds_rgbh, lds_stash, lds_cnt);
    else primary_body<false, true, true>(sc, P, lds_rgbh, lds_stash, lds_cnt);
  } else {
    if (cull) primary_body<true, false, true>(sc, P, lds_rgbh, lds_stash, lds_cnt);
    else primary_body<false, false, true>(sc, P, lds_rgbh, lds_stash, lds_cnt);
  }
}
#endif

// ------------------------------------------------------------------------------------------------
// secondary rays (reflection / refraction children of any depth), three steps per queue chunk:
//   rt_trace_kernel   one thread per queued ray: nearest hit -> (t, id) + a 30-bit Morton key of the
//                     hit point (misses get the largest key)
//   radix sort        (key, ray index) pairs, rocPRIM device sort (rt_sort.hip)
//   rt_shade_kernel   one thread per ray IN HIT-POINT ORDER: lighting, pixel accumulation, children
// After a refraction through the glass spheres the hit points of neighbouring pixels are scattered over
// the whole scene; shading them in queue order made every wavefront's shadow packets incoherent (7x
// slower per shadow ray than primary hits).  Sorted by hit point, the 64 lanes of a wavefront shade
// neighbouring surface points whatever pixel they belong to.  The image does not depend on the order
// (integer pixel accumulation).
// ------------------------------------------------------------------------------------------------
// The children of a Whitted node: calculate_reflection (raytracer_renderer.rs:526-729) and calculate_refractions (:279-524), appended
// to the ray queue (the same op sequence as the two blocks at the end of process_ray).  W0 = the node's weight with a reflection child's
// own atten(t) already in it (:722-726); every lane of the wavefront calls this (idle lanes: hit = false).
__device__ __forceinline__ void spawn_children(const RtDevScene& sc, const RtDevParams& P, bool hit, V3 d, const Surf& sf, const Mat& m, V3 W0,
                                               float n_start, int depth, uint32_t pix, uint32_t mult) {
  const V3 epsv = mk(P.eps_distance, P.eps_distance, P.eps_distance);
  const bool T = m.transmissive;
  {
    bool spawn = false;
    V3 co = mk(0, 0, 0), cd = mk(0, 0, 1), cW = mk(0, 0, 0);
    int cdepth = 0;
    const bool R = (m.metallic > 0.0f) || T;
    if (hit && (P.flags & RT_FLAG_REFLECTIONS) && R) {
      float cos_theta = dot(d, sf.n);
      bool inside = cos_theta < 0.0f;
      V3 inormal = inside ? -sf.n : sf.n;
      float n2 = inside ? m.ior : P.air_ior;
      float eta = inside ? (n2 / n_start) : (n_start / n2);
      float cos_i = fabsf(cos_theta);
      float sin2 = eta * eta * (1.0f - cos_i * cos_i);
      bool tir = sin2 >= 1.0f;
      bool reflective = (m.metallic > 0.0f) || (T && tir);
      cdepth = depth < 0 ? (int)P.max_depth_reflection : (depth > 0 ? depth - 1 : 0);
      if (reflective && cdepth > 0) {
        V3 rr = normalize(reflected(d, sf.n));
        V3 Rf = fresnel_reflectance(m, inormal, -d, n_start);
        spawn = true;
        co = sf.p + rr * epsv;
        cd = rr;
        cW = W0 * Rf;
      }
    }
    queue_push(P, spawn, co, cd, n_start, cW, cdepth, KIND_REFL, pix, mult);
  }
  {
    bool spawn = false;
    V3 co = mk(0, 0, 0), cd = mk(0, 0, 1), cW = mk(0, 0, 0);
    int cdepth = 0;
    float cior = 0.0f;
    if (hit && (P.flags & RT_FLAG_REFRACTIONS) && T) {
      float cos_theta = dot(d, sf.n);
      bool inside = cos_theta <= 0.0f;
      V3 inormal = inside ? -sf.n : sf.n;
      float n2 = inside ? m.ior : P.air_ior;
      float eta = inside ? (n2 / n_start) : (n_start / n2);
      float inv_eta = 1.0f / eta;
      V3 Rf = fresnel_reflectance(m, inormal, d, inv_eta);
      V3 Tr = mk(1.0f - Rf.x, 1.0f - Rf.y, 1.0f - Rf.z);
      V3 nn = -inormal;  // ultraviolet refracted(n = -inormal, eta = 1/eta)
      float ndi = dot(nn, d);
      float kk = 1.0f - inv_eta * inv_eta * (1.0f - ndi * ndi);
      float op = m.opacity;
      int step = (op < 0.5f) ? 2 : 1;
      int fac = (op <= 0.3f) ? 3 : ((op < 0.5f) ? 2 : 1);
      cdepth = depth < 0 ? (int)P.max_depth_refraction / fac : (depth > step ? depth - step : 0);
      if (!(kk < 0.0f) && cdepth > 0) {  // kk < 0: zero vector -> NaN direction -> miss (deviation D2)
        float sq = inv_eta * ndi + __builtin_sqrtf(kk);
        V3 q = normalize(d * inv_eta - nn * sq);
        spawn = true;
        co = sf.p + q * epsv;
        cd = q;
        cW = (W0 * (m.boost + 1.0f)) * Tr;
        cior = n2;
      }
    }
    queue_push(P, spawn, co, cd, cior, cW, cdepth, KIND_REFR, pix, mult);
  }
}

// The same children, appended with ONE atomic per WORKGROUP (both kinds together) instead of two per wavefront: every queue slot comes
// from one counter, and returning atomics on one address retire at ~3 ns each whatever else the GPU does -- two per wavefront were
// 6.9 M atomics = 20 ms of a 4K frame's trace launches (profiles/r04_phase_split.md, r04k).  Called by all 256 threads of the workgroup
// the same number of times (two barriers); lds = 2 x 8 dwords, used alternately (`parity`) so that a wavefront that runs ahead into
// the next call does not overwrite what a slower one still reads.
__device__ __forceinline__ void spawn_children_block(const RtDevScene& sc, const RtDevParams& P, bool hit, V3 d, const Surf& sf, const Mat& m,
                                                     V3 W0, float n_start, int depth, uint32_t pix, uint32_t mult, uint32_t* lds, uint32_t parity) {
  const V3 epsv = mk(P.eps_distance, P.eps_distance, P.eps_distance);
  const bool T = m.transmissive;
  // ---- which children exist: the conditions of calculate_reflection (:526-729) and calculate_refractions (:279-524)
  bool s_refl = false, s_refr = false;
  int d_refl = 0, d_refr = 0;
  float cos_theta = dot(d, sf.n);
  if (hit && (P.flags & RT_FLAG_REFLECTIONS) && ((m.metallic > 0.0f) || T)) {
    bool inside = cos_theta < 0.0f;
    float n2 = inside ? m.ior : P.air_ior;
    float eta = inside ? (n2 / n_start) : (n_start / n2);
    float cos_i = fabsf(cos_theta);
    float sin2 = eta * eta * (1.0f - cos_i * cos_i);
    bool reflective = (m.metallic > 0.0f) || (T && sin2 >= 1.0f);
    d_refl = depth < 0 ? (int)P.max_depth_reflection : (depth > 0 ? depth - 1 : 0);
    s_refl = reflective && d_refl > 0;
  }
  float inv_eta = 1.0f, ndi = 0.0f, kk = 0.0f, n2r = 0.0f;
  V3 inormal_r = sf.n;
  if (hit && (P.flags & RT_FLAG_REFRACTIONS) && T) {
    bool inside = cos_theta <= 0.0f;
    inormal_r = inside ? -sf.n : sf.n;
    n2r = inside ? m.ior : P.air_ior;
    float eta = inside ? (n2r / n_start) : (n_start / n2r);
    inv_eta = 1.0f / eta;
    ndi = dot(-inormal_r, d);
    kk = 1.0f - inv_eta * inv_eta * (1.0f - ndi * ndi);
    float op = m.opacity;
    int step = (op < 0.5f) ? 2 : 1;
    int fac = (op <= 0.3f) ? 3 : ((op < 0.5f) ? 2 : 1);
    d_refr = depth < 0 ? (int)P.max_depth_refraction / fac : (depth > step ? depth - step : 0);
    s_refr = !(kk < 0.0f) && d_refr > 0;  // kk < 0: zero vector -> NaN direction -> miss (deviation D2)
  }
  // ---- one reservation for the workgroup
  const lanemask m1 = wave_ballot(s_refl), m2 = wave_ballot(s_refr);
  const uint32_t n1 = (uint32_t)__popcll(m1), n2c = (uint32_t)__popcll(m2);
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (wave-uniform: the prefix loop below is a scalar loop)
  uint32_t* L = lds + parity * 8u;
  if ((threadIdx.x & 63u) == 0) L[wave] = n1 + n2c;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t tot = L[0] + L[1] + L[2] + L[3];
    L[4] = tot ? atomicAdd(P.q_out_count, tot) : 0u;
  }
  __syncthreads();
  uint32_t base = L[4];
  for (uint32_t w = 0; w < wave; w++) base += L[w];
  base = __builtin_amdgcn_readfirstlane(base);
  auto store = [&](uint32_t i, V3 o, V3 dd, float ns, V3 Wt, int dep, int kind) {
    if (i < P.q_capacity) {
      float4* rec = P.q_out + (size_t)i * RT_QUEUE_QUADS;
      stream_store4(rec + 0, make_float4(o.x, o.y, o.z, ns));
      stream_store4(rec + 1, make_float4(dd.x, dd.y, dd.z, __int_as_float(pack_dkm(dep, kind, mult))));
      stream_store4(rec + 2, make_float4(Wt.x, Wt.y, Wt.z, __uint_as_float(pix)));
    } else {
      atomicAdd(P.q_overflow, 1u);
    }
  };
  if (s_refl) {
    bool inside = cos_theta < 0.0f;
    V3 inormal = inside ? -sf.n : sf.n;
    V3 rr = normalize(reflected(d, sf.n));
    V3 Rf = fresnel_reflectance(m, inormal, -d, n_start);
    store(base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u)), sf.p + rr * epsv, rr, n_start, W0 * Rf, d_refl,
          KIND_REFL);
  }
  if (s_refr) {
    V3 Rf = fresnel_reflectance(m, inormal_r, d, inv_eta);
    V3 Tr = mk(1.0f - Rf.x, 1.0f - Rf.y, 1.0f - Rf.z);
    V3 nn = -inormal_r;  // ultraviolet refracted(n = -inormal, eta = 1/eta)
    float sq = inv_eta * ndi + __builtin_sqrtf(kk);
    V3 q = normalize(d * inv_eta - nn * sq);
    store(base + n1 + __builtin_amdgcn_mbcnt_hi((uint32_t)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m2, 0u)), sf.p + q * epsv, q, n2r,
          (W0 * (m.boost + 1.0f)) * Tr, d_refr, KIND_REFR);
  }
}

__device__ __forceinline__ RayIn load_queued_ray(const RtDevParams& P, size_t j) {
  const float4* rec = P.q_in + j * RT_QUEUE_QUADS;
  float4 a = stream_load4(rec + 0), b = stream_load4(rec + 1), c = stream_load4(rec + 2);
  RayIn r;
  r.o = mk(a.x, a.y, a.z);
  r.n_start = a.w;
  r.d_raw = mk(b.x, b.y, b.z);
  unpack_dkm(__float_as_int(b.w), r.depth, r.kind, r.mult);
  r.Wt = mk(c.x, c.y, c.z);
  r.pix = __float_as_uint(c.w);
  return r;
}
__device__ __forceinline__ RayIn idle_ray() {
  RayIn r;
  r.o = mk(0, 0, 0);
  r.d_raw = mk(0, 0, 1);
  r.n_start = 1.0f;
  r.Wt = mk(0, 0, 0);
  r.depth = 1;
  r.kind = KIND_REFL;
  r.pix = 0;
  r.mult = 1;
  return r;
}
__device__ __forceinline__ uint32_t morton_expand10(uint32_t v) {
  v = (v | (v << 16)) & 0x030000FFu;
  v = (v | (v << 8)) & 0x0300F00Fu;
  v = (v | (v << 4)) & 0x030C30C3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

// SPAWN (merged levels, rt_tuning.levels): the level's rays are the slice [*seg_lo, *seg_hi) of ONE append-only queue, and the kernel
// that finds a ray's hit also appends its children behind the slice -- so the next level can be traced at once, without waiting for this
// level to be shaded, and ALL levels are shaded by one launch in one hit-point order.
template <bool CULL, bool SPAWN>
__device__ __forceinline__ void trace_body(const RtDevScene& sc, const RtDevParams& P, unsigned long long* lds_cnt) {
  Wave wv;
  wave_init(wv);
  wave_flush_init(P, lds_cnt);
  // how many rays this level holds is only known on the device (the kernel before appended them)
  uint32_t n = SPAWN ? uload(P.seg_hi) : uload(P.q_in_count);
  n = n < P.q_capacity ? n : P.q_capacity;
  uint32_t first = SPAWN ? uload(P.seg_lo) : 0u;
  first = first < n ? first : n;
  uint32_t spawn_parity = 0u;
  for (uint32_t base = first + blockIdx.x * 256u; base < n; base += gridDim.x * 256u) {
    const uint32_t i = base + threadIdx.x;
    const bool have = i < n;
    RayIn r = idle_ray();
    if (have) r = load_queued_ray(P, (size_t)i);
    V3 d = normalize(r.d_raw);  // Ray::new_with_mask, ray.rs:52-57
    bool alive = have && !has_nan(d);
    unsigned long long bal = wave_ballot(alive);
    Hit h;
    h.t = INFINITY;
    h.id = -1;
    if (bal) {
      if (P.weighted) {
        wv.cnt_kind[1] += wave_sum((alive && r.kind == KIND_REFL) ? r.mult : 0u);
        wv.cnt_kind[2] += wave_sum((alive && r.kind == KIND_REFR) ? r.mult : 0u);
      } else {
        wv.cnt_kind[1] += (uint32_t)__popcll(wave_ballot(alive && r.kind == KIND_REFL));
        wv.cnt_kind[2] += (uint32_t)__popcll(wave_ballot(alive && r.kind == KIND_REFR));
      }
      wv.cnt_traced += (uint32_t)__popcll(bal);
      WSTAT(wv.cnt_pass += 1);
      WSTAT(wv.cnt_lanes += (uint32_t)__popcll(bal));
      h = nearest_hit<CULL>(sc, P, wv.ctx, alive, r.o, d);
    }
    const bool hit = alive && h.id >= 0;
    uint32_t key = 0u, bucket = 0u, rank = 0u;
    if (hit) {
      V3 p = fma_s(d, h.t, r.o);
      // 10 bits per axis over the scene's bounding box (host: prepare()); the top sort_bits bits order the shading
      uint32_t qx = (uint32_t)clampf((p.x - P.morton_lo[0]) * P.morton_scale[0], 0.0f, 1023.0f);
      uint32_t qy = (uint32_t)clampf((p.y - P.morton_lo[1]) * P.morton_scale[1], 0.0f, 1023.0f);
      uint32_t qz = (uint32_t)clampf((p.z - P.morton_lo[2]) * P.morton_scale[2], 0.0f, 1023.0f);
      key = morton_expand10(qx) | (morton_expand10(qy) << 1) | (morton_expand10(qz) << 2);
      bucket = key >> (30u - P.sort_bits);
    }
    // rank of the ray inside its bucket: one atomic per wavefront and distinct bucket (neighbouring rays mostly share one)
    for (lanemask todo = wave_ballot(hit); todo;) {
      const int fl = __ffsll((long long)todo) - 1;
      const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)bucket, fl);
      const lanemask same = wave_ballot(bucket == b) & todo;
      uint32_t first = 0;
      if ((int)(threadIdx.x & 63u) == fl) first = atomicAdd(&P.sort_hist[b], (uint32_t)__popcll(same));
      first = (uint32_t)__builtin_amdgcn_readlane((int)first, fl);
      if (lane_of(same)) rank = first + __builtin_amdgcn_mbcnt_hi((uint32_t)(same >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)same, 0u));
      todo &= ~same;
    }
    if (have) {
      stream_store4(P.q_in + (size_t)i * RT_QUEUE_QUADS + 3u, make_float4(h.t, __int_as_float(hit ? h.id : -1), __uint_as_float(key), __uint_as_float(rank)));
      // what rt_sort_place_kernel needs, 8 coalesced bytes per ray (it would pull the 64-byte record for them otherwise)
      P.sort_slot[i] = make_uint2(hit ? bucket : 0xFFFFFFFFu, rank);
    }
    if (SPAWN) {  // (every wavefront of the workgroup: the reservation below has barriers)
      Hit hh = h;
      hh.id = hit ? h.id : 0;
      Surf sf;
      sf.p = mk(0, 0, 0), sf.n = mk(0, 0, 1), sf.mat = 0;
      if (hit) sf = surface_of(sc, hh, r.o, d);
      const Mat m = load_mat(sc, sf.mat);
      V3 W0 = r.Wt;
      if (r.kind == KIND_REFL) W0 = W0 * atten(h.t);  // (a reflection child's weight carries its own atten(t), :722-726)
      spawn_children_block(sc, P, hit, d, sf, m, W0, r.n_start, r.depth, r.pix, r.mult, (uint32_t*)(lds_cnt + 20), spawn_parity);
      spawn_parity ^= 1u;
    }
  }
  wave_flush(wv, P, 0ull, lds_cnt);
}

__global__ __launch_bounds__(256, 4) void rt_trace_kernel(RtDevScene sc, RtDevParams P) {
  __shared__ unsigned long long lds_cnt[20];  // 15 counters (wave_flush), [16..19]: wave start times (cost calibration)
  if (P.flags & RT_FLAG_BACKFACE_CULLING)
    trace_body<true, false>(sc, P, lds_cnt);
  else
    trace_body<false, false>(sc, P, lds_cnt);
}
__global__ __launch_bounds__(256, 4) void rt_trace_spawn_kernel(RtDevScene sc, RtDevParams P) {
  __shared__ unsigned long long lds_cnt[28];  // [20..27]: the workgroup's queue reservation (spawn_children_block), 2 x 8 dwords
  if (P.flags & RT_FLAG_BACKFACE_CULLING)
    trace_body<true, true>(sc, P, lds_cnt);
  else
    trace_body<false, true>(sc, P, lds_cnt);
}

template <bool CULL>
__device__ __forceinline__ void shade_body(const RtDevScene& sc, const RtDevParams& P, float* lds_stash,
                                           unsigned long long* lds_cnt) {
  Wave wv;
  wave_init(wv);
  wave_flush_init(P, lds_cnt);
  const uint32_t n = uload(P.sort_hits);  // the rays of this level that hit something, in hit-point order (misses are not listed)
  const uint32_t first = P.seg_lo ? uload(P.seg_lo) : 0u;  // (pipelined levels: the level's sorted positions start at its slice of the queue)
  for (uint32_t base = blockIdx.x * 256u; base < n; base += gridDim.x * 256u) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t jr = i < n ? P.sh_idx[first + i] : 0xFFFFFFFFu;
    const bool have = jr != 0xFFFFFFFFu;
    RayIn r = idle_ray();
    Hit h;
    h.t = INFINITY;
    h.id = -1;
    if (have) {
      const size_t j = jr;  // i-th ray in hit-point order: one 64-byte record
      r = load_queued_ray(P, j);
      const float4 q3 = stream_load4(P.q_in + j * RT_QUEUE_QUADS + 3u);
      h.t = q3.x;
      h.id = __float_as_int(q3.y);
    }
    RayOut out = process_ray<CULL, true, true>(sc, P, wv, have, r, lds_stash, h);
    if (out.hit) {
      const long long* fx = stash_fix(lds_stash) + threadIdx.x;
      acc_add_fixed(P, out.pix, fx[0], fx[256], fx[512], out.mult);
    }
  }
  wave_flush(wv, P, 0ull, lds_cnt);
}

__global__ __launch_bounds__(256, RT_MIN_WAVES) void rt_shade_kernel(RtDevScene sc, RtDevParams P) {
  __shared__ __attribute__((aligned(16))) float lds_stash[RT_STASH_FIELDS * 256];
  __shared__ unsigned long long lds_cnt[20];
  if (P.flags & RT_FLAG_BACKFACE_CULLING)
    shade_body<true>(sc, P, lds_stash, lds_cnt);
  else
    shade_body<false>(sc, P, lds_stash, lds_cnt);
}

// ------------------------------------------------------------------------------------------------
// rt_hard_kernel: the soft-shadow light of one (hit point, light) pair per N lanes -- lane j traces sample j
// (PointLight::to_point_light_cloud, light.rs:183-225; has_any_intersection, raytracer.rs:24-106; calculate_lighting,
// raytracer_renderer.rs:731-874) -- for the pairs the render kernels deferred (incoherent wavefronts).  The N shadow rays
// of a pair leave one point towards a small cloud, and every lane walks the BVH for its OWN ray: stackless, over the
// threaded copy of the tree (skip links), with per-lane vector loads; a lane alone visits ~50 nodes where the union of
// 64 unrelated rays visits thousands.  The samples' contributions are summed over the N lanes in a fixed order and
// added to the pixel accumulator (linear in the light's share of `direct` and `specular`, :206-209,:251-257).
// ------------------------------------------------------------------------------------------------
template <bool CULL>
__device__ __forceinline__ void shadow_tris_lane(const RtDevScene& sc, lanemask grp, V3 o, V3 d, float tmax, Shadow& S) {
  const BoxRay br = box_ray(o, d);
  const float tl = t_limit_slack(tmax);
  uint32_t n_exact = 0;
  // entry `node` of the threaded tree: a missed box jumps to its skip link, a hit box steps to the next entry (its
  // first child or -- after a leaf -- the same as its skip link)
  uint32_t node = 0;
  for (;;) {
    const lanemask live = grp & ~S.occ & wave_ballot(node < sc.n_thr);
    if (!live) break;
    const bool on = lane_of(live);
    const uint32_t at = sc.off_nodes_thr + (on ? node : 0u) * 32u;
    const float4 b0 = vload<float4>(sc, at), b1 = vload<float4>(sc, at + 16u);
    const float lo[3] = {b0.x, b0.y, b0.z}, hi[3] = {b1.x, b1.y, b1.z};
    float tn, tm;
    box_one(lo, hi, br, tn, tm);
    const float slack = __builtin_fmaf(fabsf(tm), 4e-6f, tm + 1e-5f);
    const lanemask hitbox = live & wave_ballot(tn <= fminf(slack, tl)) & wave_ballot(slack >= 0.0f);
    const uint32_t leaf = __float_as_uint(b1.w);
    lanemask todo = hitbox & wave_ballot((leaf >> 24) != 0u);
    for (uint32_t k = 0; todo; k++) {  // the triangles of the lanes' leaves, one per lane per round
      todo &= wave_ballot(k < (leaf >> 24)) & ~S.occ;
      if (!todo) break;
      const uint32_t slot = lane_of(todo) ? (leaf & 0xFFFFFFu) + k : 0u;
      const float4 q0 = vload<float4>(sc, sc.off_tri_isect + slot * 48u);
      const float4 q1 = vload<float4>(sc, sc.off_tri_isect + slot * 48u + 16u);
      const float4 q2 = vload<float4>(sc, sc.off_tri_isect + slot * 48u + 32u);
      float t;
      lanemask h = tri_hit(q0, q1, q2, o, d, todo, tmax, t, n_exact);
      h &= wave_ballot(t <= tmax);
      if (h) {
        const float4 sh = vload<float4>(sc, sc.off_tri_shade + slot * 16u);
        const Mat m = load_mat(sc, lane_of(h) ? __float_as_uint(sh.w) : 0u);
        shadow_accumulate_lane<CULL>(S, m, mk(sh.x, sh.y, sh.z), d, h);
      }
    }
    node = on ? (lane_of(hitbox) ? node + 1u : __float_as_uint(b0.w)) : node;
  }
}

template <bool CULL>
__device__ __forceinline__ void hard_body(const RtDevScene& sc, const RtDevParams& P, uint32_t wave_index, uint32_t n_pairs) {
  const uint32_t N = P.light_mult;   // 2..64 (host)
  const uint32_t ppw = 64u / N;      // pairs per wavefront
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t pw = lane / N, j = lane - pw * N;
  const uint32_t pair = wave_index * ppw + pw;
  const bool have = pw < ppw && pair < n_pairs;
  const lanemask grp = wave_ballot(have);
  if (!grp) return;
  const size_t stride = (size_t)P.hard_capacity + 64u;
  const uint32_t at = have ? pair : 0u;
  const float4 r0 = P.hard_q[0 * stride + at], r1 = P.hard_q[1 * stride + at], r2 = P.hard_q[2 * stride + at],
               r3 = P.hard_q[3 * stride + at];
  const V3 p = mk(r0.x, r0.y, r0.z), n = mk(r1.x, r1.y, r1.z), view = mk(r2.x, r2.y, r2.z), Wa = mk(r3.x, r3.y, r3.z);
  const uint32_t l = have ? __float_as_uint(r1.w) : 0u, pix = __float_as_uint(r2.w), mult = __float_as_uint(r3.w);
  const Mat m = load_mat(sc, have ? __float_as_uint(r0.w) : 0u);
  const float4 L0 = vload<float4>(sc, sc.off_lights + l * 32u), L1 = vload<float4>(sc, sc.off_lights + l * 32u + 16u);
  const V3 lc = mk(L1.x, L1.y, L1.z);
  // sample j of the pixel's cloud set of this light (light.rs:218; the table holds offset * (fw, fh, fd))
  const uint32_t hsh = rt_cloud_hash(P.cloud_seed, pix, l);
  const uint32_t set = (P.n_cloud_sets & (P.n_cloud_sets - 1u)) == 0u ? (hsh & (P.n_cloud_sets - 1u)) : (hsh % P.n_cloud_sets);
  const float4 cs = have ? P.cloud_sets[(size_t)set * N + j] : make_float4(0, 0, 0, 0);
  const float lI = (1.0f / (float)N) * L0.w;
  // the shadow ray, exactly as the render kernels set it up (raytracer_renderer.rs:770-790)
  const V3 lp = mk(L0.x + cs.x, L0.y + cs.y, L0.z + cs.z);
  const V3 ltp = lp - p;
  const V3 ld = ltp * exact_rcp(mag(ltp));
  const V3 so = p + ld * mk(P.eps_distance, P.eps_distance, P.eps_distance);
  const float tmax = mag(lp - so);
  const V3 d = normalize_unit(ld, grp);
  Shadow S;
  shadow_init(S);
  for (uint32_t i = 0; i < sc.n_spheres; i++) {  // spheres: few, tested by the whole wavefront one by one
    const float4 s = sload<float4>(sc, sc.off_spheres + i * 16u);
    float t = 0.0f;
    lanemask h = grp & ~S.occ & wave_ballot(sphere_hit(s, so, d, t));
    h &= wave_ballot(t <= tmax);
    if (h) {
      const V3 sp = fma_s(d, t, so);
      const V3 sn = CULL ? normalize(sp - mk(s.x, s.y, s.z)) : fast_normalize(sp - mk(s.x, s.y, s.z));
      const Mat sm = load_mat_u(sc, sload<uint32_t>(sc, sc.off_sphere_mat + i * 4u));
      if (CULL && !sm.transmissive) h &= wave_ballot(dot(d, sn) < 0.75f);
      shadow_accumulate(S, sm, sn, d, h);
    }
  }
  shadow_tris_lane<CULL>(sc, grp, so, d, tmax, S);
  // the sample's colour terms: the function the render kernels call (light_sample_terms), then the pair's chain over its
  // N samples in sample order -- run by every lane of the pair on values fetched from the sample lanes, so that the sum
  // is the same bits as the inline loop's
  const bool has_spec = m.shininess > 0.0f;
  const V3 mmc_lc = m.color * (m.color * lc);
  const LightTerms T = light_sample_terms<true>(n, view, mmc_lc, lI, m.shininess, has_spec, ltp, S);
  const int ok = (have && !lane_of(S.occ) && T.lit) ? 1 : 0;
  V3 dl = mk(0.0f, 0.0f, 0.0f), ds = mk(0.0f, 0.0f, 0.0f);
  const uint32_t first = (pw < ppw ? pw : 0u) * N;
  for (uint32_t jj = 0; jj < N; jj++) {
    const int src = (int)(first + jj);
    const V3 mLc = mk(__shfl(T.mLc.x, src, 64), __shfl(T.mLc.y, src, 64), __shfl(T.mLc.z, src, 64));
    const float lf = __shfl(T.lf, src, 64), sfv = __shfl(T.sf, src, 64);
    if (__shfl(ok, src, 64)) {
      dl = fma_s(mLc, lf, dl);
      if (has_spec) ds = fma_s(lc, sfv, ds);
    }
  }
  const V3 own = m.transmissive ? ds : (dl + ds);  // :251-257
  if (have && j == 0) acc_add(P, pix, Wa * own, mult);
}

__global__ __launch_bounds__(256) void rt_hard_kernel(RtDevScene sc, RtDevParams P) {
  // the pairs the launch before deferred: their number is only known on the device
  const uint32_t n_raw = uload((const uint32_t*)P.hard_count);
  const uint32_t n_pairs = n_raw < P.hard_capacity ? n_raw : P.hard_capacity;
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicMax(&P.hard_stat[1], n_raw);  // (sizes the queue of the next frame)
  const uint32_t ppw = 64u / P.light_mult;
  const uint32_t n_waves = (n_pairs + ppw - 1u) / ppw;
  for (uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6); w < n_waves; w += gridDim.x * 4u) {
    if (P.flags & RT_FLAG_BACKFACE_CULLING)
      hard_body<true>(sc, P, w, n_pairs);
    else
      hard_body<false>(sc, P, w, n_pairs);
  }
}

// ------------------------------------------------------------------------------------------------
// resolve: fixed-point accumulator -> packed pixel (and clears the accumulator for the next frame)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rt_resolve_kernel(RtDevParams P) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= P.win_w * P.win_h) return;
  const uint32_t gx = P.win_x0 + i % P.win_w, gy = P.win_y0 + i / P.win_w;
  const uint32_t pix = gy * P.width + gx;
  long long* a = P.acc + 4 * (size_t)pix;
  long long r = a[0], g = a[1], b = a[2], f = a[3];
  unsigned long long wrote = 0;
  if (f) {
    V3 c = mk((float)r * RT_ACC_INV_SCALE, (float)g * RT_ACC_INV_SCALE, (float)b * RT_ACC_INV_SCALE);
    P.argb[out_index(P, gx, gy, pix)] = pack_argb(c);
    if (P.aux_rgb) {
      P.aux_rgb[3 * (size_t)pix + 0] = c.x;
      P.aux_rgb[3 * (size_t)pix + 1] = c.y;
      P.aux_rgb[3 * (size_t)pix + 2] = c.z;
    }
    a[0] = a[1] = a[2] = a[3] = 0;
    wrote = 1;
  }
  if (P.resolve_counts_written && P.counters) {  // (the phase kernels do not count pixels: one atomic per wavefront here)
    const uint32_t nw = (uint32_t)__popcll(wave_ballot(wrote != 0ull));
    if (nw && (threadIdx.x & 63u) == 0) atomicAdd(&P.counters[(size_t)(blockIdx.x % RT_COUNTER_REPLICAS) * 16u + 4u], (unsigned long long)nw);
  }
}

// ------------------------------------------------------------------------------------------------
// rt_flags_kernel: one thread per receiver cell (see FatBeam / COLLECT_FLAGS).  Runs when a scene is first rendered with
// soft shadows, and again when the light clouds' size changes.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rt_flags_kernel(RtDevScene sc, RtDevParams P) {
  const uint32_t c = blockIdx.x * 256u + threadIdx.x;
  const bool have = c < P.n_cells;
  FatBeam fb;
  V3 pm = mk(0.0f, 0.0f, 0.0f);
  bool active = false;
  if (have && c < P.n_tri_cells) {
    // ---- a cell of a triangle: the parallelogram [i, i+1] x [j, j+1] / R of its (u, v) coordinates -------------------
    // the triangle that owns cell c: the last one whose first cell is <= c (triangles without cells share their successor's)
    uint32_t lo = 0u, hi = sc.n_triangles;  // invariant: first(lo) <= c, first(hi) > c (first(n) = n_tri_cells)
    while (hi - lo > 1u) {
      const uint32_t mid = (lo + hi) >> 1;
      if (__float_as_uint(P.flag_geo[3u * mid + 1u].w) <= c) lo = mid; else hi = mid;
    }
    const uint32_t t = lo;
    const float4 g0 = P.flag_geo[3u * t], g1 = P.flag_geo[3u * t + 1u], g2 = P.flag_geo[3u * t + 2u];
    const uint32_t R = __float_as_uint(g0.w), first = __float_as_uint(g1.w);
    const uint32_t local = c - first;
    const uint32_t j = R ? local / R : 0u, i = local - j * R;
    active = R != 0u && i + j < R;  // (cells beyond the hypotenuse are never looked up)
    const V3 v1 = mk(g0.x, g0.y, g0.z), e1 = mk(g1.x, g1.y, g1.z), e2 = mk(g2.x, g2.y, g2.z);
    const float inv_r = R ? 1.0f / (float)R : 0.0f;
    // the cell, 5 % larger than it is on every side (a hit point's computed (u, v) may round across a cell border)
    const float u0 = ((float)i - 0.05f) * inv_r, u1 = ((float)i + 1.05f) * inv_r, w0 = ((float)j - 0.05f) * inv_r, w1 = ((float)j + 1.05f) * inv_r;
    fb.pt[0] = v1 + e1 * u0 + e2 * w0;
    fb.pt[1] = v1 + e1 * u1 + e2 * w0;
    fb.pt[2] = v1 + e1 * u0 + e2 * w1;
    fb.pt[3] = v1 + e1 * u1 + e2 * w1;
#pragma unroll
    for (int k = 0; k < 4; k++) fb.pt[4 + k] = fb.pt[k];
    pm = v1 + e1 * (((float)i + 0.5f) * inv_r) + e2 * (((float)j + 0.5f) * inv_r);
  } else if (have) {
    // ---- a cell of a sphere: a patch of its cube map of directions.  The patch lies in the frustum of the cone through
    // its four corner directions between the tangent plane at the patch centre (outer corners, r / cos) and the parallel
    // plane through its lowest corner (inner corners, r cos_min / cos): 8 points whose hull contains the patch ----------
    uint32_t k = 0u, Rs = 0u, first = 0u;
    for (uint32_t q = 0; q < sc.n_spheres; q++) {
      const uint2 sr = vload<uint2>(sc, sc.off_srecv + q * 8u);
      if (sr.x != 0u && c >= sr.y && c < sr.y + 6u * sr.x * sr.x) k = q, Rs = sr.x, first = sr.y;
    }
    if (Rs != 0u) {
      active = true;
      const float4 sp = vload<float4>(sc, sc.off_spheres + k * 16u);
      const float rad = vload<float>(sc, sc.off_sphere_rad + k * 4u);  // (upper bound of the radius)
      const uint32_t local = c - first, face = local / (Rs * Rs), rem = local - face * Rs * Rs, j = rem / Rs, i = rem - j * Rs;
      const float inv_r = 2.0f / (float)Rs;
      const float a0 = fmaxf(((float)i - 0.05f) * inv_r - 1.0f, -1.1f), a1 = fminf(((float)i + 1.05f) * inv_r - 1.0f, 1.1f);
      const float b0 = fmaxf(((float)j - 0.05f) * inv_r - 1.0f, -1.1f), b1 = fminf(((float)j + 1.05f) * inv_r - 1.0f, 1.1f);
      const uint32_t m = face >> 1;
      const float sgn = (face & 1u) ? -1.0f : 1.0f;
      auto dir = [&](float a, float b) {  // component m = sign, (m + 1) % 3 = a, (m + 2) % 3 = b; normalised
        const V3 d = m == 0u ? mk(sgn, a, b) : (m == 1u ? mk(b, sgn, a) : mk(a, b, sgn));
        return d * __builtin_amdgcn_rsqf(dot(d, d));
      };
      const V3 ctr = mk(sp.x, sp.y, sp.z);
      const V3 uc = dir(0.5f * (a0 + a1), 0.5f * (b0 + b1));
      const V3 uk[4] = {dir(a0, b0), dir(a1, b0), dir(a0, b1), dir(a1, b1)};
      float cmin = 1.0f;
#pragma unroll
      for (int q = 0; q < 4; q++) cmin = fminf(cmin, dot(uk[q], uc));
      cmin = fmaxf(cmin * 0.9999f, 0.1f);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const float cq = fmaxf(dot(uk[q], uc), 0.1f);
        fb.pt[q] = ctr + uk[q] * (rad * cmin * 0.9999f * __builtin_amdgcn_rcpf(cq));      // inner
        fb.pt[4 + q] = ctr + uk[q] * (rad * 1.0001f * __builtin_amdgcn_rcpf(cq));          // outer (tangent plane)
      }
      pm = ctr + uc * rad;
    }
  }
  if (!active) {
#pragma unroll
    for (int k = 0; k < 8; k++) fb.pt[k] = pm;
  }
  float r = 0.0f;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const V3 dk = fb.pt[k] - pm;
    r = fmaxf(r, fabsf(dk.x) + fabsf(dk.y) + fabsf(dk.z));
  }
  fb.r = __builtin_fmaf(r, 1.0001f, 4e-7f * (fabsf(pm.x) + fabsf(pm.y) + fabsf(pm.z)));
  WaveCtx W;
  W.n_nodes = W.n_tris = W.s_nodes = W.s_tris = W.s_passes = W.n_exact = W.s_exact = 0;
  uint32_t flags = 0u;
  for (uint32_t l = 0; l < sc.n_lights && l < 8u; l++) {
    const float4 L0 = sload<float4>(sc, sc.off_lights + l * 32u);
    const V3 centre = mk(L0.x + P.cloud_centre[0], L0.y + P.cloud_centre[1], L0.z + P.cloud_centre[2]);
    // (the cell's candidate list for this light: RT_CELL_LIST_SLOTS 16-bit leaf slots, preset to 0xFFFF by the host)
    fb.list = (P.cell_list_out && have) ? P.cell_list_out + ((size_t)c * sc.n_lights + l) * RT_CELL_LIST_SLOTS : nullptr;
    const CandList cl = collect_light_candidates<false, COLLECT_FLAGS>(sc, W, active, pm, centre, P, pm, 0.0f, RT_MAX_CANDIDATES, &fb,
                                                                          sc.n_triangles != 0u);  // (no triangles: no walk)
    if (fb.list && active && cl.reg > RT_CELL_LIST_SLOTS) fb.list[0] = (uint16_t)RT_CELL_LIST_OVERFLOW;
    const lanemask near_m = (lanemask)cl.count | ((lanemask)cl.spheres << 32);
    if (wave_ballot(active)) {
      flags |= lane_of(cl.umbra) ? 0u : (1u << l);
      flags |= lane_of(near_m) ? 0u : (1u << (8u + l));
    }
  }
  if (have) P.flag_out[c] = (uint16_t)(active ? flags : 0u);
}

// diagnostics (rt_selftest_exact_math): the exact sequences on arbitrary operands
__global__ __launch_bounds__(256) void rt_selftest_math_kernel(const float* in, float* out_sqrt, float* out_rcp, uint32_t n) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  out_sqrt[i] = exact_sqrt(in[i]);
  out_rcp[i] = exact_rcp(in[i]);
}

#include "rt_phases.h"

}  // namespace

bool rt_phases_arrive_inline() { return RT_ARRIVE_INLINE != 0; }
bool rt_has_cost_kernel() { return RT_COST_KERNEL != 0; }

int rt_launch_hit(const RtDevScene& sc, const RtDevParams& p, uint32_t n_wgs, void* stream) {
  if (n_wgs == 0) return 0;
  if (p.hit_spawns)  // merged levels: the camera rays' children are appended where their hits are found
    hipLaunchKernelGGL(rt_hit_spawn_kernel, dim3(n_wgs), dim3(256), 0, (hipStream_t)stream, sc, p);
  else
    hipLaunchKernelGGL(rt_hit_kernel, dim3(n_wgs), dim3(256), 0, (hipStream_t)stream, sc, p);
  return (int)hipGetLastError();
}

int rt_launch_classify(const RtDevScene& sc, const RtDevParams& p, bool level0, uint32_t n_wgs, void* stream) {
  if (n_wgs == 0) {
    if (level0) return 0;
    n_wgs = 1;
  }
  if (level0)
    hipLaunchKernelGGL(rt_classify0_kernel, dim3(n_wgs), dim3(256), 0, (hipStream_t)stream, sc, p);
  else
    hipLaunchKernelGGL(rt_classify_kernel, dim3(n_wgs), dim3(256), 0, (hipStream_t)stream, sc, p);
  return (int)hipGetLastError();
}

int rt_launch_compact(const RtDevParams& p, uint32_t n_wgs, void* stream) {
  if (n_wgs == 0) n_wgs = 1;
  hipLaunchKernelGGL(rt_compact_kernel, dim3(n_wgs), dim3(256), 0, (hipStream_t)stream, p);
  return (int)hipGetLastError();
}

int rt_launch_sets(const RtDevScene& sc, const RtDevParams& p, bool level0, int cls, uint32_t n_wgs, void* stream) {
  if (n_wgs == 0) n_wgs = 1;
  const dim3 g(n_wgs), b(256);
  hipStream_t st = (hipStream_t)stream;
  if (cls == SET_ARRIVE) {
#if !RT_ARRIVE_INLINE
    if (level0) hipLaunchKernelGGL(rt_sets0_arrive_kernel, g, b, 0, st, sc, p);
    else hipLaunchKernelGGL(rt_sets_arrive_kernel, g, b, 0, st, sc, p);
#endif
  } else if (level0) {
    if (cls == SET_LIST) hipLaunchKernelGGL(rt_sets0_list_kernel, g, b, 0, st, sc, p);
    else hipLaunchKernelGGL(rt_sets0_walk_kernel, g, b, 0, st, sc, p);
  } else {
    if (cls == SET_LIST) hipLaunchKernelGGL(rt_sets_list_kernel, g, b, 0, st, sc, p);
    else hipLaunchKernelGGL(rt_sets_walk_kernel, g, b, 0, st, sc, p);
  }
  return (int)hipGetLastError();
}

int rt_launch_flags(const RtDevScene& sc, const RtDevParams& p, void* stream) {
  if (p.n_cells == 0) return 0;
  hipLaunchKernelGGL(rt_flags_kernel, dim3((p.n_cells + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, sc, p);
  return (int)hipGetLastError();
}

int rt_launch_selftest_math(const float* in, float* out_sqrt, float* out_rcp, uint32_t n, void* stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(rt_selftest_math_kernel, dim3((n + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, in, out_sqrt, out_rcp, n);
  return (int)hipGetLastError();
}

// ---- host-side launchers ---------------------------------------------------------------------------
uint32_t rt_primary_pixels_per_wg(const RtDevParams& p) {
  const bool aa = (p.flags & RT_FLAG_ANTI_ALIASING) && p.aa_rays > 0;
  const uint32_t n_thr = aa ? p.aa_unique : 1u;
  return rt_primary_wave_local(n_thr) ? 4u * (64u / n_thr) : 256u / n_thr;
}

// workgroups of the primary kernel: the 256 pixels of every (listed) 16x16 super-tile, ppw per workgroup
uint32_t rt_primary_total_wgs(const RtDevParams& p) {
  uint32_t ppw = rt_primary_pixels_per_wg(p);
  return (uint32_t)(((uint64_t)p.n_sup * 256u + ppw - 1u) / ppw);
}

int rt_launch_primary(const RtDevScene& sc, const RtDevParams& p, uint32_t n_wgs, void* stream) {
  if (n_wgs == 0) return 0;  // nothing owned inside the window
#if RT_COST_KERNEL
  if (p.cost_map) {
    hipLaunchKernelGGL(rt_primary_cost_kernel, dim3(n_wgs), dim3(256), 0, (hipStream_t)stream, sc, p);
    return (int)hipGetLastError();
  }
#endif
  if (p.acc && p.hitrec)  // (hits and children by rt_hit_spawn_kernel)
    hipLaunchKernelGGL(rt_primary_pre_kernel, dim3(n_wgs), dim3(256), 0, (hipStream_t)stream, sc, p);
  else if (p.acc)
    hipLaunchKernelGGL(rt_primary_stream_kernel, dim3(n_wgs), dim3(256), 0, (hipStream_t)stream, sc, p);
  else
    hipLaunchKernelGGL(rt_primary_kernel, dim3(n_wgs), dim3(256), 0, (hipStream_t)stream, sc, p);
  return (int)hipGetLastError();
}

int rt_launch_trace(const RtDevScene& sc, const RtDevParams& p, uint32_t n_wgs, void* stream) {
  if (n_wgs == 0) n_wgs = 1;
  if (p.seg_lo)
    hipLaunchKernelGGL(rt_trace_spawn_kernel, dim3(n_wgs), dim3(256), 0, (hipStream_t)stream, sc, p);
  else
    hipLaunchKernelGGL(rt_trace_kernel, dim3(n_wgs), dim3(256), 0, (hipStream_t)stream, sc, p);
  return (int)hipGetLastError();
}

int rt_launch_shade(const RtDevScene& sc, const RtDevParams& p, uint32_t n_wgs, void* stream) {
  if (n_wgs == 0) n_wgs = 1;
  hipLaunchKernelGGL(rt_shade_kernel, dim3(n_wgs), dim3(256), 0, (hipStream_t)stream, sc, p);
  return (int)hipGetLastError();
}

int rt_launch_hard(const RtDevScene& sc, const RtDevParams& p, uint32_t n_wgs, void* stream) {
  if (n_wgs == 0) n_wgs = 1;
  hipLaunchKernelGGL(rt_hard_kernel, dim3(n_wgs), dim3(256), 0, (hipStream_t)stream, sc, p);
  return (int)hipGetLastError();
}

int rt_launch_resolve(const RtDevParams& p, void* stream) {
  uint32_t n = p.win_w * p.win_h;
  hipLaunchKernelGGL(rt_resolve_kernel, dim3((n + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, p);
  return (int)hipGetLastError();
}
