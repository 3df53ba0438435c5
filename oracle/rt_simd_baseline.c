/*
 * rt_simd_baseline.c -- the CPU baseline of bench.py: the reference's `simd_render` path restated as 8-lane AVX2
 * packets, brute force (every ray scans every object, like the reference: SURVEY F1), 48x48 tiles handed to a pool of
 * host threads.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY (like rt_oracle.c): nothing in the shipped package may import, link or
 * execute it; only tests/ and bench.py's cpu_baseline leg do.  The reference itself (Rust nightly + git-forked
 * crates) cannot be built in this image, so this file is what "the reference's simd_render CPU path timed on the GPU
 * box's host cores" can be here: cpu_baseline.kind = "port".
 *
 * Structure followed (paths relative to the reference repo root):
 *   tile driver          src/renderer/mod.rs:146-209, src/image_buffer.rs:48-97 (RENDER_STRIDE = 48 tiles, rayon pool
 *                        -> here: pthreads pulling tiles from an atomic counter), rows inside a tile :306-320
 *   packets              src/renderer/raytracer_renderer.rs:1199-1221 (anti-aliasing: the samples of ONE pixel in
 *                        packets of 8 = Vec3x8), :1256-1297 (no anti-aliasing: 8 consecutive pixels of a row)
 *   per-packet code      single_raytrace :147-264, calculate_lighting :731-874, calculate_reflection :526-729,
 *                        calculate_refractions :279-524, antialiased_raytrace :918-1016
 *   object loops         src/raytracing/raytracer.rs:162-220 (cast_ray: splat every object to 8 lanes, intersect,
 *                        blend), :24-106 (has_any_intersection, break when ALL lanes are occluded)
 *   intersections        src/geometry/basic/sphere.rs:78-162, triangle.rs:149-212
 *   materials / lights   src/raytracing/material.rs:468-525,213-231, src/scene/lighting/light.rs:261-299
 *
 * Numerics: the lanes run exactly the op sequence of rt_oracle.c (fused multiply-adds only where the reference calls
 * mul_add; correctly rounded div / sqrt; tanhf / powf from libm per lane), so hit ids, distances and colours are
 * bit-identical to the scalar oracle (tests/test_simd_baseline.py).  The deliberate deviations D1-D5 of rt_oracle.c
 * apply here as well (per-lane recursion depth instead of the packet's horizontal max; per-pixel light clouds).
 */
#include <immintrin.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/rt_hip.h"

#define RT_EPS 1.1920929e-7f

typedef __m256 f8;
typedef struct {
  f8 x, y, z;
} v8;

#define S8(v) _mm256_set1_ps(v)
static inline f8 add8(f8 a, f8 b) { return _mm256_add_ps(a, b); }
static inline f8 sub8(f8 a, f8 b) { return _mm256_sub_ps(a, b); }
static inline f8 mul8(f8 a, f8 b) { return _mm256_mul_ps(a, b); }
static inline f8 div8(f8 a, f8 b) { return _mm256_div_ps(a, b); }
static inline f8 fma8(f8 a, f8 b, f8 c) { return _mm256_fmadd_ps(a, b, c); }
static inline f8 neg8(f8 a) { return _mm256_xor_ps(a, S8(-0.0f)); }
static inline f8 abs8(f8 a) { return _mm256_andnot_ps(S8(-0.0f), a); }
static inline f8 and8(f8 a, f8 b) { return _mm256_and_ps(a, b); }
static inline f8 or8(f8 a, f8 b) { return _mm256_or_ps(a, b); }
static inline f8 andn8(f8 a, f8 b) { return _mm256_andnot_ps(a, b); } /* ~a & b */
static inline f8 sel8(f8 m, f8 a, f8 b) { return _mm256_blendv_ps(b, a, m); } /* m ? a : b */
static inline int any8(f8 m) { return _mm256_movemask_ps(m) != 0; }
static inline int bits8(f8 m) { return _mm256_movemask_ps(m); }
static inline f8 true8(void) { return _mm256_castsi256_ps(_mm256_set1_epi32(-1)); }
static inline f8 false8(void) { return _mm256_setzero_ps(); }
static inline f8 not8(f8 m) { return _mm256_xor_ps(m, true8()); }
/* fminf/fmaxf semantics for the non-NaN operands that occur here (clamp of finite values; NaN x -> the bound, as
 * fminf(fmaxf(x, lo), hi) gives): _mm256_max_ps(a, b) returns b when a is NaN */
static inline f8 clamp8(f8 x, f8 lo, f8 hi) { return _mm256_min_ps(_mm256_max_ps(x, lo), hi); }

static inline v8 V8(f8 x, f8 y, f8 z) {
  v8 r = {x, y, z};
  return r;
}
static inline v8 vsplat(float x, float y, float z) { return V8(S8(x), S8(y), S8(z)); }
static inline v8 vadd8(v8 a, v8 b) { return V8(add8(a.x, b.x), add8(a.y, b.y), add8(a.z, b.z)); }
static inline v8 vsub8(v8 a, v8 b) { return V8(sub8(a.x, b.x), sub8(a.y, b.y), sub8(a.z, b.z)); }
static inline v8 vmul8(v8 a, v8 b) { return V8(mul8(a.x, b.x), mul8(a.y, b.y), mul8(a.z, b.z)); }
static inline v8 vdiv8(v8 a, v8 b) { return V8(div8(a.x, b.x), div8(a.y, b.y), div8(a.z, b.z)); }
static inline v8 vscale8(v8 a, f8 s) { return V8(mul8(a.x, s), mul8(a.y, s), mul8(a.z, s)); }
static inline v8 vneg8(v8 a) { return V8(neg8(a.x), neg8(a.y), neg8(a.z)); }
static inline v8 vsel8(f8 m, v8 a, v8 b) { return V8(sel8(m, a.x, b.x), sel8(m, a.y, b.y), sel8(m, a.z, b.z)); }
/* ultraviolet Vec3x8::dot: x.mul_add(ox, y.mul_add(oy, z*oz)) */
static inline f8 vdot8(v8 a, v8 b) { return fma8(a.x, b.x, fma8(a.y, b.y, mul8(a.z, b.z))); }
static inline f8 vmag8(v8 a) { return _mm256_sqrt_ps(vdot8(a, a)); }
static inline v8 vnormalize8(v8 a) { return vscale8(a, div8(S8(1.0f), vmag8(a))); }
static inline v8 vfma_s8(v8 d, f8 t, v8 o) { return V8(fma8(d.x, t, o.x), fma8(d.y, t, o.y), fma8(d.z, t, o.z)); }
static inline v8 vreflected8(v8 v, v8 n) { return vsub8(v, vscale8(n, mul8(S8(2.0f), vdot8(v, n)))); }
static inline f8 isnan8(f8 a) { return _mm256_cmp_ps(a, a, _CMP_UNORD_Q); }

/* ---- materials (one row per object: uniform over the packet while an object is tested) ---------- */
typedef struct {
  float r, g, b, metallic, shininess, ior, opacity, boost;
  int transmissive;
} mat1;
static inline mat1 load_mat1(const rt_scene_desc* s, uint32_t idx) {
  const float* m = s->materials + (size_t)idx * RT_MATERIAL_STRIDE;
  mat1 r;
  r.r = m[RT_MAT_R], r.g = m[RT_MAT_G], r.b = m[RT_MAT_B];
  r.metallic = m[RT_MAT_METALLIC], r.shininess = m[RT_MAT_SHININESS], r.ior = m[RT_MAT_IOR];
  r.opacity = m[RT_MAT_OPACITY], r.boost = m[RT_MAT_BOOST];
  r.transmissive = (m[RT_MAT_HAS_OPACITY] != 0.0f) && !(fabsf(r.opacity - 0.0f) <= RT_EPS);
  return r;
}
/* per-lane materials of the packet's hit surfaces */
typedef struct {
  v8 color;
  f8 metallic, shininess, ior, opacity, boost, transmissive /* mask */;
} mat8;

/* Material::compute_fresnel (reflectance), material.rs:468-525; `tr` = lanes whose material is transmissive */
static inline v8 fresnel8(v8 color, f8 metallic, f8 ior, f8 tr, v8 normal, v8 view, f8 other_ior) {
  f8 n_dot_v = vdot8(normal, view);
  f8 cos_theta = abs8(n_dot_v);
  f8 inside = _mm256_cmp_ps(n_dot_v, S8(0.0f), _CMP_LT_OQ);
  f8 eta_t = sel8(inside, div8(ior, other_ior), div8(other_ior, ior));
  f8 sin2_t = mul8(mul8(eta_t, eta_t), sub8(S8(1.0f), mul8(cos_theta, cos_theta)));
  f8 reflective = _mm256_cmp_ps(metallic, S8(0.0f), _CMP_GT_OQ);
  f8 tir = or8(and8(inside, _mm256_cmp_ps(sin2_t, S8(1.0f), _CMP_GT_OQ)), reflective);
  f8 q = div8(sub8(other_ior, ior), add8(other_ior, ior));
  f8 f0 = mul8(q, q);
  f8 omt = sub8(S8(1.0f), metallic);
  v8 f0v = V8(add8(mul8(f0, omt), mul8(color.x, metallic)), add8(mul8(f0, omt), mul8(color.y, metallic)),
              add8(mul8(f0, omt), mul8(color.z, metallic)));
  f8 c1 = sub8(S8(1.0f), cos_theta);
  f8 c2 = mul8(c1, c1);
  f8 c5 = mul8(c1, mul8(c2, c2));
  v8 fres = V8(add8(f0v.x, mul8(sub8(S8(1.0f), f0v.x), c5)), add8(f0v.y, mul8(sub8(S8(1.0f), f0v.y), c5)),
               add8(f0v.z, mul8(sub8(S8(1.0f), f0v.z), c5)));
  f8 ra = sel8(reflective, metallic, S8(1.0f));
  v8 res = vsel8(tir, V8(ra, ra, ra), fres);
  return vsel8(tr, res, V8(metallic, metallic, metallic)); /* :483-488 */
}

/* attenuation_factor_based_on_distance, raytracer_renderer.rs:266-277 */
static inline f8 atten8(f8 t) {
  f8 d = abs8(t);
  f8 a = div8(S8(1.0f), add8(add8(S8(1.0f), d), mul8(mul8(S8(0.1f), d), d)));
  return clamp8(a, S8(0.0f), S8(1.0f));
}

/* ---- intersections: one object (splat) against 8 rays -------------------------------------------------------- */
/* SphereData::intersect, sphere.rs:78-162.  Returns the valid mask; *t_out, and (if n_out) the surface normal. */
static inline f8 sphere8(const rt_scene_desc* s, uint32_t i, v8 o, v8 d, int cull, f8* t_out, v8* n_out) {
  v8 c = vsplat(s->sphere_center[3 * i], s->sphere_center[3 * i + 1], s->sphere_center[3 * i + 2]);
  v8 v = vsub8(o, c);
  f8 b = mul8(S8(2.0f), vdot8(d, v));
  f8 cc = sub8(vdot8(v, v), S8(s->sphere_r_sq[i]));
  f8 disc = fma8(b, b, mul8(S8(2.0f * -2.0f), cc));
  f8 ok = _mm256_cmp_ps(disc, S8(0.0f), _CMP_GE_OQ);
  if (!any8(ok)) return ok;
  f8 sq = _mm256_sqrt_ps(disc);
  f8 mba = mul8(neg8(b), S8(0.5f));
  f8 sa = mul8(sq, S8(0.5f));
  f8 t0 = sub8(mba, sa), t1 = add8(mba, sa);
  f8 t0v = _mm256_cmp_ps(t0, S8(0.0f), _CMP_GE_OQ), t1v = _mm256_cmp_ps(t1, S8(0.0f), _CMP_GE_OQ);
  f8 use0 = and8(t0v, or8(not8(t1v), _mm256_cmp_ps(t0, t1, _CMP_LT_OQ)));
  f8 use1 = andn8(use0, t1v);
  f8 valid = and8(ok, or8(use0, use1));
  f8 t = sel8(use0, t0, t1);
  *t_out = t;
  if ((cull || n_out) && any8(valid)) {
    v8 p = vfma_s8(d, t, o);
    v8 n = vnormalize8(vsub8(p, c));
    if (cull) {
      mat1 m = load_mat1(s, s->sphere_material[i]);
      if (!m.transmissive) valid = and8(valid, _mm256_cmp_ps(vdot8(d, n), S8(0.75f), _CMP_LT_OQ));
    }
    if (n_out) *n_out = n;
  }
  return valid;
}

/* TriangleData::intersect, triangle.rs:149-212 (ultraviolet Mat3::inversed / determinant) */
static inline f8 triangle8(const rt_scene_desc* s, uint32_t i, v8 o, v8 d, int cull, f8* t_out) {
  const float *pv = s->tri_v1 + 3 * (size_t)i, *pe1 = s->tri_e1 + 3 * (size_t)i, *pe2 = s->tri_e2 + 3 * (size_t)i;
  f8 live = true8();
  if (cull) {
    const float* pn = s->tri_normal + 3 * (size_t)i;
    mat1 m = load_mat1(s, s->tri_material[i]);
    if (!m.transmissive) {
      live = _mm256_cmp_ps(vdot8(d, vsplat(pn[0], pn[1], pn[2])), S8(0.75f), _CMP_LT_OQ);
      if (!any8(live)) return live;
    }
  }
  v8 b = vsub8(vsplat(pv[0], pv[1], pv[2]), o);
  /* columns c0 = d, c1 = -e1, c2 = -e2 */
  const float c1x = -pe1[0], c1y = -pe1[1], c1z = -pe1[2], c2x = -pe2[0], c2y = -pe2[1], c2z = -pe2[2];
  /* x = c1 x c2 (ray independent; volatile keeps the compiler from contracting the scalar products) */
  volatile float xa = c1y * c2z, xb = -c1z * c2y, ya = c1z * c2x, yb = -c1x * c2z, za = c1x * c2y, zb = -c1y * c2x;
  const float xx = xa + xb, xy = ya + yb, xz = za + zb;
  /* y = c2 x c0, z = c0 x c1: (a.y*b.z) + (-a.z*b.y), ... */
  v8 y = V8(add8(mul8(S8(c2y), d.z), mul8(S8(-c2z), d.y)), add8(mul8(S8(c2z), d.x), mul8(S8(-c2x), d.z)),
            add8(mul8(S8(c2x), d.y), mul8(S8(-c2y), d.x)));
  v8 z = V8(add8(mul8(d.y, S8(c1z)), mul8(neg8(d.z), S8(c1y))), add8(mul8(d.z, S8(c1x)), mul8(neg8(d.x), S8(c1z))),
            add8(mul8(d.x, S8(c1y)), mul8(neg8(d.y), S8(c1x))));
  f8 det_i = fma8(d.x, S8(xx), fma8(d.y, S8(xy), mul8(d.z, S8(xz))));
  f8 inv_det = div8(S8(1.0f), det_i);
  v8 r0 = V8(mul8(S8(xx), inv_det), mul8(S8(xy), inv_det), mul8(S8(xz), inv_det));
  v8 r1 = vscale8(y, inv_det), r2 = vscale8(z, inv_det);
  f8 t = add8(add8(mul8(r0.x, b.x), mul8(r0.y, b.y)), mul8(r0.z, b.z));
  f8 u = add8(add8(mul8(r1.x, b.x), mul8(r1.y, b.y)), mul8(r1.z, b.z));
  f8 w = add8(add8(mul8(r2.x, b.x), mul8(r2.y, b.y)), mul8(r2.z, b.z));
  /* determinant(): c0.x*(c1.y*c2.z - c2.y*c1.z) - c1.x*(c0.y*c2.z - c2.y*c0.z) + c2.x*(c0.y*c1.z - c1.y*c0.z) */
  volatile float m0a = c1y * c2z, m0b = c2y * c1z;
  const float m0 = m0a - m0b;
  f8 det = add8(sub8(mul8(d.x, S8(m0)), mul8(S8(c1x), sub8(mul8(d.y, S8(c2z)), mul8(S8(c2y), d.z)))),
                mul8(S8(c2x), sub8(mul8(d.y, S8(c1z)), mul8(S8(c1y), d.z))));
  f8 bad = or8(or8(_mm256_cmp_ps(t, S8(RT_EPS), _CMP_LE_OQ), _mm256_cmp_ps(u, S8(0.0f), _CMP_LT_OQ)),
               or8(_mm256_cmp_ps(w, S8(0.0f), _CMP_LT_OQ), _mm256_cmp_ps(add8(u, w), S8(1.0f), _CMP_GE_OQ)));
  f8 det0 = _mm256_cmp_ps(abs8(sub8(det, S8(0.0f))), S8(RT_EPS), _CMP_LE_OQ);
  *t_out = t;
  return and8(live, andn8(or8(bad, det0), true8()));
}

/* ---- per-thread context ----------------------------------------------------------------------------------- */
typedef struct {
  const rt_scene_desc* s;
  const rt_params* p;
  int cull;
  uint32_t n_lights; /* lights x light_mult */
  /* expanded light list, per lane (a packet of 8 pixels has 8 clouds; the samples of one pixel share one) */
  float* lpos; /* [n][3][8] */
  float* lcol; /* [n][3] */
  float* lint; /* [n] */
  uint64_t rays[3], shadow;
  uint32_t literal; /* RT_SIMD_LITERAL_*: reproduce packet-coupled behaviours of the reference (rt_simd_render_ex) */
} ctx8;

/* Packet-literal modes: the two places where the reference's result depends on WHICH rays share an 8-lane packet.
 * The parity oracle and the GPU path decide them per lane / per pixel (deviations D3, D4 of DESIGN.md); these modes
 * exist to measure how much that changes an image (tests/test_simd_baseline.py). */
#define RT_SIMD_LITERAL_D3 1u /* refraction depth step / factor from the packet's horizontal max opacity, :458-491 */
#define RT_SIMD_LITERAL_D4 2u /* without anti-aliasing one light cloud per 8-pixel packet, :1256-1280 */

enum { KIND_PRIMARY = 0, KIND_REFL = 1, KIND_REFR = 2 };

typedef struct {
  f8 valid, t;
  __m256i id; /* canonical object index, -1 = none */
  v8 p, n;
  mat8 m;
} hit8;

/* Raytracer::cast_ray, raytracer.rs:162-220: linear scan, per lane replace if new.t <= cur.t (ties -> later object) */
static hit8 nearest8(const ctx8* c, v8 o, v8 d, f8 active) {
  const rt_scene_desc* s = c->s;
  f8 best_t = S8(INFINITY), best_v = false8();
  __m256i best_id = _mm256_set1_epi32(-1);
  for (uint32_t i = 0; i < s->n_spheres; i++) {
    f8 t;
    f8 h = and8(sphere8(s, i, o, d, c->cull, &t, NULL), active);
    f8 take = and8(h, or8(not8(best_v), _mm256_cmp_ps(t, best_t, _CMP_LE_OQ)));
    best_t = sel8(take, t, best_t);
    best_id = _mm256_castps_si256(sel8(take, _mm256_castsi256_ps(_mm256_set1_epi32((int)i)), _mm256_castsi256_ps(best_id)));
    best_v = or8(best_v, take);
  }
  for (uint32_t i = 0; i < s->n_triangles; i++) {
    f8 t;
    f8 h = and8(triangle8(s, i, o, d, c->cull, &t), active);
    if (!any8(h)) continue;
    f8 take = and8(h, or8(not8(best_v), _mm256_cmp_ps(t, best_t, _CMP_LE_OQ)));
    best_t = sel8(take, t, best_t);
    best_id = _mm256_castps_si256(
        sel8(take, _mm256_castsi256_ps(_mm256_set1_epi32((int)(s->n_spheres + i))), _mm256_castsi256_ps(best_id)));
    best_v = or8(best_v, take);
  }
  hit8 r;
  r.valid = best_v;
  r.t = best_t;
  r.id = best_id;
  r.p = vfma_s8(d, best_t, o);
  /* SurfaceInteraction of the winning object per lane (surface_interaction.rs:55-64): gather */
  float cx[8], cy[8], cz[8], nx[8], ny[8], nz[8], mr[8], mg[8], mb[8], mm[8], ms[8], mi[8], mo[8], mbo[8];
  int ids[8], sph[8], tr[8];
  _mm256_storeu_si256((__m256i*)ids, best_id);
  for (int l = 0; l < 8; l++) {
    cx[l] = cy[l] = cz[l] = nx[l] = ny[l] = 0.0f, nz[l] = 1.0f;
    mr[l] = mg[l] = mb[l] = mm[l] = ms[l] = mi[l] = mo[l] = mbo[l] = 0.0f;
    sph[l] = tr[l] = 0;
    if (ids[l] < 0) continue;
    uint32_t mat;
    if ((uint32_t)ids[l] < s->n_spheres) {
      sph[l] = -1;
      cx[l] = s->sphere_center[3 * ids[l]], cy[l] = s->sphere_center[3 * ids[l] + 1], cz[l] = s->sphere_center[3 * ids[l] + 2];
      mat = s->sphere_material[ids[l]];
    } else {
      uint32_t ti = (uint32_t)ids[l] - s->n_spheres;
      nx[l] = s->tri_normal[3 * (size_t)ti], ny[l] = s->tri_normal[3 * (size_t)ti + 1], nz[l] = s->tri_normal[3 * (size_t)ti + 2];
      mat = s->tri_material[ti];
    }
    mat1 m = load_mat1(s, mat);
    mr[l] = m.r, mg[l] = m.g, mb[l] = m.b, mm[l] = m.metallic, ms[l] = m.shininess, mi[l] = m.ior, mo[l] = m.opacity, mbo[l] = m.boost;
    tr[l] = m.transmissive ? -1 : 0;
  }
  f8 is_sph = _mm256_castsi256_ps(_mm256_loadu_si256((const __m256i*)sph));
  v8 ns = vnormalize8(vsub8(r.p, V8(_mm256_loadu_ps(cx), _mm256_loadu_ps(cy), _mm256_loadu_ps(cz))));
  r.n = vsel8(is_sph, ns, V8(_mm256_loadu_ps(nx), _mm256_loadu_ps(ny), _mm256_loadu_ps(nz)));
  r.m.color = V8(_mm256_loadu_ps(mr), _mm256_loadu_ps(mg), _mm256_loadu_ps(mb));
  r.m.metallic = _mm256_loadu_ps(mm), r.m.shininess = _mm256_loadu_ps(ms), r.m.ior = _mm256_loadu_ps(mi);
  r.m.opacity = _mm256_loadu_ps(mo), r.m.boost = _mm256_loadu_ps(mbo);
  r.m.transmissive = _mm256_castsi256_ps(_mm256_loadu_si256((const __m256i*)tr));
  return r;
}

typedef struct {
  f8 occluded, opacity;
  v8 filter;
} shadow8_t;

/* Raytracer::has_any_intersection, raytracer.rs:24-106: every object; a lane stops once it is completely occluded,
 * the loop breaks when all live lanes are */
static shadow8_t shadow8(ctx8* c, v8 from, v8 dir_raw, f8 tmax, f8 active) {
  const rt_scene_desc* s = c->s;
  c->shadow += (uint64_t)__builtin_popcount(bits8(active));
  v8 d = vnormalize8(dir_raw);
  shadow8_t r;
  r.occluded = false8();
  r.opacity = S8(1.0f);
  r.filter = vsplat(1.0f, 1.0f, 1.0f);
  const uint32_t n = s->n_spheres + s->n_triangles;
  for (uint32_t k = 0; k < n; k++) {
    f8 t, h;
    v8 nrm;
    uint32_t mat;
    if (k < s->n_spheres) {
      h = sphere8(s, k, from, d, c->cull, &t, &nrm);
      mat = s->sphere_material[k];
    } else {
      const uint32_t ti = k - s->n_spheres;
      h = triangle8(s, ti, from, d, c->cull, &t);
      if (!any8(h)) continue;
      nrm = vsplat(s->tri_normal[3 * (size_t)ti], s->tri_normal[3 * (size_t)ti + 1], s->tri_normal[3 * (size_t)ti + 2]);
      mat = s->tri_material[ti];
    }
    h = and8(and8(h, _mm256_cmp_ps(t, tmax, _CMP_LE_OQ)), andn8(r.occluded, active));
    if (!any8(h)) continue;
    const mat1 m = load_mat1(s, mat);
    f8 io = S8(0.0f);
    if (m.transmissive) {
      v8 refl = fresnel8(vsplat(m.r, m.g, m.b), S8(m.metallic), S8(m.ior), true8(), nrm, vneg8(d), S8(1.0f));
      io = mul8(S8(m.opacity), sub8(S8(1.0f), refl.x));
    }
    f8 nop = clamp8(sub8(r.opacity, sub8(S8(1.0f), io)), S8(0.0f), S8(1.0f));
    r.opacity = sel8(h, nop, r.opacity);
    if (!m.transmissive) r.occluded = or8(r.occluded, and8(h, _mm256_cmp_ps(abs8(sub8(nop, S8(0.0f))), S8(RT_EPS), _CMP_LE_OQ)));
    float op = m.transmissive ? m.opacity : 1.0f;
    op = fminf(fmaxf(op, 0.0f), 1.0f - RT_EPS);
    const float k1 = 1.0f - op;
    v8 nf = vsub8(r.filter, vsplat(m.r * k1, m.g * k1, m.b * k1)); /* Material::absorption, material.rs:213-231 */
    r.filter = vsel8(h, nf, r.filter);
    if (!any8(andn8(r.occluded, active))) break; /* :94-96 */
  }
  return r;
}

static inline f8 map_lanes(f8 v, f8 mask, float (*fn)(float)) {
  float a[8];
  _mm256_storeu_ps(a, v);
  const int m = bits8(mask);
  for (int l = 0; l < 8; l++) a[l] = ((m >> l) & 1) ? fn(a[l]) : 0.0f;
  return _mm256_loadu_ps(a);
}
static inline f8 pow_lanes(f8 b, f8 e, f8 mask) {
  float x[8], y[8];
  _mm256_storeu_ps(x, b);
  _mm256_storeu_ps(y, e);
  const int m = bits8(mask);
  for (int l = 0; l < 8; l++) x[l] = ((m >> l) & 1) ? powf(x[l], y[l]) : 0.0f;
  return _mm256_loadu_ps(x);
}

/* calculate_lighting, raytracer_renderer.rs:731-874 + PointLight::calculate_contribution_at, light.rs:261-299 */
static void lighting8(ctx8* c, const hit8* h, v8 view, f8 active, v8* out_direct, v8* out_spec) {
  const rt_params* P = c->p;
  const v8 mc = h->m.color;
  v8 ambient = vscale8(vmul8(mc, vsplat(1.0f, 1.0f, 1.0f)), S8(P->ambient));
  v8 light_color = vsplat(0, 0, 0), spec_color = vsplat(0, 0, 0);
  const f8 has_spec = _mm256_cmp_ps(h->m.shininess, S8(0.0f), _CMP_GT_OQ);
  const v8 epsv = vsplat(P->eps_distance, P->eps_distance, P->eps_distance);
  for (uint32_t li = 0; li < c->n_lights; li++) {
    const float* lp8 = c->lpos + (size_t)li * 24;
    v8 lp = V8(_mm256_loadu_ps(lp8), _mm256_loadu_ps(lp8 + 8), _mm256_loadu_ps(lp8 + 16));
    v8 lc = vsplat(c->lcol[3 * li], c->lcol[3 * li + 1], c->lcol[3 * li + 2]);
    f8 lI = S8(c->lint[li]);
    v8 ltp = vsub8(lp, h->p);
    v8 ld = vnormalize8(ltp);
    v8 so = vadd8(h->p, vmul8(ld, epsv));
    f8 tmax = vmag8(vsub8(lp, so));
    shadow8_t S = shadow8(c, so, ld, tmax, active);
    f8 reach = andn8(S.occluded, active);
    if (!any8(reach)) continue;
    f8 dist = add8(vmag8(ltp), S8(RT_EPS));
    f8 cosi = div8(vdot8(ltp, h->n), dist);
    f8 pos = _mm256_cmp_ps(cosi, S8(0.0f), _CMP_GT_OQ);
    f8 att = mul8(S8(0.95f), add8(add8(S8(RT_EPS), dist), mul8(dist, dist)));
    f8 sig = div8(add8(map_lanes(att, reach, tanhf), S8(1.0f)), S8(2.0f));
    f8 lf = mul8(mul8(cosi, lI), clamp8(sig, S8(0.0f), S8(1.0f)));
    v8 ccol = vsel8(pos, vmul8(mc, lc), vsplat(0, 0, 0));
    f8 cint = sel8(pos, lf, S8(0.0f));
    v8 Lc = vdiv8(ccol, S.filter);
    f8 diff = _mm256_max_ps(vdot8(h->n, ld), S8(0.0f));
    f8 specf = S8(0.0f);
    f8 spec_lanes = and8(has_spec, reach);
    if (any8(spec_lanes)) {
      v8 rr = vnormalize8(vreflected8(ld, h->n));
      f8 base = _mm256_max_ps(vdot8(rr, view), S8(0.0f));
      specf = pow_lanes(base, _mm256_max_ps(mul8(h->m.shininess, S8(512.0f)), S8(1.0f)), spec_lanes);
    }
    f8 light_factor = mul8(mul8(diff, cint), S.opacity);
    f8 spec_factor = mul8(mul8(cint, S.opacity), specf);
    f8 lit = and8(reach, _mm256_cmp_ps(diff, S8(0.0f), _CMP_GT_OQ));
    light_color = vsel8(lit, vadd8(light_color, vscale8(vmul8(mc, Lc), light_factor)), light_color);
    spec_color = vsel8(and8(lit, has_spec), vadd8(spec_color, vscale8(lc, spec_factor)), spec_color);
  }
  *out_direct = vadd8(ambient, light_color);
  *out_spec = spec_color;
}

typedef struct {
  f8 hit, t;
  v8 color;
  __m256i id;
} trace8_t;

static trace8_t trace8(ctx8* c, v8 o, v8 d_raw, f8 n_start, __m256i depth, int kind, f8 active);

static inline __m256i sel8i(f8 m, __m256i a, __m256i b) {
  return _mm256_castps_si256(sel8(m, _mm256_castsi256_ps(a), _mm256_castsi256_ps(b)));
}

/* calculate_reflection, raytracer_renderer.rs:526-729 */
static v8 reflection8(ctx8* c, const hit8* h, v8 view, f8 n_start, __m256i depth, f8 active) {
  const rt_params* P = c->p;
  f8 cos_theta = vdot8(view, h->n);
  f8 inside = _mm256_cmp_ps(cos_theta, S8(0.0f), _CMP_LT_OQ);
  v8 inormal = vsel8(inside, vneg8(h->n), h->n);
  f8 n2 = sel8(inside, h->m.ior, S8(P->air_ior));
  f8 eta = sel8(inside, div8(n2, n_start), div8(n_start, n2));
  f8 cos_i = abs8(cos_theta);
  f8 sin2 = mul8(mul8(eta, eta), sub8(S8(1.0f), mul8(cos_i, cos_i)));
  f8 tir = _mm256_cmp_ps(sin2, S8(1.0f), _CMP_GE_OQ);
  f8 reflective = or8(_mm256_cmp_ps(h->m.metallic, S8(0.0f), _CMP_GT_OQ), and8(h->m.transmissive, tir));
  f8 on = and8(active, reflective);
  if (!any8(on)) return vsplat(0, 0, 0);
  v8 r = vnormalize8(vreflected8(view, h->n));
  v8 Rf = fresnel8(h->m.color, h->m.metallic, h->m.ior, h->m.transmissive, inormal, vneg8(view), n_start);
  /* depth.map(d -> d - 1).or(REFL_MAX) */
  const __m256i zero = _mm256_setzero_si256(), one = _mm256_set1_epi32(1);
  f8 none = _mm256_castsi256_ps(_mm256_cmpgt_epi32(zero, depth));
  __m256i dec = _mm256_max_epi32(_mm256_sub_epi32(depth, one), zero);
  __m256i child = sel8i(none, _mm256_set1_epi32((int)P->max_depth_reflection), dec);
  v8 epsv = vsplat(P->eps_distance, P->eps_distance, P->eps_distance);
  trace8_t ch = trace8(c, vadd8(h->p, vmul8(r, epsv)), r, n_start, child, KIND_REFL, on);
  v8 res = vmul8(vscale8(ch.color, atten8(ch.t)), Rf);
  return vsel8(and8(on, ch.hit), res, vsplat(0, 0, 0));
}

/* calculate_refractions, raytracer_renderer.rs:279-524 */
static v8 refraction8(ctx8* c, const hit8* h, v8 view, f8 n_start, __m256i depth, f8 active) {
  const rt_params* P = c->p;
  f8 on = and8(active, h->m.transmissive);
  if (!any8(on)) return vsplat(0, 0, 0);
  f8 cos_theta = vdot8(view, h->n);
  f8 inside = _mm256_cmp_ps(cos_theta, S8(0.0f), _CMP_LE_OQ);
  v8 inormal = vsel8(inside, vneg8(h->n), h->n);
  f8 n2 = sel8(inside, h->m.ior, S8(P->air_ior));
  f8 eta = sel8(inside, div8(n2, n_start), div8(n_start, n2));
  f8 inv_eta = div8(S8(1.0f), eta);
  v8 Rf = fresnel8(h->m.color, h->m.metallic, h->m.ior, h->m.transmissive, inormal, view, inv_eta);
  v8 Tr = V8(sub8(S8(1.0f), Rf.x), sub8(S8(1.0f), Rf.y), sub8(S8(1.0f), Rf.z));
  /* ultraviolet refracted(i = view, n = -inormal, eta = 1/eta): zero vector when k < 0 -> NaN after normalisation */
  v8 nn = vneg8(inormal);
  f8 ndi = vdot8(nn, view);
  f8 k = sub8(S8(1.0f), mul8(mul8(inv_eta, inv_eta), sub8(S8(1.0f), mul8(ndi, ndi))));
  f8 kneg = _mm256_cmp_ps(k, S8(0.0f), _CMP_LT_OQ);
  f8 sq = add8(mul8(inv_eta, ndi), _mm256_sqrt_ps(k));
  v8 q0 = vsub8(vscale8(view, inv_eta), vscale8(nn, sq));
  v8 q = vnormalize8(vsel8(kneg, vsplat(0, 0, 0), q0));
  /* depth step / factor from the lane's own opacity (deviation D3), or -- packet-literal mode -- from the horizontal
   * max over the packet of opacity().simd_unwrap_or(0) (:458-463; non-transmissive and idle lanes count as 0) */
  f8 op_eff = h->m.opacity;
  if (c->literal & RT_SIMD_LITERAL_D3) {
    float ops[8], mx = 0.0f;
    _mm256_storeu_ps(ops, and8(h->m.opacity, and8(h->m.transmissive, h->valid)));
    for (int l = 0; l < 8; l++) mx = ops[l] > mx ? ops[l] : mx;
    op_eff = S8(mx);
  }
  f8 lt05 = _mm256_cmp_ps(op_eff, S8(0.5f), _CMP_LT_OQ), le03 = _mm256_cmp_ps(op_eff, S8(0.3f), _CMP_LE_OQ);
  const __m256i zero = _mm256_setzero_si256();
  __m256i step = sel8i(lt05, _mm256_set1_epi32(2), _mm256_set1_epi32(1));
  const int md = (int)P->max_depth_refraction;
  __m256i first = sel8i(le03, _mm256_set1_epi32(md / 3), sel8i(lt05, _mm256_set1_epi32(md / 2), _mm256_set1_epi32(md)));
  f8 none = _mm256_castsi256_ps(_mm256_cmpgt_epi32(zero, depth));
  __m256i dec = _mm256_max_epi32(_mm256_sub_epi32(depth, step), zero);
  __m256i child = sel8i(none, first, dec);
  v8 epsv = vsplat(P->eps_distance, P->eps_distance, P->eps_distance);
  trace8_t ch = trace8(c, vadd8(h->p, vmul8(q, epsv)), q, n2, child, KIND_REFR, on);
  v8 res = vmul8(vscale8(ch.color, add8(h->m.boost, S8(1.0f))), Tr);
  return vsel8(and8(on, ch.hit), res, vsplat(0, 0, 0));
}

/* single_raytrace, raytracer_renderer.rs:147-264 */
static trace8_t trace8(ctx8* c, v8 o, v8 d_raw, f8 n_start, __m256i depth, int kind, f8 active) {
  trace8_t res;
  res.hit = false8();
  res.t = S8(0.0f);
  res.color = vsplat(0, 0, 0);
  res.id = _mm256_set1_epi32(-1);
  /* depth == Some(0) -> None (:174-178) */
  active = andn8(_mm256_castsi256_ps(_mm256_cmpeq_epi32(depth, _mm256_setzero_si256())), active);
  v8 d = vnormalize8(d_raw);
  active = andn8(or8(or8(isnan8(d.x), isnan8(d.y)), isnan8(d.z)), active); /* deviation D2 */
  if (!any8(active)) return res;
  c->rays[kind] += (uint64_t)__builtin_popcount(bits8(active));
  hit8 h = nearest8(c, o, d, active);
  f8 hit = and8(active, h.valid);
  if (!any8(hit)) return res;
  v8 direct, spec;
  lighting8(c, &h, d, hit, &direct, &spec);
  f8 a = atten8(h.t);
  direct = vscale8(direct, a);
  spec = vscale8(spec, a);
  f8 T = h.m.transmissive;
  f8 R = or8(_mm256_cmp_ps(h.m.metallic, S8(0.0f), _CMP_GT_OQ), T);
  v8 refl = vsplat(0, 0, 0), refr = vsplat(0, 0, 0);
  if ((c->p->flags & RT_FLAG_REFLECTIONS) && any8(and8(hit, R))) refl = reflection8(c, &h, d, n_start, depth, and8(hit, R));
  if ((c->p->flags & RT_FLAG_REFRACTIONS) && any8(and8(hit, T))) refr = refraction8(c, &h, d, n_start, depth, and8(hit, T));
  res.hit = hit;
  res.t = h.t;
  res.id = sel8i(hit, h.id, _mm256_set1_epi32(-1));
  v8 ct = vadd8(vadd8(refl, refr), spec), co = vadd8(vadd8(direct, refl), spec);
  res.color = vsel8(hit, vsel8(T, ct, co), vsplat(0, 0, 0));
  return res;
}

static inline uint32_t to_u8(float x) {
  float cx = fminf(fmaxf(x, 0.0f), 1.0f);
  return (uint32_t)lrintf(cx * 255.0f);
}
static inline uint32_t pack_pixel(float r, float g, float b) { return 0xFF000000u | (to_u8(r) << 16) | (to_u8(g) << 8) | to_u8(b); }

/* light cloud of `pixel` into lane `lane` (all lanes when lane < 0): light.rs:183-225,311-324, seeded (D4) */
static void build_lights8(ctx8* c, uint32_t pixel, int lane) {
  const rt_scene_desc* s = c->s;
  const rt_params* P = c->p;
  const uint32_t N = P->light_mult < 1 ? 1 : P->light_mult;
  uint32_t k = 0;
  for (uint32_t l = 0; l < s->n_lights; l++) {
    const float* L = s->lights + (size_t)l * RT_LIGHT_STRIDE;
    const float* cs = NULL;
    if (N > 1) cs = P->cloud_sets + (size_t)(rt_cloud_hash(P->cloud_seed, pixel, l) % P->n_cloud_sets) * N * 3;
    for (uint32_t j = 0; j < N; j++, k++) {
      float pos[3] = {L[0], L[1], L[2]};
      if (N > 1) {
        pos[0] = L[0] + cs[3 * j] * P->fw;
        pos[1] = L[1] + cs[3 * j + 1] * P->fh;
        pos[2] = L[2] + cs[3 * j + 2] * P->fd;
      }
      for (int a = 0; a < 3; a++) {
        float* dst = c->lpos + (size_t)k * 24 + 8 * a;
        if (lane < 0)
          for (int q = 0; q < 8; q++) dst[q] = pos[a];
        else
          dst[lane] = pos[a];
      }
      c->lcol[3 * k] = L[3], c->lcol[3 * k + 1] = L[4], c->lcol[3 * k + 2] = L[5];
      c->lint[k] = N > 1 ? (1.0f / (float)N) * L[6] : L[6];
    }
  }
  c->n_lights = k;
}

typedef struct {
  const rt_scene_desc* s;
  const rt_params* p;
  uint32_t* argb;
  const rt_aux* aux;
  uint32_t x0, y0, w, h, tiles_x, n_tiles;
  volatile uint32_t* next_tile;
  uint64_t rays[3], shadow, written;
  uint32_t literal;
} job8;

static void write_pixel(job8* j, uint32_t pix, int any, float r, float g, float b, int id0, float t0) {
  if (j->aux && j->aux->hit_id) j->aux->hit_id[pix] = id0;
  if (j->aux && j->aux->hit_t && id0 >= 0) j->aux->hit_t[pix] = t0;
  if (!any) return;
  j->argb[pix] = pack_pixel(r, g, b);
  if (j->aux && j->aux->rgb) j->aux->rgb[3 * (size_t)pix] = r, j->aux->rgb[3 * (size_t)pix + 1] = g, j->aux->rgb[3 * (size_t)pix + 2] = b;
  j->written++;
}

/* one row segment [gx0, gx1) of a tile: render_pixel_colors, raytracer_renderer.rs:1190-1357 */
static void render_row(ctx8* c, job8* j, uint32_t gy, uint32_t gx0, uint32_t gx1) {
  const rt_params* P = c->p;
  const v8 focus = vsplat(P->focus[0], P->focus[1], P->focus[2]);
  const __m256i none = _mm256_set1_epi32(-1);
  const int aa = (P->flags & RT_FLAG_ANTI_ALIASING) && P->aa_rays > 0;
  const uint32_t ts = P->tile_size ? P->tile_size : 48u;
  if (aa) {
    /* :1199-1221: per pixel, total_rays / 8 packets of 8 sample origins; all share the un-jittered direction */
    const uint32_t n = P->aa_rays, packets = (n + 7) / 8;
    const float scale = 1.0f / (float)(packets * 8);
    for (uint32_t gx = gx0; gx < gx1; gx++) {
      if (P->n_ranks > 1 && rt_tile_owner(gx / ts, gy / ts, P->n_ranks) != P->rank) continue;
      const uint32_t pix = gy * P->width + gx;
      const float x = (float)gx * P->fw, y = (float)gy * P->fh;
      build_lights8(c, pix, -1);
      v8 D = vsub8(vsplat(x, y, 0.0f), focus);
      v8 first = vsplat(0, 0, 0), rest = vsplat(0, 0, 0);
      int any = 0, id0 = -1;
      float t0 = 0.0f;
      for (uint32_t pk = 0; pk < packets; pk++) {
        float ox[8], oy[8];
        int live[8];
        for (int l = 0; l < 8; l++) {
          const uint32_t k = pk * 8 + (uint32_t)l;
          live[l] = k < n ? -1 : 0;
          ox[l] = k < n ? x + P->aa_offsets[2 * k] : x;
          oy[l] = k < n ? y + P->aa_offsets[2 * k + 1] : y;
        }
        f8 act = _mm256_castsi256_ps(_mm256_loadu_si256((const __m256i*)live));
        trace8_t r = trace8(c, V8(_mm256_loadu_ps(ox), _mm256_loadu_ps(oy), S8(0.0f)), D, S8(P->air_ior), none, KIND_PRIMARY, act);
        if (pk == 0) {
          int ids[8];
          float tt[8];
          _mm256_storeu_si256((__m256i*)ids, r.id);
          _mm256_storeu_ps(tt, r.t);
          id0 = ids[0], t0 = tt[0];
        }
        any |= any8(r.hit);
        v8 cs = vsel8(r.hit, vscale8(r.color, S8(scale)), vsplat(0, 0, 0));
        if (pk == 0)
          first = cs;
        else
          rest = vsel8(r.hit, vadd8(cs, rest), rest); /* :998 `color + res_color` */
      }
      v8 lane = vadd8(rest, first);
      float lx[8], ly[8], lz[8];
      _mm256_storeu_ps(lx, lane.x), _mm256_storeu_ps(ly, lane.y), _mm256_storeu_ps(lz, lane.z);
      /* wide f32x8 horizontal sum: ((l0+l4)+(l2+l6)) + ((l1+l5)+(l3+l7)) */
#define HSUM(a) (((a[0] + a[4]) + (a[2] + a[6])) + ((a[1] + a[5]) + (a[3] + a[7])))
      write_pixel(j, pix, any, HSUM(lx), HSUM(ly), HSUM(lz), id0, t0);
    }
    return;
  }
  /* :1256-1297: 8 consecutive pixels of the row = one packet (tails run with the spare lanes masked off) */
  for (uint32_t gx = gx0; gx < gx1; gx += 8) {
    float ox[8], oy[8];
    int live[8];
    uint32_t pixs[8];
    int n_live = 0;
    for (int l = 0; l < 8; l++) {
      const uint32_t px = gx + (uint32_t)l;
      const int on = px < gx1 && !(P->n_ranks > 1 && rt_tile_owner(px / ts, gy / ts, P->n_ranks) != P->rank);
      live[l] = on ? -1 : 0;
      pixs[l] = gy * P->width + (on ? px : gx);
      ox[l] = (float)(on ? px : gx) * P->fw;
      oy[l] = (float)gy * P->fh;
      n_live += on;
    }
    if (!n_live) continue;
    if (c->literal & RT_SIMD_LITERAL_D4)
      build_lights8(c, gy * P->width + gx, -1); /* one cloud for the packet (here: its first pixel's set) */
    else
      for (int l = 0; l < 8; l++) build_lights8(c, pixs[l], l);
    v8 coords = V8(_mm256_loadu_ps(ox), _mm256_loadu_ps(oy), S8(0.0f));
    f8 act = _mm256_castsi256_ps(_mm256_loadu_si256((const __m256i*)live));
    trace8_t r = trace8(c, coords, vsub8(coords, focus), S8(P->air_ior), none, KIND_PRIMARY, act);
    float cr[8], cg[8], cb[8], tt[8];
    int ids[8];
    _mm256_storeu_ps(cr, r.color.x), _mm256_storeu_ps(cg, r.color.y), _mm256_storeu_ps(cb, r.color.z), _mm256_storeu_ps(tt, r.t);
    _mm256_storeu_si256((__m256i*)ids, r.id);
    const int hm = bits8(r.hit);
    for (int l = 0; l < 8; l++)
      if (live[l]) write_pixel(j, pixs[l], (hm >> l) & 1, cr[l], cg[l], cb[l], ids[l], tt[l]);
  }
}

static void* worker8(void* arg) {
  job8* j = (job8*)arg;
  ctx8 c;
  memset(&c, 0, sizeof(c));
  c.s = j->s;
  c.p = j->p;
  c.cull = (j->p->flags & RT_FLAG_BACKFACE_CULLING) != 0;
  c.literal = j->literal;
  const uint32_t N = j->p->light_mult < 1 ? 1 : j->p->light_mult;
  const size_t nl = (size_t)j->s->n_lights * N + 1;
  c.lpos = (float*)aligned_alloc(32, sizeof(float) * 24 * nl);
  c.lcol = (float*)malloc(sizeof(float) * 3 * nl);
  c.lint = (float*)malloc(sizeof(float) * nl);
  const uint32_t ts = 48u; /* RENDER_STRIDE, renderer/mod.rs:84-90 (48 for every named resolution) */
  /* work items: the rows of the tiles, tile after tile (the reference runs tiles AND the rows of a tile through
   * rayon: process_chunks_parallel -> process_rows_parallel, image_buffer.rs:48-97,306-320) */
  for (;;) {
    const uint32_t it = __atomic_fetch_add(j->next_tile, 1u, __ATOMIC_RELAXED);
    if (it >= j->n_tiles * ts) break;
    const uint32_t t = it / ts, row = it % ts;
    const uint32_t tx = t % j->tiles_x, ty = t / j->tiles_x;
    const uint32_t gx0 = j->x0 + tx * ts, gy = j->y0 + ty * ts + row;
    const uint32_t gx1 = gx0 + ts < j->x0 + j->w ? gx0 + ts : j->x0 + j->w;
    if (gy >= j->y0 + j->h) continue;
    render_row(&c, j, gy, gx0, gx1);
  }
  j->rays[0] = c.rays[0], j->rays[1] = c.rays[1], j->rays[2] = c.rays[2];
  j->shadow = c.shadow;
  free(c.lpos);
  free(c.lcol);
  free(c.lint);
  return NULL;
}

/* Same contract as rt_render (include/rt_hip.h), computed on the host with 8-lane packets over 48x48 tiles.
 * n_threads <= 0 -> 1.  The tile grid starts at the window's corner (image_buffer.rs:48-97 tiles the whole buffer). */
int rt_simd_render_ex(const rt_scene_desc* desc, const rt_params* params, uint32_t* argb, const rt_aux* aux, rt_stats* stats,
                      int n_threads, uint32_t literal);
int rt_simd_render(const rt_scene_desc* desc, const rt_params* params, uint32_t* argb, const rt_aux* aux, rt_stats* stats,
                   int n_threads) {
  return rt_simd_render_ex(desc, params, argb, aux, stats, n_threads, 0u);
}
/* literal: RT_SIMD_LITERAL_* bits (0 = the parity oracle's per-lane / per-pixel decisions) */
int rt_simd_render_ex(const rt_scene_desc* desc, const rt_params* params, uint32_t* argb, const rt_aux* aux, rt_stats* stats,
                      int n_threads, uint32_t literal) {
  if (!desc || !params || !argb) return RT_ERR_INVALID_ARG;
  if (desc->abi_version != RT_ABI_VERSION || params->abi_version != RT_ABI_VERSION) return RT_ERR_INVALID_ARG;
  if (params->light_mult > 1 && (params->n_cloud_sets == 0 || !params->cloud_sets)) return RT_ERR_INVALID_ARG;
  if ((params->flags & RT_FLAG_ANTI_ALIASING) && params->aa_rays > 0 && !params->aa_offsets) return RT_ERR_INVALID_ARG;
  uint32_t x0 = 0, y0 = 0, w = params->width, h = params->height;
  if (params->win_w) {
    x0 = params->win_x0, y0 = params->win_y0, w = params->win_w, h = params->win_h;
    if (x0 + w > params->width || y0 + h > params->height) return RT_ERR_INVALID_ARG;
  }
  if (n_threads <= 0) n_threads = 1;
  if (n_threads > 1024) n_threads = 1024;
  struct timespec ts0, ts1;
  clock_gettime(CLOCK_MONOTONIC, &ts0);
  volatile uint32_t next_tile = 0;
  job8* jobs = (job8*)calloc((size_t)n_threads, sizeof(job8));
  pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof(pthread_t));
  const uint32_t tiles_x = (w + 47u) / 48u, tiles_y = (h + 47u) / 48u;
  for (int i = 0; i < n_threads; i++) {
    jobs[i].s = desc, jobs[i].p = params, jobs[i].argb = argb, jobs[i].aux = aux;
    jobs[i].x0 = x0, jobs[i].y0 = y0, jobs[i].w = w, jobs[i].h = h;
    jobs[i].tiles_x = tiles_x, jobs[i].n_tiles = tiles_x * tiles_y;
    jobs[i].next_tile = &next_tile;
    jobs[i].literal = literal;
  }
  if (n_threads == 1) {
    worker8(&jobs[0]);
  } else {
    for (int i = 0; i < n_threads; i++) pthread_create(&th[i], NULL, worker8, &jobs[i]);
    for (int i = 0; i < n_threads; i++) pthread_join(th[i], NULL);
  }
  clock_gettime(CLOCK_MONOTONIC, &ts1);
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    for (int i = 0; i < n_threads; i++) {
      stats->rays_primary += jobs[i].rays[0];
      stats->rays_reflection += jobs[i].rays[1];
      stats->rays_refraction += jobs[i].rays[2];
      stats->rays_shadow += jobs[i].shadow;
      stats->pixels_written += jobs[i].written;
    }
    stats->rays_traced = stats->rays_primary + stats->rays_reflection + stats->rays_refraction;
    stats->kernel_ms = stats->total_ms = (ts1.tv_sec - ts0.tv_sec) * 1e3 + (ts1.tv_nsec - ts0.tv_nsec) * 1e-6;
  }
  free(jobs);
  free(th);
  return RT_OK;
}
