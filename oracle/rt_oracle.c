/*
 * rt_oracle.c -- CPU restatement of the reference's per-pixel render loop.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP path and the
 * `cpu_baseline` ("port") leg of bench.py.  Nothing in the shipped package may import, link or
 * execute it; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * PARITY PINNING: the reference (Rust nightly + git-forked crates) cannot be built here and its
 * own tests never exercise the render path, so no reference
 * golden vector exists for this path: **parity unpinned**.  What pins this file instead are the
 * hand-derived known-answer vectors in tests/golden/ (derived from the formulas below, see
 * tests/golden/make_known_answers.py) and a visual/statistical cross-check against the
 * reference's output.png.
 *
 * Style: one ray == one SIMD lane of the reference's Vec3x8 packet code, written as scalar fp32.
 * Compiled with -ffp-contract=off; every fused multiply-add below is an explicit fmaf() exactly
 * where the reference calls mul_add (Rust never contracts on its own).
 *
 * Reference files followed (paths relative to the reference repo root):
 *   src/renderer/raytracer_renderer.rs   single_raytrace, calculate_lighting,
 *                                        calculate_reflection, calculate_refractions,
 *                                        antialiased_raytrace, render_pixel_colors
 *   src/raytracing/raytracer.rs          cast_ray, has_any_intersection
 *   src/raytracing/material.rs           compute_fresnel, absorption, TransmissionProperties
 *   src/geometry/basic/sphere.rs         SphereData::intersect
 *   src/geometry/basic/triangle.rs       TriangleData::intersect
 *   src/geometry/ray.rs                  Ray::new_with_mask, Ray::at
 *   src/scene/lighting/light.rs          PointLight::calculate_contribution_at, light cloud
 *   src/output/window.rs                 WindowColorEncoder::to_output
 * Third-party arithmetic restated from crate knowledge (not vendored in the reference tree):
 *   ultraviolet 0.10.0 (dot/normalize/reflected/refracted/Mat3::inversed/determinant/lerp),
 *   wide 0.7.32 + simba 0.9.0-fork (select/clamp/powi/powf/tanh), palette 0.7.6-fork (u8 pack).
 *
 * Deliberate deviations from the literal packet code (all listed in DESIGN.md):
 *   D1  miss lanes contribute exactly 0 (the packet code can leak NaN from masked lanes);
 *   D2  a NaN ray direction (zero-vector refract() normalised, raytracer_renderer.rs:418) is a
 *       miss instead of "first triangle with NaN t";
 *   D3  refraction depth step/factor use the lane's own opacity instead of the packet's
 *       horizontal max (raytracer_renderer.rs:458-491); identical for the named scenes;
 *   D4  the unseeded per-pixel Poisson light cloud and AA table are replaced by seeded tables
 *       passed in rt_params;
 *   D5  object order is fixed: spheres (insertion order) then triangles (insertion order).
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/rt_hip.h"

#define RT_EPS 1.1920929e-7f /* f32::EPSILON, src/float_ext.rs:44-45 */

typedef struct {
  float x, y, z;
} v3;

static inline v3 V(float x, float y, float z) {
  v3 r = {x, y, z};
  return r;
}
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vdiv(v3 a, v3 b) { return V(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
/* ultraviolet Vec3::dot: x.mul_add(ox, y.mul_add(oy, z*oz)) */
static inline float vdot(v3 a, v3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
static inline float vmag(v3 a) { return sqrtf(vdot(a, a)); }
/* ultraviolet normalize: v *= 1/mag */
static inline v3 vnormalize(v3 a) {
  float r = 1.0f / vmag(a);
  return vscale(a, r);
}
/* ultraviolet cross: (a.y*b.z) + (-a.z*b.y), ... */
static inline v3 vcross(v3 a, v3 b) {
  return V((a.y * b.z) + (-a.z * b.y), (a.z * b.x) + (-a.x * b.z), (a.x * b.y) + (-a.y * b.x));
}
/* Vec3::mul_add(self, mul, add) = self*mul + add, fused per component; Ray::at ray.rs:60-66 */
static inline v3 vfma_s(v3 d, float t, v3 o) {
  return V(fmaf(d.x, t, o.x), fmaf(d.y, t, o.y), fmaf(d.z, t, o.z));
}
/* ultraviolet reflected: v - (2*dot(v,n))*n */
static inline v3 vreflected(v3 v, v3 n) {
  float k = 2.0f * vdot(v, n);
  return vsub(v, vscale(n, k));
}
/* ultraviolet refracted (GLSL refract); zero vector on total internal reflection */
static inline v3 vrefracted(v3 i, v3 n, float eta) {
  float ndi = vdot(n, i);
  float k = 1.0f - eta * eta * (1.0f - ndi * ndi);
  if (k < 0.0f) return V(0.0f, 0.0f, 0.0f);
  float s = eta * ndi + sqrtf(k);
  return vsub(vscale(i, eta), vscale(n, s));
}
static inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
static inline int has_nan3(v3 a) { return (a.x != a.x) || (a.y != a.y) || (a.z != a.z); }

/* ---- material accessors (material.rs:15-74) --------------------------------------------------- */
typedef struct {
  v3 color;
  float metallic, shininess, ior, opacity, boost;
  int has_opacity;
} mat_t;

static inline mat_t load_mat(const rt_scene_desc* s, uint32_t idx) {
  const float* m = s->materials + (size_t)idx * RT_MATERIAL_STRIDE;
  mat_t r;
  r.color = V(m[RT_MAT_R], m[RT_MAT_G], m[RT_MAT_B]);
  r.metallic = m[RT_MAT_METALLIC];
  r.shininess = m[RT_MAT_SHININESS];
  r.ior = m[RT_MAT_IOR];
  r.opacity = m[RT_MAT_OPACITY];
  r.boost = m[RT_MAT_BOOST];
  r.has_opacity = m[RT_MAT_HAS_OPACITY] != 0.0f;
  return r;
}
/* TransmissionProperties::mask, material.rs:44-50: Some(opacity) and !(|opacity - 0| <= eps) */
static inline int transmissive(const mat_t* m) {
  return m->has_opacity && !(fabsf(m->opacity - 0.0f) <= RT_EPS);
}

/* Material::compute_fresnel, material.rs:468-525 (per lane).  Returns reflectance; transmittance
 * is 1 - reflectance. */
static v3 fresnel_reflectance(const mat_t* m, v3 normal, v3 view, float other_ior) {
  if (!transmissive(m)) return V(m->metallic, m->metallic, m->metallic); /* :483-488 */
  float ior = m->ior;
  float n_dot_v = vdot(normal, view);
  float cos_theta = fabsf(n_dot_v);
  int inside = n_dot_v < 0.0f;
  float eta_t = inside ? (ior / other_ior) : (other_ior / ior);
  float sin2_t = eta_t * eta_t * (1.0f - cos_theta * cos_theta);
  int reflective = m->metallic > 0.0f;
  int tir = (inside && sin2_t > 1.0f) || reflective;
  float q = (other_ior - ior) / (other_ior + ior);
  float f0 = q * q; /* simd_powi(2) */
  /* VectorLerp::lerp(broadcast(f0), color, metallic) = a*(1-t) + b*t */
  float omt = 1.0f - m->metallic;
  v3 f0v = V(f0 * omt + m->color.x * m->metallic, f0 * omt + m->color.y * m->metallic,
             f0 * omt + m->color.z * m->metallic);
  float c1 = 1.0f - cos_theta;
  float c2 = c1 * c1;
  float c5 = c1 * (c2 * c2); /* simd_powi(5) by repeated squaring */
  v3 fres = V(f0v.x + (1.0f - f0v.x) * c5, f0v.y + (1.0f - f0v.y) * c5, f0v.z + (1.0f - f0v.z) * c5);
  float ra = reflective ? m->metallic : 1.0f;
  return tir ? V(ra, ra, ra) : fres;
}

/* Material::absorption, material.rs:213-231 */
static inline v3 absorption(const mat_t* m) {
  float op = transmissive(m) ? m->opacity : 1.0f;
  op = clampf(op, 0.0f, 1.0f - RT_EPS);
  return vscale(m->color, 1.0f - op);
}

/* ---- intersections -------------------------------------------------------------------------- */
typedef struct {
  int valid;
  float t;
  v3 p, n;
  int id;       /* canonical object index */
  uint32_t mat; /* material row */
} hit_t;

/* SphereData::intersect, sphere.rs:78-162 */
static inline int sphere_intersect(const rt_scene_desc* s, uint32_t i, v3 o, v3 d, int cull,
                                   hit_t* h) {
  v3 c = V(s->sphere_center[3 * i], s->sphere_center[3 * i + 1], s->sphere_center[3 * i + 2]);
  v3 v = vsub(o, c);
  float b = 2.0f * vdot(d, v);
  float cc = vdot(v, v) - s->sphere_r_sq[i];
  float disc = fmaf(b, b, (2.0f * -2.0f) * cc);
  if (!(disc >= 0.0f)) return 0;
  float sq = sqrtf(disc);
  float mba = (-b) * 0.5f;
  float sa = sq * 0.5f;
  float t0 = mba - sa;
  float t1 = mba + sa;
  int t0v = t0 >= 0.0f, t1v = t1 >= 0.0f;
  int use0 = t0v && (!t1v || t0 < t1);
  int use1 = t1v && !use0;
  if (!(use0 || use1)) return 0;
  float t = use0 ? t0 : t1;
  v3 p = vfma_s(d, t, o);
  v3 n = vnormalize(vsub(p, c));
  if (cull) { /* sphere.rs:137-151 */
    mat_t m = load_mat(s, s->sphere_material[i]);
    if (!(vdot(d, n) < 0.75f || transmissive(&m))) return 0;
  }
  h->valid = 1;
  h->t = t;
  h->p = p;
  h->n = n;
  h->id = (int)i;
  h->mat = s->sphere_material[i];
  return 1;
}

/* TriangleData::intersect, triangle.rs:149-212; ultraviolet Mat3::inversed/determinant */
static inline int triangle_intersect(const rt_scene_desc* s, uint32_t i, v3 o, v3 d, int cull,
                                     hit_t* h) {
  v3 v1 = V(s->tri_v1[3 * i], s->tri_v1[3 * i + 1], s->tri_v1[3 * i + 2]);
  v3 e1 = V(s->tri_e1[3 * i], s->tri_e1[3 * i + 1], s->tri_e1[3 * i + 2]);
  v3 e2 = V(s->tri_e2[3 * i], s->tri_e2[3 * i + 1], s->tri_e2[3 * i + 2]);
  v3 nrm = V(s->tri_normal[3 * i], s->tri_normal[3 * i + 1], s->tri_normal[3 * i + 2]);
  if (cull) { /* triangle.rs:154-168 */
    mat_t m = load_mat(s, s->tri_material[i]);
    if (!(vdot(d, nrm) < 0.75f || transmissive(&m))) return 0;
  }
  v3 b = vsub(v1, o);
  v3 c0 = d, c1 = vneg(e1), c2 = vneg(e2);
  /* inversed(): rows (c1 x c2, c2 x c0, c0 x c1) * (1 / c0.(c1 x c2)) */
  v3 x = vcross(c1, c2);
  v3 y = vcross(c2, c0);
  v3 z = vcross(c0, c1);
  float det_i = vdot(c0, x);
  float inv_det = 1.0f / det_i;
  v3 r0 = vscale(x, inv_det), r1 = vscale(y, inv_det), r2 = vscale(z, inv_det);
  /* Mat3 * Vec3: a.x*v.x + b.x*v.y + c.x*v.z (columns of the transposed matrix) */
  float t = r0.x * b.x + r0.y * b.y + r0.z * b.z;
  float u = r1.x * b.x + r1.y * b.y + r1.z * b.z;
  float vv = r2.x * b.x + r2.y * b.y + r2.z * b.z;
  /* determinant(): cofactor expansion along the first row of columns */
  float det = c0.x * (c1.y * c2.z - c2.y * c1.z) - c1.x * (c0.y * c2.z - c2.y * c0.z) +
              c2.x * (c0.y * c1.z - c1.y * c0.z);
  int t_invalid = t <= RT_EPS;
  int uv_invalid = (u < 0.0f) || (vv < 0.0f) || ((u + vv) >= 1.0f);
  int valid = !(t_invalid || uv_invalid) && !(fabsf(det - 0.0f) <= RT_EPS);
  if (!valid) return 0;
  h->valid = 1;
  h->t = t;
  h->p = vfma_s(d, t, o);
  h->n = nrm;
  h->id = (int)(s->n_spheres + i);
  h->mat = s->tri_material[i];
  return 1;
}

/* ---- render context --------------------------------------------------------------------------- */
typedef struct {
  const rt_scene_desc* s;
  const rt_params* p;
  int cull;
  /* expanded light list for the current pixel (lights x light_mult) */
  uint32_t n_lights;
  float* lpos; /* [n][3] */
  float* lcol; /* [n][3] */
  float* lint; /* [n] */
  uint64_t rays[3]; /* primary, reflection, refraction */
  uint64_t shadow;
} ctx_t;

enum { KIND_PRIMARY = 0, KIND_REFL = 1, KIND_REFR = 2 };

/* Raytracer::cast_ray, raytracer.rs:162-220: nearest valid hit, ties go to the LATER object */
static hit_t nearest(const ctx_t* c, v3 o, v3 d) {
  const rt_scene_desc* s = c->s;
  hit_t best;
  best.valid = 0;
  best.t = INFINITY;
  best.id = -1;
  hit_t h;
  for (uint32_t i = 0; i < s->n_spheres; i++) {
    h.valid = 0;
    if (sphere_intersect(s, i, o, d, c->cull, &h)) {
      if (!best.valid || h.t <= best.t) best = h;
    }
  }
  for (uint32_t i = 0; i < s->n_triangles; i++) {
    h.valid = 0;
    if (triangle_intersect(s, i, o, d, c->cull, &h)) {
      if (!best.valid || h.t <= best.t) best = h;
    }
  }
  return best;
}

typedef struct {
  int occluded;
  float opacity;
  v3 filter;
} shadow_t;

/* Raytracer::has_any_intersection, raytracer.rs:24-106 */
static shadow_t shadow_test(ctx_t* c, v3 from, v3 dir_raw, float tmax) {
  const rt_scene_desc* s = c->s;
  c->shadow++;
  v3 d = vnormalize(dir_raw); /* Ray::new_with_mask re-normalises, ray.rs:52-57 */
  shadow_t r;
  r.occluded = 0;
  r.opacity = 1.0f;
  r.filter = V(1.0f, 1.0f, 1.0f);
  uint32_t n = s->n_spheres + s->n_triangles;
  for (uint32_t k = 0; k < n; k++) {
    hit_t h;
    h.valid = 0;
    int ok = (k < s->n_spheres) ? sphere_intersect(s, k, from, d, c->cull, &h)
                                : triangle_intersect(s, k - s->n_spheres, from, d, c->cull, &h);
    if (!ok) continue;
    if (!(h.t <= tmax)) continue; /* :53-55 (the `& t >= 0` line is a no-op statement) */
    mat_t m = load_mat(s, h.mat);
    int tr = transmissive(&m);
    float io = 0.0f;
    if (tr) {
      v3 refl = fresnel_reflectance(&m, h.n, vneg(d), 1.0f);
      float trans_red = 1.0f - refl.x;
      io = m.opacity * trans_red;
    }
    r.opacity = clampf(r.opacity - (1.0f - io), 0.0f, 1.0f);
    if (!tr && fabsf(r.opacity - 0.0f) <= RT_EPS) r.occluded = 1;
    r.filter = vsub(r.filter, absorption(&m));
    if (r.occluded) break; /* :94-96, per lane */
  }
  return r;
}

/* attenuation_factor_based_on_distance, raytracer_renderer.rs:266-277 */
static inline float atten(float t) {
  float d = fabsf(t);
  float a = 1.0f / (1.0f + d + 0.1f * d * d);
  return clampf(a, 0.0f, 1.0f);
}

/* calculate_lighting, raytracer_renderer.rs:731-874 + PointLight::calculate_contribution_at,
 * light.rs:261-299 */
static void lighting(ctx_t* c, const hit_t* h, const mat_t* m, v3 view, v3* out_direct,
                     v3* out_spec) {
  const rt_params* P = c->p;
  v3 mc = m->color;
  v3 ambient = vscale(vmul(mc, V(1.0f, 1.0f, 1.0f)), P->ambient);
  v3 light_color = V(0, 0, 0), spec_color = V(0, 0, 0);
  int has_spec = m->shininess > 0.0f;
  v3 epsv = V(P->eps_distance, P->eps_distance, P->eps_distance);
  for (uint32_t li = 0; li < c->n_lights; li++) {
    v3 lp = V(c->lpos[3 * li], c->lpos[3 * li + 1], c->lpos[3 * li + 2]);
    v3 lc = V(c->lcol[3 * li], c->lcol[3 * li + 1], c->lcol[3 * li + 2]);
    float lI = c->lint[li];
    v3 ltp = vsub(lp, h->p);
    v3 ld = vnormalize(ltp);
    v3 so = vadd(h->p, vmul(ld, epsv));
    float tmax = vmag(vsub(lp, so));
    shadow_t S = shadow_test(c, so, ld, tmax);
    if (S.occluded) continue;
    /* contribution */
    float dist = vmag(ltp) + RT_EPS;
    float cosi = vdot(ltp, h->n) / dist;
    int pos = cosi > 0.0f;
    float att = 0.95f * (RT_EPS + dist + dist * dist);
    float sig = (tanhf(att) + 1.0f) / 2.0f;
    float lf = cosi * lI * clampf(sig, 0.0f, 1.0f);
    v3 ccol = pos ? vmul(mc, lc) : V(0, 0, 0);
    float cint = pos ? lf : 0.0f;
    v3 Lc = vdiv(ccol, S.filter);
    float diff = fmaxf(vdot(h->n, ld), 0.0f);
    float specf = 0.0f;
    if (has_spec) {
      v3 rr = vnormalize(vreflected(ld, h->n));
      float base = fmaxf(vdot(rr, view), 0.0f);
      specf = powf(base, fmaxf(m->shininess * 512.0f, 1.0f));
    }
    float light_factor = diff * cint * S.opacity;
    float spec_factor = cint * S.opacity * specf;
    if (diff > 0.0f) {
      light_color = vadd(light_color, vscale(vmul(mc, Lc), light_factor));
      if (has_spec) spec_color = vadd(spec_color, vscale(lc, spec_factor));
    }
  }
  *out_direct = vadd(ambient, light_color);
  *out_spec = spec_color;
}

typedef struct {
  int hit;
  v3 color;
  float t;
  int id;
} trace_t;

static trace_t trace(ctx_t* c, v3 o, v3 d_raw, float n_start, int depth, int kind);

/* calculate_reflection, raytracer_renderer.rs:526-729 */
static v3 reflection(ctx_t* c, const hit_t* h, const mat_t* m, v3 view, float n_start, int depth) {
  const rt_params* P = c->p;
  float cos_theta = vdot(view, h->n);
  int inside = cos_theta < 0.0f;
  v3 inormal = inside ? vneg(h->n) : h->n;
  float n2 = inside ? m->ior : P->air_ior;
  float eta = inside ? (n2 / n_start) : (n_start / n2);
  float cos_i = fabsf(cos_theta);
  float sin2 = eta * eta * (1.0f - cos_i * cos_i);
  int tir = sin2 >= 1.0f;
  int reflective = (m->metallic > 0.0f) || (transmissive(m) && tir);
  if (!reflective) return V(0, 0, 0);
  v3 r = vnormalize(vreflected(view, h->n));
  v3 Rf = fresnel_reflectance(m, inormal, vneg(view), n_start);
  int child_depth = depth < 0 ? (int)P->max_depth_reflection : (depth > 0 ? depth - 1 : 0);
  v3 epsv = V(P->eps_distance, P->eps_distance, P->eps_distance);
  trace_t ch = trace(c, vadd(h->p, vmul(r, epsv)), r, n_start, child_depth, KIND_REFL);
  if (!ch.hit) return V(0, 0, 0);
  float df = atten(ch.t);
  return vmul(vscale(ch.color, df), Rf);
}

/* calculate_refractions, raytracer_renderer.rs:279-524 */
static v3 refraction(ctx_t* c, const hit_t* h, const mat_t* m, v3 view, float n_start, int depth) {
  const rt_params* P = c->p;
  if (!transmissive(m)) return V(0, 0, 0);
  float cos_theta = vdot(view, h->n);
  int inside = cos_theta <= 0.0f;
  v3 inormal = inside ? vneg(h->n) : h->n;
  float n2 = inside ? m->ior : P->air_ior;
  float eta = inside ? (n2 / n_start) : (n_start / n2);
  float inv_eta = 1.0f / eta;
  v3 Rf = fresnel_reflectance(m, inormal, view, inv_eta);
  v3 Tr = V(1.0f - Rf.x, 1.0f - Rf.y, 1.0f - Rf.z);
  v3 q = vnormalize(vrefracted(view, vneg(inormal), inv_eta));
  float op = m->opacity; /* transmissive here */
  int step = (op < 0.5f) ? 2 : 1;
  int fac = (op <= 0.3f) ? 3 : ((op < 0.5f) ? 2 : 1);
  int child_depth;
  if (depth < 0)
    child_depth = (int)P->max_depth_refraction / fac;
  else
    child_depth = depth > step ? depth - step : 0;
  v3 epsv = V(P->eps_distance, P->eps_distance, P->eps_distance);
  trace_t ch = trace(c, vadd(h->p, vmul(q, epsv)), q, n2, child_depth, KIND_REFR);
  if (!ch.hit) return V(0, 0, 0);
  float boost1 = m->boost + 1.0f;
  return vmul(vscale(ch.color, boost1), Tr);
}

/* single_raytrace, raytracer_renderer.rs:147-264 */
static trace_t trace(ctx_t* c, v3 o, v3 d_raw, float n_start, int depth, int kind) {
  trace_t res;
  res.hit = 0;
  res.color = V(0, 0, 0);
  res.t = 0.0f;
  res.id = -1;
  if (depth == 0) return res; /* :174-178 */
  v3 d = vnormalize(d_raw);
  if (has_nan3(d)) return res; /* deviation D2 */
  c->rays[kind]++;
  hit_t h = nearest(c, o, d);
  if (!h.valid) return res;
  mat_t m = load_mat(c->s, h.mat);
  v3 direct, spec;
  lighting(c, &h, &m, d, &direct, &spec);
  float a = atten(h.t);
  direct = vscale(direct, a);
  spec = vscale(spec, a);
  int T = transmissive(&m);
  int R = (m.metallic > 0.0f) || T;
  v3 refl = V(0, 0, 0), refr = V(0, 0, 0);
  if ((c->p->flags & RT_FLAG_REFLECTIONS) && R) refl = reflection(c, &h, &m, d, n_start, depth);
  if ((c->p->flags & RT_FLAG_REFRACTIONS) && T) refr = refraction(c, &h, &m, d, n_start, depth);
  res.hit = 1;
  res.t = h.t;
  res.id = h.id;
  res.color = T ? vadd(vadd(refl, refr), spec) : vadd(vadd(direct, refl), spec);
  return res;
}

/* palette Rgb<f32> -> Rgb<u8>: clamp to [0,1], *255, round half to even; window.rs:105-109 */
static inline uint32_t to_u8(float x) {
  float cx = fminf(fmaxf(x, 0.0f), 1.0f); /* NaN -> 0 */
  return (uint32_t)lrintf(cx * 255.0f);   /* default rounding mode = nearest-even */
}
static inline uint32_t pack_pixel(v3 c) {
  return 0xFF000000u | (to_u8(c.x) << 16) | (to_u8(c.y) << 8) | to_u8(c.z);
}

/* light cloud for pixel p: PointLight::to_point_light_cloud + SceneLightSource::preprocess,
 * light.rs:183-225,311-324, seeded (deviation D4) */
static void build_lights(ctx_t* c, uint32_t pixel) {
  const rt_scene_desc* s = c->s;
  const rt_params* P = c->p;
  uint32_t N = P->light_mult < 1 ? 1 : P->light_mult;
  uint32_t k = 0;
  for (uint32_t l = 0; l < s->n_lights; l++) {
    const float* L = s->lights + (size_t)l * RT_LIGHT_STRIDE;
    if (N == 1) {
      c->lpos[3 * k] = L[0];
      c->lpos[3 * k + 1] = L[1];
      c->lpos[3 * k + 2] = L[2];
      c->lcol[3 * k] = L[3];
      c->lcol[3 * k + 1] = L[4];
      c->lcol[3 * k + 2] = L[5];
      c->lint[k] = L[6];
      k++;
      continue;
    }
    uint32_t set = rt_cloud_hash(P->cloud_seed, pixel, l) % P->n_cloud_sets;
    const float* cs = P->cloud_sets + (size_t)set * N * 3;
    float scale = 1.0f / (float)N;
    for (uint32_t j = 0; j < N; j++) {
      /* position + random_point * window_to_scene_scale (plain mul + add), light.rs:218 */
      c->lpos[3 * k] = L[0] + cs[3 * j] * P->fw;
      c->lpos[3 * k + 1] = L[1] + cs[3 * j + 1] * P->fh;
      c->lpos[3 * k + 2] = L[2] + cs[3 * j + 2] * P->fd;
      c->lcol[3 * k] = L[3];
      c->lcol[3 * k + 1] = L[4];
      c->lcol[3 * k + 2] = L[5];
      c->lint[k] = scale * L[6];
      k++;
    }
  }
  c->n_lights = k;
}

/* one pixel: render_pixel_colors / antialiased_raytrace, raytracer_renderer.rs:918-1016,1190-1357 */
static void render_pixel(ctx_t* c, uint32_t gx, uint32_t gy, uint32_t* argb, const rt_aux* aux,
                         uint64_t* written) {
  const rt_params* P = c->p;
  uint32_t pix = gy * P->width + gx;
  float x = (float)gx * P->fw; /* renderer/mod.rs:176-180 */
  float y = (float)gy * P->fh;
  v3 coords = V(x, y, 0.0f);
  v3 D = vsub(coords, V(P->focus[0], P->focus[1], P->focus[2]));
  build_lights(c, pix);
  int any = 0;
  v3 color = V(0, 0, 0);
  int id0 = -1;
  float t0 = 0.0f;
  if ((P->flags & RT_FLAG_ANTI_ALIASING) && P->aa_rays > 0) {
    uint32_t n = P->aa_rays;
    uint32_t packets = (n + 7) / 8;
    float scale = 1.0f / (float)(packets * 8); /* :936-937 */
    v3 first[8], rest[8];
    for (int l = 0; l < 8; l++) first[l] = rest[l] = V(0, 0, 0);
    for (uint32_t k = 0; k < n; k++) {
      v3 o = V(coords.x + P->aa_offsets[2 * k], coords.y + P->aa_offsets[2 * k + 1], coords.z);
      trace_t r = trace(c, o, D, P->air_ior, -1, KIND_PRIMARY);
      if (k == 0) {
        id0 = r.id;
        t0 = r.t;
      }
      if (r.hit) {
        any = 1;
        v3 cs = vscale(r.color, scale);
        if (k < 8)
          first[k] = cs; /* packet 0, added last (:1001) */
        else
          rest[k & 7] = vadd(cs, rest[k & 7]); /* :998 `color + res_color` */
      }
    }
    v3 lane[8];
    for (int l = 0; l < 8; l++) lane[l] = vadd(rest[l], first[l]);
    /* wide f32x8 horizontal sum: ((l0+l4)+(l2+l6)) + ((l1+l5)+(l3+l7)) */
    v3 a04 = vadd(lane[0], lane[4]), a15 = vadd(lane[1], lane[5]);
    v3 a26 = vadd(lane[2], lane[6]), a37 = vadd(lane[3], lane[7]);
    color = vadd(vadd(a04, a26), vadd(a15, a37));
  } else {
    trace_t r = trace(c, coords, D, P->air_ior, -1, KIND_PRIMARY);
    any = r.hit;
    color = r.color;
    id0 = r.id;
    t0 = r.t;
  }
  if (aux && aux->hit_id) aux->hit_id[pix] = id0;
  if (aux && aux->hit_t && id0 >= 0) aux->hit_t[pix] = t0;
  if (any) {
    argb[pix] = pack_pixel(color);
    if (aux && aux->rgb) {
      aux->rgb[3 * (size_t)pix] = color.x;
      aux->rgb[3 * (size_t)pix + 1] = color.y;
      aux->rgb[3 * (size_t)pix + 2] = color.z;
    }
    (*written)++;
  }
}

/* ---- threaded driver (stands in for rayon over tiles/rows, image_buffer.rs:48-97,306-320) ----- */
typedef struct {
  const rt_scene_desc* s;
  const rt_params* p;
  uint32_t* argb;
  const rt_aux* aux;
  uint32_t x0, y0, w, h, chunk;
  volatile uint32_t* next_row;
  uint64_t rays[3], shadow, written;
} job_t;

static int tile_owned(const rt_params* P, uint32_t gx, uint32_t gy) {
  if (P->n_ranks <= 1) return 1;
  uint32_t ts = P->tile_size ? P->tile_size : 48u;
  return rt_tile_owner(gx / ts, gy / ts, P->n_ranks) == P->rank;
}

static void* worker(void* arg) {
  job_t* j = (job_t*)arg;
  ctx_t c;
  memset(&c, 0, sizeof(c));
  c.s = j->s;
  c.p = j->p;
  c.cull = (j->p->flags & RT_FLAG_BACKFACE_CULLING) != 0;
  uint32_t N = j->p->light_mult < 1 ? 1 : j->p->light_mult;
  size_t nl = (size_t)j->s->n_lights * N;
  c.lpos = (float*)malloc(sizeof(float) * 3 * (nl + 1));
  c.lcol = (float*)malloc(sizeof(float) * 3 * (nl + 1));
  c.lint = (float*)malloc(sizeof(float) * (nl + 1));
  /* work items: runs of RT_CHUNK pixels in row-major window order (fine-grained so that small
   * windows still spread over all threads) */
  const uint32_t total = j->w * j->h;
  const uint32_t chunk = j->chunk;
  for (;;) {
    uint32_t first = __atomic_fetch_add(j->next_row, chunk, __ATOMIC_RELAXED);
    if (first >= total) break;
    uint32_t last = first + chunk < total ? first + chunk : total;
    for (uint32_t k = first; k < last; k++) {
      uint32_t gx = j->x0 + k % j->w, gy = j->y0 + k / j->w;
      if (!tile_owned(j->p, gx, gy)) continue;
      render_pixel(&c, gx, gy, j->argb, j->aux, &j->written);
    }
  }
  j->rays[0] = c.rays[0];
  j->rays[1] = c.rays[1];
  j->rays[2] = c.rays[2];
  j->shadow = c.shadow;
  free(c.lpos);
  free(c.lcol);
  free(c.lint);
  return NULL;
}

static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

/* Same contract as rt_render (include/rt_hip.h) but computed on the host, brute force, like the
 * reference.  n_threads <= 0 -> 1. */
int rt_cpu_render(const rt_scene_desc* desc, const rt_params* params, uint32_t* argb,
                  const rt_aux* aux, rt_stats* stats, int n_threads) {
  if (!desc || !params || !argb) return RT_ERR_INVALID_ARG;
  if (desc->abi_version != RT_ABI_VERSION || params->abi_version != RT_ABI_VERSION)
    return RT_ERR_INVALID_ARG;
  if (params->light_mult > 1 && (params->n_cloud_sets == 0 || !params->cloud_sets))
    return RT_ERR_INVALID_ARG;
  if ((params->flags & RT_FLAG_ANTI_ALIASING) && params->aa_rays > 0 && !params->aa_offsets)
    return RT_ERR_INVALID_ARG;
  uint32_t x0 = 0, y0 = 0, w = params->width, h = params->height;
  if (params->win_w) {
    x0 = params->win_x0;
    y0 = params->win_y0;
    w = params->win_w;
    h = params->win_h;
    if (x0 + w > params->width || y0 + h > params->height) return RT_ERR_INVALID_ARG;
  }
  if (n_threads <= 0) n_threads = 1;
  if (n_threads > 256) n_threads = 256;
  double t0 = now_ms();
  volatile uint32_t next_row = 0;
  job_t* jobs = (job_t*)calloc((size_t)n_threads, sizeof(job_t));
  pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof(pthread_t));
  for (int i = 0; i < n_threads; i++) {
    jobs[i].s = desc;
    jobs[i].p = params;
    jobs[i].argb = argb;
    jobs[i].aux = aux;
    jobs[i].x0 = x0;
    jobs[i].y0 = y0;
    jobs[i].w = w;
    jobs[i].h = h;
    jobs[i].next_row = &next_row;
    /* runs of pixels small enough that every thread gets >= 8 work items */
    jobs[i].chunk = (w * h) / ((uint32_t)n_threads * 8u) < 1u ? 1u : ((w * h) / ((uint32_t)n_threads * 8u) > 8u ? 8u : (w * h) / ((uint32_t)n_threads * 8u));
  }
  if (n_threads == 1) {
    worker(&jobs[0]);
  } else {
    for (int i = 0; i < n_threads; i++) pthread_create(&th[i], NULL, worker, &jobs[i]);
    for (int i = 0; i < n_threads; i++) pthread_join(th[i], NULL);
  }
  double t1 = now_ms();
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
    stats->kernel_ms = t1 - t0;
    stats->total_ms = t1 - t0;
  }
  free(jobs);
  free(th);
  return RT_OK;
}

/* ---- single-function probes for the known-answer tests (tests/golden) ------------------------- */
int rt_oracle_sphere(const rt_scene_desc* s, uint32_t i, const float* o, const float* d, int cull,
                     float* out /* t, px,py,pz, nx,ny,nz */) {
  hit_t h;
  h.valid = 0;
  if (!sphere_intersect(s, i, V(o[0], o[1], o[2]), V(d[0], d[1], d[2]), cull, &h)) return 0;
  out[0] = h.t;
  out[1] = h.p.x;
  out[2] = h.p.y;
  out[3] = h.p.z;
  out[4] = h.n.x;
  out[5] = h.n.y;
  out[6] = h.n.z;
  return 1;
}
int rt_oracle_triangle(const rt_scene_desc* s, uint32_t i, const float* o, const float* d,
                       int cull, float* out /* t, px,py,pz */) {
  hit_t h;
  h.valid = 0;
  if (!triangle_intersect(s, i, V(o[0], o[1], o[2]), V(d[0], d[1], d[2]), cull, &h)) return 0;
  out[0] = h.t;
  out[1] = h.p.x;
  out[2] = h.p.y;
  out[3] = h.p.z;
  return 1;
}
void rt_oracle_fresnel(const float* mat9, const float* n, const float* v, float other_ior,
                       float* out_refl3) {
  rt_scene_desc s;
  memset(&s, 0, sizeof(s));
  s.materials = mat9;
  s.n_materials = 1;
  mat_t m = load_mat(&s, 0);
  v3 r = fresnel_reflectance(&m, V(n[0], n[1], n[2]), V(v[0], v[1], v[2]), other_ior);
  out_refl3[0] = r.x;
  out_refl3[1] = r.y;
  out_refl3[2] = r.z;
}
float rt_oracle_atten(float t) { return atten(t); }
uint32_t rt_oracle_pack(float r, float g, float b) { return pack_pixel(V(r, g, b)); }
void rt_oracle_refract(const float* i, const float* n, float eta, float* out3) {
  v3 r = vrefracted(V(i[0], i[1], i[2]), V(n[0], n[1], n[2]), eta);
  out3[0] = r.x;
  out3[1] = r.y;
  out3[2] = r.z;
}
