// rt_phases.h -- the render loop as PHASE kernels (included by rt_kernels.hip inside its anonymous namespace; it uses that
// file's device functions: nearest_hit, collect_light_candidates, shadow_ray, light_sample_terms, queue_push, hard_push).
//
// The fused kernels (process_ray) run nearest hit -> 5 x (classify, collect, N-sample loop) -> shade -> spawn in ONE kernel
// at 80 VGPRs with ~110 B of scratch per lane, and a wavefront pays for its slowest light.  Here the same work is cut at
// the points where the state that must survive is small, and what crosses a cut goes through HBM (which this path leaves
// 97 % idle):
//
//   K1 rt_hit_kernel        one thread per (pixel, distinct AA sample): camera ray -> nearest hit -> 8-byte hit record
//                           (raytracer_renderer.rs:1190-1357 ray set-up, raytracer.rs:162-220 cast_ray); secondary levels:
//                           rt_trace_kernel + the counting sort, as before
//   K2 rt_classify*_kernel  one thread per hit: surface + material, the ambient term, the children of the Whitted node
//                           (:526-729 reflection, :279-524 refraction -> ray queue), and per (wavefront, light) the DECISION how
//                           its N shadow samples are to be traced -- horizon, receiver flags, per-cell lists or the candidate
//                           walk, umbra (calculate_lighting :731-874 up to the sample loop) -- written as a 32-byte SET record
//                           (+ the shared candidate list, one dword per lane) and appended to one of three class queues:
//                             ARRIVE  nothing can touch the rays: no shadow ray is set up at all
//                             LIST    the samples test the set's shared candidate list
//                             WALK    one BVH walk per sample (hard shadows, linear scan, sharing switched off)
//                           (incoherent sets still go to rt_hard_kernel as (hit point, light) pairs)
//   K3 rt_sets*_kernel      one wavefront per set of ONE class: re-derives the 64 hit points of the set from the hit records
//                           (cheaper than carrying them: ~150 vector instructions against the ~4 000 of a LIST set), runs the
//                           N-sample loop of that class (has_any_intersection raytracer.rs:24-106, light.rs:261-299) and adds
//                           the (hit point, light) share W * atten * own_l to the pixel as ONE fixed-point term -- the term
//                           the fused streaming kernels and rt_hard_kernel add, so the frame is the same integers
//   K4 rt_resolve_kernel    fixed-point pixel sums -> packed pixels
//
// Work items of K3 are (wavefront, light) sets, so the longest item is a fifth of a fused wavefront's, and each kernel is
// register-allocated for its own phase only.
#pragma once

enum { SET_ARRIVE = 0, SET_LIST = 1, SET_WALK = 2, SET_CLASSES = 3 };

struct PixelMap {
  uint32_t slot, k, gx, gy, pix;
  bool on;
};
// work item `tid` of workgroup `wg` of the frame's list -> (pixel, AA sample): the mapping of primary_body
__device__ __forceinline__ PixelMap map_item(const RtDevParams& P, uint32_t wg, uint32_t tid, uint32_t n_thr) {
  PixelMap m;
  const bool wave_local = rt_primary_wave_local(n_thr);
  const uint32_t ppwave = wave_local ? 64u / n_thr : 0u;
  const uint32_t ppw = wave_local ? 4u * ppwave : 256u / n_thr;
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
  const uint32_t g = wg * ppw + m.slot;
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
}

// the camera ray of a work item (render_pixel_colors, raytracer_renderer.rs:1190-1357; renderer/mod.rs:176-180)
__device__ __forceinline__ RayIn primary_ray(const RtDevParams& P, const PixelMap& pm) {
  const bool aa = (P.flags & RT_FLAG_ANTI_ALIASING) && P.aa_rays > 0;
  const uint32_t n_samples = aa ? P.aa_rays : 1u;
  const float x = (float)pm.gx * P.fw, y = (float)pm.gy * P.fh;
  const V3 coords = mk(x, y, 0.0f);
  RayIn r;
  r.o = coords;
  if (aa && pm.on) {
    r.o.x = coords.x + P.aa_offsets[2 * pm.k];
    r.o.y = coords.y + P.aa_offsets[2 * pm.k + 1];
  }
  r.d_raw = coords - mk(P.focus[0], P.focus[1], P.focus[2]);  // un-jittered for every sample (:1204)
  r.n_start = P.air_ior;
  const float scale = aa ? 1.0f / (float)(((n_samples + 7u) / 8u) * 8u) : 1.0f;  // :936-937
  r.Wt = mk(scale, scale, scale);
  r.depth = -1;
  r.kind = KIND_PRIMARY;
  r.pix = pm.pix;
  r.mult = (P.weighted && pm.on) ? P.aa_mult[pm.k] : 1u;
  return r;
}

// ---- K1: camera ray -> nearest hit -> hit record ---------------------------------------------------------------------
// SPAWN (merged levels of the fused pipeline): the children of the camera rays are appended here too, so that the levels below can be
// traced while rt_primary_pre_kernel shades these hits on another stream.
template <bool CULL, bool SPAWN = false>
__device__ __forceinline__ void hit_body(const RtDevScene& sc, const RtDevParams& P, unsigned long long* lds_cnt) {
  Wave wv;
  wave_init(wv);
  wave_flush_init(P, lds_cnt);
  const bool aa = (P.flags & RT_FLAG_ANTI_ALIASING) && P.aa_rays > 0;
  const uint32_t n_thr = aa ? P.aa_unique : 1u;
  const PixelMap pm = map_item(P, rt_batch_wg(P, blockIdx.x), threadIdx.x, n_thr);
  const RayIn r = primary_ray(P, pm);
  const V3 d = normalize(r.d_raw);  // Ray::new_with_mask, ray.rs:52-57
  const bool alive = pm.on && !has_nan(d);
  const lanemask bal = wave_ballot(alive);
  Hit h;
  h.t = INFINITY;
  h.id = -1;
  if (bal) {
    wv.cnt_kind[0] += P.weighted ? wave_sum(alive ? r.mult : 0u) : (uint32_t)__popcll(bal);
    wv.cnt_traced += (uint32_t)__popcll(bal);
    WSTAT(wv.cnt_pass += 1);
    WSTAT(wv.cnt_lanes += (uint32_t)__popcll(bal));
    h = nearest_hit<CULL>(sc, P, wv.ctx, alive, r.o, d);
  }
  const bool hit = alive && h.id >= 0;
  P.hitrec[(size_t)blockIdx.x * 256u + threadIdx.x] = make_uint2(__float_as_uint(h.t), (uint32_t)(hit ? h.id : -1));
  if (pm.k == 0 && pm.on) {
    if (P.aux_hit_id) P.aux_hit_id[pm.pix] = hit ? h.id : -1;
    if (P.aux_hit_t && hit) P.aux_hit_t[pm.pix] = h.t;
  }
  if (SPAWN) {  // (all 256 threads: the workgroup's queue reservation has barriers)
    Hit hh = h;
    hh.id = hit ? h.id : 0;
    Surf sf;
    sf.p = mk(0, 0, 0), sf.n = mk(0, 0, 1), sf.mat = 0;
    if (hit) sf = surface_of(sc, hh, r.o, d);
    const Mat m = load_mat(sc, sf.mat);
    spawn_children_block(sc, P, hit, d, sf, m, r.Wt, r.n_start, r.depth, r.pix, r.mult, (uint32_t*)(lds_cnt + 20), 0u);
  }
  wave_flush(wv, P, 0ull, lds_cnt);
}

__global__ __launch_bounds__(256, 8) void rt_hit_kernel(RtDevScene sc, RtDevParams P) {
  __shared__ unsigned long long lds_cnt[20];
  if (P.flags & RT_FLAG_BACKFACE_CULLING)
    hit_body<true>(sc, P, lds_cnt);
  else
    hit_body<false>(sc, P, lds_cnt);
}

__global__ __launch_bounds__(256, 8) void rt_hit_spawn_kernel(RtDevScene sc, RtDevParams P) {
  __shared__ unsigned long long lds_cnt[28];  // [20..27]: the workgroup's queue reservation (spawn_children_block)
  if (P.flags & RT_FLAG_BACKFACE_CULLING)
    hit_body<true, true>(sc, P, lds_cnt);
  else
    hit_body<false, true>(sc, P, lds_cnt);
}

// ---- the hit of a work item, re-derived where it is needed (K2, K3) ------------------------------------------------------
struct ItemHit {
  bool hit;       // the item exists and its ray hit something
  V3 d;           // unit direction of the ray
  Surf sf;        // hit point, shading normal, material row
  V3 Wa;          // W * atten(t) (a reflection child's W carries its own atten(t) too, :722-726)
  V3 W0;          // weight its children inherit
  float n_start;
  int depth, kind;
  uint32_t pix, mult;
  uint32_t k;     // level 0: AA sample index of the lane inside its pixel
  bool on;        // level 0: the lane stands for a pixel of the window
  int id;
};
// item = index in the launch's item space.  Level 0: launched workgroup * 256 + thread (hit record written by K1); levels
// >= 1: position in hit-point order (sh_idx -> ray record, quad 3 = the hit rt_trace_kernel found).
template <bool L0>
__device__ __forceinline__ ItemHit load_item(const RtDevScene& sc, const RtDevParams& P, uint32_t item, uint32_t n_items) {
  ItemHit it;
  RayIn r = idle_ray();
  Hit h;
  h.t = INFINITY;
  h.id = -1;
  it.k = 0;
  it.on = false;
  bool have = false;
  if (L0) {
    const bool aa = (P.flags & RT_FLAG_ANTI_ALIASING) && P.aa_rays > 0;
    const uint32_t n_thr = aa ? P.aa_unique : 1u;
    const PixelMap pm = map_item(P, rt_batch_wg(P, item >> 8), item & 255u, n_thr);
    r = primary_ray(P, pm);
    it.k = pm.k;
    it.on = pm.on;
    have = pm.on && item < n_items;
    if (have) {
      const uint2 hr = P.hitrec[item];
      h.t = __uint_as_float(hr.x);
      h.id = (int)hr.y;
    }
  } else {
    const uint32_t jr = item < n_items ? P.sh_idx[item] : 0xFFFFFFFFu;
    have = jr != 0xFFFFFFFFu;
    if (have) {
      r = load_queued_ray(P, (size_t)jr);
      const float4 q3 = stream_load4(P.q_in + (size_t)jr * RT_QUEUE_QUADS + 3u);
      h.t = q3.x;
      h.id = __float_as_int(q3.y);
    }
  }
  it.d = normalize(r.d_raw);
  it.hit = have && h.id >= 0;
  it.id = it.hit ? h.id : -1;
  it.sf.p = mk(0, 0, 0);
  it.sf.n = mk(0, 0, 1);
  it.sf.mat = 0;
  if (it.hit) it.sf = surface_of(sc, h, r.o, it.d);
  const float a0 = atten(h.t);
  it.W0 = r.Wt;
  if (r.kind == KIND_REFL) it.W0 = it.W0 * a0;
  it.Wa = mk(it.W0.x * a0, it.W0.y * a0, it.W0.z * a0);
  it.n_start = r.n_start;
  it.depth = r.depth;
  it.kind = r.kind;
  it.pix = r.pix;
  it.mult = r.mult;
  return it;
}

// Adds a lane's fixed-point term to its pixel.  Level 0 with the samples of a pixel in one wavefront: the terms of a pixel
// meet in the wavefront's LDS block and its first lane issues the atomics (3 per pixel instead of 3 per sample; integer
// sums, so the grouping cannot change the result).  Otherwise one lane, one pixel.
#ifndef RT_AB_NO_ATOMICS
#define RT_AB_NO_ATOMICS 0  /* timing experiment only (wrong image): the phase kernels add nothing to the pixel accumulator */
#endif
template <bool L0>
__device__ __forceinline__ void add_terms(const RtDevParams& P, long long* lds_fx /* this wavefront's [3][64] */, const ItemHit& it, bool on,
                                          long long fx, long long fy, long long fz) {
  const bool aa = (P.flags & RT_FLAG_ANTI_ALIASING) && P.aa_rays > 0;
  const uint32_t n_thr = aa ? P.aa_unique : 1u;
  if (RT_AB_NO_ATOMICS) {  // (the terms stay live -- the store never happens -- so the sample loops are not optimised away)
    if (on && (fx ^ fy ^ fz) == 0x7FFFFFFFFFFFFFF1ll) P.acc[0] = fx;
    return;
  }
  if (L0 && n_thr > 1u && rt_primary_wave_local(n_thr)) {
    const uint32_t lane = threadIdx.x & 63u;
    lds_fx[lane] = on ? fx * (long long)it.mult : 0ll;
    lds_fx[64u + lane] = on ? fy * (long long)it.mult : 0ll;
    lds_fx[128u + lane] = on ? fz * (long long)it.mult : 0ll;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (it.k == 0 && it.on) {
      long long sx = 0, sy = 0, sz = 0;
      for (uint32_t u = 0; u < n_thr; u++) sx += lds_fx[lane + u], sy += lds_fx[64u + lane + u], sz += lds_fx[128u + lane + u];
      acc_add_fixed(P, it.pix, sx, sy, sz, 1u);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // (the block is reused by the wavefront's next set)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else if (on) {
    acc_add_fixed(P, it.pix, fx, fy, fz, it.mult);
  }
}

// one set = one (wavefront of 64 items, light): what K2 decided, what K3 needs
struct SetRec {
  uint32_t item_base, light, count, spheres;
  lanemask use_m;
};
__device__ __forceinline__ void set_write(const RtDevParams& P, uint32_t set_id, int cls, const SetRec& s, uint32_t cand_reg) {
  // (no queue slot is taken here: one returning atomic per set on ONE counter serialises the whole launch -- 1.5 M of them
  // cost config 3 more than 5 ms, r04c.  The set's class goes into a dense byte array; rt_compact_kernel builds the class
  // queues from it with one atomic per 64 sets.)
  if ((threadIdx.x & 63u) == 0) {
    P.set_hdr[2u * (size_t)set_id] = make_uint4(s.item_base, s.light, s.count, s.spheres);
    P.set_hdr[2u * (size_t)set_id + 1u] = make_uint4((uint32_t)s.use_m, (uint32_t)(s.use_m >> 32), 0u, 0u);
    P.set_cls[set_id] = (uint8_t)(cls + 1);
  }
  if (cls == SET_LIST) P.set_list[(size_t)set_id * 64u + (threadIdx.x & 63u)] = cand_reg;
}

// class bytes (0 = no set; every wavefront of K2 clears its own before it classifies) -> the three class queues: one thread per set id, ballot compaction, one atomic per wavefront
// and class.  set_n = 0: the level's set ids end at ceil(hits / 64) * lights (device-side count).
__global__ __launch_bounds__(256) void rt_compact_kernel(RtDevParams P) {
  // 8 set ids per thread (one 8-byte load): 512 ids per wavefront and one atomic per wavefront and class -- the atomics all
  // hit three counters, ~3 ns each whatever else happens (r04d: 0.21 ms for 1.5 M ids at 64 per wavefront)
  const uint32_t n = P.set_n ? P.set_n : ((uload(P.sort_hits) + 63u) / 64u) * P.set_lights;
  const uint32_t n_up = (n + 511u) & ~511u;  // (whole wavefronts: the votes below need every lane)
  for (uint32_t i = (blockIdx.x * 256u + threadIdx.x) * 8u; i < n_up; i += gridDim.x * 2048u) {
    const uint2 w = i < n ? *(const uint2*)(P.set_cls + i) : make_uint2(0u, 0u);  // (set_cls is padded to a multiple of 8)
    uint32_t cls[8];
#pragma unroll
    for (uint32_t b = 0; b < 8u; b++) cls[b] = i + b < n ? (((b < 4u ? w.x : w.y) >> (8u * (b & 3u))) & 0xFFu) : 0u;
#pragma unroll
    for (uint32_t c = 0; c < 3u; c++) {
      uint32_t mine = 0;
#pragma unroll
      for (uint32_t b = 0; b < 8u; b++) mine += cls[b] == c + 1u ? 1u : 0u;
      // exclusive prefix of `mine` over the lanes + the wavefront's total
      uint32_t incl = mine;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)incl, o, 64);
        if ((int)(threadIdx.x & 63u) >= o) incl += v;
      }
      const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
      if (!total) continue;
      uint32_t base = 0;
      if ((threadIdx.x & 63u) == 0) base = atomicAdd(&P.set_count[c], total);
      base = __builtin_amdgcn_readfirstlane(base) + incl - mine;
#pragma unroll
      for (uint32_t b = 0; b < 8u; b++)
        if (cls[b] == c + 1u) P.set_q[(size_t)c * P.set_cap + base++] = i + b;
    }
  }
}

// ---- the N-sample loops of a (wavefront, light) set ----------------------------------------------------------------------
// Returns the lane's fixed-point term W * atten * (transmissive ? specular_l : direct_l + specular_l) of light l.  The chain --
// from zero, in sample order, through light_sample_terms -- is the one the fused streaming kernels and rt_hard_kernel run.
template <bool CULL, int CLS>
__device__ __forceinline__ void set_samples(const RtDevScene& sc, const RtDevParams& P, WaveCtx& W, const ItemHit& it, lanemask use_m,
                                            uint32_t l, const CandList& cand, long long& fx, long long& fy, long long& fz) {
  const uint32_t N = P.light_mult < 1u ? 1u : P.light_mult;
  const V3 epsv = mk(P.eps_distance, P.eps_distance, P.eps_distance);
  const float4 L0v = sload<float4>(sc, sc.off_lights + l * 32u);
  const float4 L1v = sload<float4>(sc, sc.off_lights + l * 32u + 16u);
  const V3 lc = mk(L1v.x, L1v.y, L1v.z);
  const bool use = lane_of(use_m);
  const Mat m = load_mat(sc, it.sf.mat);
  const bool has_spec = m.shininess > 0.0f;
  const V3 mmc_lc = m.color * (m.color * lc);
  const float4* cs = nullptr;
  float lI = L0v.w;
  if (N > 1) {
    const uint32_t hsh = rt_cloud_hash(P.cloud_seed, it.pix, l);
    const uint32_t set = (P.n_cloud_sets & (P.n_cloud_sets - 1u)) == 0u ? (hsh & (P.n_cloud_sets - 1u)) : (hsh % P.n_cloud_sets);
    cs = P.cloud_sets + (size_t)set * N;
    lI = (1.0f / (float)N) * L0v.w;
  }
  V3 dl = mk(0.0f, 0.0f, 0.0f), ds = mk(0.0f, 0.0f, 0.0f);
  float4 cnext = make_float4(0, 0, 0, 0);
  if (N > 1 && use) cnext = cs[0];
  auto light_position = [&](uint32_t j) {
    V3 lp = mk(L0v.x, L0v.y, L0v.z);
    if (N > 1) {
      lp.x = L0v.x + cnext.x;  // light.rs:218; the table holds offset * (fw, fh, fd)
      lp.y = L0v.y + cnext.y;
      lp.z = L0v.z + cnext.z;
      if (use && j + 1 < N) cnext = cs[j + 1];
    }
    return lp;
  };
  if (CLS == SET_ARRIVE) {
    Shadow S;
    shadow_init(S);
    for (uint32_t j = 0; j < N; j++) {
      const V3 ltp = light_position(j) - it.sf.p;
      WSTAT(W.s_passes++);
      const LightTerms T = light_sample_terms<false>(it.sf.n, it.d, mmc_lc, lI, m.shininess, has_spec, ltp, S);
      if (use && T.lit) {
        dl = fma_s(T.mLc, T.lf, dl);
        if (has_spec) ds = fma_s(lc, T.sf, ds);
      }
    }
  } else {
    for (uint32_t j = 0; j < N; j++) {
      const V3 lp = light_position(j);
      const V3 ltp = lp - it.sf.p;
      const V3 ld = ltp * exact_rcp(mag(ltp));  // normalize(ltp): the shadow ray's geometry is exact
      const V3 so = it.sf.p + ld * epsv;
      const float tmax = mag(lp - so);
#if RT_SKIP  // removal ablations (timing / instruction counts only, wrong image): 1 no sphere tests, 2 no triangle tests, 32 no lighting
      CandList cand_ab = cand;
      if (RT_SKIP & 1) cand_ab.spheres = 0;
      if (RT_SKIP & 2) cand_ab.count = 0;
      const Shadow S = shadow_ray<CULL, CLS == SET_LIST>(sc, P, W, use_m, so, ld, tmax, cand_ab);
#else
      const Shadow S = shadow_ray<CULL, CLS == SET_LIST>(sc, P, W, use_m, so, ld, tmax, cand);
#endif
      const lanemask reach_m = use_m & ~S.occ;
      if (!reach_m) continue;
      if (RT_SKIP & 32) {  // (keeps the shadow result live)
        if (lane_of(reach_m)) dl.x += (float)S.dec + (float)S.fr, ds.x += tmax;
        continue;
      }
      const LightTerms T = light_sample_terms<true>(it.sf.n, it.d, mmc_lc, lI, m.shininess, has_spec, ltp, S);
      if (lane_of(reach_m) && T.lit) {
        dl = fma_s(T.mLc, T.lf, dl);
        if (has_spec) ds = fma_s(lc, T.sf, ds);
      }
    }
  }
  const V3 c = it.Wa * (m.transmissive ? ds : (dl + ds));
  fx = __float2ll_rn(c.x * RT_ACC_SCALE);
  fy = __float2ll_rn(c.y * RT_ACC_SCALE);
  fz = __float2ll_rn(c.z * RT_ACC_SCALE);
}

// ---- K2: per hit -- ambient, children, and the class of every (wavefront, light) set --------------------------------------
// ARRIVE_INLINE: sets nothing can touch are finished here (their loop needs no shadow ray and little state) instead of
// being queued for K3.
#ifndef RT_ARRIVE_INLINE
#define RT_ARRIVE_INLINE 1
#endif
template <bool CULL, bool L0>
__device__ __forceinline__ void classify_wave(const RtDevScene& sc, const RtDevParams& P, Wave& wv, uint32_t item_base, uint32_t n_items,
                                              long long* lds_fx) {
  WaveCtx& W = wv.ctx;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t N = P.light_mult < 1u ? 1u : P.light_mult;
  const ItemHit it = load_item<L0>(sc, P, item_base + lane, n_items);
  const bool hit = it.hit;
  const lanemask hit_m = wave_ballot(hit);
  // this wavefront's set ids start out as "no set" (the lane that may overwrite one later: same lane, program order)
  if (lane == 0)
    for (uint32_t l = 0; l < sc.n_lights; l++) P.set_cls[(item_base >> 6) * sc.n_lights + l] = 0;
  if (!hit_m) return;
  const Mat m = load_mat(sc, it.sf.mat);
  const bool T = m.transmissive;
  // ---- the node's own ambient term (part of `direct`, :754; own = transmissive ? spec : direct + spec, :251-257) --------
  // (what this kernel itself contributes to the lane's pixel -- the ambient term and the shares of the lights whose sets it
  // finishes inline -- is summed here as integers and added ONCE at the end: a fifth of the atomics, same sum)
  long long own_x = 0, own_y = 0, own_z = 0;
  {
    const V3 c = it.Wa * ((m.color * mk(1.0f, 1.0f, 1.0f)) * P.ambient);
    if (hit && !T) own_x = __float2ll_rn(c.x * RT_ACC_SCALE), own_y = __float2ll_rn(c.y * RT_ACC_SCALE), own_z = __float2ll_rn(c.z * RT_ACC_SCALE);
    if (L0 && hit) P.acc[4 * (size_t)it.pix + 3] = 1;  // the pixel is written (any sample hit: antialiased_raytrace :1001-1015)
  }
  // ---- children: calculate_reflection :526-729, calculate_refractions :279-524 -------------------------------------------------
  if (P.q_out) spawn_children(sc, P, hit, it.d, it.sf, m, it.W0, it.n_start, it.depth, it.pix, it.mult);
  // every hit point casts lights x N shadow rays in the reference (raytracer.rs:24); counted once, here
  wv.cnt_shadow += sc.n_lights * N * (P.weighted ? wave_sum(hit ? it.mult : 0u) : (uint32_t)__popcll(hit_m));

  // ---- receiver cell of the hit point (as process_ray) ---------------------------------------------------------------------
  uint32_t rflags = 0u, cell = RT_NO_CELL;
  if (N > 1 && P.recv_flags && hit) {
    if (it.id >= (int)sc.n_spheres) {
      const uint32_t ro = sc.off_recv + (uint32_t)(it.id - (int)sc.n_spheres) * 48u;
      const float4 ru = vload<float4>(sc, ro), rv = vload<float4>(sc, ro + 16u);
      const uint2 rr = vload<uint2>(sc, ro + 32u);  // {R, first cell}
      const float Rf = (float)rr.x;
      const float cu = (__builtin_fmaf(ru.x, it.sf.p.x, __builtin_fmaf(ru.y, it.sf.p.y, ru.z * it.sf.p.z)) + ru.w) * Rf;
      const float cv = (__builtin_fmaf(rv.x, it.sf.p.x, __builtin_fmaf(rv.y, it.sf.p.y, rv.z * it.sf.p.z)) + rv.w) * Rf;
      const uint32_t ci = (uint32_t)fminf(fmaxf(cu, 0.0f), Rf - 1.0f), cj = (uint32_t)fminf(fmaxf(cv, 0.0f), Rf - 1.0f);
      if (rr.x != 0u && ci + cj < rr.x) cell = rr.y + ci + rr.x * cj, rflags = P.recv_flags[cell];
    } else {
      const uint2 sr = vload<uint2>(sc, sc.off_srecv + (uint32_t)it.id * 8u);  // {Rs, first cell}
      const float4 sp = vload<float4>(sc, sc.off_spheres + (uint32_t)it.id * 16u);
      const V3 dd = it.sf.p - mk(sp.x, sp.y, sp.z);
      const float ax = fabsf(dd.x), ay = fabsf(dd.y), az = fabsf(dd.z);
      const bool mx = ax >= ay && ax >= az, my = !mx && ay >= az;
      const float dm = mx ? dd.x : (my ? dd.y : dd.z), du = mx ? dd.y : (my ? dd.z : dd.x), dv = mx ? dd.z : (my ? dd.x : dd.y);
      const uint32_t face = (mx ? 0u : (my ? 2u : 4u)) + (dm < 0.0f ? 1u : 0u);
      const float inv = __builtin_amdgcn_rcpf(fmaxf(fabsf(dm), 1e-30f)), Rf = (float)sr.x;
      const uint32_t ci = (uint32_t)fminf(fmaxf((__builtin_fmaf(du, inv, 1.0f)) * 0.5f * Rf, 0.0f), Rf - 1.0f);
      const uint32_t cj = (uint32_t)fminf(fmaxf((__builtin_fmaf(dv, inv, 1.0f)) * 0.5f * Rf, 0.0f), Rf - 1.0f);
      if (sr.x != 0u) cell = sr.y + (face * sr.x + cj) * sr.x + ci, rflags = P.recv_flags[cell];
    }
  }
  // first hit point of the wavefront and how far the others are from it (sphere pre-selection of the candidate collection)
  V3 p_first = mk(0, 0, 0);
  float p_spread = 0.0f;
  if (N > 1 && sc.n_spheres) {
    const int fl = __ffsll((long long)hit_m) - 1;
    p_first = mk(__uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(it.sf.p.x), fl)),
                 __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(it.sf.p.y), fl)),
                 __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(it.sf.p.z), fl)));
    const V3 dp = it.sf.p - p_first;
    float mm = hit ? (fabsf(dp.x) + fabsf(dp.y) + fabsf(dp.z)) : 0.0f;  // 1-norm >= distance
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mm = fmaxf(mm, __shfl_xor(mm, o, 64));
    p_spread = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(mm))) * 1.000001f;
  }
  const bool hard_ok = P.hard_q && N > 1 && N <= 64u && P.traversal == RT_TRAVERSAL_BVH && sc.n_triangles && P.cand_cap != 0u;
  const uint32_t set_base = (item_base >> 6) * sc.n_lights;
  // The per-cell candidate lists are a gather from a table of hundreds of megabytes (HBM latency, and the first thing a light's
  // classification needs): the list of light l + 1 is requested while light l is processed.  (A cell's lists for all lights lie
  // side by side: mostly the same 64-byte line.)
  const bool lists_on = !CULL && N > 1 && P.cell_lists && P.cloud_delta > 0.0f && P.traversal == RT_TRAVERSAL_BVH && sc.n_triangles;
  const uint4 no_list = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
  uint4 lst_next = no_list;
  if (lists_on && hit && cell != RT_NO_CELL) lst_next = P.cell_lists[(size_t)cell * sc.n_lights];

  for (uint32_t l = 0; l < sc.n_lights; l++) {
    const uint4 lst_now = lst_next;
    if (lists_on && hit && cell != RT_NO_CELL && l + 1u < sc.n_lights) lst_next = P.cell_lists[(size_t)cell * sc.n_lights + l + 1u];
    const float4 L0v = sload<float4>(sc, sc.off_lights + l * 32u);
    // lights below the horizon of the hit point add nothing whatever their shadow rays would find (as process_ray)
    bool use = hit;
    {
      const float dlt = (N > 1) ? P.cloud_delta : 0.0f;
      V3 dc = mk(L0v.x, L0v.y, L0v.z) - it.sf.p;
      if (N > 1) dc = mk(L0v.x + P.cloud_centre[0], L0v.y + P.cloud_centre[1], L0v.z + P.cloud_centre[2]) - it.sf.p;
      const float n1 = fabsf(it.sf.n.x) + fabsf(it.sf.n.y) + fabsf(it.sf.n.z);
      const float scale = fabsf(dc.x) + fabsf(dc.y) + fabsf(dc.z) + fabsf(it.sf.p.x) + fabsf(it.sf.p.y) + fabsf(it.sf.p.z) + 1.0f;
      use = hit && (dot(it.sf.n, dc) + n1 * __builtin_fmaf(4e-6f, scale, dlt) > 0.0f);
    }
    lanemask use_m = wave_ballot(use);
    if (!use_m) continue;
    CandList cand;
    cand.reg = 0;
    cand.count = RT_CAND_OVERFLOW;
    cand.spheres = 0xFFFFFFFFu;
    cand.umbra = 0ull;
    if (N > 1 && P.cloud_delta > 0.0f && P.traversal == RT_TRAVERSAL_BVH && sc.n_triangles) {
      const V3 centre = mk(L0v.x + P.cloud_centre[0], L0v.y + P.cloud_centre[1], L0v.z + P.cloud_centre[2]);
      bool walk_tris = true, test_spheres = true;
      if (!CULL && P.recv_flags) {
        const uint32_t rf = rflags >> l;
        walk_tris = (use_m & ~wave_ballot(rf & 1u)) != 0ull;
        test_spheres = walk_tris || (use_m & ~wave_ballot((rf >> 8) & 1u)) != 0ull;
      }
      // per-cell candidate lists: the union of the lanes' lists replaces the BVH walk (as process_ray)
      uint32_t pre_reg = 0, pre_count = 0;
      bool have_pre = false;
      if (!CULL && P.cell_lists && walk_tris) {
        const uint4 lst = use ? lst_now : no_list;
        const lanemask unusable = use_m & (wave_ballot(cell == RT_NO_CELL) | wave_ballot((lst.x & 0xFFFFu) == RT_CELL_LIST_OVERFLOW));
        if (!unusable) {
          have_pre = true;
          for (lanemask todo = use_m & ~wave_ballot((lst.x & 0xFFFFu) == RT_CELL_LIST_END); todo;) {
            const int fl = __ffsll((long long)todo) - 1;
            const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cell, fl);
            auto add_slot = [&](uint32_t slot) {  // (uniform)
              if (slot >= RT_CELL_LIST_OVERFLOW) return;  // end marker
              const lanemask filled = pre_count >= 64u ? ~0ull : ((1ull << pre_count) - 1ull);
              if (wave_ballot(pre_reg == slot) & filled) return;  // another cell listed it already
              lane_put(pre_reg, pre_count, slot, lane);
              pre_count++;
            };
            auto add_pair = [&](uint32_t v) {
              const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)v, fl);
              add_slot(w & 0xFFFFu);
              add_slot(w >> 16);
            };
            add_pair(lst.x), add_pair(lst.y), add_pair(lst.z), add_pair(lst.w);
            todo &= ~wave_ballot(cell == c);
          }
          if (pre_count > 64u) have_pre = false;  // (more than a VGPR's worth of lanes: walk)
        }
      }
      if (test_spheres) {
        cand = collect_light_candidates<CULL>(sc, W, use, it.sf.p, centre, P, p_first, p_spread, P.cand_cap, nullptr, walk_tris, have_pre,
                                              pre_reg, pre_count);
      } else {
        cand.count = 0;  // every lane's cell is clear of triangles and spheres for this light: nothing to test
        cand.spheres = 0;
      }
    }
    if (cand.umbra) {  // lanes in the full shadow of an opaque triangle are done with this light
      use_m &= ~cand.umbra;
      if (!use_m) continue;
    }
    if (cand.count == RT_CAND_OVERFLOW && hard_ok) {
      // an incoherent wavefront: its (hit point, light) pairs go to rt_hard_kernel (N samples of a pair on N lanes)
      hard_push(P, use_m, it.sf.p, it.sf.n, it.d, it.sf.mat, l, it.pix, it.Wa, it.mult);
      continue;
    }
    SetRec s;
    s.item_base = item_base, s.light = l, s.count = cand.count, s.spheres = cand.spheres, s.use_m = use_m;
    const bool nothing = cand.count == 0 && cand.spheres == 0;
    if (nothing && RT_ARRIVE_INLINE) {
      long long fx, fy, fz;
      set_samples<CULL, SET_ARRIVE>(sc, P, W, it, use_m, l, cand, fx, fy, fz);
      if (lane_of(use_m)) own_x += fx, own_y += fy, own_z += fz;
      continue;
    }
    const int cls = nothing ? SET_ARRIVE : ((cand.count != RT_CAND_OVERFLOW && P.traversal != RT_TRAVERSAL_LINEAR && sc.n_triangles) ? SET_LIST : SET_WALK);
    set_write(P, set_base + l, cls, s, cand.reg);
  }
  add_terms<L0>(P, lds_fx, it, hit, own_x, own_y, own_z);
}

template <bool CULL, bool L0>
__device__ __forceinline__ void classify_body(const RtDevScene& sc, const RtDevParams& P, long long* lds_fx_all, unsigned long long* lds_cnt) {
  Wave wv;
  wave_init(wv);
  wave_flush_init(P, lds_cnt);
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (wave-uniform: item_base lives in an SGPR)
  long long* lds_fx = lds_fx_all + wave * 192u;
  if (L0) {
    // (one workgroup = one workgroup of K1's launch: its items are hitrec[blockIdx.x * 256 ...])
    classify_wave<CULL, true>(sc, P, wv, blockIdx.x * 256u + wave * 64u, gridDim.x * 256u, lds_fx);
  } else {
    const uint32_t n = uload(P.sort_hits);  // the rays of this level that hit something, in hit-point order
    for (uint32_t base = blockIdx.x * 256u; base < n; base += gridDim.x * 256u)
      classify_wave<CULL, false>(sc, P, wv, base + wave * 64u, n, lds_fx);
  }
  wave_flush(wv, P, 0ull, lds_cnt);
}

#ifndef RT_CLASSIFY_WAVES
#define RT_CLASSIFY_WAVES 6
#endif
__global__ __launch_bounds__(256, RT_CLASSIFY_WAVES) void rt_classify0_kernel(RtDevScene sc, RtDevParams P) {
  __shared__ long long lds_fx[4 * 192];
  __shared__ unsigned long long lds_cnt[20];
  if (P.flags & RT_FLAG_BACKFACE_CULLING)
    classify_body<true, true>(sc, P, lds_fx, lds_cnt);
  else
    classify_body<false, true>(sc, P, lds_fx, lds_cnt);
}
__global__ __launch_bounds__(256, RT_CLASSIFY_WAVES) void rt_classify_kernel(RtDevScene sc, RtDevParams P) {
  __shared__ long long lds_fx[4 * 192];
  __shared__ unsigned long long lds_cnt[20];
  if (P.flags & RT_FLAG_BACKFACE_CULLING)
    classify_body<true, false>(sc, P, lds_fx, lds_cnt);
  else
    classify_body<false, false>(sc, P, lds_fx, lds_cnt);
}

// ---- K3: one wavefront per (wavefront, light) set of one class -----------------------------------------------------------------
template <bool CULL, bool L0, int CLS>
__device__ __forceinline__ void sets_body(const RtDevScene& sc, const RtDevParams& P, long long* lds_fx_all, unsigned long long* lds_cnt) {
  Wave wv;
  wave_init(wv);
  wave_flush_init(P, lds_cnt);
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (wave-uniform: the loop below is a scalar loop)
  long long* lds_fx = lds_fx_all + wave * 192u;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t n_sets = uload(P.set_count + CLS);
  const uint32_t n_items = L0 ? P.set_items : uload(P.sort_hits);
  for (uint32_t s = blockIdx.x * 4u + wave; s < n_sets; s += gridDim.x * 4u) {
    const uint32_t set_id = uload(P.set_q + (size_t)CLS * P.set_cap + s);
    const uint4 h0 = uload(P.set_hdr + 2u * (size_t)set_id), h1 = uload(P.set_hdr + 2u * (size_t)set_id + 1u);
    const lanemask use_m = (lanemask)h1.x | ((lanemask)h1.y << 32);
    const ItemHit it = load_item<L0>(sc, P, h0.x + lane, n_items);
    CandList cand;
    cand.reg = 0;
    cand.count = CLS == SET_LIST ? h0.z : (CLS == SET_ARRIVE ? 0u : RT_CAND_OVERFLOW);
    cand.spheres = h0.w;
    cand.umbra = 0ull;
    if (CLS == SET_LIST) cand.reg = P.set_list[(size_t)set_id * 64u + lane];
    long long fx, fy, fz;
    set_samples<CULL, CLS>(sc, P, wv.ctx, it, use_m, h0.y, cand, fx, fy, fz);
    add_terms<L0>(P, lds_fx, it, lane_of(use_m), fx, fy, fz);
  }
  wave_flush(wv, P, 0ull, lds_cnt);
}

#ifndef RT_SETS_WAVES
#define RT_SETS_WAVES 8
#endif
#define RT_SETS_KERNEL(name, L0, CLS, waves)                                                  \
  __global__ __launch_bounds__(256, waves) void name(RtDevScene sc, RtDevParams P) {          \
    __shared__ long long lds_fx[4 * 192];                                                     \
    __shared__ unsigned long long lds_cnt[20];                                                \
    if (P.flags & RT_FLAG_BACKFACE_CULLING)                                                   \
      sets_body<true, L0, CLS>(sc, P, lds_fx, lds_cnt);                                       \
    else                                                                                      \
      sets_body<false, L0, CLS>(sc, P, lds_fx, lds_cnt);                                      \
  }
#if !RT_ARRIVE_INLINE  // (K2 finishes ARRIVE sets itself in the shipped build: their queue is never filled)
RT_SETS_KERNEL(rt_sets0_arrive_kernel, true, SET_ARRIVE, RT_SETS_WAVES)
RT_SETS_KERNEL(rt_sets_arrive_kernel, false, SET_ARRIVE, RT_SETS_WAVES)
#endif
RT_SETS_KERNEL(rt_sets0_list_kernel, true, SET_LIST, RT_SETS_WAVES)
RT_SETS_KERNEL(rt_sets0_walk_kernel, true, SET_WALK, RT_SETS_WAVES)
RT_SETS_KERNEL(rt_sets_list_kernel, false, SET_LIST, RT_SETS_WAVES)
RT_SETS_KERNEL(rt_sets_walk_kernel, false, SET_WALK, RT_SETS_WAVES)
