/*
 * rt_hip.h -- C ABI of the MI355X-native render loop (drop-in for the reference's
 * `Renderer::render` path).
 *
 * This is the boundary a Rust `impl Renderer<W,H,C> for HipRenderer` (or any other FFI host)
 * binds.  Plain pointers and sizes only; no C++/torch types cross it.
 *
 * What each entry point replaces in the reference (paths relative to the reference repo):
 *
 *   rt_scene_create   <- Scene<Vec3>{scene_objects, scene_lights}      src/scene/scene.rs:24-27
 *                        (flattened: spheres src/geometry/basic/sphere.rs:20-30,
 *                         triangles src/geometry/basic/triangle.rs:22-48,
 *                         materials src/raytracing/material.rs:15-19,78-89,
 *                         point lights src/scene/lighting/light.rs:162-169)
 *   rt_render         <- <RaytracerRenderer<C> as Renderer<W,H,C>>::render
 *                        src/renderer/raytracer_renderer.rs:1360-1378, driver
 *                        src/renderer/mod.rs:146-209; output layout = ImageBuffer<W,H>
 *                        src/image_buffer.rs:8-15 packed by OutputColorEncoder::to_output
 *                        src/output/window.rs:105-109
 *   rt_render_device  <- same, but the packed pixels stay in a caller-provided DEVICE buffer
 *                        (used by the multi-GPU gather and by bench.py, HBM-resident I/O)
 *   rt_params         <- the compile-time feature/const table: src/lib.rs:30-92,
 *                        src/renderer/raytracer_renderer.rs:55-127
 *   rt_scene_destroy  <- Drop of Scene
 *   rt_last_error     <- (reference panics: `unwrap()/expect()`); here: error codes + message
 *
 * All arithmetic on the path is fp32.  Hit ids are canonical object indices: spheres first
 * (insertion order), then triangles (insertion order); -1 = miss.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 4u

/* ---- error codes -------------------------------------------------------------------------- */
#define RT_OK 0
#define RT_ERR_INVALID_ARG (-1)
#define RT_ERR_NO_DEVICE (-2)
#define RT_ERR_HIP (-3)
#define RT_ERR_OOM (-4)
#define RT_ERR_UNSUPPORTED (-5)

/* ---- feature flags (reference: Cargo features read through cfg!()) ------------------------ */
#define RT_FLAG_REFLECTIONS 0x1u      /* feature "reflections"      raytracer_renderer.rs:216 */
#define RT_FLAG_REFRACTIONS 0x2u      /* feature "refractions"      raytracer_renderer.rs:232 */
#define RT_FLAG_BACKFACE_CULLING 0x4u /* feature "backface_culling" sphere.rs:137, triangle.rs:154 */
#define RT_FLAG_ANTI_ALIASING 0x8u    /* feature "anti_aliasing"    raytracer_renderer.rs:1199 */

/* ---- traversal selector (no reference counterpart: the reference scans linearly) ---------- */
#define RT_TRAVERSAL_BVH 0u    /* BVH over the triangles (result-preserving w.r.t. the scan) */
#define RT_TRAVERSAL_LINEAR 1u /* literal linear scan of all objects, raytracer.rs:48,180 */

/* material row layout inside rt_scene_desc.materials (stride RT_MATERIAL_STRIDE floats) */
#define RT_MATERIAL_STRIDE 9u
#define RT_MAT_R 0
#define RT_MAT_G 1
#define RT_MAT_B 2
#define RT_MAT_METALLIC 3
#define RT_MAT_SHININESS 4
#define RT_MAT_IOR 5         /* TransmissionProperties.refraction_index (raw field) */
#define RT_MAT_OPACITY 6     /* TransmissionProperties.opacity value */
#define RT_MAT_BOOST 7       /* TransmissionProperties.boost */
#define RT_MAT_HAS_OPACITY 8 /* SimdOption mask of opacity: 1.0f = Some, 0.0f = None */

/* light row layout inside rt_scene_desc.lights (stride RT_LIGHT_STRIDE floats).  The colour is
 * the one PointLight::new stores, i.e. ALREADY passed through maximize_value (light.rs:175-181). */
#define RT_LIGHT_STRIDE 7u

/* BVH builder knobs (no reference counterpart: the reference has no acceleration structure).  0 = default. */
typedef struct rt_bvh_tuning {
  uint32_t max_leaf;    /* triangles per leaf, default 4 */
  float tri_cost;       /* SAH cost of one triangle test relative to one node visit, default 2 */
  uint32_t split_depth; /* early split clipping: at most 2^depth references per triangle, default 0 = off */
  float split_gain;     /* split only if area(left) + area(right) < gain * area(whole), default 0.8 */
} rt_bvh_tuning;

typedef struct rt_scene_desc {
  uint32_t abi_version; /* RT_ABI_VERSION */

  uint32_t n_spheres;
  const float* sphere_center;      /* [n_spheres][3] */
  const float* sphere_r_sq;        /* [n_spheres]  radius*radius   sphere.rs:43 */
  const float* sphere_r_inv;       /* [n_spheres]  1/radius        sphere.rs:44 (unused by intersect) */
  const uint32_t* sphere_material; /* [n_spheres]  row in materials */

  uint32_t n_triangles;
  const float* tri_v1;          /* [n_triangles][3] vertex1           triangle.rs:35 */
  const float* tri_e1;          /* [n_triangles][3] vertex2 - vertex1 triangle.rs:65,89 */
  const float* tri_e2;          /* [n_triangles][3] vertex3 - vertex1 triangle.rs:66,90 */
  const float* tri_normal;      /* [n_triangles][3] stored face normal (may be non-unit) */
  const uint32_t* tri_material; /* [n_triangles] */

  uint32_t n_materials;
  const float* materials; /* [n_materials][RT_MATERIAL_STRIDE] */

  uint32_t n_lights;
  const float* lights; /* [n_lights][RT_LIGHT_STRIDE]: x,y,z, r,g,b, intensity */

  rt_bvh_tuning bvh; /* all 0 = defaults */

  /* Device memory the scene may spend on OPTIONAL acceleration tables on top of the scene proper (geometry + BVH, a few MB):
   * the receiver flags (2 bytes per receiver cell) and the per-cell candidate lists (16 bytes per cell and light).  The
   * reference's whole scene is < 1 MB (src/scene/scene.rs:24-27); a drop-in that shares a device should not quietly take a
   * gigabyte for a few percent.  0 = RT_SCENE_BUDGET_DEFAULT (128 MiB).  Tables that do not fit are coarsened (flags) or not
   * built (lists; rt_stats.notes says so) -- the image is the same either way.  rt_scene_memory_info reports what is held. */
  uint64_t device_budget_bytes;
} rt_scene_desc;
#define RT_SCENE_BUDGET_DEFAULT ((uint64_t)128 << 20)

/* Execution knobs that never change the image (no reference counterpart).  All 0 = defaults. */
#define RT_CAND_CAP_NONE 0xFFFFFFFFu
typedef struct rt_tuning {
  /* Soft shadows share one BVH walk per (wavefront, light): the walk collects at most this many candidate
   * triangles (1..64; 0 = default 64).  RT_CAND_CAP_NONE: no sharing, one BVH walk per shadow sample. */
  uint32_t shadow_candidate_cap;
  uint32_t chunk_log2;  /* log2 of the rays per secondary launch / primary batch; 0 = sized from free HBM */
  uint32_t no_aa_dedup; /* 1: trace every AA sample, also the bit-identical repeats of the sample table */
  uint32_t no_counters; /* 1: skip the ray counters of rt_stats (timing experiments) */
  /* rt_render_multi, testing only: use the RCCL calls even when several ranks share one GPU (real RCCL refuses such a
   * communicator; tests/mock_rccl checks the call sequence on a one-GPU box) */
  uint32_t multi_force_rccl;
  /* 1: no receiver flags.  Default 0: with soft shadows every triangle carries a grid of receiver cells, flagged per light
   * when no triangle / sphere can touch a shadow ray that starts in the cell (computed once per scene and light-cloud
   * size by rt_flags_kernel); wavefronts whose hit points all lie in clear cells skip the candidate walk -- same image */
  uint32_t no_receiver_flags;
  /* Launch order of the 16x16-pixel super-tiles of a frame (RT_TILE_ORDER_*).  The reference hands its tiles to a
   * work-stealing pool in shuffled order (src/image_buffer.rs:48-97); a GPU launch runs its workgroups in list order, and
   * cannot end before its longest wavefront does.  COST: the cost of every super-tile is measured once per scene and
   * frame shape (one calibration frame, shader-clock sums per super-tile) and the list is launched heaviest first. */
  uint32_t tile_order;
  /* Secondary rays are shaded in the order of their hit points: a counting sort on the top `sort_bits` bits of the 30-bit
   * Morton key of the hit point (12..24; 0 = default: 22, 24 for frames of more than 32 Mi primary work items).  More bits = neighbouring rays in a wavefront lie closer together
   * (their soft-shadow candidate walks are shared), at 8 bytes of device memory per bucket. */
  uint32_t sort_bits;
  /* 1: no per-cell candidate lists.  Default 0: rt_flags_kernel also LISTS, per receiver cell and light, the (up to 8)
   * triangles that survive the cell's fat beam; a wavefront whose hit points all lie in cells with complete lists takes
   * the union of those lists instead of walking the BVH for its soft-shadow candidates -- same candidates after the
   * lanes' own beam tests, same image.  16 bytes per cell and light of device memory. */
  uint32_t no_cell_lists;
  /* Chains a frame with secondary rays is split into (1..2; 0 = default 2).  The ray tree of a frame is a chain of launches,
   * one per level, and a launch cannot end before its longest wavefront does; two halves of the frame's primary work list
   * run as independent chains on two streams (the caller's and one of the library's, forked and joined with events) so that
   * the head of one chain's launch fills the drain of the other's.  Twice the queues of half the size -- same memory, same
   * image.  A host that keeps several frames in flight itself may prefer 1. */
  uint32_t sub_frames;
  /* Form of the render loop (RT_PHASES_*).  FUSED: one kernel per ray-tree level runs nearest hit, the five lights' shadow
   * classification and N-sample loops, shading and child spawning.  SPLIT: the same work as phase kernels -- hit -> per-hit
   * classification into queues of (wavefront, light) sets -> one kernel per class of set -> resolve (csrc/rt_phases.h); the
   * work item of the dominant kernels is a (wavefront, light) set, a fifth of a fused wavefront's.  Same integers in the pixel
   * accumulator, hence the same frame.  0 = the library's choice for the frame shape. */
  uint32_t phases;
  /* How the levels of a frame's ray tree are run (RT_LEVELS_*; fused kernels only).  CHAINED: per level trace -> sort -> shade,
   * the shade kernel appending the next level's rays.  MERGED: the kernel that finds a ray's hit also appends its children, so the
   * levels are traced back to back (level k+1 does not wait for level k to be shaded) and ALL levels are then shaded by ONE launch in
   * ONE hit-point order: one drain instead of one per level, and rays of every depth that hit neighbouring points share a wavefront.
   * PIPELINED: the levels are traced back to back as in MERGED, each sorted on its own, and level k is shaded -- on one of two streams
   * of the library's, alternately -- as soon as it has been traced and sorted: the trace launches (latency bound) run under the shade
   * launches (issue bound), and the head of level k+1's shading fills the compute units the drain of level k's leaves idle.
   * Same integer pixel sums, same frame.  0 = the library's choice. */
  uint32_t levels;
} rt_tuning;
#define RT_LEVELS_DEFAULT 0u
#define RT_LEVELS_CHAINED 1u
#define RT_LEVELS_MERGED 2u
#define RT_LEVELS_PIPELINED 3u
#define RT_PHASES_DEFAULT 0u
#define RT_PHASES_FUSED 1u
#define RT_PHASES_SPLIT 2u
#define RT_PHASES_FUSED_DEFER 3u /* fused kernels; frames without secondary rays also sum through the accumulator and defer incoherent soft-shadow sets to rt_hard_kernel */
#define RT_TILE_ORDER_DEFAULT 0u
#define RT_TILE_ORDER_ROW_MAJOR 1u
#define RT_TILE_ORDER_COST 2u

typedef struct rt_params {
  uint32_t abi_version; /* RT_ABI_VERSION */
  uint32_t width;       /* WINDOW_WIDTH  lib.rs:50 */
  uint32_t height;      /* WINDOW_HEIGHT lib.rs:61 */

  float focus[3];     /* RENDER_RAY_FOCUS lib.rs:88-89 */
  float fw;           /* WINDOW_TO_SCENE_WIDTH_FACTOR  lib.rs:81 */
  float fh;           /* WINDOW_TO_SCENE_HEIGHT_FACTOR lib.rs:82 */
  float fd;           /* WINDOW_TO_SCENE_DEPTH_FACTOR  lib.rs:83 */
  float eps_distance; /* Vector3DOperations::default_epsilon_distance vector.rs:697-700 */
  float air_ior;      /* DEFAULT_REFRACTION_INDEX lib.rs:92 */
  float ambient;      /* ambient intensity 0.08, raytracer_renderer.rs:754 */

  uint32_t flags; /* RT_FLAG_* */

  /* anti-aliasing sample table, already scaled and direction-multiplied
   * (bundle_rays_for_simd_antialiased_raytracing, raytracer_renderer.rs:1021-1138):
   * sample k origin = (x + aa_offsets[2k], y + aa_offsets[2k+1], 0).  Ignored (one centre ray)
   * unless RT_FLAG_ANTI_ALIASING is set. */
  uint32_t aa_rays;
  const float* aa_offsets; /* [aa_rays][2] */

  /* soft-shadow light cloud (PointLight::to_point_light_cloud<N>, light.rs:183-225).
   * light_mult = N.  N == 1: the light itself.  N > 1: light j of the cloud of light l at pixel
   * p sits at  pos_l + cloud_sets[set][j] * (fw, fh, fd)  with intensity (1/N)*I_l, where
   * set = rt_cloud_hash(cloud_seed, p, l) % n_cloud_sets  -- the seeded, reproducible stand-in
   * for the reference's unseeded per-pixel Poisson3D set. */
  uint32_t light_mult;
  uint32_t cloud_seed;
  uint32_t n_cloud_sets;
  const float* cloud_sets; /* [n_cloud_sets][light_mult][3], in "pixel units" */

  uint32_t max_depth_reflection; /* RAYTRACE_REFLECTION_MAX_DEPTH raytracer_renderer.rs:55 */
  uint32_t max_depth_refraction; /* RAYTRACE_REFRACTION_MAX_DEPTH raytracer_renderer.rs:65 */

  /* sub-rectangle to render (ChunkView, image_buffer.rs:178-251).  win_w == 0 -> full frame. */
  uint32_t win_x0, win_y0, win_w, win_h;

  /* tile ownership for multi-GPU: RENDER_STRIDE x RENDER_STRIDE tiles (renderer/mod.rs:84-90,
   * image_buffer.rs:48-97); this call renders the tiles with rt_tile_owner(tx, ty, n_ranks) == rank.
   * n_ranks <= 1 -> all tiles. */
  uint32_t tile_size; /* 0 -> 48 */
  uint32_t n_ranks;
  uint32_t rank;

  uint32_t traversal; /* RT_TRAVERSAL_* */

  rt_tuning tuning; /* all 0 = defaults */
} rt_params;

/* optional per-pixel debug planes for parity checks (all nullable, caller-owned, W*H each) */
typedef struct rt_aux {
  float* rgb;      /* [H*W][3] un-quantised linear RGB of written pixels (else untouched) */
  int32_t* hit_id; /* [H*W] canonical object index hit by the first sample's primary ray, -1 miss */
  float* hit_t;    /* [H*W] its distance (untouched on miss) */
} rt_aux;

typedef struct rt_stats {
  uint64_t rays_primary;    /* lanes entering cast_ray as camera rays  raytracer.rs:162 */
  uint64_t rays_reflection; /* ... as reflection children (raytracer_renderer.rs:698) */
  uint64_t rays_refraction; /* ... as refraction children (raytracer_renderer.rs:493) */
  uint64_t rays_shadow;     /* has_any_intersection calls, raytracer.rs:24 */
  uint64_t pixels_written;
  /* Rays actually traced on the GPU (nearest-hit searches).  rays_primary/reflection/refraction/shadow count what
   * the reference casts; AA samples whose origin offsets are bit-identical repeats of another sample (7 of 16 with
   * the deterministic table, 15 of 24 with extreme_quality: raytracer_renderer.rs:107-122,1111-1116) are traced once
   * and weighted by their multiplicity, so rays_traced <= the sum of the three.  CPU oracle: equal to the sum. */
  uint64_t rays_traced;
  double kernel_ms; /* device time of the render kernel(s) (CPU oracle: wall time) */
  double total_ms;  /* wall time of the call incl. copies */
  double d2h_ms;    /* rt_render / rt_render_multi: wall time of the device -> host copy of the packed pixels */
  double gather_ms; /* rt_render_multi: device time of the tile gather on the root (RCCL recv + scatter kernel) */
  /* The wave_* work statistics below are filled only by the statistics build of the library
   * (`make STATS=1` -> librt_hip_stats.so, used by tools/perf_ab.py); the shipped kernels leave them 0.
   * GPU only (0 from the CPU oracle): SIMD efficiency of the ray loop.  wave_ray_passes = number of
   * wavefront-level trips through cast_ray + shading; wave_ray_lanes = live lanes summed over those
   * trips.  lanes / (64 * passes) = fraction of the 64-wide machine doing useful ray work. */
  uint64_t wave_ray_passes;
  uint64_t wave_ray_lanes;
  /* GPU only: wavefront-level BVH work (one count per wave, not per lane) */
  uint64_t wave_nearest_nodes; /* BVH nodes fetched by nearest-hit traversals */
  uint64_t wave_nearest_tris;  /* triangle records tested by nearest-hit traversals */
  uint64_t wave_shadow_nodes;  /* same for shadow rays */
  uint64_t wave_shadow_tris;
  uint64_t wave_shadow_passes; /* wavefront-level shadow-ray traversals */
  uint64_t wave_nearest_tris_exact; /* triangle tests that passed the conservative pre-filter */
  uint64_t wave_shadow_tris_exact;
  /* RT_NOTE_* bits: fast paths this frame did NOT take, and why (the image is the same either way) */
  uint32_t notes;
  uint32_t reserved0;
  uint64_t queue_bytes; /* device memory the frame's ray queues, hard-pair queue and sort workspace hold (0 without secondary rays) */
  /* rt_render: device time between the start of the call's device work and its first render kernel -- sample-table uploads and, on
   * the first frame that needs them, rt_flags_kernel (receiver flags + per-cell lists).  NOT part of kernel_ms. */
  double setup_ms;
  uint64_t scene_bytes; /* device memory the scene handle holds in total right now (rt_scene_memory_info.bytes_total) */
} rt_stats;
#define RT_NOTE_RECV_FLAGS_OFF_LIGHTS 0x1u    /* receiver flags need n_lights <= 8 */
#define RT_NOTE_RECV_FLAGS_OFF_CULLING 0x2u   /* ... and no backface culling */
#define RT_NOTE_RECV_FLAGS_OFF_TRAVERSAL 0x4u /* ... and RT_TRAVERSAL_BVH */
#define RT_NOTE_RECV_FLAGS_OFF_TUNING 0x8u    /* switched off by rt_tuning (no_receiver_flags / shadow_candidate_cap) */
#define RT_NOTE_RECV_FLAGS_OFF_SCENE 0x10u    /* no receiver cells (no triangles / degenerate scene) or no light cloud */
#define RT_NOTE_HARD_PAIRS_OFF 0x20u          /* incoherent soft-shadow sets are traced inline (light_mult > 64, linear, cap) */
#define RT_NOTE_FRAME_BATCHED 0x40u           /* the ray queues did not fit: the frame ran in several primary batches */
#define RT_NOTE_CELL_LISTS_OFF 0x80u          /* no per-cell candidate lists (receiver flags off, > 65 533 leaf slots, over the scene's budget, tuning) */
#define RT_NOTE_TILE_ORDER_COST_OFF 0x100u    /* RT_TILE_ORDER_COST asked for, library built without the calibration kernel (make COST=1): row-major */
#define RT_NOTE_FRAME_DROPPED_WORK 0x200u     /* an asynchronously rendered frame of this shape overflowed a ray / pair queue (its counters came back
                                                 later): some of its secondary terms are missing; the queues have grown and the next frame is verified */

typedef struct rt_scene rt_scene; /* opaque: device copies + BVH */

/* the two ABI-spec hashes below are also called from the HIP kernels */
#if defined(__HIPCC__)
#define RT_HOSTDEV __host__ __device__
#else
#define RT_HOSTDEV
#endif

/* deterministic hash that selects the per-(pixel, light) cloud set; part of the ABI spec */
RT_HOSTDEV static inline uint32_t rt_cloud_hash(uint32_t seed, uint32_t pixel, uint32_t light) {
  uint32_t h = seed * 0x9E3779B1u;
  h ^= (pixel + 0x7F4A7C15u) * 0x85EBCA6Bu;
  h ^= (light + 0x165667B1u) * 0xC2B2AE35u;
  h ^= h >> 16;
  h *= 0x7FEB352Du;
  h ^= h >> 15;
  h *= 0x846CA68Bu;
  h ^= h >> 16;
  return h;
}

/* tile -> owning rank for multi-GPU interleaving; part of the ABI spec.  A lattice interleave
 * (tx + S*ty) mod n with S the smallest odd number >= 3 coprime to n: neighbouring tiles always
 * belong to different ranks, so spatial cost hot-spots (glass sphere vs background) spread evenly,
 * and every rank owns the same number of tiles +-1 per row. */
RT_HOSTDEV static inline uint32_t rt_tile_owner(uint32_t tile_x, uint32_t tile_y, uint32_t n_ranks) {
  if (n_ranks <= 1u) return 0u;
  uint32_t s = 3u;
  for (;;) {
    uint32_t a = s, b = n_ranks;
    while (b) {
      uint32_t t = a % b;
      a = b;
      b = t;
    }
    if (a == 1u) break;
    s += 2u;
  }
  return (tile_x + s * tile_y) % n_ranks;
}

/* ---- entry points --------------------------------------------------------------------------- */

/* number of HIP devices visible (0 if none); never fails */
int rt_device_count(void);

/* Uploads the scene to `device`, builds the triangle BVH.  Caller keeps ownership of all host
 * arrays (they may be freed after the call returns). */
int rt_scene_create(const rt_scene_desc* desc, int device, rt_scene** out);

/* Renders into a HOST buffer of width*height packed 0xFFRRGGBB pixels.  Only pixels whose ray
 * hit something are written (miss pixels keep the caller's fill, image_buffer.rs:27-37).
 * Blocks until the buffer is complete.  aux/stats may be NULL; aux pointers are HOST pointers. */
int rt_render(rt_scene* scene, const rt_params* params, uint32_t* argb, const rt_aux* aux,
              rt_stats* stats);

/* Same, but `argb_dev` (and aux pointers) are DEVICE pointers on the scene's device and all work
 * is enqueued on `hip_stream` (a hipStream_t, NULL = default stream).  Without reflections /
 * refractions this is one asynchronous kernel launch; with them the call drives the ray-streaming
 * passes and returns when the last pass has been enqueued (it synchronises the stream in between to
 * read queue sizes).  Ray counters: after the caller has synchronised, rt_render_collect_stats (those of the frame
 * enqueued last).
 * Consecutive frames of one scene may be enqueued on DIFFERENT streams: the library orders what they share (two
 * counter blocks used alternately; frames with secondary rays own the ray queues and wait for every earlier frame), so
 * two frames without secondary rays overlap -- the head of one fills the compute units the drain of the other leaves
 * idle. */
int rt_render_device(rt_scene* scene, const rt_params* params, uint32_t* argb_dev,
                     const rt_aux* aux_dev, void* hip_stream);
int rt_render_collect_stats(rt_scene* scene, rt_stats* stats);

void rt_scene_destroy(rt_scene* scene);

/* ---- progressive read-back: the frame lands in the caller's buffer band by band WHILE it is being rendered ----------------
 *
 * Reference: main() spawns a thread that calls render(&buffer, &scene) and meanwhile blits the same buffer in its window loop
 * (src/main.rs:327-347); the buffer is [AtomicU32], written tile by tile with relaxed stores and read with relaxed loads
 * (src/image_buffer.rs:39-44,243-250).  Here `begin` starts the library's own render thread: the frame (or params' window) is
 * rendered in bands of `band_rows` rows (0 = tile_size, one row of RENDER_STRIDE tiles: renderer/mod.rs:84-90) on a stream of
 * its own, every finished band is copied to pinned host memory, and a counter says how many rows have landed.  `poll` -- the
 * UI thread's side; it never blocks -- copies the rows that are new since the last poll into `argb` and reports the count; rows
 * [win_y0, win_y0 + rows_done) of `argb` are then final, the rows below still hold the caller's fill.  `end` waits for the
 * rest, hands it over and frees the handle; its stats are summed over the bands.  `argb` (HOST, W*H, pre-filled by the caller)
 * must stay valid until `end`; params and its tables are copied by `begin`.  One progressive render per scene at a time, and no
 * other render call on that scene until `end`.  The final buffer equals rt_render's. */
typedef struct rt_progress rt_progress;
int rt_render_begin(rt_scene* scene, const rt_params* params, uint32_t* argb, uint32_t band_rows, rt_progress** out);
int rt_render_poll(rt_progress* progress, uint32_t* rows_done, int* finished);
int rt_render_end(rt_progress* progress, rt_stats* stats);

/* thread-local message for the last non-RT_OK return on this thread */
const char* rt_last_error(void);

/* hash of the sources and flags this library was built from (profiles/ summaries record it: a rocprof summary is
 * only quoted for the build it was taken on) */
const char* rt_build_id(void);

/* diagnostics: the kernels' correctly rounded sqrt and reciprocal (normal-range sequences, csrc/rt_kernels.hip
 * exact_sqrt / exact_rcp) evaluated on `n` host values on `device` -- tests compare them with IEEE sqrtf / division */
int rt_selftest_exact_math(int device, const float* in, float* out_sqrt, float* out_rcp, uint32_t n);

/* introspection for DESIGN.md / tests: BVH size of a created scene */
typedef struct rt_bvh_info {
  uint32_t n_nodes;
  uint32_t n_leaves;
  uint32_t max_depth;
  uint32_t max_leaf_size;
  uint64_t bytes_nodes;
  uint64_t bytes_triangles;
  uint32_t n_references; /* triangle references in the leaves (>= n_triangles: split clipping) */
} rt_bvh_info;
int rt_scene_bvh_info(const rt_scene* scene, rt_bvh_info* out);

/* What a scene handle holds in device memory, by purpose (the reference's Scene: src/scene/scene.rs:24-27, < 1 MB of host memory). */
typedef struct rt_scene_info {
  uint64_t bytes_geometry;    /* spheres, triangle records (intersection + shading), ids, materials, lights, receiver records */
  uint64_t bytes_bvh;         /* BVH nodes + the 8 per-octant copies + the threaded copy */
  uint64_t bytes_flags;       /* receiver flags + the cell geometry rt_flags_kernel reads (optional table, under the budget) */
  uint64_t bytes_cell_lists;  /* per-cell candidate lists (optional table, under the budget; 0 = not built) */
  uint64_t bytes_tables;      /* AA samples, light clouds, tile lists, counters */
  uint64_t bytes_workspace;   /* ray queues, sort workspace, pair queues, set records, pixel accumulators of every frame slot in use */
  uint64_t bytes_frames;      /* frame buffer + aux planes rt_render keeps for host callers */
  uint64_t bytes_total;
  uint64_t budget_bytes;      /* rt_scene_desc.device_budget_bytes as applied (bounds bytes_flags + bytes_cell_lists) */
  uint32_t n_receiver_cells;
  uint32_t cell_lists_built;  /* 1 = the lists exist (they are built by the first frame with soft shadows) */
} rt_scene_info;
int rt_scene_memory_info(const rt_scene* scene, rt_scene_info* out);

/* ---- multi-GPU: tile-partitioned frame + ONE gather of the packed pixels to rank 0 (RCCL over xGMI) ---------
 *
 * Reference: Renderer::render hands RENDER_STRIDE tiles to rayon workers that all write one ImageBuffer
 * (src/renderer/mod.rs:80-94,146-209, src/image_buffer.rs:48-97).  Here rank r renders the tiles with
 * rt_tile_owner(tx, ty, n_ranks) == r on its own GPU (scene replicated), straight into a rank-compact staging
 * buffer; the gather is one ncclGroupStart / ncclRecv x (n-1) on the root, ncclSend on the others / ncclGroupEnd --
 * every peer sends over its own xGMI link, no ring -- followed by a scatter kernel on the root.  No other
 * communication exists on the path.
 *
 * Staging layout (ABI spec): a rank's staging buffer holds its tiles in row-major tile order, each as
 * tile_size x tile_size pixels, row-major inside the tile (edge tiles padded); a staged 0 = "no hit" (the pixel
 * keeps the caller's fill).  rt_gather_layout computes it on the host (no GPU needed). */

/* tile_slot[ty * tiles_x + tx] = index of tile (tx, ty) inside its owner's staging buffer; tiles_per_rank[r] =
 * number of tiles rank r owns.  tiles_x = ceil(width / tile_size).  Either output may be NULL.  tile_size 0 -> 48. */
int rt_gather_layout(uint32_t width, uint32_t height, uint32_t tile_size, uint32_t n_ranks, uint32_t* tile_slot,
                     uint32_t* tiles_per_rank);

/* (a) ONE process drives all GPUs -- the shape of the reference's single-process Renderer::render.  per_gpu[i] is
 * the same scene created on GPU i (rt_scene_create(desc, device_i, ...)); per_gpu[0]'s GPU is the root.  argb is a
 * HOST buffer, W*H, as in rt_render; params->n_ranks / rank are ignored (n_gpu and i are used).  Blocks until argb is
 * complete.  stats (nullable): ray counters summed over the GPUs, kernel_ms = slowest GPU's render, gather_ms, d2h_ms.
 * n_gpu == 1 is rt_render without aux planes.  Scenes on distinct GPUs use RCCL; several scenes on ONE GPU (only
 * useful to rehearse the tile logic on a single-GPU machine) are gathered with device-to-device copies. */
int rt_render_multi(rt_scene* const* per_gpu, int n_gpu, const rt_params* params, uint32_t* argb, rt_stats* stats);
/* The same frame in two halves, for a host that renders frame after frame (the reference's window loop calls render()
 * once per displayed frame, src/main.rs:342-347): `begin` uploads the caller's fill, enqueues every GPU's render and
 * the gather, and returns a ticket (frames with reflections / refractions block until their ray-queue levels are through,
 * as rt_render_device does); `end` waits for that frame and copies it into the argb given to `begin` (which must stay
 * valid until then).  TWO frames may be in flight: the head of frame k+1 fills the compute units that the drain of
 * frame k leaves idle -- a rank's share of a frame is a sub-millisecond launch that cannot end before its longest
 * wavefront does.  A third `begin` before an `end` is refused (RT_ERR_INVALID_ARG).  rt_render_multi = begin + end. */
int rt_render_multi_begin(rt_scene* const* per_gpu, int n_gpu, const rt_params* params, uint32_t* argb, int* ticket);
int rt_render_multi_end(int ticket, rt_stats* stats);
/* frees the communicators / staging buffers rt_render_multi caches between calls.  The cache is deliberately NOT freed
 * at process exit (the HIP / RCCL runtimes may be gone by then); call this to release it earlier. */
void rt_multi_release(void);

/* (b) one process per GPU (torch.distributed / MPI style launchers): rank 0 makes an id, the host ships its 128 bytes
 * to the other ranks by any out-of-band means, every rank creates its communicator (collective call). */
#define RT_COMM_ID_BYTES 128
typedef struct rt_comm rt_comm;
int rt_comm_unique_id(uint8_t id[RT_COMM_ID_BYTES]);
int rt_comm_create(const uint8_t* id, uint32_t n_ranks, uint32_t rank, int device, rt_comm** out); /* n_ranks 1: id may be NULL */
void rt_comm_destroy(rt_comm* comm);
/* Renders this rank's tiles and takes part in the gather; everything is enqueued on hip_stream (with reflections /
 * refractions the render part synchronises the stream between ray-queue levels, as rt_render_device does).  On rank 0
 * argb_dev (DEVICE, W*H, pre-filled by the caller) holds the complete frame once the stream has drained; on the
 * other ranks it is not touched and may be NULL.  params->n_ranks / rank are ignored (the communicator's are used). */
/* The render runs on hip_stream, the gather on a stream the communicator owns (in call order, behind this rank's render);
 * hip_stream then waits for the gather, so work enqueued on it afterwards sees the complete frame.  Staging and receive
 * buffers are double buffered: frame k+1 may be enqueued on ANOTHER stream while frame k drains and travels (two frames
 * in flight; give rank 0 a second argb_dev for it).
 * Failure on one rank: a rank whose render fails still sends its (zeroed) tiles / posts its receives and returns the
 * error afterwards, so its peers are not left waiting; a rank that cannot even set the frame up (invalid arguments,
 * no memory for the staging buffers) aborts the communicator (ncclCommAbort) -- the peers' calls fail instead of hanging,
 * and the communicator must be re-created. */
int rt_render_gather_device(rt_scene* scene, rt_comm* comm, const rt_params* params, uint32_t* argb_dev, void* hip_stream);

#define RT_TRANSPORT_NONE 0u  /* one rank: nothing to gather */
#define RT_TRANSPORT_RCCL 1u  /* ncclSend / ncclRecv */
#define RT_TRANSPORT_LOCAL 2u /* rt_render_multi with several scenes on one GPU: device-to-device copies */
typedef struct rt_gather_info {
  double render_ms;        /* device time of this rank's render kernels in the last call */
  double gather_ms;        /* device time from the end of the render to the end of send / recv (+ scatter on the root) */
  uint64_t bytes_sent;     /* by this rank in the last call */
  uint64_t bytes_received; /* root only */
  uint32_t n_ranks, rank;  /* as RCCL reports them (ncclCommCount / ncclCommUserRank) when transport is RCCL */
  uint32_t tiles_owned;
  uint32_t transport; /* RT_TRANSPORT_* */
} rt_gather_info;
/* timings / sizes of the last rt_render_gather_device on this communicator; call after the stream has drained */
int rt_comm_last_gather(rt_comm* comm, rt_gather_info* out);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
