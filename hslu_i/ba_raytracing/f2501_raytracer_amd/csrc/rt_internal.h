// rt_internal.h -- structures shared by the host API, the BVH builder and the HIP kernels.
// Not part of the public ABI (that is include/rt_hip.h).
#pragma once

#include <stdint.h>

#include <vector>

#include "../../../../include/rt_hip.h"

// ---- device-side scene layout (all arrays live in HBM, read-only during a render) -------------
//
// spheres      float4 {cx, cy, cz, r_sq}                      16 B / sphere, then float radius bound[n_spheres]
// sphere_mat   uint32 material row
// tri_isect    3 x float4 per triangle REFERENCE (BVH leaf order, n_slots entries):  48 B each
//                {v1.xyz, e1.x} {e1.yz, e2.xy} {e2.z, x.xyz}   x = e1 x e2 (ray independent part of
//                                                              Mat3::inversed, triangle.rs:174-177)
// tri_shade    float4 {n.xyz, bits(material row)}: [0,n_slots) leaf order, then [n_slots, +n_triangles) canonical
// tri_id       uint32 canonical triangle index (insertion order) of the triangle in leaf slot i
// materials    3 x float4 per material: {r,g,b,metallic} {shininess, ior, opacity, boost}
//                {has_opacity,0,0,0}
// lights       2 x float4 per light: {x,y,z,intensity} {r,g,b,0}
// nodes        BVH2, 64 B per node (two child boxes + two child references)

struct RtNode {
  float lo0[3];
  uint32_t c0;  // count0 == 0: child node index; else first triangle slot
  float hi0[3];
  uint32_t n0;  // triangle count of child 0 (0 = internal)
  float lo1[3];
  uint32_t c1;
  float hi1[3];
  uint32_t n1;
};
static_assert(sizeof(RtNode) == 64, "node must be 64 bytes");

// The same tree in depth-first order with skip links ("threaded"): one box per entry, `skip` = the entry to continue with
// when the box is missed (or after a leaf), so a walk needs NO stack -- the per-lane (divergent) any-hit traversal of
// incoherent wavefronts.  leaf = triangle count << 24 | first slot (0 = internal node: its children follow).
struct RtThrNode {
  float lo[3];
  uint32_t skip;
  float hi[3];
  uint32_t leaf;
};
static_assert(sizeof(RtThrNode) == 32, "threaded node must be 32 bytes");

#define RT_NODE_EMPTY 0xFFFFFFFFu  // c* value of an absent child (its box is NaN: no slab test ever passes)
#define RT_TRI_DUPLICATE 0x80000000u  // tri_id flag: not the first reference of its triangle (slot order)
#define RT_TRI_TRANSMISSIVE 0x40000000u  // tri_id flag (device copy): the triangle's material lets light through
#define RT_TRI_INDEX_MASK 0x3FFFFFFFu

struct RtBvh {
  std::vector<RtNode> nodes;        // nodes[0] is the root
  std::vector<uint32_t> tri_order;  // leaf slot -> canonical triangle index (| RT_TRI_DUPLICATE)
  uint32_t n_leaves = 0;
  uint32_t max_depth = 0;
  uint32_t max_leaf = 0;
};

// Builds a binned-SAH BVH2 over the triangles (v1, e1, e2 as in rt_scene_desc).  Boxes are padded
// so that the fp32 slab test can never cull a triangle the literal intersection test accepts.
void rt_build_bvh(const float* v1, const float* e1, const float* e2, const uint8_t* no_split, uint32_t n,
                  const rt_bvh_tuning& tuning, RtBvh* out);

// ---- kernel argument block ---------------------------------------------------------------------
struct RtDevScene {
  // One allocation; every array is addressed as base + 32-bit byte offset (an SGPR offset of the scalar loads).
  const char* base;
  uint32_t off_spheres;     // float4 {cx, cy, cz, r_sq}
  uint32_t off_sphere_rad;  // float: upper bound of the radius (candidate culling)
  uint32_t off_sphere_mat;  // uint32 material row
  uint32_t off_tri_isect;   // 3 x float4 per leaf slot {v1.xyz, e1.x} {e1.yz, e2.xy} {e2.z, X.xyz}
  uint32_t off_tri_shade;   // float4 {normal, material} per leaf slot, then per canonical triangle
  uint32_t off_tri_id;      // uint32 per leaf slot: canonical index | RT_TRI_* flags
  uint32_t off_materials;   // 3 x float4
  uint32_t off_lights;      // 2 x float4
  uint32_t off_nodes;       // RtNode
  // 8 copies of nodes, one per direction octant o (bit a = direction negative along axis a): lo* hold the entry
  // planes and hi* the exit planes for that octant, children are in near-first order (octant o: [o * n_nodes ...])
  uint32_t off_nodes_oct;
  uint32_t off_nodes_thr;   // RtThrNode[n_thr]: the tree threaded for stackless per-lane walks
  // receiver records, canonical triangle order, 48 B: {Au, au0} {Av, av0} {bits(R), bits(first cell), 0, 0} -- barycentric
  // coordinates of a hit point p are u = Au.p + au0, v = Av.p + av0; the triangle's R x R cells start at `first cell`
  uint32_t off_recv;
  // sphere receivers: {Rs, first cell} per sphere -- a cube map of 6 x Rs x Rs cells over the directions from its centre
  uint32_t off_srecv;
  uint32_t n_thr;
  uint32_t n_spheres, n_triangles, n_lights, n_nodes;
  uint32_t n_slots;  // triangle references in leaf order (>= n_triangles with split clipping)
};

struct RtDevParams {
  uint32_t width, height;
  float focus[3];
  float fw, fh, fd, eps_distance, air_ior, ambient;
  uint32_t flags;
  uint32_t aa_rays;          // samples per pixel the reference casts; 0 = no anti-aliasing (one centre ray)
  // Bit-identical repeats of a sample offset are traced once (host: prepare()): aa_unique distinct offsets, thread u
  // of a pixel traces aa_offsets[u] with weight aa_mult[u]; sample q of the reference's sum is thread aa_src[q].
  uint32_t aa_unique;        // threads per pixel (== aa_rays when nothing repeats; 1 without anti-aliasing)
  uint32_t weighted;         // 1: some ray carries a multiplicity > 1 (counters need per-lane sums)
  const float* aa_offsets;   // device, [aa_unique][2]
  const uint32_t* aa_mult;   // device, [aa_unique]
  const uint32_t* aa_src;    // device, [aa_rays]
  uint32_t light_mult, cloud_seed, n_cloud_sets;
  // host-computed constants of the soft-shadow beam tests (uniform float arithmetic would sit in VGPRs):
  // delta = cloud_delta + 2 eps, delta + 1e-5, 0.998 eps, (1.3e-7 + 2.5e-6) eps, 1.01 eps + 2 * that, 1.98 eps
  float beam_delta, beam_delta_e5, beam_eps_push, beam_eps_ulp, beam_eps_o, beam_eps_198;
  const float4* cloud_sets;  // device, [n_sets][light_mult] {dx*fw, dy*fh, dz*fd, 0}: pre-scaled light-cloud offsets
  // bounding ball of all cloud offsets (scene units): offset of its centre from the light position and
  // its radius (computed on the host over every set; 0 = unknown -> no candidate sharing)
  float cloud_centre[3];
  float cloud_delta;
  uint32_t cand_cap;  // give up candidate sharing for a (wavefront, light) above this many slots (<= 64)
  // Receiver flags, one uint16 per receiver cell: bit l = no triangle can touch a soft-shadow ray from this cell towards
  // light l, bit 8 + l = no sphere can (rt_flags_kernel; nullptr = not in use).  flag_*: inputs of that kernel.
  const uint16_t* recv_flags;
  // Per-cell candidate lists, one uint4 (8 x 16-bit leaf slots; 0xFFFF = end, entry 0 == 0xFFFE = more than 8) per
  // (receiver cell, light): what survives the cell's fat beam, written by rt_flags_kernel together with the flags.  The
  // union of the lists of a wavefront's cells replaces its candidate walk (nullptr = not in use).
  const uint4* cell_lists;
  uint16_t* cell_list_out;
  uint16_t* flag_out;
  const float4* flag_geo;         // canonical triangles: {v1, bits(R)} {e1, bits(first cell)} {e2, 0}
  uint32_t n_cells;
  uint32_t n_tri_cells;  // cells [0, n_tri_cells) belong to triangles, the rest to spheres
  uint32_t max_depth_reflection, max_depth_refraction;
  uint32_t win_x0, win_y0, win_w, win_h;
  uint32_t tile_size, n_ranks, rank;
  uint32_t traversal;
  // outputs
  // multi-GPU: with stage_slot != nullptr `argb` is this rank's compact staging buffer and pixel (gx, gy) of tile
  // (tx, ty) goes to stage_slot[ty * stage_tiles_x + tx] * tile_size^2 + (gy % tile_size) * tile_size + gx % tile_size
  const uint32_t* stage_slot;
  uint32_t stage_tiles_x;
  uint32_t* argb;
  float* aux_rgb;
  int32_t* aux_hit_id;
  float* aux_hit_t;
  unsigned long long* counters;  // [RT_COUNTER_REPLICAS][16]: primary, reflection, refraction, shadow, written, wave stats
  // ---- ray streaming (only when reflections / refractions are enabled) ---------------------------
  // A queue holds 64-byte ray records, 4 float4 each (one HBM line per ray: the shade kernel fetches rays in hit-point
  // order, and a gathered record costs one line, not one line per field):
  //   {o.xyz, n_start}  {d.xyz, bits(depth, kind, multiplicity)}  {W.rgb, bits(pixel)}  {t, bits(hit id), bits(Morton key of the hit point), bits(rank in its bucket)}
  // the last quad is written by rt_trace_kernel.  Two queues are used alternately: level k reads one and appends its
  // children to the other.  ALL sizes stay on the device: a kernel reads how many rays / pairs / hits it has to process
  // from the counters the kernel before it wrote, and walks them with a grid-stride loop (the host only guesses the grid).
  float4* q_out;          // children of this launch are appended here (nullptr: no children wanted)
  uint32_t* q_out_count;  // device counter of q_out (counts dropped children too: it can exceed q_capacity)
  uint32_t* q_overflow;   // device counter of dropped children (must stay 0; checked by the host after the frame)
  uint32_t q_capacity;
  float4* q_in;           // secondary kernels: the rays of this level
  const uint32_t* q_in_count;  // device: how many (clamped to q_capacity by the reader)
  long long* acc;         // [W*H][4] fixed-point RGB accumulator + primary-hit flag (nullptr: direct write)
  // "Hard" (hit point, light) pairs: soft-shadow sets of INCOHERENT wavefronts (their shared candidate list overflows)
  // are not traced where they are found; every lane's pair is appended here and rt_hard_kernel traces it with the N
  // samples of the pair spread over N lanes.  4 float4 planes of hard_capacity + 64 entries (the last 64: dump slots
  // of idle lanes): {p, bits(material row)} {n, bits(light)} {view d, bits(pixel)} {W * atten, bits(multiplicity)}.
  float4* hard_q;         // nullptr: no deferral (no accumulator to add to, or no soft shadows)
  uint32_t* hard_count;   // device: pairs appended since rt_hard_kernel last ran (producers add, rt_hard_kernel reads)
  uint32_t* hard_stat;    // device: [0] dropped pairs (must stay 0), [1] largest hard_count seen this frame
  uint32_t hard_capacity;
  // hit-point ordering of a level's rays: a counting sort on the top sort_bits bits of the Morton key of the hit point
  // (misses are not sorted at all: they are not shaded).  rt_trace_kernel takes a ray's rank inside its bucket from the
  // histogram (one atomic per wavefront and bucket), two small kernels turn the histogram into offsets (and clear it),
  // rt_sort_place_kernel writes ray index -> sorted position.  Sizes never leave the device.
  uint32_t* sort_hist;    // [1 << sort_bits]: zero between uses
  uint32_t* sort_offs;    // [1 << sort_bits]: bucket -> first sorted position inside its tile of RT_SORT_TILE buckets
  uint32_t* sort_tile;    // [(1 << sort_bits) / RT_SORT_TILE]: tile -> first sorted position
  uint32_t* sort_hits;    // device scalar: rays of this level that hit something (= rays rt_shade_kernel shades)
  uint32_t* sh_idx;       // [q_capacity] sorted position -> ray index
  uint2* sort_slot;       // [q_capacity] ray -> {bucket or 0xFFFFFFFF for a miss, rank inside the bucket}
  uint32_t sort_bits;
  uint32_t batch_first_wg;  // primary kernel: workgroup offset of this batch
  // ... and its stride: launched workgroup b stands for workgroup first + ((b >> group_log2) * stride << group_log2) + (b & group mask) of
  // the frame's list -- stride 1: a contiguous range; stride 2: every other group of 2^group_log2 workgroups (the chains of a
  // frame with secondary rays interleave, so that each gets its share of the expensive regions)
  uint32_t batch_stride, batch_group_log2;
  // multi-GPU: the 16x16 super-tiles (window-relative index) that contain pixels of this rank's tiles;
  // nullptr = all super-tiles of the window
  const uint32_t* sup_list;
  uint32_t n_sup;         // number of super-tiles to render (listed, or all of the window)
  // calibration frame of RT_TILE_ORDER_COST: every wavefront of the primary kernel adds its run time (shader clock / 64)
  // to the entry of the super-tile (window-relative index) its first pixel lies in; nullptr = not measuring
  uint32_t* cost_map;
  // Morton key of a secondary hit point: q = (p - morton_lo) * morton_scale in [0, 1024)^3 (scene AABB, host)
  float morton_lo[3], morton_scale[3];
  // ---- phase-split pipeline (rt_phases.h; all nullptr / 0 in the fused pipeline) -----------------------------------------
  uint2* hitrec;        // K1 -> K2 / K3: {bits(t), hit id or -1} per primary work item of the launch (launched workgroup * 256 + thread)
  uint4* set_hdr;       // K2 -> K3: 2 x uint4 per (wavefront, light) set: {first item, light, candidate count, sphere mask} {lanes in use, 0, 0}
  uint32_t* set_list;   // 64 dwords per set: lane i = i-th candidate leaf slot (LIST sets only)
  uint32_t* set_q;      // [3][set_cap]: the set ids of each class (ARRIVE, LIST, WALK), appended by K2
  uint8_t* set_cls;     // [set_cap]: class + 1 of every set id K2 wrote (0 = none; every wavefront of K2 clears its own first), input of rt_compact_kernel
  uint32_t set_n;       // rt_compact_kernel: set ids of the launch (level 0), 0 = from the level's device-side hit count
  uint32_t set_lights;  // ... = n_lights
  uint32_t* set_count;  // device, [3]: sets per class of this level
  uint32_t set_cap;     // set ids are (first item / 64) * n_lights + light < set_cap: a class queue cannot overflow
  uint32_t set_items;   // level 0: work items of the launch (K3 re-derives a lane's camera ray from its item index)
  uint32_t resolve_counts_written;  // 1: rt_resolve_kernel counts the written pixels (the phase kernels do not)
  // ---- merged levels (rt_tuning.levels = RT_LEVELS_MERGED): ONE append-only queue; level k = its slice [*seg_lo, *seg_hi); nullptr otherwise
  const uint32_t* seg_lo;
  const uint32_t* seg_hi;
  uint32_t hit_spawns;  // rt_launch_hit: 1 = rt_hit_spawn_kernel (the camera rays' children are appended where their hits are found)
};

#define RT_QUEUE_QUADS 4u   // float4 per ray record
#define RT_SORT_TILE 4096u  // buckets per workgroup of the offset scan
#define RT_SORT_BITS_DEFAULT 22u
#define RT_COUNTER_REPLICAS 64u

// primary kernel: do the n_thr sample threads of a pixel share one wavefront (wasting at most 4 of its lanes)?
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline bool rt_primary_wave_local(uint32_t n_thr) { return n_thr <= 64u && (64u / n_thr) * n_thr >= 60u; }

// kernel launchers (rt_kernels.hip); return hipError_t as int
uint32_t rt_primary_pixels_per_wg(const RtDevParams& p);
uint32_t rt_primary_total_wgs(const RtDevParams& p);
int rt_launch_primary(const RtDevScene& sc, const RtDevParams& p, uint32_t n_wgs, void* stream);
// secondary kernels: n_wgs = the host's guess of the grid (every kernel walks its device-side count with a grid-stride loop)
int rt_launch_trace(const RtDevScene& sc, const RtDevParams& p, uint32_t n_wgs, void* stream);
int rt_launch_sort(const RtDevParams& p, uint32_t n_wgs_place, void* stream);  // histogram -> offsets -> sh_idx (rt_sort.hip)
int rt_launch_shade(const RtDevScene& sc, const RtDevParams& p, uint32_t n_wgs, void* stream);
int rt_launch_hard(const RtDevScene& sc, const RtDevParams& p, uint32_t n_wgs, void* stream);
int rt_launch_resolve(const RtDevParams& p, void* stream);
int rt_launch_flags(const RtDevScene& sc, const RtDevParams& p, void* stream);
// phase-split pipeline (rt_phases.h): level0 = the items are primary work items (else: the level's hits in hit-point order)
int rt_launch_hit(const RtDevScene& sc, const RtDevParams& p, uint32_t n_wgs, void* stream);
int rt_launch_classify(const RtDevScene& sc, const RtDevParams& p, bool level0, uint32_t n_wgs, void* stream);
int rt_launch_compact(const RtDevParams& p, uint32_t n_wgs, void* stream);
int rt_launch_sets(const RtDevScene& sc, const RtDevParams& p, bool level0, int cls, uint32_t n_wgs, void* stream);
bool rt_has_cost_kernel();       // built with COST=1 (rt_primary_cost_kernel: calibration frames of RT_TILE_ORDER_COST)
bool rt_phases_arrive_inline();  // this build's K2 finishes ARRIVE sets itself (no ARRIVE launches)
int rt_launch_selftest_math(const float* in, float* out_sqrt, float* out_rcp, uint32_t n, void* stream);
// multi-GPU gather, root side (rt_gather.hip): copies the other ranks' staged tiles (recv + rank_off[owner]) into the frame
int rt_launch_scatter(uint32_t* argb, const uint32_t* recv, const uint32_t* rank_off, const uint32_t* tile_slot, uint32_t width,
                      uint32_t height, uint32_t tile_size, uint32_t tiles_x, uint32_t n_ranks, void* stream);
