// rt_internal.h -- structures shared by the host API, the BVH builder and the HIP kernels.
// Not part of the public ABI (that is include/rt_hip.h).
#pragma once

#include <stdint.h>

#include <vector>

#include "../../../../include/rt_hip.h"

// ---- device-side scene layout (all arrays live in HBM, read-only during a render) -------------
//
// spheres      float4 {cx, cy, cz, r_sq}                      16 B / sphere
// sphere_mat   uint32 material row
// tri_isect    3 x float4 per triangle (BVH leaf order):      48 B / triangle
//                {v1.xyz, e1.x} {e1.yz, e2.xy} {e2.z, x.xyz}   x = e1 x e2 (ray independent part of
//                                                              Mat3::inversed, triangle.rs:174-177)
// tri_shade    float4 {n.xyz, bits(material row)}  (BVH leaf order)   16 B / triangle
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

#define RT_NODE_EMPTY 0xFFFFFFFFu  // c* value of an absent child (box is inverted, never hit)

struct RtBvh {
  std::vector<RtNode> nodes;        // nodes[0] is the root
  std::vector<uint32_t> tri_order;  // leaf slot -> canonical triangle index
  uint32_t n_leaves = 0;
  uint32_t max_depth = 0;
  uint32_t max_leaf = 0;
};

// Builds a binned-SAH BVH2 over the triangles (v1, e1, e2 as in rt_scene_desc).  Boxes are padded
// so that the fp32 slab test can never cull a triangle the literal intersection test accepts.
void rt_build_bvh(const float* v1, const float* e1, const float* e2, uint32_t n, RtBvh* out);

// ---- kernel argument block ---------------------------------------------------------------------
struct RtDevScene {
  const float4* spheres;
  const uint32_t* sphere_mat;
  const float4* tri_isect;
  const float4* tri_shade;
  const uint32_t* tri_id;
  const float4* materials;
  const float4* lights;
  const RtNode* nodes;
  uint32_t n_spheres, n_triangles, n_lights, n_nodes;
};

struct RtDevParams {
  uint32_t width, height;
  float focus[3];
  float fw, fh, fd, eps_distance, air_ior, ambient;
  uint32_t flags;
  uint32_t aa_rays;          // 0 = no anti-aliasing (one centre ray)
  const float* aa_offsets;   // device, [aa_rays][2]
  uint32_t light_mult, cloud_seed, n_cloud_sets;
  const float* cloud_sets;   // device, [n_sets][light_mult][3]
  uint32_t max_depth_reflection, max_depth_refraction;
  uint32_t win_x0, win_y0, win_w, win_h;
  uint32_t tile_size, n_ranks, rank;
  uint32_t traversal;
  // outputs
  uint32_t* argb;
  float* aux_rgb;
  int32_t* aux_hit_id;
  float* aux_hit_t;
  unsigned long long* counters;  // [5]: primary, reflection, refraction, shadow, pixels written
  // per-thread path stack (SoA, RT_PATH_FIELDS dwords per level), only when secondary rays are on
  float* path_stack;
  uint32_t path_levels;
  uint32_t path_threads;  // total threads of the launch (stride of the SoA)
};

#define RT_PATH_FIELDS 12u
#define RT_BLOCK_W 16u
#define RT_BLOCK_H 16u

// launches the render kernel on `stream`; returns hipError_t as int
int rt_launch_render(const RtDevScene& sc, const RtDevParams& p, void* stream);
