/*
 * c_abi_example.c -- the drop-in boundary used from plain C (what a Rust/cgo/JNI binding does):
 * builds a two-object scene, renders 96x80 through rt_scene_create / rt_render and prints a checksum; then renders
 * the same frame through the multi-GPU entry point rt_render_multi (n_gpu = 1, and 3 tile-partitioned "ranks" on the one
 * GPU as a rehearsal of the gather) and checks that the images are identical.
 *
 *   gcc -I include examples/c_abi_example.c -L hslu_i/ba_raytracing/f2501_raytracer_amd -lrt_hip \
 *       -Wl,-rpath,$PWD/hslu_i/ba_raytracing/f2501_raytracer_amd -o /tmp/c_abi_example
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rt_hip.h"

int main(void) {
  /* one diffuse sphere in front of one big triangle, one light */
  const float sphere_center[3] = {0.5f, 0.4f, 0.5f};
  const float r = 0.2f;
  const float sphere_r_sq[1] = {r * r}, sphere_r_inv[1] = {1.0f / r};
  const uint32_t sphere_material[1] = {0};
  const float tri_v1[3] = {-1.0f, -1.0f, 0.9f}, tri_e1[3] = {3.0f, 0.0f, 0.0f}, tri_e2[3] = {0.0f, 3.0f, 0.0f};
  const float tri_normal[3] = {0.0f, 0.0f, -1.0f};
  const uint32_t tri_material[1] = {1};
  const float materials[2 * RT_MATERIAL_STRIDE] = {
      1.0f, 0.2f, 0.2f, 0.0f, 0.3f, 1.0f, 0.0f, 0.0f, 0.0f, /* red, shiny */
      0.5f, 0.75f, 0.75f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f /* wall */
  };
  const float lights[RT_LIGHT_STRIDE] = {0.3f, 0.1f, 0.0f, 1.0f, 1.0f, 1.0f, 0.8f};

  rt_scene_desc d;
  memset(&d, 0, sizeof(d));
  d.abi_version = RT_ABI_VERSION;
  d.n_spheres = 1;
  d.sphere_center = sphere_center;
  d.sphere_r_sq = sphere_r_sq;
  d.sphere_r_inv = sphere_r_inv;
  d.sphere_material = sphere_material;
  d.n_triangles = 1;
  d.tri_v1 = tri_v1;
  d.tri_e1 = tri_e1;
  d.tri_e2 = tri_e2;
  d.tri_normal = tri_normal;
  d.tri_material = tri_material;
  d.n_materials = 2;
  d.materials = materials;
  d.n_lights = 1;
  d.lights = lights;

  rt_params p;
  memset(&p, 0, sizeof(p));
  p.abi_version = RT_ABI_VERSION;
  p.width = 96;
  p.height = 80;
  const float sh = 80.0f / 96.0f, sd = (1.0f + sh) / 2.0f;
  p.focus[0] = 0.5f;
  p.focus[1] = sh / 2.0f;
  p.focus[2] = -1.9f * sd;
  p.fw = 1.0f / 96.0f;
  p.fh = sh / 80.0f;
  p.fd = sd / 88.0f;
  p.eps_distance = 1.1920929e-7f * (100.0f * (1.0f + sh + sd) / 3.0f);
  p.air_ior = 1.000293f;
  p.ambient = 0.08f;
  p.light_mult = 1;
  p.max_depth_reflection = 9;
  p.max_depth_refraction = 8;
  p.tile_size = 48;

  if (rt_device_count() <= 0) {
    printf("no HIP device: ABI links, nothing rendered\n");
    return 0;
  }
  rt_scene* scene = NULL;
  if (rt_scene_create(&d, 0, &scene) != RT_OK) {
    fprintf(stderr, "rt_scene_create: %s\n", rt_last_error());
    return 1;
  }
  uint32_t* argb = (uint32_t*)calloc((size_t)p.width * p.height, 4);
  rt_stats st;
  if (rt_render(scene, &p, argb, NULL, &st) != RT_OK) {
    fprintf(stderr, "rt_render: %s\n", rt_last_error());
    return 1;
  }
  unsigned long long sum = 0;
  for (uint32_t i = 0; i < p.width * p.height; i++) sum += argb[i];
  printf("pixels written %llu of %u, rays %llu, shadow rays %llu, checksum %llx, kernel %.3f ms\n",
         (unsigned long long)st.pixels_written, p.width * p.height, (unsigned long long)st.rays_primary,
         (unsigned long long)st.rays_shadow, sum, st.kernel_ms);
  /* the multi-GPU side of Renderer::render: one process, n scenes (one per GPU), one gather to the root */
  int bad = 0;
  {
    uint32_t* multi = (uint32_t*)calloc((size_t)p.width * p.height, 4);
    rt_scene* one[1] = {scene};
    rt_stats sm;
    if (rt_render_multi(one, 1, &p, multi, &sm) != RT_OK) {
      fprintf(stderr, "rt_render_multi(1): %s\n", rt_last_error());
      return 1;
    }
    bad |= memcmp(multi, argb, (size_t)p.width * p.height * 4) != 0 || sm.pixels_written != st.pixels_written;
    /* three tile-partitioned ranks rehearsed on the one GPU (16x16 tiles so that every rank owns some) */
    rt_scene* three[3] = {scene, NULL, NULL};
    if (rt_scene_create(&d, 0, &three[1]) != RT_OK || rt_scene_create(&d, 0, &three[2]) != RT_OK) {
      fprintf(stderr, "rt_scene_create: %s\n", rt_last_error());
      return 1;
    }
    rt_params p3 = p;
    p3.tile_size = 16;
    memset(multi, 0, (size_t)p.width * p.height * 4);
    if (rt_render_multi(three, 3, &p3, multi, &sm) != RT_OK) {
      fprintf(stderr, "rt_render_multi(3): %s\n", rt_last_error());
      return 1;
    }
    bad |= memcmp(multi, argb, (size_t)p.width * p.height * 4) != 0 || sm.pixels_written != st.pixels_written ||
           sm.rays_primary != st.rays_primary;
    printf("rt_render_multi: 1 GPU and 3 ranks on one GPU %s rt_render (gather %.3f ms, d2h %.3f ms)\n",
           bad ? "DIFFER from" : "match", sm.gather_ms, sm.d2h_ms);
    /* a host that renders frame after frame: two frames in flight (begin, begin, end, end) */
    uint32_t* second = (uint32_t*)calloc((size_t)p.width * p.height, 4);
    int t0 = -1, t1 = -1;
    memset(multi, 0, (size_t)p.width * p.height * 4);
    if (rt_render_multi_begin(three, 3, &p3, multi, &t0) != RT_OK || rt_render_multi_begin(three, 3, &p3, second, &t1) != RT_OK ||
        rt_render_multi_end(t0, NULL) != RT_OK || rt_render_multi_end(t1, NULL) != RT_OK) {
      fprintf(stderr, "rt_render_multi_begin/_end: %s\n", rt_last_error());
      return 1;
    }
    const int bad2 = memcmp(multi, argb, (size_t)p.width * p.height * 4) != 0 || memcmp(second, argb, (size_t)p.width * p.height * 4) != 0;
    printf("rt_render_multi_begin/_end: two frames in flight %s rt_render\n", bad2 ? "DIFFER from" : "match");
    bad |= bad2;
    free(second);
    rt_multi_release();
    rt_scene_destroy(three[1]);
    rt_scene_destroy(three[2]);
    free(multi);
  }
  /* the reference's window loop shows the buffer WHILE the render thread fills it (src/main.rs:327-347): begin / poll / end */
  {
    uint32_t* prog = (uint32_t*)calloc((size_t)p.width * p.height, 4);
    rt_progress* h = NULL;
    if (rt_render_begin(scene, &p, prog, 16 /* rows per band */, &h) != RT_OK) {
      fprintf(stderr, "rt_render_begin: %s\n", rt_last_error());
      return 1;
    }
    uint32_t rows = 0, polls = 0;
    int finished = 0;
    while (!finished) {
      if (rt_render_poll(h, &rows, &finished) != RT_OK) return 1; /* never blocks: rows [0, rows) of prog are final */
      polls++;
    }
    rt_stats sp;
    if (rt_render_end(h, &sp) != RT_OK) {
      fprintf(stderr, "rt_render_end: %s\n", rt_last_error());
      return 1;
    }
    const int bad3 = memcmp(prog, argb, (size_t)p.width * p.height * 4) != 0 || sp.pixels_written != st.pixels_written || rows != p.height;
    printf("rt_render_begin/_poll/_end: %u polls, progressive frame %s rt_render\n", polls, bad3 ? "DIFFERS from" : "matches");
    bad |= bad3;
    free(prog);
    rt_scene_info mi;
    if (rt_scene_memory_info(scene, &mi) != RT_OK) return 1;
    printf("scene holds %llu bytes on the device (geometry %llu, BVH %llu, optional tables %llu of a budget of %llu)\n",
           (unsigned long long)mi.bytes_total, (unsigned long long)mi.bytes_geometry, (unsigned long long)mi.bytes_bvh,
           (unsigned long long)(mi.bytes_flags + mi.bytes_cell_lists), (unsigned long long)mi.budget_bytes);
  }
  rt_scene_destroy(scene);
  free(argb);
  /* pixels whose ray misses everything keep the caller's background, as in the reference */
  return (!bad && st.pixels_written > 0 && st.rays_primary == (uint64_t)p.width * p.height) ? 0 : 2;
}
