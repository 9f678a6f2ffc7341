/* Minimal C host for the drop-in boundary: build a scene through the host API, hand the compiled records to the
 * device API, render, write an EXR.  Pure C99 — what a Rust/Go/... FFI binding sees is exactly this surface.
 *
 *   gcc -std=c99 -Iinclude examples/render_cbox.c -Lhijiki_amd/lib -lhijiki_hip -lhijiki_host \
 *       -Wl,-rpath,$PWD/hijiki_amd/lib -o render_cbox && ./render_cbox out.exr
 */
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>

#include "hijiki_hip.h"
#include "hijiki_host.h"

/* the layouts the reference's #[repr(C)] structs and GLSL std430 blocks have */
#define HJ_STATIC_ASSERT(cond, name) typedef char hj_static_assert_##name[(cond) ? 1 : -1]
HJ_STATIC_ASSERT(sizeof(hj_camera) == 48, camera);
HJ_STATIC_ASSERT(sizeof(hj_scene_info) == 64 && offsetof(hj_scene_info, num_spheres) == 48, scene_info);
HJ_STATIC_ASSERT(sizeof(hj_bvh_node) == 32 && offsetof(hj_bvh_node, shape_index) == 12 && offsetof(hj_bvh_node, exit_index) == 28, bvh_node);
HJ_STATIC_ASSERT(sizeof(hj_sphere) == 16 && sizeof(hj_quad) == 48 && offsetof(hj_quad, edge2) == 32, shapes);
HJ_STATIC_ASSERT(sizeof(hj_triangle) == 12 && sizeof(hj_vertex) == 32 && offsetof(hj_vertex, normal) == 16, mesh);
HJ_STATIC_ASSERT(sizeof(hj_emitter) == 16 && sizeof(hj_diffuse) == 16 && sizeof(hj_diffuse_cb) == 32, records);
HJ_STATIC_ASSERT(sizeof(hj_dielectric) == 16 && offsetof(hj_dielectric, eta) == 12 && sizeof(hj_emissive) == 16, materials);
HJ_STATIC_ASSERT(sizeof(hj_image_block) == 40 && offsetof(hj_image_block, sample_offset) == 32, image_block);

static void die(const char* what, const char* why) {
  fprintf(stderr, "%s: %s\n", what, why);
  exit(1);
}

int main(int argc, char** argv) {
  const char* out = argc > 1 ? argv[1] : "/tmp/output.exr";
  const uint32_t W = 512, H = 512, spp = 32;
  hjh_scene* scene = NULL;
  hjh_compiled* compiled = NULL;
  hj_scene_desc desc;
  hj_context* ctx = NULL;
  hj_render_stats stats;
  float* rgb;

  if (hjh_scene_make_synthetic(HJH_SYNTH_CBOX, 0, 1, &scene) != HJ_OK) die("scene", hjh_last_error());
  if (hjh_scene_compile(scene, &compiled) != HJ_OK) die("compile", hjh_last_error());
  if (hjh_compiled_desc(compiled, &desc) != HJ_OK) die("desc", hjh_last_error());

  if (hj_context_create(0, &ctx) != HJ_OK) die("context", hj_last_error(NULL));
  if (hj_scene_upload(ctx, &desc) != HJ_OK) die("upload", hj_last_error(ctx));
  if (hj_framebuffer_create(ctx, W, H, NULL) != HJ_OK) die("framebuffer", hj_last_error(ctx));
  if (hj_render_frame(ctx, spp, 1u, 0, spp, 0, 1, NULL, &stats) != HJ_OK) die("render", hj_last_error(ctx));
  printf("%llu paths, %llu closest + %llu shadow rays in %.2f ms\n", (unsigned long long)stats.paths,
         (unsigned long long)stats.closest_rays, (unsigned long long)stats.shadow_rays, stats.total_ms);

  rgb = (float*)malloc((size_t)W * H * 3 * sizeof(float));
  if (!rgb) die("malloc", "out of memory");
  if (hj_framebuffer_resolve(ctx, rgb) != HJ_OK) die("resolve", hj_last_error(ctx));
  if (hjh_write_exr(out, W, H, rgb) != HJ_OK) die("write", hjh_last_error());
  free(rgb);
  hj_context_destroy(ctx);
  hjh_compiled_destroy(compiled);
  hjh_scene_destroy(scene);
  return 0;
}
