/*
 * hijiki_host.h — C ABI of the host-side (CPU, no GPU dependency) mirror of
 * Hijiki's scene model and scene compiler.
 *
 * Mirrors `Scene` / `Shape` / `Material` / `Scene::compile` /
 * `ImageBlockGenerator` of /root/reference/src/main.rs (line numbers below)
 * so that a host written against the reference's Rust types maps 1:1.  The
 * output of hjh_scene_compile is exactly the hj_scene_desc the device
 * library (hijiki_hip.h) uploads.
 */
#ifndef HIJIKI_HOST_H
#define HIJIKI_HOST_H

#include "hijiki_hip.h"

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)   /* libraries are built with -fvisibility=hidden */
#endif

typedef struct hjh_scene hjh_scene;        /* `struct Scene`,         src/main.rs:162-170 */
typedef struct hjh_compiled hjh_compiled;  /* `struct CompiledScene`, src/main.rs:376-397 */

/* --- Scene construction (the "Shape API / material enum") ----------------- */
int  hjh_scene_create(hjh_scene** out);
void hjh_scene_destroy(hjh_scene* s);
const char* hjh_last_error(void);          /* thread-local message of the last failure */

/* Camera, src/main.rs:154-160.  `hjh_scene_set_camera_cbox` installs the
 * hard-coded camera of Scene::from_obj (src/main.rs:417-425). */
int hjh_scene_set_camera(hjh_scene* s, const float position[3], const float rotation_xyzw[4], float fov_deg);
int hjh_scene_set_camera_cbox(hjh_scene* s);

/* Materials, `enum Material`, src/main.rs:38-44.  Each returns the material
 * index (>= 0) that shapes refer to, or a negative hj_status. */
int hjh_scene_add_diffuse(hjh_scene* s, const float color[3]);
int hjh_scene_add_diffuse_cboard(hjh_scene* s, const float color1[3], float scale_u, const float color2[3], float scale_v);
int hjh_scene_add_mirror(hjh_scene* s);
int hjh_scene_add_dielectric(hjh_scene* s, const float extinction[3], float eta_ratio); /* clear(): extinction 0 */
int hjh_scene_add_emissive(hjh_scene* s, const float power[3]);

/* Vertices (src/main.rs:54-60, pushed at :465-474).  Returns the index of
 * the first vertex added, or negative status. */
long hjh_scene_add_vertices(hjh_scene* s, const hj_vertex* v, size_t n);

/* Shapes, `enum Shape`, src/main.rs:47-52; pushed with a material index
 * (src/main.rs:484-488,1473-1482). */
int hjh_scene_add_sphere(hjh_scene* s, const float center[3], float radius, int material);
int hjh_scene_add_quad(hjh_scene* s, const float origin[3], const float edge1[3], const float edge2[3], int material);
int hjh_scene_add_triangle(hjh_scene* s, uint32_t a, uint32_t b, uint32_t c, int material);
int hjh_scene_add_triangles(hjh_scene* s, const uint32_t* abc, size_t ntris, int material);
size_t hjh_scene_num_shapes(const hjh_scene* s);

/* --- Scene::compile, src/main.rs:173-357 ---------------------------------- */
/* Splits shapes per kind, builds the BVH (own binned-SAH builder standing in
 * for the `bvh` 0.3.1 crate, whose output the reference never pins), flattens
 * it depth-first with skip links (src/main.rs:203-243), builds the material
 * words (:246-287) and the uniform emitter table (:289-307).  Scenes with
 * fewer than 2 shapes are rejected (the reference panics at :230). */
int  hjh_scene_compile(const hjh_scene* s, hjh_compiled** out);
/* The same without the tree (desc.bvh == NULL, num_bvh_nodes == 0): per-kind shape lists, material words, emitter table - for a
 * host that lets the device build the tree (hj_build_bvh_device, then hj_scene_upload with bvh == NULL or hjh_compiled_set_bvh). */
int  hjh_scene_compile_shapes(const hjh_scene* s, hjh_compiled** out);
void hjh_compiled_destroy(hjh_compiled* c);
/* Borrowed view; valid while `c` lives. */
int  hjh_compiled_desc(const hjh_compiled* c, hj_scene_desc* out);
/* Replace the tree (e.g. by the one hj_build_bvh_device made from this scene's shapes); n must be 2 * shapes - 1. */
int  hjh_compiled_set_bvh(hjh_compiled* c, const hj_bvh_node* nodes, size_t n);
/* The tree passes of hjh_scene_compile on the INSTALLED tree (e.g. hj_build_bvh_device's): `reinsert_passes` passes of the
 * insertion-based optimisation, then - when vote_paths != 0 - the order of every node's two children voted by that many sampled
 * camera paths (60000 is what compile uses).  The tree must be a pre-order skip-link tree; the image changes at most in epsilon
 * ties, as with any other tree over the same shapes. */
int  hjh_compiled_tune_bvh(hjh_compiled* c, int reinsert_passes, size_t vote_paths);
/* K = hj_direction_classes(mode) link orderings of the installed tree, one per direction class of the rays (hijiki_hip.h:
 * hj_ray_direction_class): same boxes, same leaves, the child order each class of a sample of `vote_paths` camera paths votes
 * for; where a class has no opinion: the installed order (fallback 0) or the class's geometric near-first order (fallback 1);
 * geometric_only skips the vote.  out receives K arrays of num_bvh_nodes records each, every one a valid pre-order skip-link
 * tree (array 0 first).  No counterpart upstream (shader/scene.glsl:97-133 walks one fixed order). */
int  hjh_compiled_directional_bvh(const hjh_compiled* c, int mode, size_t vote_paths, int fallback, int geometric_only,
                                  hj_bvh_node* out, size_t capacity);
/* Size of the reference's packed scene buffer (12 sub-buffers padded to
 * 256 B, src/main.rs:314-339) and the packing itself (src/main.rs:561-605),
 * for tools that want the reference's exact buffer image. */
size_t hjh_compiled_packed_size(const hjh_compiled* c);
int    hjh_compiled_pack(const hjh_compiled* c, void* buffer, size_t size);

/* --- ImageBlockGenerator, src/main.rs:619-682 (deterministic) ------------- */
size_t hjh_num_blocks_per_pass(uint32_t width, uint32_t height, uint32_t block_size);
/* Blocks of passes [pass_begin, pass_end) in generator order; writes at most
 * `cap`, returns how many the range holds.  block_size must be a multiple of
 * 64 (assert at src/main.rs:633). */
size_t hjh_make_blocks(uint32_t width, uint32_t height, uint32_t block_size, uint64_t master_seed,
                       uint32_t pass_begin, uint32_t pass_end, hj_image_block* out, size_t cap);

/* --- Scene::from_obj, src/main.rs:414-530 ----------------------------------- */
/* OBJ/MTL load with the semantics of the `tobj` 0.1.11 crate the reference calls (one model per o/g, (v,vt,vn)
 * re-indexing, fan triangulation, unknown MTL statements kept as strings), material kind by NAME PREFIX
 * (`light*` -> emissive with `Ke`, `glass*` -> dielectric 1.5, `mirror*` -> mirror, else diffuse `Kd`), every
 * vertex needs a normal, models without material contribute vertices only, hard-coded camera. */
int hjh_scene_from_obj(const char* path, hjh_scene** out);
/* `--put-cbox-spheres`, src/main.rs:1463-1483. */
int hjh_scene_put_cbox_spheres(hjh_scene* s);

/* --- Renderer::save_image, src/main.rs:1395-1419 ----------------------------- */
/* rgb = width*height*3 floats (row 0 on top), e.g. from hj_framebuffer_resolve.  EXR: three FLOAT channels
 * R,G,B, scan-line, uncompressed.  PFM: little-endian "PF". */
int hjh_write_exr(const char* path, uint32_t width, uint32_t height, const float* rgb);
int hjh_write_pfm(const char* path, uint32_t width, uint32_t height, const float* rgb);
/* 8-bit sRGB PNG (stored deflate): the image the reference's preview window shows (shader/preview.glsl:9-12). */
int hjh_write_png(const char* path, uint32_t width, uint32_t height, const float* rgb);

/* --- Synthetic scenes (bench inputs) ---------- */
enum hjh_synth_kind {
  HJH_SYNTH_CBOX = 0,         /* Cornell-box-shaped: 12 wall/light triangles + 6320-triangle smooth object */
  HJH_SYNTH_CBOX_SPHERES = 1, /* + mirror sphere + clear dielectric sphere (eta 1.5)                       */
  HJH_SYNTH_CBOX_MESH = 2,    /* walls + light + `mesh_triangles`-triangle bumpy closed mesh                */
  HJH_SYNTH_CBOX_CBOARD = 3   /* reference's --put-cbox-spheres: mirror + checkerboard-diffuse spheres     */
};
/* mesh_triangles: object tessellation (0 = 6320, the reference teapot's count). */
int hjh_scene_make_synthetic(int kind, uint32_t mesh_triangles, uint32_t gen_seed, hjh_scene** out);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
