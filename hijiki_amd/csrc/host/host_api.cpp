// extern "C" surface of libhijiki_host.so (see include/hijiki_host.h).
#include <cstring>
#include <exception>
#include <new>
#include <string>

#include "../../../include/hijiki_host.h"
#include "blockgen.hpp"
#include "scene.hpp"

using namespace hijiki;

struct hjh_scene { Scene scene; };
struct hjh_compiled { CompiledScene cs; };

namespace {
thread_local std::string g_error;
int fail(int code, const std::string& msg) {
  g_error = msg;
  return code;
}
template <class F>
int guarded(F&& f) {
  try {
    return f();
  } catch (const std::bad_alloc&) {
    return fail(HJ_ERR_NOMEM, "out of host memory");
  } catch (const std::exception& e) {
    return fail(HJ_ERR_INVALID, e.what());
  }
}
}  // namespace

extern "C" {

const char* hjh_last_error(void) { return g_error.c_str(); }

int hjh_scene_create(hjh_scene** out) {
  if (!out) return fail(HJ_ERR_INVALID, "null out pointer");
  return guarded([&] {
    *out = new hjh_scene();
    (*out)->scene.camera.rotation[3] = 1.f;
    (*out)->scene.camera.fov = 27.7f;
    return (int)HJ_OK;
  });
}
void hjh_scene_destroy(hjh_scene* s) { delete s; }

int hjh_scene_set_camera(hjh_scene* s, const float position[3], const float rotation[4], float fov_deg) {
  if (!s || !position || !rotation) return fail(HJ_ERR_INVALID, "null argument");
  hj_camera& c = s->scene.camera;
  c = hj_camera{};
  for (int i = 0; i < 3; i++) c.position[i] = position[i];
  for (int i = 0; i < 4; i++) c.rotation[i] = rotation[i];
  c.fov = fov_deg;
  return HJ_OK;
}

int hjh_scene_set_camera_cbox(hjh_scene* s) {
  if (!s) return fail(HJ_ERR_INVALID, "null scene");
  s->scene.camera = make_synthetic(HJH_SYNTH_CBOX, 16, 0).camera;
  return HJ_OK;
}

static int push_material(hjh_scene* s, const Material& m) {
  if (!s) return -fail(HJ_ERR_INVALID, "null scene");
  if (s->scene.materials.size() >= HJ_MATERIAL_INDEX_MASK) return -fail(HJ_ERR_INVALID, "too many materials");
  s->scene.materials.push_back(m);
  return (int)s->scene.materials.size() - 1;
}

int hjh_scene_add_diffuse(hjh_scene* s, const float color[3]) {
  if (!color) return -fail(HJ_ERR_INVALID, "null color");
  Material m{};
  m.tag = HJ_MAT_DIFFUSE;
  m.diffuse = hj_diffuse{{color[0], color[1], color[2]}, 0.f};
  return push_material(s, m);
}
int hjh_scene_add_diffuse_cboard(hjh_scene* s, const float c1[3], float su, const float c2[3], float sv) {
  if (!c1 || !c2) return -fail(HJ_ERR_INVALID, "null color");
  Material m{};
  m.tag = HJ_MAT_DIFFUSECBOARD;
  m.cboard = hj_diffuse_cb{{c1[0], c1[1], c1[2]}, su, {c2[0], c2[1], c2[2]}, sv};
  return push_material(s, m);
}
int hjh_scene_add_mirror(hjh_scene* s) {
  Material m{};
  m.tag = HJ_MAT_MIRROR;
  return push_material(s, m);
}
int hjh_scene_add_dielectric(hjh_scene* s, const float ext[3], float eta) {
  Material m{};
  m.tag = HJ_MAT_DIELECTRIC;
  m.dielectric = hj_dielectric{{ext ? ext[0] : 0.f, ext ? ext[1] : 0.f, ext ? ext[2] : 0.f}, eta};
  return push_material(s, m);
}
int hjh_scene_add_emissive(hjh_scene* s, const float power[3]) {
  if (!power) return -fail(HJ_ERR_INVALID, "null power");
  Material m{};
  m.tag = HJ_MAT_EMISSIVE;
  m.emissive = hj_emissive{{power[0], power[1], power[2]}, 0.f};
  return push_material(s, m);
}

long hjh_scene_add_vertices(hjh_scene* s, const hj_vertex* v, size_t n) {
  if (!s || (n && !v)) return -(long)fail(HJ_ERR_INVALID, "null argument");
  long first = (long)s->scene.vertices.size();
  int rc = guarded([&] {
    s->scene.vertices.insert(s->scene.vertices.end(), v, v + n);
    return (int)HJ_OK;
  });
  return rc == HJ_OK ? first : -(long)rc;
}

static int push_shape(hjh_scene* s, const Shape& sh, int material) {
  if (!s) return fail(HJ_ERR_INVALID, "null scene");
  if (material < 0 || (size_t)material >= s->scene.materials.size()) return fail(HJ_ERR_INVALID, "unknown material index");
  return guarded([&] {
    s->scene.objects.emplace_back(sh, material);
    return (int)HJ_OK;
  });
}

int hjh_scene_add_sphere(hjh_scene* s, const float c[3], float radius, int material) {
  if (!c) return fail(HJ_ERR_INVALID, "null center");
  Shape sh{};
  sh.kind = ShapeKind::Sphere;
  sh.sphere = hj_sphere{{c[0], c[1], c[2]}, radius};
  return push_shape(s, sh, material);
}
int hjh_scene_add_quad(hjh_scene* s, const float o[3], const float e1[3], const float e2[3], int material) {
  if (!o || !e1 || !e2) return fail(HJ_ERR_INVALID, "null argument");
  Shape sh{};
  sh.kind = ShapeKind::Quad;
  sh.quad = hj_quad{{o[0], o[1], o[2]}, 0.f, {e1[0], e1[1], e1[2]}, 0.f, {e2[0], e2[1], e2[2]}, 0.f};
  return push_shape(s, sh, material);
}
int hjh_scene_add_triangle(hjh_scene* s, uint32_t a, uint32_t b, uint32_t c, int material) {
  if (s && (a >= s->scene.vertices.size() || b >= s->scene.vertices.size() || c >= s->scene.vertices.size()))
    return fail(HJ_ERR_INVALID, "triangle refers to unknown vertex");
  Shape sh{};
  sh.kind = ShapeKind::Triangle;
  sh.tri = hj_triangle{{a, b, c}};
  return push_shape(s, sh, material);
}
int hjh_scene_add_triangles(hjh_scene* s, const uint32_t* abc, size_t ntris, int material) {
  if (!s || (ntris && !abc)) return fail(HJ_ERR_INVALID, "null argument");
  for (size_t i = 0; i < ntris; i++) {
    int rc = hjh_scene_add_triangle(s, abc[3 * i], abc[3 * i + 1], abc[3 * i + 2], material);
    if (rc != HJ_OK) return rc;
  }
  return HJ_OK;
}
size_t hjh_scene_num_shapes(const hjh_scene* s) { return s ? s->scene.objects.size() : 0; }

int hjh_scene_compile(const hjh_scene* s, hjh_compiled** out) {
  if (!s || !out) return fail(HJ_ERR_INVALID, "null argument");
  return guarded([&] {
    auto* c = new hjh_compiled();
    try {
      c->cs = compile(s->scene);
    } catch (...) {
      delete c;
      throw;
    }
    *out = c;
    return (int)HJ_OK;
  });
}
int hjh_scene_compile_shapes(const hjh_scene* s, hjh_compiled** out) {
  if (!s || !out) return fail(HJ_ERR_INVALID, "null argument");
  return guarded([&] {
    auto* c = new hjh_compiled();
    try {
      c->cs = compile(s->scene, false);
    } catch (...) {
      delete c;
      throw;
    }
    *out = c;
    return (int)HJ_OK;
  });
}
void hjh_compiled_destroy(hjh_compiled* c) { delete c; }

int hjh_compiled_desc(const hjh_compiled* c, hj_scene_desc* out) {
  if (!c || !out) return fail(HJ_ERR_INVALID, "null argument");
  *out = c->cs.desc();
  return HJ_OK;
}
int hjh_compiled_set_bvh(hjh_compiled* c, const hj_bvh_node* nodes, size_t n) {
  if (!c || !nodes) return fail(HJ_ERR_INVALID, "null argument");
  const size_t shapes = c->cs.spheres.size() + c->cs.quads.size() + c->cs.triangles.size(), want = shapes ? 2 * shapes - 1 : 0;
  if (n != want) return fail(HJ_ERR_INVALID, "a tree over these shapes has " + std::to_string(want) + " nodes");
  return guarded([&] {
    c->cs.bvh.assign(nodes, nodes + n);
    return (int)HJ_OK;
  });
}
int hjh_compiled_tune_bvh(hjh_compiled* c, int reinsert_passes, size_t vote_paths) {
  if (!c) return fail(HJ_ERR_INVALID, "null argument");
  return guarded([&] {
    tune_bvh(c->cs, reinsert_passes, vote_paths);
    return (int)HJ_OK;
  });
}
int hjh_compiled_directional_bvh(const hjh_compiled* c, int mode, size_t vote_paths, int fallback, int geometric_only,
                                 hj_bvh_node* out, size_t capacity) {
  if (!c || !out) return fail(HJ_ERR_INVALID, "null argument");
  if (mode < 0 || mode > 8) return fail(HJ_ERR_INVALID, "direction mode must be 0 ... 8");
  if (capacity < c->cs.bvh.size() * (size_t)hj_direction_classes(mode)) return fail(HJ_ERR_INVALID, "out holds fewer than classes x nodes records");
  return guarded([&] {
    std::vector<hj_bvh_node> arrays;
    directional_bvh(c->cs, mode, vote_paths, fallback, geometric_only != 0, arrays);
    std::memcpy(out, arrays.data(), arrays.size() * sizeof(hj_bvh_node));
    return (int)HJ_OK;
  });
}
size_t hjh_compiled_packed_size(const hjh_compiled* c) { return c ? c->cs.packed_size() : 0; }
int hjh_compiled_pack(const hjh_compiled* c, void* buffer, size_t size) {
  if (!c || !buffer) return fail(HJ_ERR_INVALID, "null argument");
  return c->cs.pack(buffer, size) ? (int)HJ_OK : fail(HJ_ERR_INVALID, "buffer size != packed size (assert at src/main.rs:604)");
}

size_t hjh_num_blocks_per_pass(uint32_t w, uint32_t h, uint32_t block) {
  if (!w || !h || !block) return 0;
  return BlockGrid(w, h, block).per_pass();
}

size_t hjh_make_blocks(uint32_t w, uint32_t h, uint32_t block, uint64_t master, uint32_t pass_begin, uint32_t pass_end,
                       hj_image_block* out, size_t cap) {
  if (!w || !h || !block || (block & 63u) != 0) {  // assert!(block_size & 63 == 0), src/main.rs:633
    fail(HJ_ERR_INVALID, "block size must be a non-zero multiple of 64");
    return 0;
  }
  BlockGrid g(w, h, block);
  size_t n = 0;
  for (uint32_t p = pass_begin; p < pass_end; p++)
    for (uint32_t j = 0; j < g.per_pass(); j++, n++)
      if (out && n < cap) out[n] = g.make(master, p, j);
  return n;
}

int hjh_scene_make_synthetic(int kind, uint32_t mesh_triangles, uint32_t gen_seed, hjh_scene** out) {
  if (!out) return fail(HJ_ERR_INVALID, "null out pointer");
  return guarded([&] {
    auto* s = new hjh_scene();
    try {
      s->scene = make_synthetic(kind, mesh_triangles, gen_seed);
    } catch (...) {
      delete s;
      throw;
    }
    *out = s;
    return (int)HJ_OK;
  });
}

int hjh_scene_from_obj(const char* path, hjh_scene** out) {
  if (!path || !out) return fail(HJ_ERR_INVALID, "null argument");
  return guarded([&] {
    auto* s = new hjh_scene();
    try {
      s->scene = scene_from_obj(path);
    } catch (...) {
      delete s;
      throw;
    }
    *out = s;
    return (int)HJ_OK;
  });
}

int hjh_scene_put_cbox_spheres(hjh_scene* s) {
  if (!s) return fail(HJ_ERR_INVALID, "null scene");
  return guarded([&] {
    put_cbox_spheres(s->scene);
    return (int)HJ_OK;
  });
}

int hjh_write_exr(const char* path, uint32_t w, uint32_t h, const float* rgb) {
  if (!path || !rgb || !w || !h) return fail(HJ_ERR_INVALID, "bad argument");
  return guarded([&] {
    write_exr(path, w, h, rgb);
    return (int)HJ_OK;
  });
}

int hjh_write_png(const char* path, uint32_t w, uint32_t h, const float* rgb) {
  if (!path || !rgb || !w || !h) return fail(HJ_ERR_INVALID, "bad argument");
  return guarded([&] {
    write_png(path, w, h, rgb);
    return (int)HJ_OK;
  });
}

int hjh_write_pfm(const char* path, uint32_t w, uint32_t h, const float* rgb) {
  if (!path || !rgb || !w || !h) return fail(HJ_ERR_INVALID, "bad argument");
  return guarded([&] {
    write_pfm(path, w, h, rgb);
    return (int)HJ_OK;
  });
}

// hj_block_seed / hj_pass_offset are declared in hijiki_hip.h; the host
// library exports them too so that a CPU-only host can build block lists.
uint32_t hj_block_seed(uint64_t master, uint32_t pass, uint32_t j) { return block_seed(master, pass, j); }
uint32_t hj_block_owner(uint32_t width, uint32_t height, uint32_t pass, uint32_t j, uint32_t world) {
  if (!width || !height || !world) return 0;
  return BlockGrid(width, height, HJ_BLOCK_SIZE).owner(pass, j, world);
}
void hj_pass_offset(uint64_t master, uint32_t k, float out[2]) { pass_offset(master, k, out); }

}  // extern "C"
