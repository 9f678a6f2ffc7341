// Host-side scene model: mirror of Hijiki's `Scene`, `Shape`, `Material`
// (reference src/main.rs:34-170) and of `CompiledScene` (src/main.rs:376-397).
#pragma once

#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "../../../include/hijiki_hip.h"

namespace hijiki {

// `enum Material` (src/main.rs:38-44); the payloads are the device records.
struct Material {
  hj_material_tag tag;
  hj_diffuse diffuse{};
  hj_diffuse_cb cboard{};
  hj_dielectric dielectric{};
  hj_emissive emissive{};
};

// `enum Shape` (src/main.rs:47-52).
enum class ShapeKind : uint8_t { Sphere, Quad, Triangle };
struct Shape {
  ShapeKind kind;
  hj_sphere sphere{};
  hj_quad quad{};
  hj_triangle tri{};
};

struct Aabb {
  float lo[3], hi[3];
  static Aabb empty();
  void grow(const float p[3]);
  void join(const Aabb& o);
  float half_area() const;
};

// `struct Scene` (src/main.rs:162-170).
struct Scene {
  hj_camera camera{};
  std::vector<std::pair<Shape, int>> objects;  // (shape, material index)
  std::vector<hj_vertex> vertices;
  std::vector<Material> materials;

  Aabb shape_aabb(const Shape& s) const;  // src/main.rs:69-82, src/shape.rs:13-20,46-54
};

// `struct CompiledScene` (src/main.rs:376-397).
struct CompiledScene {
  hj_camera camera{};
  std::vector<hj_bvh_node> bvh;
  std::vector<hj_sphere> spheres;
  std::vector<hj_quad> quads;
  std::vector<hj_triangle> triangles;
  std::vector<hj_vertex> vertices;
  std::vector<uint32_t> materials;
  std::vector<hj_emitter> emitters;
  std::vector<hj_diffuse> diffuse;
  std::vector<hj_diffuse_cb> diffusecb;
  std::vector<hj_dielectric> dielectric;
  std::vector<hj_emissive> emissive;

  hj_scene_desc desc() const;
  size_t packed_size() const;                        // src/main.rs:314-339
  bool pack(void* buffer, size_t size) const;        // src/main.rs:561-605
};

// `Scene::compile` (src/main.rs:173-357).  Throws std::runtime_error.
CompiledScene compile(const Scene& scene, bool with_tree = true);

// The scene compiler's environment switches (tree quality experiments; the defaults are the measured optima),
// read in ONE place - BuildTuning::get(), once per process: a compiled tree must not depend on when a variable was set.
struct BuildTuning {
  int rotate_passes;        // HJ_BVH_ROTATE            rotation passes over the SAH tree (8)
  int reinsert_passes;      // HJ_BVH_REINSERT          insertion-based optimisation passes (-1: 3 up to 400 000 nodes, none beyond)
  long reinsert_max;        // HJ_BVH_REINSERT_MAX      candidates per pass (0: all of a small tree, 1/16 of a large one)
  int reinsert_large;       // HJ_BVH_REINSERT_LARGE    batched passes over a tree beyond 400 000 nodes (6: c4 +4 ... 5 %; the first pass searches all nodes, 1.6 s on 8 cores at 1 M triangles, the others the neighbourhood of what moved: 1 s together; 0 for a fast start)
  int child_order;          // HJ_BVH_CHILD_ORDER       0 as built, 3 fewer shapes first, 4 + voted by sampled rays (4)
  long vote_paths;          // HJ_BVH_VOTE_PATHS        camera paths of the vote's sample (0: 60 000)
  int vote_shadow;          // HJ_BVH_VOTE_SHADOW       a shadow ray's vote in quarters of a closest-hit ray's (-1: 1, 4 from 300 000 nodes on)
  bool verbose, rotate_verbose;   // HJ_BVH_VERBOSE, HJ_BVH_ROTATE_VERBOSE
  static const BuildTuning& get();
};

// Binary BVH with one shape per leaf, as the `bvh` crate hands it to the
// flattening step (src/main.rs:199-231).
struct BuildNode {
  int32_t left = -1, right = -1;  // children (inner) ...
  int32_t shape = -1;             // ... or the shape (leaf)
  Aabb left_box, right_box;       // child_l_aabb / child_r_aabb
};
std::vector<BuildNode> build_bvh(const std::vector<Aabb>& boxes);  // node 0 is the root
// Optimisation passes over the finished tree (tree_opt.cpp): insertion-based optimisation of the surface-area cost, and the
// order of every node's two children voted by a sample of the rays the renderer will trace through `scene`.
double optimize_by_reinsertion(std::vector<BuildNode>& nodes, int passes);
double optimize_by_reinsertion_batched(std::vector<BuildNode>& nodes, int passes);   // large trees: parallel searches, serial moves
size_t order_children_by_rays(std::vector<BuildNode>& nodes, const Scene& scene, size_t num_paths);
// Flattening (src/main.rs:203-231) and its inverse; the tree passes on an installed tree (scene.cpp).
void flatten_bvh(const std::vector<BuildNode>& tree, const std::function<uint32_t(int32_t)>& global_index, std::vector<hj_bvh_node>& out_bvh);
std::vector<BuildNode> unflatten_bvh(const std::vector<hj_bvh_node>& bvh);
Scene scene_of(const CompiledScene& cs);
void tune_bvh(CompiledScene& cs, int reinsert_passes, size_t vote_paths);
// One child order per direction class of the rays (tree_opt.cpp, scene.cpp): K = hj_direction_classes(mode) link orderings of the installed tree.
void directional_child_orders(const std::vector<BuildNode>& nodes, const Scene& scene, size_t num_paths, int mode, int fallback,
                              bool geometric_only, std::vector<uint8_t>& orders);
void directional_bvh(const CompiledScene& cs, int mode, size_t vote_paths, int fallback, bool geometric_only, std::vector<hj_bvh_node>& out);

// Synthetic bench scenes.
Scene make_synthetic(int kind, uint32_t mesh_triangles, uint32_t gen_seed);

// `Scene::from_obj` (src/main.rs:414-530) and `--put-cbox-spheres` (src/main.rs:1463-1483).
Scene scene_from_obj(const std::string& path);
void put_cbox_spheres(Scene& scene);

// Image output (src/main.rs:1395-1419): rgb = W*H*3 floats, row 0 on top.
void write_pfm(const std::string& path, uint32_t w, uint32_t h, const float* rgb);
void write_exr(const std::string& path, uint32_t w, uint32_t h, const float* rgb);
// 8-bit sRGB PNG of the same image: what the reference's preview window shows (shader/preview.glsl:9-12).
void write_png(const std::string& path, uint32_t w, uint32_t h, const float* rgb);

}  // namespace hijiki
