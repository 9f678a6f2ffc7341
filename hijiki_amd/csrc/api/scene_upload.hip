// hj_scene_upload: validation of the reference's scene arrays (src/main.rs:561-605) and their re-layout for the kernels
// (kernels/hj_device.h: collapsed tree, pair nodes, hot-first node order, pre-gathered triangle and emitter records).
#include "hj_internal.h"
#include "light_grid.hpp"
#include "scene_relayout.hpp"

#pragma clang fp contract(off)

using namespace hjapi;

namespace hjapi {

// Same invariants the reference asserts while packing (src/main.rs:562-565)
// plus every index range a kernel dereferences, and the monotonic-exit
// property that makes the skip-link walk terminate on any input.
int validate_scene(hj_context* ctx, const hj_scene_desc* s) {
  const size_t shapes = s->num_spheres + s->num_quads + s->num_triangles;
  if (s->num_materials != shapes)
    return set_error(ctx, HJ_ERR_INVALID, "materials (%zu) != spheres+quads+triangles (%zu) (assert src/main.rs:562-565)",
                     s->num_materials, shapes);
  if (shapes >= 0x7FFFFFFFu || s->num_bvh_nodes >= 0x7FFFFFFFu) return set_error(ctx, HJ_ERR_INVALID, "scene too large");
  auto need = [&](const void* p, size_t n) { return n == 0 || p != nullptr; };
  if (!need(s->bvh, s->num_bvh_nodes) || !need(s->spheres, s->num_spheres) || !need(s->quads, s->num_quads) ||
      !need(s->triangles, s->num_triangles) || !need(s->vertices, s->num_vertices) ||
      !need(s->materials, s->num_materials) || !need(s->emitters, s->num_emitters) ||
      !need(s->diffuse, s->num_diffuse) || !need(s->diffusecb, s->num_diffusecb) ||
      !need(s->dielectric, s->num_dielectric) || !need(s->emissive, s->num_emissive))
    return set_error(ctx, HJ_ERR_INVALID, "null array with non-zero count");
  for (size_t i = 0; i < s->num_bvh_nodes; i++) {
    const hj_bvh_node& n = s->bvh[i];
    if (n.exit_index <= i) return set_error(ctx, HJ_ERR_INVALID, "bvh node %zu: exit index %u does not move forward", i, n.exit_index);
    if (n.shape_index != HJ_BVH_INNER && n.shape_index >= shapes)
      return set_error(ctx, HJ_ERR_INVALID, "bvh node %zu: shape index %u out of range", i, n.shape_index);
  }
  // The array must be what Scene::compile flattens (src/main.rs:203-231): a binary tree in pre-order - an inner node's left
  // child is the next record, the left child's exit is the right child, a node's exit is the record behind its subtree, or
  // beyond the array on the right spine.  The kernels lean on it (a subtree is left only through its root's exit: camera
  // packets, the collapse, the sibling-group order); boxes may be anything, links may not.
  if (s->num_bvh_nodes) {
    const size_t N = s->num_bvh_nodes;
    std::vector<uint32_t> size;
    try { size.assign(N, 1); } catch (const std::bad_alloc&) { return set_error(ctx, HJ_ERR_NOMEM, "out of host memory"); }
    for (size_t i = N; i-- > 0;) {
      const hj_bvh_node& n = s->bvh[i];
      if (n.shape_index == HJ_BVH_INNER) {
        const size_t l = i + 1;
        if (l >= N) return set_error(ctx, HJ_ERR_INVALID, "bvh node %zu: an inner node without children", i);
        const size_t r = l + size[l];
        if (r >= N || s->bvh[l].exit_index != r)
          return set_error(ctx, HJ_ERR_INVALID, "bvh node %zu: not a pre-order skip-link tree (the left child's exit %u is not its sibling %zu)", i, s->bvh[l].exit_index, r);
        size[i] = 1 + size[l] + size[r];
        if (s->bvh[r].exit_index != n.exit_index)
          return set_error(ctx, HJ_ERR_INVALID, "bvh node %zu: not a pre-order skip-link tree (the right child leaves through %u, the node through %u)", i, s->bvh[r].exit_index, n.exit_index);
      }
      const size_t end = i + size[i];
      if (end < N ? n.exit_index != end : n.exit_index < N)
        return set_error(ctx, HJ_ERR_INVALID, "bvh node %zu: not a pre-order skip-link tree (exit %u, its subtree ends at %zu)", i, n.exit_index, end);
    }
    if (size[0] != N) return set_error(ctx, HJ_ERR_INVALID, "bvh: %zu records, the tree under record 0 has %u", N, size[0]);
  }
  for (size_t i = 0; i < s->num_triangles; i++)
    for (int k = 0; k < 3; k++)
      if (s->triangles[i].v[k] >= s->num_vertices)
        return set_error(ctx, HJ_ERR_INVALID, "triangle %zu refers to vertex %u of %zu", i, s->triangles[i].v[k], s->num_vertices);
  for (size_t i = 0; i < s->num_materials; i++) {
    const uint32_t tag = s->materials[i] >> HJ_MATERIAL_TAG_SHIFT, idx = s->materials[i] & HJ_MATERIAL_INDEX_MASK;
    size_t lim = 0;
    switch (tag) {
      case HJ_MAT_DIFFUSE: lim = s->num_diffuse; break;
      case HJ_MAT_DIFFUSECBOARD: lim = s->num_diffusecb; break;
      case HJ_MAT_MIRROR: lim = 1; break;
      case HJ_MAT_DIELECTRIC: lim = s->num_dielectric; break;
      case HJ_MAT_EMISSIVE: lim = s->num_emissive; break;
      default: return set_error(ctx, HJ_ERR_INVALID, "shape %zu: unknown material tag %u", i, tag);
    }
    if (idx >= lim) return set_error(ctx, HJ_ERR_INVALID, "shape %zu: material index %u out of range for tag %u", i, idx, tag);
  }
  for (size_t i = 0; i < s->num_emitters; i++) {
    const uint32_t sh = s->emitters[i].shape;
    if (sh >= shapes) return set_error(ctx, HJ_ERR_INVALID, "emitter %zu: shape %u out of range", i, sh);
    if ((s->materials[sh] >> HJ_MATERIAL_TAG_SHIFT) != HJ_MAT_EMISSIVE)
      return set_error(ctx, HJ_ERR_INVALID, "emitter %zu: shape %u is not emissive", i, sh);
  }
  return HJ_OK;
}

}  // namespace hjapi

namespace {

// What depends on the tree's size once its records exist (host and device re-layout alike): N = nodes of the uploaded array,
// M = device records without padding.
void finish_tree_settings(hj::DeviceScene& d, const hjapi::Tuning& tn, size_t N, size_t M, bool has_pairs) {
  using hjapi::Tuning;
  // Large trees (their nodes and triangles do not fit the caches): the path-state streams bypass the caches so that they
  // do not evict scene data (1 M triangles +4.4 %; cache-resident scenes lose 0.5 ... 3 % with it).  HJ_STREAM_STATE = 0 / 1 forces.
  {
    const int nt_env = tn.stream_state;
    d.stream_state = (nt_env == 1 || (nt_env < 0 && N >= (size_t)tn.stream_min_nodes)) ? 1u : 0u;
  }
  {   // camera packets of 8 x 8 pixels instead of 64 x 1 on the same large trees.  HJ_GROUP_TILE = 0 / 1 forces.
    const int gt_env = tn.group_tile;
    d.group_tile = (gt_env == 1 || (gt_env < 0 && N >= (size_t)tn.stream_min_nodes)) ? 1u : 0u;
  }
  // steps per round of the walk loop (the first one is the merged step that also runs the leaf tests) and free lanes at
  // which a wave fetches new rays: without pair nodes 4 / 32 (6: -0.4 %, 8: -4 % on cbox); with them 7 / 24 on small trees
  // (the rays of the rotated, child-ordered trees are shorter: 5 / 32, the optimum before those passes, is 3 % slower on
  // cbox now; 6 or 8 steps, 20 or 28 lanes: -1 ... -2 %), 8 / 24 up to 600 000 records and 8 / 32 beyond (1 M triangles:
  // 6 .. 10 steps the same, 24 lanes -1 %).  Round 5: with the light-shaft grid most of the short shadow rays are gone and the
  // small trees want 6 steps (c2 +2.3 %, c3 +0.8 % against 7; 5: the same; 8: -1 %; refill at 16 / 32 lanes: -0 ... -2 %)
  const bool small_tree = M < 50000;
  d.inner_burst = (uint32_t)Tuning::pick(tn.inner_burst, !has_pairs ? 4 : (small_tree ? 6 : 8));   // >= 1, or the walk would never advance
  d.refill_min = (uint32_t)Tuning::pick(tn.refill_min, has_pairs && M < 600000 ? 24 : (int)hj::kRefillMin);   // (20 k / 60 k / 200 k triangles: 24 lanes +2 / +3 / +1 %)
}

}  // namespace

extern "C" {

int hj_scene_upload(hj_context* ctx, const hj_scene_desc* s) {
  if (!ctx) return HJ_ERR_INVALID;
  HJ_NOT_BUSY(ctx);
  HJ_NOT_PIPELINED(ctx);
  if (!s) return set_error(ctx, HJ_ERR_INVALID, "null scene");
  // HJ_UPLOAD_TIMING=1: wall time of the stages below on stderr
  const Tuning tn = ctx->tuning = Tuning::from_env();
  const bool timing = tn.upload_timing != 0;
  auto t_last = std::chrono::steady_clock::now();
  auto mark = [&](const char* what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "hj_scene_upload: %-34s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  // scene->bvh == NULL: the tree hj_build_bvh_device left on this context's device, built over these very shape arrays
  // (hj_context::resident) - no trip through the host for the tree, and the triangle / vertex arrays are on the device already
  const bool resident = s->bvh == nullptr && s->num_bvh_nodes == 0;
  hj_scene_desc with_tree;
  if (resident) {
    const hj_context::ResidentTree& rt = ctx->resident;
    if (!rt.valid) return set_error(ctx, HJ_ERR_STATE, "scene->bvh is NULL and no tree is on the device: hj_build_bvh_device builds one");
    if (rt.ns != s->num_spheres || rt.nq != s->num_quads || rt.nt != s->num_triangles || rt.nv != s->num_vertices ||
        ((s->num_spheres == 0 || s->spheres) && (s->num_quads == 0 || s->quads) && (s->num_triangles == 0 || s->triangles) &&
         (s->num_vertices == 0 || s->vertices) && rt.shapes_hash != shape_arrays_hash(s)))
      return set_error(ctx, HJ_ERR_INVALID, "scene->bvh is NULL, but the tree on the device was built over other shape arrays (%zu / %zu / %zu shapes, %zu vertices)", rt.ns, rt.nq, rt.nt, rt.nv);
  }
  int rc = validate_scene(ctx, s);
  if (rc != HJ_OK) return rc;
  if (resident) { with_tree = *s; with_tree.num_bvh_nodes = ctx->resident.total; s = &with_tree; }   // (bvh stays NULL: nothing below reads it on this route)
  mark("validation");
  HJ_HIP(ctx, hipSetDevice(ctx->device));
  rc = sync_all(ctx);
  if (rc != HJ_OK) return rc;
  release_scene(ctx);

  hj::DeviceScene d{};
  d.camera = s->camera;
  d.tan_half_fov = (float)std::tan((double)(0.5f * s->camera.fov) * (3.14159265358979323846 / 180.0));
  d.ns = (uint32_t)s->num_spheres;
  d.nq = (uint32_t)s->num_quads;
  d.nt = (uint32_t)s->num_triangles;
  d.num_emitters = (uint32_t)s->num_emitters;
  d.num_nodes = (uint32_t)s->num_bvh_nodes;
  d.has_extinction = 0;
  for (size_t i = 0; i < s->num_dielectric; i++)
    if (s->dielectric[i].extinction[0] != 0.f || s->dielectric[i].extinction[1] != 0.f || s->dielectric[i].extinction[2] != 0.f)
      d.has_extinction = 1;

#define HJ_UP(expr) do { rc = (expr); if (rc != HJ_OK) { release_scene(ctx); return rc; } } while (0)
  // Large trees: the whole re-layout below runs on the device (api/scene_relayout.hip: the arrays go up as they are, a dozen
  // kernels derive the kernels' records) - 0.19 s of host work at 1 M triangles otherwise.  HJ_UPLOAD_DEVICE = 0 / 1 forces;
  // default: from HJ_UPLOAD_DEVICE_MIN (100 000) nodes on.  An array that is not a tree takes the host path.
  bool on_device = false;
  {
    const int env = tn.upload_device;
    const bool want = resident || env == 1 || (env < 0 && s->num_bvh_nodes >= (size_t)tn.upload_device_min);
    if (want && s->num_bvh_nodes >= 3) {
      const size_t mark_bufs = ctx->scene_bufs.size();
      if (resident) {                                         // the build's own copies become the scene's
        ctx->scene_bufs.push_back(ctx->resident.triangles); d.triangles = static_cast<const hj_triangle*>(ctx->resident.triangles.p);
        ctx->scene_bufs.push_back(ctx->resident.vertices); d.vertices = static_cast<const hj_vertex*>(ctx->resident.vertices.p);
        ctx->resident.triangles = DevBuf{}; ctx->resident.vertices = DevBuf{};
      } else {
        HJ_UP(upload(ctx, s->triangles, s->num_triangles, &d.triangles));
        HJ_UP(upload(ctx, s->vertices, s->num_vertices, &d.vertices));
      }
      mark("triangle + vertex upload");
      const int pair_env = tn.pair_leaves;
      const bool pairs_on = pair_env == 1 || (pair_env < 0 && s->num_bvh_nodes >= (size_t)tn.pair_min_nodes);
      RelayoutOut ro;
      rc = relayout_on_device(ctx, s, d.triangles, d.vertices, pairs_on, tn.node_order,
                              (float)tn.collapse_pct / 100.0f, timing, ro,
                              resident ? static_cast<const hj_bvh_node*>(ctx->resident.nodes.p) : nullptr);
      if (resident && rc != HJ_OK) {                          // (no host array to fall back to)
        release_scene(ctx);
        ctx->resident.release();
        return rc == HJ_ERR_UNSUPPORTED ? set_error(ctx, rc, "the tree on the device cannot be re-laid out there (fewer than 3 or too many records)") : rc;
      }
      if (rc == HJ_OK) {
        on_device = true;
        d.nodes = ro.nodes; d.tri_isect = ro.tri_isect; d.tri_shade = ro.tri_shade; d.tri_pair = ro.tri_pair;
        d.num_nodes = ro.num_nodes; d.root = ro.root; d.root2 = ro.root2; d.num_hot = ro.num_hot; d.has_pairs = ro.num_pairs ? 1u : 0u;
        finish_tree_settings(d, tn, s->num_bvh_nodes, ro.kept, ro.num_pairs != 0);
      } else if (rc == HJ_ERR_UNSUPPORTED) {
        while (ctx->scene_bufs.size() > mark_bufs) { ctx->scene_bufs.back().release(); ctx->scene_bufs.pop_back(); }
        d.triangles = nullptr; d.vertices = nullptr;
      } else {
        release_scene(ctx);
        return rc;
      }
    }
  }
  if (!on_device) {
  // pre-gathered triangle records (see kernels/hj_device.h)
  std::vector<float4> isect, shade;
  try {
    isect.resize(3 * s->num_triangles);
    shade.resize(4 * s->num_triangles);
  } catch (const std::bad_alloc&) {
    return set_error(ctx, HJ_ERR_NOMEM, "out of host memory");
  }
  // (a plain loop per thread over its share of the triangles: 26 ms on one core at 1 M triangles)
  auto gather = [&](size_t i0, size_t i1) {
  for (size_t i = i0; i < i1; i++) {
    const hj_vertex& A = s->vertices[s->triangles[i].v[0]];
    const hj_vertex& B = s->vertices[s->triangles[i].v[1]];
    const hj_vertex& C = s->vertices[s->triangles[i].v[2]];
    isect[3 * i + 0] = make_float4(A.pos[0], A.pos[1], A.pos[2], 0.f);
    isect[3 * i + 1] = make_float4(B.pos[0] - A.pos[0], B.pos[1] - A.pos[1], B.pos[2] - A.pos[2], 0.f);
    isect[3 * i + 2] = make_float4(C.pos[0] - A.pos[0], C.pos[1] - A.pos[1], C.pos[2] - A.pos[2], 0.f);
    shade[4 * i + 0] = make_float4(A.normal[0], A.normal[1], A.normal[2], A.u);
    shade[4 * i + 1] = make_float4(B.normal[0], B.normal[1], B.normal[2], B.u);
    shade[4 * i + 2] = make_float4(C.normal[0], C.normal[1], C.normal[2], C.u);
    shade[4 * i + 3] = make_float4(A.v, B.v, C.v, 0.f);
  }
  };
  {
    const size_t nt = s->num_triangles;
    const unsigned hw = nt >= 100000 ? std::max(1u, std::min(8u, std::thread::hardware_concurrency())) : 1u;
    std::vector<std::thread> pool;
    try {
      for (unsigned w = 1; w < hw; w++) pool.emplace_back(gather, nt * w / hw, nt * (w + 1) / hw);
    } catch (const std::exception&) {}
    gather(0, nt / hw);
    for (unsigned w = (unsigned)pool.size() + 1; w < hw; w++) gather(nt * w / hw, nt * (w + 1) / hw);   // (threads that could not be started)
    for (auto& th : pool) th.join();
  }
  mark("triangle records");
  static_assert(sizeof(hj_bvh_node) == 2 * sizeof(float4), "node = 2 x float4");
  static_assert(sizeof(hj_quad) == 3 * sizeof(float4) && sizeof(hj_sphere) == sizeof(float4), "shape records");
  static_assert(sizeof(hj_diffuse_cb) == 2 * sizeof(float4), "checkerboard record");
  // device node array (kernels/hj_device.h): redundant inner nodes dropped, hottest (largest surface area) nodes
  // first, explicit left/exit links.
  {
    // Guard nodes for single leaves (HJ_LEAF_GUARDS).  The reference never tests a leaf's box: every ray that
    // enters the box of a leaf's parent stops at the leaf and runs the shape test - a leaf stop ends the lane's burst -
    // and the parent's box, the union with a sibling, is much larger than the shape.  Here the leaf gets a one-child inner
    // node in front of it whose box is the shape's OWN bounds padded by a thousandth of its size (plus an absolute 2e-4):
    // an ordinary box step (no new code in the walk) that fails for most of those rays and sends them to the leaf's exit.
    // Exact for TRIANGLES and QUADS: their tests (triangle.glsl:15-52, quad.glsl:7-25) find the true intersection of the ray's
    // line with the shape's plane whatever the length of the direction, so a ray that fails the padded box - it misses it, or
    // enters it behind tMax, or leaves it before tMin - cannot pass the shape test, with the padding orders of magnitude above
    // float rounding in either test.
    // NOT exact for SPHERES: sphere.glsl:18-41 solves t^2 + 2 t (d . l) + |l|^2 - r^2 = 0, the intersection only for |d| = 1, and
    // the reference does not keep |d| = 1: a hit point is o + t d, the next normal (p - c) / r, the next direction built on it -
    // among small spheres the error of each step is amplified by |l| / r at the next, and paths a few bounces deep carry directions
    // of length 1.3 or 3 (found by the 9000-sphere chain test of round 5: 3 of 56 000 rays of a 20-sphere frame).  For those rays
    // the reference's "sphere" is not where the guard's box is, and hits the reference accepts were culled.  Sphere leaves keep
    // their parent's box as their only test, as upstream (modes 1 and 3 remain for measurements; c3, whose two large spheres never
    // saw such a ray in 10^9 paths, loses the 1 ... 2.7 % the sphere guards gave it).
    std::vector<hj_bvh_node> guarded;
    std::vector<uint32_t> gidx;                               // index of original node i in `guarded` (empty: no guards, identity)
    const hj_bvh_node* bvh = s->bvh;
    size_t N = s->num_bvh_nodes;
    // HJ_LEAF_GUARDS: 0 none; 2 (default) every triangle or quad leaf that does not become half of a pair node: +1 % on c2 and c3
    // (profiles/r05_ab_sphere_guards.txt); 1 sphere leaves only, 3 all leaves: NOT exact, see above.
    const int guard_mode = tn.leaf_guards;
    const bool guard_spheres = guard_mode == 1 || guard_mode == 3, guard_flat = guard_mode >= 2;
    if ((guard_spheres && s->num_spheres != 0) || (guard_flat && s->num_quads + s->num_triangles != 0)) {
      const size_t n0 = N, first_tri = s->num_spheres + s->num_quads;
      std::vector<uint8_t> want(n0, 0);
      for (size_t i = 0; i < n0; i++) {
        const uint32_t sh = s->bvh[i].shape_index;
        if (sh == HJ_BVH_INNER) continue;
        want[i] = sh < s->num_spheres ? guard_spheres : guard_flat;
      }
      if (guard_flat)                                         // the two triangle leaves of a future pair node keep their parent's box as their guard
        for (size_t i = 0; i + 2 < n0; i++) {
          if (s->bvh[i].shape_index != HJ_BVH_INNER) continue;
          const size_t l = i + 1, r = s->bvh[l].exit_index;
          if (r != l + 1 || s->bvh[l].shape_index == HJ_BVH_INNER || s->bvh[r].shape_index == HJ_BVH_INNER) continue;
          if (s->bvh[l].shape_index >= first_tri && s->bvh[r].shape_index >= first_tri) want[l] = want[r] = 0;
        }
      // The absolute part of the padding covers the rounding of the shape test itself: its (u, v) move by some 1e-6 of the
      // distance between ray origin and shape (divided by the cosine of incidence - and so does the length of the ray inside the
      // padded slab), so 2e-4 is good for 50 units; larger scenes (root box + camera) get 4e-6 of their extent.
      float pad_abs = 2e-4f;
      {
        float ext = 0.f;
        for (int k = 0; k < 3; k++) {
          const float lo = std::min(s->bvh[0].aabb_min[k], s->camera.position[k]), hi = std::max(s->bvh[0].aabb_max[k], s->camera.position[k]);
          if (hi - lo == hi - lo) ext = std::max(ext, hi - lo);
        }
        if (std::isfinite(ext)) pad_abs = std::max(pad_abs, 4e-6f * ext);
      }
      std::vector<uint32_t> before(n0 + 1, 0);               // guards in front of node i
      for (size_t i = 0; i < n0; i++) before[i + 1] = before[i] + want[i];
      if (before[n0] != 0 && n0 + before[n0] < 0x3FFFFFFFu) {
        guarded.reserve(n0 + before[n0]);
        auto moved = [&](uint32_t e) { return e < n0 ? e + before[e] : e + before[n0]; };      // (an exit beyond the array stays beyond it)
        for (size_t i = 0; i < n0; i++) {
          hj_bvh_node nd = s->bvh[i];
          nd.exit_index = moved(nd.exit_index);
          if (want[i]) {
            // the shape's own bounds, outward, padded
            float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, size = 0.f;
            auto grow = [&](float x, float y, float z) { const float p[3] = {x, y, z}; for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); } };
            const uint32_t sh = nd.shape_index;
            if (sh < s->num_spheres) {
              const hj_sphere& sp = s->spheres[sh];
              const float r = std::fabs(sp.radius);
              grow(sp.center[0] - r, sp.center[1] - r, sp.center[2] - r); grow(sp.center[0] + r, sp.center[1] + r, sp.center[2] + r);
              size = r;
            } else if (sh < first_tri) {
              const hj_quad& q = s->quads[sh - s->num_spheres];
              for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++)
                grow(q.origin[0] + a * q.edge1[0] + b * q.edge2[0], q.origin[1] + a * q.edge1[1] + b * q.edge2[1], q.origin[2] + a * q.edge1[2] + b * q.edge2[2]);
            } else {
              const hj_triangle& t = s->triangles[sh - first_tri];
              for (int c = 0; c < 3; c++) grow(s->vertices[t.v[c]].pos[0], s->vertices[t.v[c]].pos[1], s->vertices[t.v[c]].pos[2]);
            }
            for (int k = 0; k < 3; k++) size = std::max(size, hi[k] - lo[k]);
            const float pad = size * 1e-3f + pad_abs;
            hj_bvh_node g = nd;
            g.shape_index = HJ_BVH_INNER;
            bool ok = std::isfinite(pad);
            for (int k = 0; k < 3; k++) {
              g.aabb_min[k] = std::nextafter(lo[k] - pad, -INFINITY);
              g.aabb_max[k] = std::nextafter(hi[k] + pad, INFINITY);
              ok = ok && g.aabb_min[k] <= g.aabb_max[k];
            }
            if (!ok) for (int k = 0; k < 3; k++) { g.aabb_min[k] = -INFINITY; g.aabb_max[k] = INFINITY; }   // (NaN geometry: a box every ray enters)
            guarded.push_back(g);
          }
          guarded.push_back(nd);
        }
        bvh = guarded.data();
        N = guarded.size();
        gidx.resize(n0);
        for (size_t i = 0; i < n0; i++) gidx[i] = (uint32_t)(i + before[i] + want[i]);
      }
    }
    std::vector<float> sa(N);
    for (size_t i = 0; i < N; i++) {
      const float dx = bvh[i].aabb_max[0] - bvh[i].aabb_min[0], dy = bvh[i].aabb_max[1] - bvh[i].aabb_min[1],
                  dz = bvh[i].aabb_max[2] - bvh[i].aabb_min[2];
      sa[i] = (dx >= 0 && dy >= 0 && dz >= 0) ? dx * dy + dy * dz + dz * dx : 0.f;
      if (!(sa[i] == sa[i])) sa[i] = 0.f;                       // (inf * 0: keep the sort's comparison a strict weak order)
    }
    // Collapse: an inner node P whose two children are inner nodes can be removed from the walk without changing
    // which leaves are tested, in which order, with which tMax: a child box lies inside P's box and every term of
    // the slab test is monotone in the bounds, so "child passes => P passes" and "P fails => both children fail";
    // the children keep their own box tests and exits.  (Not for leaf children: a leaf's box is never tested, so
    // P's test is the only guard in front of its shape test.)  It pays when P usually passes: with pass
    // probability p ~ area(P) / area(nearest kept ancestor), testing P costs 1 + 2p box tests against 2 without.
    std::vector<char> del(N, 0);
    {
      const float thr = (float)tn.collapse_pct / 100.0f;
      auto inner = [&](size_t i) { return bvh[i].shape_index == HJ_BVH_INNER; };
      auto inside = [&](size_t c, size_t p) {   // false for NaN bounds
        bool ok = true;
        for (int k = 0; k < 3; k++)
          ok = ok && bvh[c].aabb_min[k] >= bvh[p].aabb_min[k] && bvh[c].aabb_max[k] <= bvh[p].aabb_max[k];
        return ok;
      };
      std::vector<float> anc(N, 0.f);   // area of the nearest kept ancestor
      for (size_t i = 0; i < N; i++) {  // pre-order: ancestors come first
        if (!inner(i) || i + 1 >= N) continue;
        const size_t l = i + 1, r = bvh[l].exit_index;
        if (bvh[l].exit_index == bvh[i].exit_index) { anc[l] = sa[i]; continue; }   // a sphere leaf's guard node: one child
        if (r >= N || r <= l) continue;                       // not a well-formed pre-order pair: leave it alone
        // (an uploaded tree whose child boxes stick out of P's box keeps P: the argument above needs containment)
        if (i != 0 && inner(l) && inner(r) && anc[i] > 0.f && sa[i] > thr * anc[i] && inside(l, i) && inside(r, i)) del[i] = 1;
        anc[l] = anc[r] = del[i] ? anc[i] : sa[i];
      }
    }
    mark("areas + collapse");
    // Pair nodes (kernels/hj_intersect.h leaf_test): an inner node whose two children are triangle leaves keeps its
    // record, the two leaves lose theirs (nothing but the pair's own walk ever reaches them: the left one is the
    // node's first child, the right one the left one's exit) and their triangles go side by side into `pairs`.
    std::vector<uint32_t> pair_of(N, 0xFFFFFFFFu);
    std::vector<float4> pairs;
    pairs.reserve(N / 4 * 6 + 6);              // (a pair per three records at most; typically 0.43 per two leaves)
    // A fifth fewer dependent fetch rounds per ray.  Before the walk's merged first step (hj_walk.h) the longer leaf phase
    // - two tests while the rest of the wave waits - cost more than the rounds saved on cache-resident scenes (-3 % on the
    // 6 k-triangle box against +12 % at 1 M triangles); with the shape fetch riding along with the other lanes' node fetch
    // they pay everywhere: 6 k triangles +3 %, with the spheres +5 %, 60 k +7 %, 200 k +8 %.  HJ_PAIR_LEAVES = 0 / 1 forces;
    // default: trees of >= HJ_PAIR_MIN_NODES records (0: all).
    const int pair_env = tn.pair_leaves;
    if (pair_env == 1 || (pair_env < 0 && N >= (size_t)tn.pair_min_nodes)) {
      const size_t first_tri = s->num_spheres + s->num_quads;
      for (size_t i = 0; i + 2 < N; i++) {
        if (bvh[i].shape_index != HJ_BVH_INNER) continue;
        const size_t l = i + 1, r = bvh[l].exit_index;
        if (r >= N || r != l + 1) continue;
        const uint32_t sl = bvh[l].shape_index, sr = bvh[r].shape_index;
        if (sl == HJ_BVH_INNER || sr == HJ_BVH_INNER || sl < first_tri || sr < first_tri) continue;
        if (bvh[r].exit_index != bvh[i].exit_index) continue;      // (a well-formed tree: the right child's exit is its parent's)
        pair_of[i] = (uint32_t)(pairs.size() / 6);
        for (uint32_t sh : {sl, sr}) {
          const size_t t = sh - first_tri;
          float4 a = isect[3 * t];
          a.w = __builtin_bit_cast(float, sh);
          pairs.push_back(a); pairs.push_back(isect[3 * t + 1]); pairs.push_back(isect[3 * t + 2]);
        }
        del[l] = del[r] = 1;                                              // no records for the two leaves
      }
    }
    mark("pair nodes");
    auto resolve = [&](size_t i) { while (i < N && del[i]) i++; return i; };   // first kept node of a subtree
    std::vector<uint32_t> order, map(N, 0);
    order.reserve(N);
    for (size_t i = 0; i < N; i++) if (!del[i]) order.push_back((uint32_t)i);
    const size_t M = order.size();
    const uint32_t hot = (uint32_t)std::min<size_t>(hj::kHotNodes, M);
    // (only the first `hot` places matter: a partial sort with the index as tie-break gives exactly the prefix a stable sort
    // of the whole list by area would - 110 ms less at 1 M triangles)
    std::partial_sort(order.begin(), order.begin() + hot, order.end(),
                      [&](uint32_t a, uint32_t b) { return sa[a] > sa[b] || (sa[a] == sa[b] && a < b); });
    std::vector<char> is_hot(N, 0);
    for (size_t k = 0; k < hot; k++) { map[order[k]] = (uint32_t)k; is_hot[order[k]] = 1; }
    // (a treelet-blocked order - a node and its largest descendants per 128-byte line - was measured on the 1 M-triangle
    // scene before: within 1 % at 4, 8 and 16 records per treelet)
    mark("hot-first sort");
    uint32_t next = hot;
    const int node_order = tn.node_order;                      // -1: by tree size (with the pair nodes)
    if (node_order == 0 || (node_order < 0 && pairs.empty())) {
      for (size_t i = 0; i < N; i++) if (!del[i] && !is_hot[i]) map[i] = next++;   // small trees (cache-resident): pre-order
    } else {
      // Large trees: the children of a node side by side, a group of two or more starting on a 64-byte sector - the walk
      // always goes from a child to its sibling (the child's exit), so the sibling's record comes with the child's; the
      // group of the larger child (the likelier visit) follows directly.  Unused slots (padding) are never referenced.
      // 1 M triangles: +2.4 %, 200 k: +1 %; cbox (forced): -0.6 %.
      std::vector<uint32_t> stack, kids;
      if (N && !is_hot[0]) map[0] = next++;
      if (N) stack.push_back(0);
      while (!stack.empty()) {
        const uint32_t i = stack.back();
        stack.pop_back();
        const hj_bvh_node& nd = bvh[i];
        if (nd.shape_index != HJ_BVH_INNER || pair_of[i] != 0xFFFFFFFFu) continue;
        const size_t end = nd.exit_index < N ? resolve(nd.exit_index) : N;
        kids.clear();
        for (size_t c = resolve((size_t)i + 1); c < N && c != end;) {
          kids.push_back((uint32_t)c);
          const uint32_t e = bvh[c].exit_index;
          c = e < N ? resolve(e) : N;
        }
        uint32_t cold = 0;
        for (uint32_t c : kids) cold += is_hot[c] ? 0u : 1u;
        if (cold >= 2 && (next & 1u)) next++;
        for (uint32_t c : kids) if (!is_hot[c]) map[c] = next++;
        std::stable_sort(kids.begin(), kids.end(), [&](uint32_t x, uint32_t y) { return sa[x] > sa[y]; });
        for (size_t k = kids.size(); k-- > 0;) stack.push_back(kids[k]);
      }
    }
    mark("node order");
    const size_t M_all = next;                                                       // records incl. padding
    std::vector<float4> dev(2 * M_all, make_float4(0.f, 0.f, 0.f, 0.f));
    for (size_t i = 0; i < N; i++) {
      if (del[i]) continue;
      const hj_bvh_node& nd = bvh[i];
      uint32_t a;
      if (nd.shape_index != HJ_BVH_INNER) a = nd.shape_index;
      else if (pair_of[i] != 0xFFFFFFFFu) a = hj::kInnerFlag | hj::kPairFlag | pair_of[i];
      else {
        const size_t l = resolve(i + 1);                                             // left child = next pre-order record
        a = hj::kInnerFlag | (l < N ? map[l] : hj::kEndOfWalk);
      }
      const size_t e = nd.exit_index < N ? resolve(nd.exit_index) : N;
      const uint32_t b = e < N ? map[e] : hj::kEndOfWalk;                             // >= the record count ends the walk
      float4* rec = &dev[2 * (size_t)map[i]];
      rec[0] = make_float4(nd.aabb_min[0], nd.aabb_min[1], nd.aabb_min[2], __builtin_bit_cast(float, a));
      rec[1] = make_float4(nd.aabb_max[0], nd.aabb_max[1], nd.aabb_max[2], __builtin_bit_cast(float, b));
    }
    // The second copy of the tree: the reference's own array, record i at M_all + i - every node, no guards, nothing collapsed,
    // the boxes as uploaded; pair nodes keep their mark (the same box test decides about the same two triangles; their two leaves'
    // records are never reached).  Rays that are not in general position start here (kernels/hj_intersect.h general_position):
    // for them the slab test is not monotone in a box's bounds, which the collapse and the guards rest on.
    const size_t N0 = s->num_bvh_nodes;
    if (M_all + N0 >= hj::kEndOfWalk) { release_scene(ctx); return set_error(ctx, HJ_ERR_UNSUPPORTED, "BVH of %zu records: too large", M_all + N0); }
    dev.resize(2 * (M_all + N0), make_float4(0.f, 0.f, 0.f, 0.f));
    for (size_t i = 0; i < N0; i++) {
      const hj_bvh_node& nd = s->bvh[i];
      const uint32_t pr = nd.shape_index == HJ_BVH_INNER ? pair_of[gidx.empty() ? i : gidx[i]] : 0xFFFFFFFFu;
      uint32_t a;
      if (nd.shape_index != HJ_BVH_INNER) a = nd.shape_index;
      else if (pr != 0xFFFFFFFFu) a = hj::kInnerFlag | hj::kPairFlag | pr;
      else a = hj::kInnerFlag | (i + 1 < N0 ? (uint32_t)(M_all + i + 1) : hj::kEndOfWalk);
      const uint32_t b = nd.exit_index < N0 ? (uint32_t)(M_all + nd.exit_index) : hj::kEndOfWalk;
      float4* rec = &dev[2 * (M_all + i)];
      rec[0] = make_float4(nd.aabb_min[0], nd.aabb_min[1], nd.aabb_min[2], __builtin_bit_cast(float, a));
      rec[1] = make_float4(nd.aabb_max[0], nd.aabb_max[1], nd.aabb_max[2], __builtin_bit_cast(float, b));
    }
    mark("device records");
    HJ_UP(upload(ctx, pairs.data(), pairs.size(), &d.tri_pair));
    d.has_pairs = pairs.empty() ? 0u : 1u;
    d.num_nodes = (uint32_t)(M_all + N0);
    d.root = N ? map[0] : 0u;
    d.root2 = (uint32_t)M_all;
    d.num_hot = hot;
    finish_tree_settings(d, tn, s->num_bvh_nodes, M, !pairs.empty());
    // The walk adds 32 * index to the low word of the array's address without a carry (kernels/hj_walk.h): the
    // array must not cross a 4 GiB boundary.  Allocate twice the size and start at the boundary if it would.
    {
      const size_t bytes = std::max<size_t>(dev.size() * sizeof(float4), 16) + 128;   // (slack: a whole 128-byte line may be read around the last record)
      if (bytes >= (1ull << 32)) { release_scene(ctx); return set_error(ctx, HJ_ERR_UNSUPPORTED, "BVH of %zu records: the device node array is limited to 4 GiB", M); }
      ctx->scene_bufs.emplace_back();
      DevBuf& b = ctx->scene_bufs.back();
      HJ_UP(dev_alloc(ctx, b, bytes));
      uintptr_t start = reinterpret_cast<uintptr_t>(b.p);
      if ((start >> 32) != ((start + bytes - 1) >> 32)) {
        b.release();
        HJ_UP(dev_alloc(ctx, b, 2 * bytes));
        start = reinterpret_cast<uintptr_t>(b.p);
        if ((start >> 32) != ((start + bytes - 1) >> 32)) start = ((start >> 32) + 1) << 32;
      }
      if (hipMemcpy(reinterpret_cast<void*>(start), dev.data(), dev.size() * sizeof(float4), hipMemcpyHostToDevice) != hipSuccess) {
        release_scene(ctx);
        return set_error(ctx, HJ_ERR_DEVICE, "node upload failed");
      }
      d.nodes = reinterpret_cast<const float4*>(start);
    }
  }
  mark("pair + node upload");
  HJ_UP(upload(ctx, isect.data(), isect.size(), &d.tri_isect));
  HJ_UP(upload(ctx, shade.data(), shade.size(), &d.tri_shade));
  HJ_UP(upload(ctx, s->triangles, s->num_triangles, &d.triangles));
  HJ_UP(upload(ctx, s->vertices, s->num_vertices, &d.vertices));
  }   // (!on_device)
  HJ_UP(upload(ctx, reinterpret_cast<const float4*>(s->spheres), s->num_spheres, &d.spheres));
  HJ_UP(upload(ctx, reinterpret_cast<const float4*>(s->quads), 3 * s->num_quads, &d.quads));
  HJ_UP(upload(ctx, s->materials, s->num_materials, &d.materials));
  HJ_UP(upload(ctx, s->emitters, s->num_emitters, &d.emitters));
  {
    std::vector<float4> rec((size_t)hj::kEmitRecF4 * s->num_emitters, make_float4(0.f, 0.f, 0.f, 0.f));
    for (size_t i = 0; i < s->num_emitters; i++) {
      float4* r = &rec[(size_t)hj::kEmitRecF4 * i];
      const uint32_t shape = s->emitters[i].shape;
      const hj_emissive& em = s->emissive[s->materials[shape] & HJ_MATERIAL_INDEX_MASK];
      uint32_t kind;
      if (shape < s->num_spheres) {
        kind = 0;
        const hj_sphere& sp = s->spheres[shape];
        r[1] = make_float4(sp.center[0], sp.center[1], sp.center[2], 0.f);
        r[0].z = sp.radius;
      } else if (shape < s->num_spheres + s->num_quads) {
        kind = 1;
        const hj_quad& q = s->quads[shape - s->num_spheres];
        r[1] = make_float4(q.origin[0], q.origin[1], q.origin[2], 0.f);
        r[2] = make_float4(q.edge1[0], q.edge1[1], q.edge1[2], 0.f);
        r[3] = make_float4(q.edge2[0], q.edge2[1], q.edge2[2], 0.f);
      } else {
        kind = 2;
        const hj_triangle& t = s->triangles[shape - s->num_spheres - s->num_quads];
        for (int k = 0; k < 3; k++) {
          const hj_vertex& v = s->vertices[t.v[k]];
          r[1 + k] = make_float4(v.pos[0], v.pos[1], v.pos[2], 0.f);
          r[4 + k] = make_float4(v.normal[0], v.normal[1], v.normal[2], 0.f);
        }
      }
      r[0].x = s->emitters[i].pdf;
      r[0].y = __builtin_bit_cast(float, kind);
      r[1].w = em.power[0]; r[2].w = em.power[1]; r[3].w = em.power[2];
    }
    HJ_UP(upload(ctx, rec.data(), rec.size(), &d.emit_rec));
  }
  HJ_UP(upload(ctx, reinterpret_cast<const float4*>(s->diffuse), s->num_diffuse, &d.diffuse));
  HJ_UP(upload(ctx, reinterpret_cast<const float4*>(s->diffusecb), 2 * s->num_diffusecb, &d.diffusecb));
  HJ_UP(upload(ctx, reinterpret_cast<const float4*>(s->dielectric), s->num_dielectric, &d.dielectric));
  HJ_UP(upload(ctx, reinterpret_cast<const float4*>(s->emissive), s->num_emissive, &d.emissive));
  mark("other uploads");
  // Light-shaft visibility grid (api/light_grid.cpp): which next-event shadow rays are unoccluded whatever happens.  HJ_LIGHT_GRID =
  // cells per axis (0: none).  Default: 64 up to HJ_STREAM_MIN_NODES (300 000) tree nodes, none beyond - on the 1 M-triangle scene
  // a third of the shadow rays is proven free and the frame rate does not move (they are the cheap rays: a dozen steps on
  // LDS-resident nodes against the mesh rays' seventeen cold ones), while the build costs 50 ms of start-up there.
  {
    const int big = (int)(s->num_bvh_nodes >= (size_t)tn.stream_min_nodes);
    const int res = Tuning::pick(tn.light_grid, big ? 0 : 64);
    LightGrid lg;
    bool have_grid = false;
    std::vector<hj_bvh_node> host_tree;                       // (the grid is host code: a tree that lives on the device comes back for it)
    hj_scene_desc for_grid = *s;
    try {
      if (res >= 2 && resident) {
        host_tree.resize(ctx->resident.total);
        if (hipMemcpy(host_tree.data(), ctx->resident.nodes.p, sizeof(hj_bvh_node) * host_tree.size(), hipMemcpyDeviceToHost) != hipSuccess) host_tree.clear();
        for_grid.bvh = host_tree.data();
      }
      have_grid = res >= 2 && (!resident || !host_tree.empty()) && build_light_grid(&for_grid, (uint32_t)res, lg);
    } catch (const std::exception&) { have_grid = false; }   // (out of host memory: no grid)
    if (have_grid) {
      // cells (low byte: planar proofs; high byte: bundle proofs, for hits that were not grazing), then - 16-byte aligned - the
      // records the shade stage checks a hit point against its shape with (kernels/hj_shade.h shadow_ray_proven_free)
      const size_t ncell = lg.bits.size(), cells_u16 = (ncell + 7) & ~(size_t)7;
      std::vector<uint16_t> cells(cells_u16 + lg.shape_recs.size() * 2, 0);
      for (size_t i = 0; i < ncell; i++) cells[i] = (uint16_t)(lg.bits[i] | (lg.mesh_bits.empty() ? 0u : (uint32_t)lg.mesh_bits[i] << 8));
      if (!lg.shape_recs.empty()) std::memcpy(cells.data() + cells_u16, lg.shape_recs.data(), lg.shape_recs.size() * sizeof(float));
      HJ_UP(upload(ctx, cells.data(), cells.size(), &d.light_grid));
      d.lg_res = lg.res | (lg.shape_recs.empty() ? 0u : hj::kLightGridHasRecords);
      for (int k = 0; k < 3; k++) { d.lg_lo[k] = lg.lo[k]; d.lg_inv[k] = lg.inv[k]; }
      if (timing) std::fprintf(stderr, "hj_scene_upload: light grid %u^3: %zu cells hold a surface, %zu a planar one, %zu (cell, emitter) pairs proven free\n",
                               lg.res, lg.cells_surface, lg.cells_planar, lg.pairs_clear);
    }
  }
  mark("light-shaft grid");
#undef HJ_UP
  if (resident) ctx->resident.release();                     // consumed (hipFree waits for the kernels that read the tree)
  ctx->scene = d;
  ctx->have_scene = true;
  return HJ_OK;
}

}  // extern "C"
