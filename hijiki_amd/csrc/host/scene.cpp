// Scene compiler: the host step that defines what the kernels read.
// Mirrors `Scene::compile` (reference src/main.rs:173-357) with an own
// binned-SAH builder in place of the un-vendored `bvh` 0.3.1 crate.
#include "scene.hpp"

#include <algorithm>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <limits>
#include <functional>
#include <future>
#include <numeric>
#include <stdexcept>

namespace hijiki {

// ---------------------------------------------------------------- Aabb

Aabb Aabb::empty() {
  Aabb b;
  for (int i = 0; i < 3; i++) {
    b.lo[i] = std::numeric_limits<float>::infinity();
    b.hi[i] = -std::numeric_limits<float>::infinity();
  }
  return b;
}
void Aabb::grow(const float p[3]) {
  for (int i = 0; i < 3; i++) {
    lo[i] = std::min(lo[i], p[i]);
    hi[i] = std::max(hi[i], p[i]);
  }
}
void Aabb::join(const Aabb& o) {
  for (int i = 0; i < 3; i++) {
    lo[i] = std::min(lo[i], o.lo[i]);
    hi[i] = std::max(hi[i], o.hi[i]);
  }
}
float Aabb::half_area() const {
  float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
  if (dx < 0 || dy < 0 || dz < 0) return 0.f;
  return dx * dy + dy * dz + dz * dx;
}

// Shape bounds: sphere centre +- r (src/shape.rs:13-20), quad = 4 corners
// (src/shape.rs:46-54), triangle = 3 vertices (src/main.rs:74-79).
Aabb Scene::shape_aabb(const Shape& s) const {
  Aabb b = Aabb::empty();
  switch (s.kind) {
    case ShapeKind::Sphere: {
      float lo[3], hi[3];
      for (int i = 0; i < 3; i++) {
        lo[i] = s.sphere.center[i] - s.sphere.radius;
        hi[i] = s.sphere.center[i] + s.sphere.radius;
      }
      b.grow(lo);
      b.grow(hi);
      break;
    }
    case ShapeKind::Quad: {
      float p[3];
      b.grow(s.quad.origin);
      for (int i = 0; i < 3; i++) p[i] = s.quad.origin[i] + s.quad.edge1[i];
      b.grow(p);
      for (int i = 0; i < 3; i++) p[i] = s.quad.origin[i] + s.quad.edge2[i];
      b.grow(p);
      for (int i = 0; i < 3; i++) p[i] = (s.quad.origin[i] + s.quad.edge1[i]) + s.quad.edge2[i];
      b.grow(p);
      break;
    }
    case ShapeKind::Triangle:
      for (int k = 0; k < 3; k++) b.grow(vertices[s.tri.v[k]].pos);
      break;
  }
  return b;
}

// ---------------------------------------------------------------- BVH build

namespace {

constexpr int kMaxBins = 16;
constexpr int kBins = 16;   // 32 bins, and ordering the children by surface area or size, measured no better on cbox

struct Builder {
  const std::vector<Aabb>& boxes;
  std::vector<float> cx, cy, cz;  // centroids
  std::vector<uint32_t> order;    // shape permutation being partitioned
  std::vector<BuildNode> nodes;

  explicit Builder(const std::vector<Aabb>& b) : boxes(b) {
    size_t n = b.size();
    cx.resize(n), cy.resize(n), cz.resize(n), order.resize(n);
    for (size_t i = 0; i < n; i++) {
      cx[i] = 0.5f * (b[i].lo[0] + b[i].hi[0]);
      cy[i] = 0.5f * (b[i].lo[1] + b[i].hi[1]);
      cz[i] = 0.5f * (b[i].lo[2] + b[i].hi[2]);
    }
    std::iota(order.begin(), order.end(), 0u);
    nodes.resize(n ? 2 * n - 1 : 0);   // one shape per leaf: a subtree over k shapes has exactly 2k - 1 nodes
  }

  float centroid(uint32_t s, int axis) const { return axis == 0 ? cx[s] : axis == 1 ? cy[s] : cz[s]; }

  Aabb bounds(size_t lo, size_t hi) const {
    Aabb b = Aabb::empty();
    for (size_t i = lo; i < hi; i++) b.join(boxes[order[i]]);
    return b;
  }

  // Builds the subtree over order[lo, hi) at node index `me` (pre-order: the left subtree follows its parent, the
  // right one starts 2 * (left shapes) records after it) and returns `me`.  Because every subtree's node range is
  // known beforehand, large subtrees are built by their own threads with the same result as the serial recursion.  Degenerate inputs (all centroids equal) fall back
  // to median splits, so the depth stays O(log n) there.
  int32_t build(size_t lo, size_t hi, int32_t me = 0, int depth = 0) {
    if (hi - lo == 1) {
      nodes[me].shape = (int32_t)order[lo];
      return me;
    }
    // centroid bounds -> split axis
    float cmin[3] = {INFINITY, INFINITY, INFINITY}, cmax[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = lo; i < hi; i++) {
      uint32_t s = order[i];
      float c[3] = {cx[s], cy[s], cz[s]};
      for (int a = 0; a < 3; a++) {
        cmin[a] = std::min(cmin[a], c[a]);
        cmax[a] = std::max(cmax[a], c[a]);
      }
    }
    // binned SAH, all three axes; fall back to a median split of the widest axis
    size_t mid = lo + (hi - lo) / 2;
    bool have_split = false;
    int axis = 0;
    float ext = cmax[0] - cmin[0];
    for (int a = 1; a < 3; a++)
      if (cmax[a] - cmin[a] > ext) ext = cmax[a] - cmin[a], axis = a;
    if (hi - lo > 2 && depth < 96) {   // (past that depth - adversarial size progressions - median splits bound the recursion)
      float best = INFINITY;
      int best_axis = -1, best_split = -1;
      for (int ax = 0; ax < 3; ax++) {
        const float e = cmax[ax] - cmin[ax];
        if (!(e > 1e-7f)) continue;
        Aabb bin_box[kMaxBins];
        size_t bin_cnt[kMaxBins] = {};
        for (auto& bb : bin_box) bb = Aabb::empty();
        const float scale = (float)kBins / e;
        for (size_t i = lo; i < hi; i++) {
          int bi = (int)((centroid(order[i], ax) - cmin[ax]) * scale);
          bi = std::min(std::max(bi, 0), kBins - 1);
          bin_cnt[bi]++;
          bin_box[bi].join(boxes[order[i]]);
        }
        float right_area[kMaxBins];
        size_t right_cnt[kMaxBins];
        Aabb acc = Aabb::empty();
        size_t cnt = 0;
        for (int bi = kBins - 1; bi > 0; bi--) {
          acc.join(bin_box[bi]);
          cnt += bin_cnt[bi];
          right_area[bi] = acc.half_area();
          right_cnt[bi] = cnt;
        }
        acc = Aabb::empty();
        cnt = 0;
        for (int bi = 0; bi < kBins - 1; bi++) {
          acc.join(bin_box[bi]);
          cnt += bin_cnt[bi];
          if (cnt == 0 || right_cnt[bi + 1] == 0) continue;
          const float cost = acc.half_area() * (float)cnt + right_area[bi + 1] * (float)right_cnt[bi + 1];
          if (cost < best) best = cost, best_axis = ax, best_split = bi;
        }
      }
      if (best_axis >= 0) {
        const float scale = (float)kBins / (cmax[best_axis] - cmin[best_axis]);
        auto bin_of = [&](uint32_t sh) {
          int bi = (int)((centroid(sh, best_axis) - cmin[best_axis]) * scale);
          return std::min(std::max(bi, 0), kBins - 1);
        };
        auto it = std::stable_partition(order.begin() + lo, order.begin() + hi,
                                        [&](uint32_t sh) { return bin_of(sh) <= best_split; });
        mid = (size_t)(it - order.begin());
        have_split = mid > lo && mid < hi;
      }
    }
    if (!have_split) {
      // median split along the widest axis (stable for equal keys)
      mid = lo + (hi - lo) / 2;
      std::stable_sort(order.begin() + lo, order.begin() + hi,
                       [&](uint32_t a, uint32_t b) { return centroid(a, axis) < centroid(b, axis); });
    }
    Aabb lb = bounds(lo, mid), rb = bounds(mid, hi);
    const int32_t l = me + 1, r = me + (int32_t)(2 * (mid - lo));
    if (depth < 4 && hi - lo >= 65536) {          // up to 16 threads at the top of a large tree (1 M triangles: 1.46 -> 0.50 s on 8 cores)
      auto left = std::async(std::launch::async, [&] { build(lo, mid, l, depth + 1); });
      build(mid, hi, r, depth + 1);
      left.get();
    } else {
      build(lo, mid, l, depth + 1);
      build(mid, hi, r, depth + 1);
    }
    nodes[me].left = l;
    nodes[me].right = r;
    nodes[me].left_box = lb;
    nodes[me].right_box = rb;
    return me;
  }
};

}  // namespace

namespace {
// Which child comes first matters to a fixed-order (skip-link) walk: a hit found in the first child shrinks tMax and
// culls the second one's box (t0 < tMax), never the other way round.  The SAH split does not depend on the order, so
// this is a pass over the finished tree that swaps children: the child with FEWER SHAPES first (equal counts: the smaller
// box) - it is the cheaper one to walk and about as likely to hold the nearest hit, and what it finds spares the walk
// the expensive one; a leaf (always tested once its parent is entered) so comes before a subtree.  Closest-hit node
// visits per ray with the reference walk (oracle counters): cbox 21.8 -> 21.1, with the two spheres 23.5 -> 21.6,
// 100 k-triangle mesh 54.9 -> 50.5.  Shadow rays do not care (any-hit; unoccluded ones visit the same boxes in any order).
// HJ_BVH_CHILD_ORDER: 0 as split (lower coordinates first), 1 larger box first, 2 smaller box first, 3 (default) fewer shapes first.
// Tree rotations (Kensler 2008): the greedy top-down SAH leaves local improvements on the table - for a node with children
// A and B, handing B down into A in exchange for one of A's children (or the other way round) changes only A's box; the
// exchange that shrinks it most is applied, bottom-up, for a few passes; so is an exchange of two grandchildren across
// (both children's boxes change).  HJ_BVH_ROTATE = passes (default 8, 0: off).
// Node visits per ray of the reference walk (oracle counters, closest / shadow): cbox 21.1 / 18.8 -> 19.5 / 16.8, with the
// spheres 21.6 / 17.5 -> 20.4 / 17.2, 100 k-triangle mesh 50.5 / 63.0 -> 42.7 / 56.2.  On the GPU the gain is smaller - lanes
// per wave-step fall with the steps per ray (cbox 13.9 -> 13.1 lane-steps, 41.7 -> 38.8 lanes: the same wave-steps) -:
// c3 +5 %, c4 +2 %, 60 k triangles +2 %, c2 the same.
struct Rotator {
  std::vector<BuildNode>& nodes;
  double gain = 0;
  static Aabb join(const Aabb& a, const Aabb& b) { Aabb r = a; r.join(b); return r; }
  // bottom-up: the children's subtrees first, then this node's own exchange.  The two subtrees are disjoint, so at the top of a large
  // tree they go to two threads (parallel_depth levels: up to 16 tasks); the result is the serial pass's, node for node.
  void visit(int32_t nd, int parallel_depth = 0) {
    BuildNode& n = nodes[nd];
    if (n.shape >= 0) return;
    if (parallel_depth > 0) {
      Rotator l{nodes}, r{nodes};
      auto left = std::async(std::launch::async, [&] { l.visit(n.left, parallel_depth - 1); });
      r.visit(n.right, parallel_depth - 1);
      left.get();
      gain += l.gain + r.gain;
    } else {
      visit(n.left); visit(n.right);
    }
    // candidates: (which child of n keeps its place and is opened: 0 left, 1 right) x (which grandchild goes up: 0 left, 1 right)
    float best = 0.f; int bo = -1, bg = -1;
    for (int o = 0; o < 2; o++) {
      const int32_t open = o == 0 ? n.left : n.right;
      const BuildNode& a = nodes[open];
      if (a.shape >= 0) continue;
      const Aabb& abox = o == 0 ? n.left_box : n.right_box;
      const Aabb& other = o == 0 ? n.right_box : n.left_box;
      for (int g = 0; g < 2; g++) {
        const Aabb& stays = g == 0 ? a.right_box : a.left_box;        // the grandchild that stays under `open`
        const float delta = join(stays, other).half_area() - abox.half_area();
        if (delta < best) { best = delta; bo = o; bg = g; }
      }
    }
    // ... and the two exchanges of grandchildren across (left's g-th child with right's h-th child: both boxes change)
    int xg = -1, xh = -1;
    if (nodes[n.left].shape < 0 && nodes[n.right].shape < 0) {
      BuildNode &a = nodes[n.left], &c = nodes[n.right];
      for (int g = 0; g < 2; g++) {
        const int h = 0;                                             // (g, 1) is (1 - g, 0) with the children's names swapped
        const Aabb& a_up = g == 0 ? a.left_box : a.right_box;        // leaves a
        const Aabb& a_stay = g == 0 ? a.right_box : a.left_box;
        const Aabb& c_up = h == 0 ? c.left_box : c.right_box;        // leaves c
        const Aabb& c_stay = h == 0 ? c.right_box : c.left_box;
        const float delta = join(a_stay, c_up).half_area() + join(c_stay, a_up).half_area() - n.left_box.half_area() - n.right_box.half_area();
        if (delta < best) { best = delta; bo = -1; xg = g; xh = h; }
      }
    }
    if (xg >= 0) {
      BuildNode &a = nodes[n.left], &c = nodes[n.right];
      int32_t& ai = xg == 0 ? a.left : a.right;  Aabb& ab = xg == 0 ? a.left_box : a.right_box;
      int32_t& ci = xh == 0 ? c.left : c.right;  Aabb& cb = xh == 0 ? c.left_box : c.right_box;
      std::swap(ai, ci); std::swap(ab, cb);
      n.left_box = join(a.left_box, a.right_box); n.right_box = join(c.left_box, c.right_box);
      gain -= best;
      return;
    }
    if (bo < 0) return;
    const int32_t open = bo == 0 ? n.left : n.right;
    BuildNode& a = nodes[open];
    int32_t& n_other = bo == 0 ? n.right : n.left;
    Aabb& n_other_box = bo == 0 ? n.right_box : n.left_box;
    Aabb& n_open_box = bo == 0 ? n.left_box : n.right_box;
    int32_t& a_up = bg == 0 ? a.left : a.right;
    Aabb& a_up_box = bg == 0 ? a.left_box : a.right_box;
    std::swap(n_other, a_up);
    std::swap(n_other_box, a_up_box);
    n_open_box = join(a.left_box, a.right_box);
    gain -= best;
  }
};

size_t order_children(std::vector<BuildNode>& nodes, int32_t nd, int mode) {
  BuildNode& b = nodes[nd];
  if (b.shape >= 0) return 1;
  const size_t nl = order_children(nodes, b.left, mode), nr = order_children(nodes, b.right, mode);
  const float al = b.left_box.half_area(), ar = b.right_box.half_area();
  const bool swap = mode == 1 ? ar > al : mode == 2 ? ar < al : (nr < nl || (nr == nl && ar < al));
  if (swap) { std::swap(b.left, b.right); std::swap(b.left_box, b.right_box); }
  return nl + nr;
}
}  // namespace

const BuildTuning& BuildTuning::get() {
  static const BuildTuning t = [] {
    auto num = [](const char* name, long dflt) { const char* e = std::getenv(name); return e ? std::atol(e) : dflt; };
    BuildTuning b;
    b.rotate_passes = (int)num("HJ_BVH_ROTATE", 8);
    b.reinsert_passes = (int)num("HJ_BVH_REINSERT", -1);
    b.reinsert_max = num("HJ_BVH_REINSERT_MAX", 0);
    b.reinsert_large = (int)num("HJ_BVH_REINSERT_LARGE", 6);
    b.child_order = (int)num("HJ_BVH_CHILD_ORDER", 4);
    b.vote_paths = num("HJ_BVH_VOTE_PATHS", 0);
    b.vote_shadow = std::getenv("HJ_BVH_VOTE_SHADOW") ? (int)std::min(16l, std::max(0l, num("HJ_BVH_VOTE_SHADOW", 0))) : -1;
    b.verbose = std::getenv("HJ_BVH_VERBOSE") != nullptr;
    b.rotate_verbose = std::getenv("HJ_BVH_ROTATE_VERBOSE") != nullptr;
    return b;
  }();
  return t;
}

std::vector<BuildNode> build_bvh(const std::vector<Aabb>& boxes) {
  if (boxes.empty()) return {};
  const BuildTuning& tn = BuildTuning::get();
  auto t_last = std::chrono::steady_clock::now();
  auto mark = [&](const char* what) {          // HJ_BVH_VERBOSE: wall time of the compiler's stages
    if (!tn.verbose) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "build_bvh: %-22s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  Builder b(boxes);
  b.build(0, boxes.size());
  mark("binned SAH");
  const int rotate_passes = tn.rotate_passes;
  for (int p = 0; p < rotate_passes && b.nodes[0].shape < 0; p++) {
    Rotator r{b.nodes};
    r.visit(0, b.nodes.size() > 400000 ? 4 : 0);
    if (tn.rotate_verbose) std::fprintf(stderr, "rotation pass %d: half-area gain %.4f\n", p, r.gain);
    if (r.gain <= 0) break;
  }
  mark("rotation passes");
  // Insertion-based optimisation (tree_opt.cpp).  HJ_BVH_REINSERT = passes, whatever the size; by default 3 serial passes over all
  // nodes up to 400 000 nodes and HJ_BVH_REINSERT_LARGE (6) batched passes beyond: parallel searches, serial moves, the later passes
  // from a work list - 1 M triangles: oracle node visits per ray -6.4 %, c4 +4 ... 5 %, 1 s on 16 cores (round 5 ran none there: its
  // serial pass over a sixteenth of the nodes bought nothing, over all of them +1 ... 3 % for 25 s).
  const int reinsert_passes = tn.reinsert_passes;
  const bool large = b.nodes.size() > 400000;
  const int passes = reinsert_passes >= 0 ? reinsert_passes : (large ? tn.reinsert_large : 3);
  if (passes > 0) { if (large) optimize_by_reinsertion_batched(b.nodes, passes); else optimize_by_reinsertion(b.nodes, passes); }
  mark("reinsertion");
  const int child_order = tn.child_order;
  if (child_order != 0) order_children(b.nodes, 0, std::min(child_order, 3));
  mark("children by shape count");
  return std::move(b.nodes);
}

// The flattening step of Scene::compile (src/main.rs:203-231): pre-order numbering, every node stores the box its PARENT kept
// for it, exit links.  `global_index` maps a leaf's shape to the index the kernels use.
void flatten_bvh(const std::vector<BuildNode>& tree, const std::function<uint32_t(int32_t)>& global_index, std::vector<hj_bvh_node>& out_bvh) {
  // pass 1: pre-order numbering, left before right (src/main.rs:203-213)
  std::vector<uint32_t> slot(tree.size(), 0);
  {
    std::vector<int32_t> st{0};
    uint32_t next = 0;
    while (!st.empty()) {
      int32_t nd = st.back();
      st.pop_back();
      slot[nd] = next++;
      if (tree[nd].shape < 0) {
        st.push_back(tree[nd].right);
        st.push_back(tree[nd].left);
      }
    }
  }
  // pass 2: every node stores the box its PARENT kept for it; exit of a left
  // child = its right sibling, of a right child = the parent's exit, of the
  // root = 1 000 000 (src/main.rs:214-231).
  out_bvh.resize(tree.size());
  struct Work { int32_t node; Aabb box; uint32_t exit; };
  Aabb root_box = tree[0].left_box;
  root_box.join(tree[0].right_box);  // src/main.rs:230
  std::vector<Work> st;
  // The reference hard-codes 1 000 000 (src/main.rs:231); with more than a million nodes that
  // index lies INSIDE the array and the shader's walk would never terminate, so larger trees
  // get the node count instead (any value >= the node count ends the walk).
  const uint32_t root_exit = tree.size() > HJ_BVH_ROOT_EXIT ? (uint32_t)tree.size() : HJ_BVH_ROOT_EXIT;
  st.push_back({0, root_box, root_exit});
  while (!st.empty()) {
    Work w = st.back();
    st.pop_back();
    const BuildNode& bn = tree[w.node];
    hj_bvh_node& nd = out_bvh[slot[w.node]];
    std::memcpy(nd.aabb_min, w.box.lo, 12);
    std::memcpy(nd.aabb_max, w.box.hi, 12);
    nd.shape_index = bn.shape >= 0 ? global_index(bn.shape) : HJ_BVH_INNER;
    nd.exit_index = w.exit;
    if (bn.shape < 0) {
      st.push_back({bn.right, bn.right_box, w.exit});
      st.push_back({bn.left, bn.left_box, slot[bn.right]});
    }
  }
}

// The inverse: the binary tree of a flattened pre-order skip-link array (leaf shapes = the array's global shape indices).
// Throws when the array is not such a tree.
std::vector<BuildNode> unflatten_bvh(const std::vector<hj_bvh_node>& bvh) {
  const size_t N = bvh.size();
  if (N < 3) throw std::runtime_error("tree of fewer than three nodes");
  std::vector<BuildNode> tree(N);
  std::vector<uint32_t> size(N, 1);
  auto box_of = [&](size_t i) { Aabb b; std::memcpy(b.lo, bvh[i].aabb_min, 12); std::memcpy(b.hi, bvh[i].aabb_max, 12); return b; };
  for (size_t i = N; i-- > 0;) {
    if (bvh[i].shape_index != HJ_BVH_INNER) { tree[i].shape = (int32_t)bvh[i].shape_index; continue; }
    const size_t l = i + 1;
    if (l >= N) throw std::runtime_error("not a pre-order skip-link tree");
    const size_t r = l + size[l];
    if (r >= N || bvh[l].exit_index != r) throw std::runtime_error("not a pre-order skip-link tree");
    size[i] = 1 + size[l] + size[r];
    tree[i].left = (int32_t)l; tree[i].right = (int32_t)r;
    tree[i].left_box = box_of(l); tree[i].right_box = box_of(r);
  }
  if (size[0] != N) throw std::runtime_error("not a pre-order skip-link tree");
  return tree;
}

// The scene model behind a compiled scene, shapes in the kernels' global order (spheres, quads, triangles): what the ray vote
// needs to trace its sample when only the compiled arrays are at hand.
Scene scene_of(const CompiledScene& cs) {
  Scene s;
  s.camera = cs.camera;
  s.vertices = cs.vertices;
  const size_t total = cs.spheres.size() + cs.quads.size() + cs.triangles.size();
  if (cs.materials.size() != total) throw std::runtime_error("materials != shapes");
  s.objects.reserve(total);
  s.materials.reserve(total);
  auto material = [&](uint32_t word) {
    Material m{};
    m.tag = (hj_material_tag)(word >> HJ_MATERIAL_TAG_SHIFT);
    const uint32_t ix = word & HJ_MATERIAL_INDEX_MASK;
    switch (m.tag) {
      case HJ_MAT_DIFFUSE: if (ix < cs.diffuse.size()) m.diffuse = cs.diffuse[ix]; break;
      case HJ_MAT_DIFFUSECBOARD: if (ix < cs.diffusecb.size()) m.cboard = cs.diffusecb[ix]; break;
      case HJ_MAT_DIELECTRIC: if (ix < cs.dielectric.size()) m.dielectric = cs.dielectric[ix]; break;
      case HJ_MAT_EMISSIVE: if (ix < cs.emissive.size()) m.emissive = cs.emissive[ix]; break;
      default: break;
    }
    return m;
  };
  size_t k = 0;
  auto add = [&](Shape sh) { s.materials.push_back(material(cs.materials[k])); s.objects.emplace_back(sh, (int)k); k++; };
  for (const hj_sphere& sp : cs.spheres) { Shape sh{}; sh.kind = ShapeKind::Sphere; sh.sphere = sp; add(sh); }
  for (const hj_quad& q : cs.quads) { Shape sh{}; sh.kind = ShapeKind::Quad; sh.quad = q; add(sh); }
  for (const hj_triangle& t : cs.triangles) {
    for (int c = 0; c < 3; c++) if (t.v[c] >= cs.vertices.size()) throw std::runtime_error("triangle refers to unknown vertex");
    Shape sh{}; sh.kind = ShapeKind::Triangle; sh.tri = t; add(sh);
  }
  return s;
}

// The tree passes of compile() on an INSTALLED tree (hj_build_bvh_device's, or any tree a host brought): optional reinsertion,
// then the ray-voted child order.  The image changes at most in epsilon ties, like with any other tree.
void tune_bvh(CompiledScene& cs, int reinsert_passes, size_t vote_paths) {
  std::vector<BuildNode> tree = unflatten_bvh(cs.bvh);
  if (reinsert_passes > 0) { if (tree.size() > 400000) optimize_by_reinsertion_batched(tree, reinsert_passes); else optimize_by_reinsertion(tree, reinsert_passes); }
  if (vote_paths > 0) order_children_by_rays(tree, scene_of(cs), vote_paths);
  flatten_bvh(tree, [](int32_t shape) { return (uint32_t)shape; }, cs.bvh);
}

// K link orderings of an installed tree, one per direction class of the rays (include/hijiki_hip.h: hj_ray_direction_class): the
// same boxes and leaves, flattened K times with the child order each class voted for.  out = K arrays of cs.bvh.size() records.
void directional_bvh(const CompiledScene& cs, int mode, size_t vote_paths, int fallback, bool geometric_only, std::vector<hj_bvh_node>& out) {
  const std::vector<BuildNode> tree = unflatten_bvh(cs.bvh);
  const size_t n = tree.size(), K = (size_t)hj_direction_classes(mode);
  std::vector<uint8_t> orders;
  directional_child_orders(tree, scene_of(cs), vote_paths, mode, fallback, geometric_only, orders);
  out.resize(n * K);
  std::vector<hj_bvh_node> one;
  for (size_t c = 0; c < K; c++) {
    std::vector<BuildNode> t = tree;
    for (size_t i = 0; i < n; i++)
      if (orders[c * n + i]) { std::swap(t[i].left, t[i].right); std::swap(t[i].left_box, t[i].right_box); }
    flatten_bvh(t, [](int32_t shape) { return (uint32_t)shape; }, one);
    std::memcpy(out.data() + c * n, one.data(), n * sizeof(hj_bvh_node));
  }
}

// ---------------------------------------------------------------- compile

CompiledScene compile(const Scene& scene, bool with_tree) {
  CompiledScene out;
  out.camera = scene.camera;
  out.vertices = scene.vertices;

  const size_t n = scene.objects.size();
  if (n < 2)
    throw std::runtime_error("scene needs at least 2 shapes (reference panics: root would be a leaf, src/main.rs:230)");

  // per-kind lists + index of each object inside its list (src/main.rs:180-196)
  std::vector<uint32_t> index_in_kind(n);
  std::vector<int> sphere_mat, quad_mat, tri_mat;
  std::vector<Aabb> boxes(n);
  for (size_t i = 0; i < n; i++) {
    const Shape& s = scene.objects[i].first;
    int mat = scene.objects[i].second;
    if (mat < 0 || (size_t)mat >= scene.materials.size()) throw std::runtime_error("shape refers to unknown material");
    switch (s.kind) {
      case ShapeKind::Sphere:
        index_in_kind[i] = (uint32_t)out.spheres.size();
        out.spheres.push_back(s.sphere);
        sphere_mat.push_back(mat);
        break;
      case ShapeKind::Quad:
        index_in_kind[i] = (uint32_t)out.quads.size();
        out.quads.push_back(s.quad);
        quad_mat.push_back(mat);
        break;
      case ShapeKind::Triangle:
        for (int k = 0; k < 3; k++)
          if (s.tri.v[k] >= scene.vertices.size()) throw std::runtime_error("triangle refers to unknown vertex");
        index_in_kind[i] = (uint32_t)out.triangles.size();
        out.triangles.push_back(s.tri);
        tri_mat.push_back(mat);
        break;
    }
    boxes[i] = scene.shape_aabb(s);
  }

  // BVH::build (src/main.rs:199) -> depth-first flatten with skip links (src/main.rs:203-231)
  // (with_tree == false: the arrays without a tree, for a host that lets the device build it - hj_build_bvh_device)
  std::vector<BuildNode> tree;
  if (with_tree) tree = build_bvh(boxes);
  if (with_tree) {
    // HJ_BVH_CHILD_ORDER = 4: on top of "fewer shapes first", the order a sample of the renderer's own rays votes for
    // (tree_opt.cpp); HJ_BVH_VOTE_PATHS = camera paths of the sample
    const int child_order = BuildTuning::get().child_order;
    const long vote_paths = BuildTuning::get().vote_paths;
    if (child_order >= 4) {
      const size_t paths = vote_paths > 0 ? (size_t)vote_paths : 60000;   // (more buys nothing: 8 k paths order cbox as 60 k do, 50 k the 1 M-triangle tree as 1 M do)
      order_children_by_rays(tree, scene, paths);
    }
  }
  const uint32_t ns = (uint32_t)out.spheres.size(), nq = (uint32_t)out.quads.size();
  auto global_index = [&](int32_t obj) -> uint32_t {  // src/main.rs:232-243
    switch (scene.objects[obj].first.kind) {
      case ShapeKind::Sphere: return index_in_kind[obj];
      case ShapeKind::Quad: return ns + index_in_kind[obj];
      default: return ns + nq + index_in_kind[obj];
    }
  };
  if (with_tree) flatten_bvh(tree, global_index, out.bvh);

  // material words (src/main.rs:246-287)
  std::vector<uint32_t> reprs;
  for (const Material& m : scene.materials) {
    uint32_t ix = 0;
    switch (m.tag) {
      case HJ_MAT_DIFFUSE: out.diffuse.push_back(m.diffuse); ix = (uint32_t)out.diffuse.size() - 1; break;
      case HJ_MAT_DIFFUSECBOARD: out.diffusecb.push_back(m.cboard); ix = (uint32_t)out.diffusecb.size() - 1; break;
      case HJ_MAT_MIRROR: ix = 0; break;
      case HJ_MAT_DIELECTRIC: out.dielectric.push_back(m.dielectric); ix = (uint32_t)out.dielectric.size() - 1; break;
      case HJ_MAT_EMISSIVE: out.emissive.push_back(m.emissive); ix = (uint32_t)out.emissive.size() - 1; break;
    }
    reprs.push_back(((uint32_t)m.tag << HJ_MATERIAL_TAG_SHIFT) + ix);
  }
  for (int m : sphere_mat) out.materials.push_back(reprs[m]);
  for (int m : quad_mat) out.materials.push_back(reprs[m]);
  for (int m : tri_mat) out.materials.push_back(reprs[m]);

  // uniform emitter table (src/main.rs:289-307)
  for (size_t ix = 0; ix < out.materials.size(); ix++)
    if ((out.materials[ix] >> HJ_MATERIAL_TAG_SHIFT) == (uint32_t)HJ_MAT_EMISSIVE)
      out.emitters.push_back(hj_emitter{(uint32_t)ix, 0.f, 0.f, 0.f});
  float pdf = 1.0f / (float)out.emitters.size();
  float cdf = 0.f;
  for (auto& e : out.emitters) {
    cdf += pdf;
    e.pdf = pdf;
    e.cdf = cdf;
  }
  return out;
}

hj_scene_desc CompiledScene::desc() const {
  hj_scene_desc d{};
  d.camera = camera;
  d.bvh = bvh.data(); d.num_bvh_nodes = bvh.size();
  d.spheres = spheres.data(); d.num_spheres = spheres.size();
  d.quads = quads.data(); d.num_quads = quads.size();
  d.triangles = triangles.data(); d.num_triangles = triangles.size();
  d.vertices = vertices.data(); d.num_vertices = vertices.size();
  d.materials = materials.data(); d.num_materials = materials.size();
  d.emitters = emitters.data(); d.num_emitters = emitters.size();
  d.diffuse = diffuse.data(); d.num_diffuse = diffuse.size();
  d.diffusecb = diffusecb.data(); d.num_diffusecb = diffusecb.size();
  d.dielectric = dielectric.data(); d.num_dielectric = dielectric.size();
  d.emissive = emissive.data(); d.num_emissive = emissive.size();
  return d;
}

namespace {
constexpr size_t kAlign = 256;  // BUFFER_ALIGNMENT, src/main.rs:411
size_t padded(size_t bytes) { return (bytes + kAlign - 1) & ~(kAlign - 1); }
}  // namespace

size_t CompiledScene::packed_size() const {
  return padded(sizeof(hj_scene_info)) + padded(bvh.size() * sizeof(hj_bvh_node)) +
         padded(spheres.size() * sizeof(hj_sphere)) + padded(quads.size() * sizeof(hj_quad)) +
         padded(triangles.size() * sizeof(hj_triangle)) + padded(vertices.size() * sizeof(hj_vertex)) +
         padded(materials.size() * 4) + padded(emitters.size() * sizeof(hj_emitter)) +
         padded(diffuse.size() * sizeof(hj_diffuse)) + padded(diffusecb.size() * sizeof(hj_diffuse_cb)) +
         padded(dielectric.size() * sizeof(hj_dielectric)) + padded(emissive.size() * sizeof(hj_emissive));
}

bool CompiledScene::pack(void* buffer, size_t size) const {
  if (size != packed_size()) return false;
  std::memset(buffer, 0, size);
  uint8_t* p = static_cast<uint8_t*>(buffer);
  auto put = [&](const void* src, size_t bytes) {
    if (bytes) std::memcpy(p, src, bytes);
    p += padded(bytes);
  };
  hj_scene_info info{};
  info.camera = camera;
  info.num_spheres = (uint32_t)spheres.size();
  info.num_quads = (uint32_t)quads.size();
  info.num_triangles = (uint32_t)triangles.size();
  info.num_emitters = (uint32_t)emitters.size();
  put(&info, sizeof info);
  put(bvh.data(), bvh.size() * sizeof(hj_bvh_node));
  put(spheres.data(), spheres.size() * sizeof(hj_sphere));
  put(quads.data(), quads.size() * sizeof(hj_quad));
  put(triangles.data(), triangles.size() * sizeof(hj_triangle));
  put(vertices.data(), vertices.size() * sizeof(hj_vertex));
  put(materials.data(), materials.size() * 4);
  put(emitters.data(), emitters.size() * sizeof(hj_emitter));
  put(diffuse.data(), diffuse.size() * sizeof(hj_diffuse));
  put(diffusecb.data(), diffusecb.size() * sizeof(hj_diffuse_cb));
  put(dielectric.data(), dielectric.size() * sizeof(hj_dielectric));
  put(emissive.data(), emissive.size() * sizeof(hj_emissive));
  return p == static_cast<uint8_t*>(buffer) + size;
}

}  // namespace hijiki
