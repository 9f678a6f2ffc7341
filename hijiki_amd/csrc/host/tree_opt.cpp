// Tree optimisation passes of the scene compiler, run on the finished binary tree (one shape per leaf) before it is
// flattened into the reference's skip-link array (src/main.rs:203-231).  Nothing here changes WHAT a walk finds except at
// epsilon ties: the oracle and the kernels walk whatever tree the compiler hands them.
//
//   optimize_by_reinsertion   Bittner, Hapala, Havran 2013 ("Fast insertion-based optimization of bounding volume
//                             hierarchies"): take a subtree out, let its sibling take the parent's place, and put it back
//                             where the surface-area cost of the whole tree grows least (branch and bound over the tree).
//   order_children_by_rays    the reference's walk is FIXED-order (scene.glsl:97-133): a hit in the first child shrinks tMax
//                             and culls the second child's box, never the other way round.  Which child should come first is a
//                             property of the rays the renderer will trace, so a sample of them is traced here (camera paths
//                             through the scene: closest-hit rays and next-event shadow rays) and every inner node is given
//                             the order that saves that sample the most node visits.
#include "scene.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <queue>
#include <system_error>
#include <thread>

namespace hijiki {

namespace {

inline Aabb join2(const Aabb& a, const Aabb& b) { Aabb r = a; r.join(b); return r; }
inline bool same_box(const Aabb& a, const Aabb& b) { return std::memcmp(&a, &b, sizeof(Aabb)) == 0; }

// The tree with a box and a parent per node (BuildNode keeps the children's boxes in the parent).
struct WorkTree {
  std::vector<int32_t> parent, c0, c1, shape;
  std::vector<Aabb> box;
  explicit WorkTree(const std::vector<BuildNode>& nodes) {
    const size_t n = nodes.size();
    parent.assign(n, -1); c0.assign(n, -1); c1.assign(n, -1); shape.assign(n, -1); box.resize(n);
    box[0] = nodes[0].shape >= 0 ? Aabb::empty() : join2(nodes[0].left_box, nodes[0].right_box);
    for (size_t i = 0; i < n; i++) {
      shape[i] = nodes[i].shape;
      if (nodes[i].shape >= 0) continue;
      c0[i] = nodes[i].left; c1[i] = nodes[i].right;
      parent[c0[i]] = (int32_t)i; parent[c1[i]] = (int32_t)i;
      box[c0[i]] = nodes[i].left_box; box[c1[i]] = nodes[i].right_box;
    }
  }
  void store(std::vector<BuildNode>& nodes) const {
    for (size_t i = 0; i < nodes.size(); i++) {
      nodes[i].shape = shape[i];
      nodes[i].left = c0[i]; nodes[i].right = c1[i];
      if (shape[i] < 0) { nodes[i].left_box = box[c0[i]]; nodes[i].right_box = box[c1[i]]; }
    }
  }
  void refit_from(int32_t nd) {               // boxes of nd and its ancestors, until one does not change
    while (nd >= 0) {
      const Aabb b = join2(box[c0[nd]], box[c1[nd]]);
      if (same_box(b, box[nd])) break;
      box[nd] = b;
      nd = parent[nd];
    }
  }
  double sah() const {                         // sum of the inner nodes' half areas (the leaves' is a constant)
    double s = 0;
    for (size_t i = 0; i < box.size(); i++) if (shape[i] < 0) s += box[i].half_area();
    return s;
  }
};

}  // namespace

// One pass = every node but the root and its two children, largest box first, is taken out and put back at its best place.
// Returns the surface-area cost (sum of the inner nodes' half areas) after the last pass.
double optimize_by_reinsertion(std::vector<BuildNode>& nodes, int passes) {
  if (nodes.size() < 7 || passes <= 0) return 0.0;
  WorkTree t(nodes);
  const size_t n = nodes.size();
  const bool verbose = BuildTuning::get().verbose;
  if (verbose) std::fprintf(stderr, "reinsertion: %zu nodes, cost %.4f\n", n, t.sah());
  struct Cand { float induced; int32_t node; bool operator<(const Cand& o) const { return induced > o.induced; } };
  std::vector<Cand> heap;
  std::vector<int32_t> order(n);
  for (int pass = 0; pass < passes; pass++) {
    for (size_t i = 0; i < n; i++) order[i] = (int32_t)i;
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return t.box[a].half_area() > t.box[b].half_area(); });
    size_t moved = 0, tried = 0;
    // candidates of a pass: the nodes with the largest boxes (most rays pass them, and their searches are the short ones: the bound
    // prunes with the candidate's own area); HJ_BVH_REINSERT_MAX of them, by default all of a small tree and 1/16 of a large one
    const long cap = BuildTuning::get().reinsert_max;
    const size_t limit = cap > 0 ? (size_t)cap : std::max<size_t>(65536, n / 16);
    for (int32_t N : order) {
      if (tried++ >= limit) break;
      const int32_t P = t.parent[N];
      if (P <= 0) continue;                                     // the root, or a child of the root (index 0 stays the root)
      const int32_t G = t.parent[P];
      const int32_t S = t.c0[P] == N ? t.c1[P] : t.c0[P];
      // take N (and P) out: S takes P's place
      (t.c0[G] == P ? t.c0[G] : t.c1[G]) = S;
      t.parent[S] = G;
      t.refit_from(G);
      // best node X to become N's sibling: minimise (area of X u N) + (growth of X's ancestors)
      const Aabb nb = t.box[N];
      const float area_n = nb.half_area();
      float best_cost = std::numeric_limits<float>::infinity();
      int32_t best = S;
      heap.clear();
      heap.push_back({0.f, 0});
      while (!heap.empty()) {
        std::pop_heap(heap.begin(), heap.end());
        const Cand c = heap.back();
        heap.pop_back();
        if (c.induced + area_n >= best_cost) break;
        const int32_t X = c.node;
        const float direct = join2(t.box[X], nb).half_area();
        const float total = c.induced + direct;
        if (X != 0 && total < best_cost) { best_cost = total; best = X; }
        const float down = total - t.box[X].half_area();       // what X's children inherit
        if (t.shape[X] < 0 && down + area_n < best_cost) {
          heap.push_back({down, t.c0[X]}); std::push_heap(heap.begin(), heap.end());
          heap.push_back({down, t.c1[X]}); std::push_heap(heap.begin(), heap.end());
        }
      }
      // put P back as the parent of (best, N), in best's place
      const int32_t X = best, XP = t.parent[X];
      (t.c0[XP] == X ? t.c0[XP] : t.c1[XP]) = P;
      t.parent[P] = XP;
      t.c0[P] = X; t.c1[P] = N;
      t.parent[X] = P; t.parent[N] = P;
      t.box[P] = join2(t.box[X], nb);
      t.refit_from(XP);
      if (X != S) moved++;
    }
    if (verbose) std::fprintf(stderr, "reinsertion pass %d: %zu subtrees moved, cost %.4f\n", pass, moved, t.sah());
    if (moved == 0) break;
  }
  t.store(nodes);
  return t.sah();
}

namespace {
// std::stable_sort over `parts` threads: the parts sorted side by side, then merged pairwise (std::inplace_merge keeps equal keys
// in order): the same permutation as the serial call.
template <class It, class Cmp>
void stable_sort_parallel(It first, It last, Cmp cmp, unsigned parts) {
  const size_t n = (size_t)(last - first);
  if (parts < 2 || n < 65536) { std::stable_sort(first, last, cmp); return; }
  std::vector<size_t> cut(parts + 1);
  for (unsigned k = 0; k <= parts; k++) cut[k] = n * k / parts;
  {
    std::vector<std::thread> th;
    try { for (unsigned k = 1; k < parts; k++) th.emplace_back([&, k] { std::stable_sort(first + (long)cut[k], first + (long)cut[k + 1], cmp); }); }
    catch (const std::system_error&) {}
    const unsigned started = (unsigned)th.size() + 1;
    std::stable_sort(first, first + (long)cut[1], cmp);
    for (unsigned k = started; k < parts; k++) std::stable_sort(first + (long)cut[k], first + (long)cut[k + 1], cmp);   // (threads that could not be started)
    for (auto& t : th) t.join();
  }
  for (unsigned width = 1; width < parts; width *= 2) {
    std::vector<std::thread> th;
    for (unsigned k = 0; k + width < parts; k += 2 * width) {
      const size_t a = cut[k], m = cut[k + width], b = cut[std::min(parts, k + 2 * width)];
      auto job = [=] { std::inplace_merge(first + (long)a, first + (long)m, first + (long)b, cmp); };
      try { th.emplace_back(job); } catch (const std::system_error&) { job(); }
    }
    for (auto& t : th) t.join();
  }
}
}  // namespace

// The same optimisation for LARGE trees (beyond 400 000 nodes the serial pass over all nodes takes half a minute: round 5 ran none
// there).  Candidates are taken in the same order, in batches: phase 1 - every candidate of a batch searches its best place on the
// tree AS IT STANDS at the start of the batch, in parallel, read-only, with its own removal emulated on the fly (its parent P is
// skipped, the sibling S stands in P's place, the ancestors' boxes are refitted along the path) - exactly the serial search;
// phase 2 - the moves are applied one after the other in candidate order, each checked against what the earlier moves of the
// batch did (the target must not have moved under the candidate, nor be the parent that leaves with it).  Batches start small
// (the largest boxes, whose moves interact most) and grow.  The result depends on the batch sizes, NOT on the thread count.
double optimize_by_reinsertion_batched(std::vector<BuildNode>& nodes, int passes) {
  if (nodes.size() < 7 || passes <= 0) return 0.0;
  WorkTree t(nodes);
  const size_t n = nodes.size();
  const bool verbose = BuildTuning::get().verbose;
  if (verbose) std::fprintf(stderr, "batched reinsertion: %zu nodes, cost %.4f\n", n, t.sah());
  std::vector<int32_t> order(n), target(n);
  const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  // one candidate's search: the best sibling X for N on the tree without N and its parent; -1: stay.  (Per thread: the heap.  The
  // path - the ancestors whose boxes shrink when N leaves - is a handful of nodes: a linear look-up beats a stamp per node, whose
  // 8 MB per thread cost more cache misses than the look-up costs compares.)
  struct Cand { float induced; int32_t node; bool operator<(const Cand& o) const { return induced > o.induced; } };
  struct alignas(128) Scratch { std::vector<Cand> heap; };      // (a cache line of its own: the vector's end pointer moves with every push)
  auto search = [&t](int32_t N, Scratch& sc) -> int32_t {
    const int32_t P = t.parent[N];
    if (P <= 0) return -1;
    const int32_t G = t.parent[P], S = t.c0[P] == N ? t.c1[P] : t.c0[P];
    // boxes of the ancestors of P once N is gone (the serial pass refits them before it searches)
    int32_t path_node[192];
    Aabb path_box[192];
    int np = 0;
    {
      int32_t below = P;
      Aabb below_box = t.box[S];
      for (int32_t a = G; a >= 0 && np < 192; a = t.parent[a]) {
        const int32_t other = t.c0[a] == below ? t.c1[a] : t.c0[a];
        const Aabb b = join2(below_box, t.box[other]);
        if (same_box(b, t.box[a])) break;                      // nothing changes from here up
        path_node[np] = a; path_box[np] = b; np++;
        below = a; below_box = b;
      }
    }
    auto box_of = [&](int32_t x) -> const Aabb& { for (int k = 0; k < np; k++) if (path_node[k] == x) return path_box[k]; return t.box[x]; };
    auto child = [&](int32_t x, int which) { const int32_t c = which ? t.c1[x] : t.c0[x]; return c == P ? S : c; };
    const Aabb nb = t.box[N];
    const float area_n = nb.half_area();
    float best_cost = std::numeric_limits<float>::infinity();
    int32_t best = S;
    std::vector<Cand>& heap = sc.heap;
    heap.clear();
    heap.push_back({0.f, 0});
    while (!heap.empty()) {
      std::pop_heap(heap.begin(), heap.end());
      const Cand c = heap.back();
      heap.pop_back();
      if (c.induced + area_n >= best_cost) break;
      const int32_t X = c.node;
      const Aabb& xb = box_of(X);
      const float direct = join2(xb, nb).half_area();
      const float total = c.induced + direct;
      if (X != 0 && total < best_cost) { best_cost = total; best = X; }
      const float down = total - xb.half_area();
      if (t.shape[X] < 0 && down + area_n < best_cost) {
        heap.push_back({down, child(X, 0)}); std::push_heap(heap.begin(), heap.end());
        heap.push_back({down, child(X, 1)}); std::push_heap(heap.begin(), heap.end());
      }
    }
    return best == S ? -1 : best;
  };
  std::vector<Scratch> scratch(hw);
  // From the second pass on only the NEIGHBOURHOOD of what the previous pass changed is searched again: the nodes a move touched
  // (the subtree's root, its old and new parent and sibling, every ancestor whose box changed), their children and their siblings.
  // (A better place may also have opened up far away; the full re-search finds 1 in 10^3 such and costs ten times as much.)
  std::vector<uint8_t> touched(n, 1), cand(n, 0);
  std::vector<float> area(n, 0.f);
  auto refit_marking = [&](int32_t nd) {
    while (nd >= 0) {
      const Aabb b = join2(t.box[t.c0[nd]], t.box[t.c1[nd]]);
      if (same_box(b, t.box[nd])) break;
      t.box[nd] = b;
      touched[nd] = 1;
      nd = t.parent[nd];
    }
  };
  for (int pass = 0; pass < passes; pass++) {
    size_t count = 0;
    if (pass == 0) {
      for (size_t i = 0; i < n; i++) order[i] = (int32_t)i;
      count = n;
    } else {
      std::fill(cand.begin(), cand.end(), 0);
      for (size_t i = 0; i < n; i++) {
        if (!touched[i]) continue;
        cand[i] = 1;
        if (t.shape[i] < 0) { cand[t.c0[i]] = 1; cand[t.c1[i]] = 1; }
        const int32_t P = t.parent[i];
        if (P >= 0) cand[t.c0[P] == (int32_t)i ? t.c1[P] : t.c0[P]] = 1;
      }
      for (size_t i = 0; i < n; i++) if (cand[i]) order[count++] = (int32_t)i;
    }
    std::fill(touched.begin(), touched.end(), 0);
    for (size_t i = 0; i < count; i++) area[order[i]] = t.box[order[i]].half_area();      // (the sort's key, once per candidate instead of once per compare)
    stable_sort_parallel(order.begin(), order.begin() + (long)count, [&](int32_t a, int32_t b) { return area[a] > area[b]; }, std::min(8u, hw));
    const long cap = BuildTuning::get().reinsert_max;
    const size_t limit = cap > 0 ? std::min<size_t>((size_t)cap, count) : count;
    size_t moved = 0, stale = 0;
    const auto pass_t0 = std::chrono::steady_clock::now();
    for (size_t begin = 0; begin < limit;) {
      const size_t batch = std::min(limit - begin, std::min<size_t>(8192, std::max<size_t>(64, begin / 4)));
      // phase 1
      std::atomic<size_t> next{begin};
      auto run = [&](unsigned me) {
        for (;;) {
          const size_t i0 = next.fetch_add(16);
          if (i0 >= begin + batch) break;
          for (size_t i = i0; i < std::min(begin + batch, i0 + 16); i++) target[i] = search(order[i], scratch[me]);
        }
      };
      std::vector<std::thread> pool;
      if (batch >= 512) { try { for (unsigned k = 1; k < hw; k++) pool.emplace_back(run, k); } catch (const std::system_error&) {} }
      run(0);
      for (auto& th : pool) th.join();
      // phase 2
      for (size_t i = begin; i < begin + batch; i++) {
        const int32_t N = order[i], X = target[i];
        if (X < 0) continue;
        const int32_t P = t.parent[N];
        if (P <= 0 || X == P || X == 0) { stale++; continue; }
        bool inside = false;                                   // an earlier move of this batch may have put X under N
        for (int32_t a = X; a >= 0; a = t.parent[a]) if (a == N) { inside = true; break; }
        if (inside) { stale++; continue; }
        const int32_t G = t.parent[P], S = t.c0[P] == N ? t.c1[P] : t.c0[P];
        if (X == S) continue;
        (t.c0[G] == P ? t.c0[G] : t.c1[G]) = S;
        t.parent[S] = G;
        refit_marking(G);
        const int32_t XP = t.parent[X];
        (t.c0[XP] == X ? t.c0[XP] : t.c1[XP]) = P;
        t.parent[P] = XP;
        t.c0[P] = X; t.c1[P] = N;
        t.parent[X] = P; t.parent[N] = P;
        t.box[P] = join2(t.box[X], t.box[N]);
        refit_marking(XP);
        touched[N] = touched[P] = touched[S] = touched[X] = touched[G] = touched[XP] = 1;
        moved++;
      }
      begin += batch;
    }
    if (verbose) std::fprintf(stderr, "batched reinsertion pass %d: %zu candidates, %zu subtrees moved (%zu targets gone stale), cost %.4f, %.2f s on %u threads\n", pass, limit, moved, stale, t.sah(),
                              std::chrono::duration<double>(std::chrono::steady_clock::now() - pass_t0).count(), hw);
    if (moved == 0) break;
  }
  t.store(nodes);
  return t.sah();
}

// ---------------------------------------------------------------------------------------------------------------------
// Child order by sampled rays.

namespace {

inline int direction_classes(int mode) { return hj_direction_classes(mode); }
inline int ray_direction_class(int mode, const float d[3]) { return hj_ray_direction_class(mode, d); }

struct V3 { float x, y, z; };
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline V3 norm(V3 a) { const float l = std::sqrt(dot(a, a)); return l > 0 ? a * (1.0f / l) : a; }
inline V3 v3(const float* p) { return {p[0], p[1], p[2]}; }

struct Rng {                       // SplitMix64: the sample must not depend on the thread count
  uint64_t s;
  uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
  float uni() { return (float)(next() >> 40) * (1.0f / 16777216.0f); }
};

struct SRay { V3 o, d; float tmin, tmax; bool any; int32_t hit; float hit_t; };

constexpr float kEpsF = 1e-4f;
constexpr float kInfF = std::numeric_limits<float>::infinity();

struct Voter {
  const std::vector<BuildNode>& nodes;
  const Scene& scene;
  std::vector<Aabb> box;                       // per node
  std::vector<uint32_t> tin, tout;             // DFS interval of every node; a leaf's tin locates its shape
  std::vector<uint32_t> leaf_tin;              // per object
  std::vector<int32_t> emitters;               // objects with an emissive material
  // node visits the sample saves when the left / right child comes first: [class * nodes + node], 64-bit (a node near the
  // root collects up to 4 x its subtree's size per ray: a million paths would wrap 32 bits)
  std::vector<std::atomic<uint64_t>> gain_l, gain_r;
  int dir_mode = 0;                            // direction classes of the rays (ray_class): 0 = one class
  size_t nclass = 1;
  // Weight of a shadow ray's vote in quarters of a closest-hit ray's (HJ_BVH_VOTE_SHADOW).  On the trees hj_scene_upload builds its
  // light-shaft grid for (fewer than 300 000 records) the renderer walks only the shadow rays the grid cannot prove free, a quarter
  // of them on the box scenes: 1; on larger trees all of them: 4.  (c3 +1.9 % with 1 against 4, c2 the same, c4 -1.5 %.)
  uint32_t w_shadow = 4;

  Voter(const std::vector<BuildNode>& nd, const Scene& sc, int mode = 0) : nodes(nd), scene(sc), box(nd.size()), tin(nd.size()), tout(nd.size()),
        leaf_tin(sc.objects.size(), 0), gain_l(nd.size() * (size_t)direction_classes(mode)), gain_r(nd.size() * (size_t)direction_classes(mode)),
        dir_mode(mode), nclass((size_t)direction_classes(mode)) {
    box[0] = join2(nodes[0].left_box, nodes[0].right_box);
    uint32_t clock = 0;
    std::vector<std::pair<int32_t, bool>> st{{0, false}};
    while (!st.empty()) {
      auto [nd_, done] = st.back();
      st.pop_back();
      if (done) { tout[nd_] = clock; continue; }
      tin[nd_] = clock++;
      const BuildNode& b = nodes[nd_];
      if (b.shape >= 0) { leaf_tin[b.shape] = tin[nd_]; tout[nd_] = clock; continue; }
      box[b.left] = b.left_box; box[b.right] = b.right_box;
      st.push_back({nd_, true});
      st.push_back({b.right, false});
      st.push_back({b.left, false});
    }
    for (size_t i = 0; i < gain_l.size(); i++) { gain_l[i].store(0, std::memory_order_relaxed); gain_r[i].store(0, std::memory_order_relaxed); }
    for (size_t i = 0; i < sc.objects.size(); i++)
      if (sc.materials[sc.objects[i].second].tag == HJ_MAT_EMISSIVE) emitters.push_back((int32_t)i);
    const int ws = BuildTuning::get().vote_shadow;
    w_shadow = ws >= 0 ? (uint32_t)ws : (nd.size() >= 300000 ? 4u : 1u);
  }

  // the reference's slab test (scene.glsl:120-131): entry distance, or +inf when the box is not entered
  struct Prep { V3 inv, off; };
  static Prep prep(const SRay& r) {
    Prep p;
    p.inv = {1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z};
    p.off = {-(r.o.x * p.inv.x), -(r.o.y * p.inv.y), -(r.o.z * p.inv.z)};
    return p;
  }
  static float entry(const Aabb& b, const Prep& p, float tmin, float tmax) {
    const float tnx = b.lo[0] * p.inv.x + p.off.x, tpx = b.hi[0] * p.inv.x + p.off.x;
    const float tny = b.lo[1] * p.inv.y + p.off.y, tpy = b.hi[1] * p.inv.y + p.off.y;
    const float tnz = b.lo[2] * p.inv.z + p.off.z, tpz = b.hi[2] * p.inv.z + p.off.z;
    const float t0 = std::fmax(std::fmax(std::fmin(tnx, tpx), std::fmin(tny, tpy)), std::fmin(tnz, tpz));
    const float t1 = std::fmin(std::fmin(std::fmax(tnx, tpx), std::fmax(tny, tpy)), std::fmax(tnz, tpz));
    return (t0 < t1 + kEpsF && t0 < tmax && t1 > tmin) ? t0 : kInfF;
  }

  // shape tests (shapes/*.glsl; plain float: this is a heuristic's sample, not the renderer)
  bool hit_shape(int32_t obj, const SRay& r, float tmax, float& t, float& u, float& v) const {
    const Shape& s = scene.objects[obj].first;
    if (s.kind == ShapeKind::Sphere) {
      const V3 oc = r.o - v3(s.sphere.center);
      const float b = dot(oc, r.d), c = dot(oc, oc) - s.sphere.radius * s.sphere.radius, disc = b * b - c;
      if (disc < 0) return false;
      const float sq = std::sqrt(disc);
      float tt = -b - sq;
      if (tt < r.tmin || tt > tmax) tt = -b + sq;
      if (tt < r.tmin || tt > tmax) return false;
      t = tt; u = v = 0;
      return true;
    }
    V3 a, ab, ac;
    const bool quad = s.kind == ShapeKind::Quad;
    if (quad) { a = v3(s.quad.origin); ab = v3(s.quad.edge1); ac = v3(s.quad.edge2); }
    else { a = v3(scene.vertices[s.tri.v[0]].pos); ab = v3(scene.vertices[s.tri.v[1]].pos) - a; ac = v3(scene.vertices[s.tri.v[2]].pos) - a; }
    const V3 n = cross(ab, ac), ro = r.o - a, q = cross(ro, r.d);
    const float inv = 1.0f / dot(r.d, n);
    const float uu = inv * -dot(q, ac), vv = inv * dot(q, ab), tt = inv * -dot(n, ro);
    if (!(uu >= 0 && vv >= 0 && (quad ? (uu <= 1 && vv <= 1) : uu + vv <= 1) && tt >= r.tmin && tt <= tmax)) return false;
    t = tt; u = uu; v = vv;
    return true;
  }

  // closest hit by an ordered stack walk (nearer child first): the sample's own rays are traced with this
  int32_t closest(const SRay& r, float& t_out, float& u_out, float& v_out) const {
    const Prep p = prep(r);
    float tmax = r.tmax;
    int32_t best = -1;
    int32_t stack[192];
    int sp = 0;
    if (entry(box[0], p, r.tmin, tmax) == kInfF) return -1;
    stack[sp++] = 0;
    while (sp) {
      const int32_t nd = stack[--sp];
      const BuildNode& b = nodes[nd];
      if (b.shape >= 0) {
        float t, u, v;
        if (hit_shape(b.shape, r, tmax, t, u, v)) { tmax = t; best = b.shape; t_out = t; u_out = u; v_out = v; }
        continue;
      }
      const float el = entry(b.left_box, p, r.tmin, tmax), er = entry(b.right_box, p, r.tmin, tmax);
      if (sp + 2 > 190) continue;
      if (el <= er) { if (er < kInfF) stack[sp++] = b.right; if (el < kInfF) stack[sp++] = b.left; }
      else { if (el < kInfF) stack[sp++] = b.left; if (er < kInfF) stack[sp++] = b.right; }
    }
    return best;
  }

  // One ray's votes.  Per subtree entered by the ray: ci = its nodes whose boxes the ray enters at all, ct = those entered
  // before the ray's hit.  At a node whose child X holds the hit: putting X first spares a closest-hit ray the nodes of the
  // other child that lie behind the hit (ci - ct of the other child: tMax = t_hit culls them), an any-hit ray all of them (ci).
  struct Cnt { uint32_t ci, ct; };
  Cnt vote(int32_t nd, const SRay& r, const Prep& p, bool before, uint32_t hit_pos, int depth, size_t cls) {
    const BuildNode& b = nodes[nd];
    if (b.shape >= 0 || depth > 160) return {1u, before ? 1u : 0u};
    Cnt cl{0, 0}, cr{0, 0};
    const float el = entry(b.left_box, p, r.tmin, r.tmax), er = entry(b.right_box, p, r.tmin, r.tmax);
    if (el < kInfF) cl = vote(b.left, r, p, el < r.hit_t, hit_pos, depth + 1, cls);
    if (er < kInfF) cr = vote(b.right, r, p, er < r.hit_t, hit_pos, depth + 1, cls);
    if (r.hit >= 0) {
      if (hit_pos >= tin[b.left] && hit_pos < tout[b.left]) {
        const uint32_t g = r.any ? cr.ci * w_shadow : (cr.ci - cr.ct) * 4u;
        if (g) gain_l[cls + (size_t)nd].fetch_add(g, std::memory_order_relaxed);
      } else if (hit_pos >= tin[b.right] && hit_pos < tout[b.right]) {
        const uint32_t g = r.any ? cl.ci * w_shadow : (cl.ci - cl.ct) * 4u;
        if (g) gain_r[cls + (size_t)nd].fetch_add(g, std::memory_order_relaxed);
      }
    }
    return {1u + cl.ci + cr.ci, before ? 1u + cl.ct + cr.ct : 0u};
  }
  void cast(SRay& r) {
    const Prep p = prep(r);
    const float e0 = entry(box[0], p, r.tmin, r.tmax);
    if (e0 == kInfF) return;
    const float d[3] = {r.d.x, r.d.y, r.d.z};
    (void)vote(0, r, p, e0 < r.hit_t, r.hit >= 0 ? leaf_tin[r.hit] : 0u, 0, (size_t)ray_direction_class(dir_mode, d) * nodes.size());
  }

  // surface point, shading normal and material of a hit (populate*, scene.glsl:160-175, without the tangent frames)
  void surface(int32_t obj, const SRay& r, float t, float u, float v, V3& pos, V3& n) const {
    const Shape& s = scene.objects[obj].first;
    pos = r.o + r.d * t;
    if (s.kind == ShapeKind::Sphere) n = (pos - v3(s.sphere.center)) * (1.0f / s.sphere.radius);
    else if (s.kind == ShapeKind::Quad) n = norm(cross(norm(v3(s.quad.edge1)), norm(v3(s.quad.edge2))));
    else n = norm(v3(scene.vertices[s.tri.v[0]].normal) * (1 - u - v) + v3(scene.vertices[s.tri.v[1]].normal) * u + v3(scene.vertices[s.tri.v[2]].normal) * v);
  }
  V3 point_on(int32_t obj, Rng& g) const {
    const Shape& s = scene.objects[obj].first;
    const float a = g.uni(), b = g.uni();
    if (s.kind == ShapeKind::Sphere) {
      const float z = 2 * a - 1, ph = 6.2831853f * b, rr = std::sqrt(std::fmax(0.f, 1 - z * z));
      return v3(s.sphere.center) + V3{rr * std::cos(ph), rr * std::sin(ph), z} * s.sphere.radius;
    }
    if (s.kind == ShapeKind::Quad) return v3(s.quad.origin) + v3(s.quad.edge1) * a + v3(s.quad.edge2) * b;
    float uu = a, vv = b;
    if (uu + vv > 1) { uu = 1 - uu; vv = 1 - vv; }
    const V3 p0 = v3(scene.vertices[s.tri.v[0]].pos);
    return p0 + (v3(scene.vertices[s.tri.v[1]].pos) - p0) * uu + (v3(scene.vertices[s.tri.v[2]].pos) - p0) * vv;
  }

  // One camera path (render.glsl:81-147 in outline): its closest-hit rays and next-event shadow rays vote as they are found.
  void path(uint64_t index) {
    Rng g{0x48494A494B49ull ^ (index * 0xD1342543DE82EF95ull)};
    const hj_camera& cam = scene.camera;
    const float th = std::tan(0.5f * cam.fov * 0.017453292f);
    const float x = (2 * g.uni() - 1) * th, y = (2 * g.uni() - 1) * th;
    const V3 qv = {cam.rotation[0], cam.rotation[1], cam.rotation[2]};
    const float qw = cam.rotation[3];
    const V3 vv = {x, -y, -1.f};
    const V3 tq = cross(qv, vv) * 2.0f;                                   // v + w t + q x t, t = 2 q x v
    SRay r;
    r.o = v3(cam.position);
    r.d = norm(vv + tq * qw + cross(qv, tq));
    r.tmin = kEpsF; r.tmax = kInfF; r.any = false;
    for (int bounce = 0; bounce < 12; bounce++) {
      float t = 0, u = 0, v = 0;
      r.hit = closest(r, t, u, v);
      r.hit_t = r.hit >= 0 ? t : kInfF;
      cast(r);
      if (r.hit < 0) return;
      const Material& m = scene.materials[scene.objects[r.hit].second];
      if (m.tag == HJ_MAT_EMISSIVE) return;
      V3 pos, n;
      surface(r.hit, r, t, u, v, pos, n);
      V3 wo;
      if (m.tag == HJ_MAT_DIFFUSE || m.tag == HJ_MAT_DIFFUSECBOARD) {
        if (!emitters.empty()) {                                          // scene.glsl:54-89
          const int32_t e = emitters[std::min<size_t>(emitters.size() - 1, (size_t)(g.uni() * (float)emitters.size()))];
          const V3 lp = point_on(e, g);
          V3 d = lp - pos;
          const float dist = std::sqrt(dot(d, d));
          d = d * (1.0f / dist);
          if (dist > 3 * kEpsF && dot(d, n) > 0) {
            SRay s;
            s.o = pos; s.d = d; s.tmin = 2 * kEpsF; s.tmax = dist - kEpsF; s.any = true;
            float st = 0, su = 0, sv = 0;
            s.hit = closest(s, st, su, sv);
            s.hit_t = s.hit >= 0 ? st : kInfF;
            cast(s);
          }
        }
        const float a = g.uni(), b = g.uni(), rr = std::sqrt(a), ph = 6.2831853f * b;
        const V3 bt = std::fabs(n.x) > std::fabs(n.y) ? V3{0, 1, 0} : V3{1, 0, 0};
        const V3 tx = norm(cross(n, bt)), ty = cross(n, tx);
        wo = tx * (rr * std::cos(ph)) + ty * (rr * std::sin(ph)) + n * std::sqrt(std::fmax(0.f, 1 - a));
      } else if (m.tag == HJ_MAT_MIRROR) {
        wo = r.d - n * (2 * dot(n, r.d));
      } else {                                                            // dielectric: material.glsl:50-87
        float eta = m.dielectric.eta, cos_i = -dot(n, r.d);
        V3 nn = n;
        float eta_inv = 1.0f / eta;
        if (cos_i < 0) { eta = eta_inv; eta_inv = 1.0f / eta; nn = n * -1.0f; cos_i = -cos_i; }
        const float k = 1 - eta_inv * eta_inv * (1 - cos_i * cos_i);
        bool reflect = k <= 0;
        if (!reflect) {
          const float cos_o = std::sqrt(k);
          const float rp = (eta * cos_i - cos_o) / (eta * cos_i + cos_o), ro = (cos_i - eta * cos_o) / (cos_i + eta * cos_o);
          reflect = g.uni() < 0.5f * (rp * rp + ro * ro);
          if (!reflect) wo = (r.d - nn * dot(r.d, nn)) * eta_inv - nn * cos_o;
        }
        if (reflect) wo = r.d - nn * (2 * dot(nn, r.d));
      }
      if (bounce > 3 && g.uni() > 0.75f) return;                          // (roulette at about the renderer's survival rate)
      r.o = pos; r.d = norm(wo); r.tmin = 2 * kEpsF; r.tmax = kInfF;
    }
  }
};

}  // namespace

namespace {

// The sample's paths over the host's threads (the gains are integer sums: the result does not depend on the thread count).  A
// thread that cannot be started leaves its paths to the caller's thread.
void run_paths(Voter& v, size_t num_paths) {
  unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  if (num_paths < 4096) nt = 1;
  std::vector<std::thread> th;
  std::vector<unsigned> inline_shares;
  for (unsigned k = 0; k < nt; k++) {
    try {
      th.emplace_back([&v, k, nt, num_paths] { for (size_t i = k; i < num_paths; i += nt) v.path(i); });
    } catch (const std::system_error&) {
      inline_shares.push_back(k);
    }
  }
  for (unsigned k : inline_shares) for (size_t i = k; i < num_paths; i += nt) v.path(i);
  for (auto& t : th) t.join();
}

}  // namespace

// Returns the number of inner nodes whose children were exchanged.
size_t order_children_by_rays(std::vector<BuildNode>& nodes, const Scene& scene, size_t num_paths) {
  if (nodes.size() < 3 || num_paths == 0) return 0;
  Voter v(nodes, scene);
  run_paths(v, num_paths);
  size_t swapped = 0;
  for (size_t i = 0; i < nodes.size(); i++) {
    BuildNode& b = nodes[i];
    if (b.shape >= 0) continue;
    if (v.gain_r[i].load(std::memory_order_relaxed) > v.gain_l[i].load(std::memory_order_relaxed)) {   // (ties and nodes no ray voted on keep their order)
      std::swap(b.left, b.right);
      std::swap(b.left_box, b.right_box);
      swapped++;
    }
  }
  if (BuildTuning::get().verbose) std::fprintf(stderr, "ray-voted child order: %zu paths, %zu of %zu inner nodes exchanged\n", num_paths, swapped, nodes.size() / 2);
  return swapped;
}

// One child order per DIRECTION CLASS of the rays (hj_ray_direction_class): the sample is partitioned by class and every class
// votes for itself; a node on which a class has no opinion (no ray of the class that hit below it, or a tie) takes
//   fallback 0: the order it has in `nodes` (the static order, voted by all rays or not);
//   fallback 1: the class's geometric near-first order (the child whose box centre comes first along the class's mean direction).
// orders[c * nodes.size() + i] = 1 when node i's children change places in class c.  `geometric_only` skips the vote.
void directional_child_orders(const std::vector<BuildNode>& nodes, const Scene& scene, size_t num_paths, int mode, int fallback,
                              bool geometric_only, std::vector<uint8_t>& orders) {
  const size_t n = nodes.size(), K = (size_t)direction_classes(mode);
  orders.assign(n * K, 0);
  if (n < 3) return;
  Voter v(nodes, scene, mode);
  if (!geometric_only && num_paths > 0) run_paths(v, num_paths);
  // mean direction of a class: signs (+-1, +-1, +-1) on the selected axes, 0 elsewhere; the major-axis classes: +-e_axis
  auto class_dir = [&](size_t c, float out[3]) {
    out[0] = out[1] = out[2] = 0.f;
    if (mode == HJ_DIR_MODE_MAJOR_AXIS) { out[c / 2] = (c & 1) ? -1.f : 1.f; return; }
    int k = 0;
    for (int a = 0; a < 3; a++) if (mode & (1 << a)) out[a] = ((c >> k++) & 1) ? -1.f : 1.f;
  };
  for (size_t c = 0; c < K; c++) {
    float dir[3];
    class_dir(c, dir);
    for (size_t i = 0; i < n; i++) {
      const BuildNode& b = nodes[i];
      if (b.shape >= 0) continue;
      const uint64_t gl = v.gain_l[c * n + i].load(std::memory_order_relaxed), gr = v.gain_r[c * n + i].load(std::memory_order_relaxed);
      bool swap = false;
      if (gl != gr) swap = gr > gl;
      else if (fallback == 1 || geometric_only) {
        float s = 0.f;
        for (int a = 0; a < 3; a++) s += dir[a] * ((b.right_box.lo[a] + b.right_box.hi[a]) - (b.left_box.lo[a] + b.left_box.hi[a]));
        swap = s < 0.f;      // the right child's centre comes first along the class direction
      }
      orders[c * n + i] = swap ? 1 : 0;
    }
  }
}

}  // namespace hijiki
