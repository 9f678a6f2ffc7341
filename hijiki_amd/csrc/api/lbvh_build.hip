// hj_build_bvh_device: the host half of the device BVH build (kernels/hj_lbvh.h) - SURVEY.md 8(f) #2.
#include "hj_internal.h"
#include "tree_vote.hpp"
#include "../kernels/hj_lbvh.h"

#include <rocprim/device/device_radix_sort.hpp>

#pragma clang fp contract(off)

using namespace hjapi;

extern "C" {

// the tree of Scene::compile (src/main.rs:199-231), built on the device (kernels/hj_lbvh.h).
int hj_build_bvh_device(hj_context* ctx, const hj_scene_desc* s, hj_bvh_node* out_nodes, size_t capacity, size_t* out_num_nodes) {
  if (!ctx) return HJ_ERR_INVALID;
  HJ_NOT_BUSY(ctx);
  HJ_NOT_PIPELINED(ctx);
  if (!s) return set_error(ctx, HJ_ERR_INVALID, "null argument");
  const size_t n = s->num_spheres + s->num_quads + s->num_triangles;
  if (n < 2) return set_error(ctx, HJ_ERR_INVALID, "scene needs at least 2 shapes (reference panics: root would be a leaf, src/main.rs:230)");
  if (n >= hj::kInnerFlag / 4) return set_error(ctx, HJ_ERR_INVALID, "scene too large");
  const size_t total = 2 * n - 1;
  if (out_nodes && capacity < total) return set_error(ctx, HJ_ERR_INVALID, "node buffer holds %zu records, the tree has %zu", capacity, total);
  if ((s->num_spheres && !s->spheres) || (s->num_quads && !s->quads) || (s->num_triangles && (!s->triangles || !s->vertices)))
    return set_error(ctx, HJ_ERR_INVALID, "null shape array");
  for (size_t i = 0; i < s->num_triangles; i++)
    for (int k = 0; k < 3; k++)
      if (s->triangles[i].v[k] >= s->num_vertices) return set_error(ctx, HJ_ERR_INVALID, "triangle %zu refers to unknown vertex", i);
  HJ_HIP(ctx, hipSetDevice(ctx->device));
  // HJ_LBVH_TIMING=1: wall time of the build's stages on stderr (the stream is drained at every mark)
  const Tuning tn = ctx->tuning = Tuning::from_env();
  const bool timing = tn.lbvh_timing != 0;
  auto t_last = std::chrono::steady_clock::now();
  auto mark = [&](const char* what) {
    if (!timing) return;
    (void)hipStreamSynchronize(ctx->stream);
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "hj_build_bvh_device: %-28s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  mark("argument checks");
  std::vector<DevBuf> bufs;
  struct Release { std::vector<DevBuf>& b; ~Release() { for (auto& x : b) x.release(); } } release{bufs};
  bufs.reserve(32);
  auto dev = [&](size_t bytes, void** p) -> int {
    bufs.emplace_back();
    const int rc = dev_alloc(ctx, bufs.back(), std::max<size_t>(bytes, 16));
    *p = bufs.back().p;
    return rc;
  };
  int rc = HJ_OK;
#define HJ_DEVBUF(ptr, type, count) do { void* p_ = nullptr; rc = dev(sizeof(type) * (count), &p_); if (rc != HJ_OK) return rc; ptr = static_cast<type*>(p_); } while (0)
  hipStream_t st = ctx->stream;
  ctx->resident.release();                                  // (a tree of an earlier build that nobody took over)
  void *keep_sp = nullptr, *keep_qd = nullptr, *keep_tr = nullptr, *keep_vx = nullptr;
  hj::lbvh::Shapes sh{};
  {
    float4* sp = nullptr; float4* qd = nullptr; hj_triangle* tr = nullptr; hj_vertex* vx = nullptr;
    HJ_DEVBUF(sp, float4, s->num_spheres);
    HJ_DEVBUF(qd, float4, 3 * s->num_quads);
    HJ_DEVBUF(tr, hj_triangle, s->num_triangles);
    HJ_DEVBUF(vx, hj_vertex, s->num_vertices);
    if (s->num_spheres) HJ_HIP(ctx, hipMemcpyAsync(sp, s->spheres, sizeof(float4) * s->num_spheres, hipMemcpyHostToDevice, st));
    if (s->num_quads) HJ_HIP(ctx, hipMemcpyAsync(qd, s->quads, sizeof(float4) * 3 * s->num_quads, hipMemcpyHostToDevice, st));
    if (s->num_triangles) HJ_HIP(ctx, hipMemcpyAsync(tr, s->triangles, sizeof(hj_triangle) * s->num_triangles, hipMemcpyHostToDevice, st));
    if (s->num_vertices) HJ_HIP(ctx, hipMemcpyAsync(vx, s->vertices, sizeof(hj_vertex) * s->num_vertices, hipMemcpyHostToDevice, st));
    sh.spheres = sp; sh.quads = qd; sh.triangles = tr; sh.vertices = vx;
    keep_sp = sp; keep_qd = qd; keep_tr = tr; keep_vx = vx;
    sh.ns = (uint32_t)s->num_spheres; sh.nq = (uint32_t)s->num_quads; sh.nt = (uint32_t)s->num_triangles;
  }
  hj::lbvh::Tree t{};
  unsigned long long* keys_in = nullptr;
  hj_bvh_node* d_out = nullptr;
  HJ_DEVBUF(t.leaf_lo, float4, n);
  HJ_DEVBUF(t.leaf_hi, float4, n);
  HJ_DEVBUF(t.bounds, int, 12);
  HJ_DEVBUF(keys_in, unsigned long long, n);
  HJ_DEVBUF(t.keys, unsigned long long, n);
  HJ_DEVBUF(t.child, uint32_t, 2 * (n - 1));
  HJ_DEVBUF(t.first, uint32_t, n - 1);
  HJ_DEVBUF(t.count, uint32_t, n - 1);
  HJ_DEVBUF(t.parent, uint32_t, total);
  HJ_DEVBUF(t.node_lo, float4, n - 1);
  HJ_DEVBUF(t.node_hi, float4, n - 1);
  HJ_DEVBUF(t.arrived, uint32_t, n - 1);
  HJ_DEVBUF(d_out, hj_bvh_node, total);
  const uint32_t N = (uint32_t)n;
  const dim3 blk(256), grid_n((N + 255u) / 256u);
  uint32_t* d_nbig = nullptr;
  HJ_DEVBUF(d_nbig, uint32_t, 1);
  mark("allocations + shape upload");
  hipLaunchKernelGGL(hj::lbvh::k_init_bounds, dim3(1), dim3(64), 0, st, t.bounds);
  hipLaunchKernelGGL(hj::lbvh::k_shape_boxes, grid_n, blk, 0, st, sh, t, N);
  uint32_t idx_bits = 1;
  while ((1ull << idx_bits) < n) idx_bits++;
  const uint32_t axis_bits = std::min<uint32_t>(20u, (63u - idx_bits) / 3u);
  const unsigned long long idx_mask = (1ull << idx_bits) - 1ull;
  void* sort_tmp = nullptr;
  size_t sort_bytes = 0;
  HJ_HIP(ctx, rocprim::radix_sort_keys(nullptr, sort_bytes, keys_in, t.keys, n, 0, 64, st));
  HJ_DEVBUF(sort_tmp, char, sort_bytes);
  // Large shapes (hj_lbvh.h) stay out of the Morton tree; HJ_LBVH_BIG_PCT = threshold in per cent of the scene's box area
  // (0 = everything goes into the Morton tree).  They sort behind everything else (bit 63 of the key).
  float big_frac = (float)tn.lbvh_big_pct / 100.0f;
  uint32_t nbig = 0;
  for (int attempt = 0; attempt < 2; attempt++) {
    HJ_HIP(ctx, hipMemsetAsync(d_nbig, 0, sizeof(uint32_t), st));
    hj::lbvh::Tree unsorted = t;
    unsorted.keys = keys_in;
    hipLaunchKernelGGL(hj::lbvh::k_morton_keys, grid_n, blk, 0, st, unsorted, N, idx_bits, axis_bits, big_frac, d_nbig);
    HJ_HIP(ctx, rocprim::radix_sort_keys(sort_tmp, sort_bytes, keys_in, t.keys, n, 0, 64, st));
    HJ_HIP(ctx, hipMemcpyAsync(&nbig, d_nbig, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HJ_HIP(ctx, hipStreamSynchronize(st));
    if (nbig == 0 || (nbig <= 256 && n - nbig >= 2)) break;
    big_frac = 0.f;                                          // too many large shapes (or nothing else): one Morton tree
  }
  mark("boxes, keys, sort");
  // ---- the Morton tree over the m = n - nbig small shapes
  const uint32_t m = N - nbig;
  const size_t sub_total = 2 * (size_t)m - 1;
  // src/main.rs:231 hard-codes 1 000 000 for the root's exit; larger trees get the node count (see host/scene.cpp)
  const uint32_t root_exit = total > HJ_BVH_ROOT_EXIT ? (uint32_t)total : HJ_BVH_ROOT_EXIT;
  const dim3 grid_m((m + 255u) / 256u), grid_sub(((uint32_t)sub_total + 255u) / 256u);
  hipLaunchKernelGGL(hj::lbvh::k_hierarchy, grid_m, blk, 0, st, t, m);
  // ---- clusters of the Morton tree (HJ_LBVH_CLUSTER leaves at most; 0 = the whole tree is one cluster)
  // (clusters of up to 64 leaves are re-split by one thread each, k_emit_clusters_sah - rounds 2-3's default -, larger ones
  // up to 512 by one wave each, k_emit_clusters_sah_wave)
  const uint32_t cmax_env = (uint32_t)tn.lbvh_cluster;
  const uint32_t cmax = cmax_env == 0 ? m : cmax_env;
  // inside the clusters: SAH re-split (one thread per cluster, which needs no boxes of the Morton tree's internal nodes) or the
  // Morton topology as it is (HJ_LBVH_SAH=0, or clusters larger than the kernel's arrays: bottom-up refit first)
  const bool sah_clusters = cmax <= hj::lbvh::kWaveClusterMax && tn.lbvh_sah != 0;
  const bool sah_wave = sah_clusters && cmax > hj::lbvh::kClusterMax;
  if (!sah_clusters) hipLaunchKernelGGL(hj::lbvh::k_refit, grid_m, blk, 0, st, t, m, idx_mask);
  hj::lbvh::Clusters cl{};
  {
    uint32_t* base_w = nullptr; uint32_t* exit_w = nullptr;
    HJ_DEVBUF(cl.count, uint32_t, 1);
    HJ_DEVBUF(cl.slot_of, uint32_t, sub_total);
    HJ_DEVBUF(cl.node, uint32_t, m);
    HJ_DEVBUF(cl.lo, float4, m);
    HJ_DEVBUF(cl.hi, float4, m);
    HJ_DEVBUF(base_w, uint32_t, m);
    HJ_DEVBUF(exit_w, uint32_t, m);
    cl.base = base_w; cl.exit = exit_w;
    HJ_HIP(ctx, hipMemsetAsync(cl.count, 0, sizeof(uint32_t), st));
    hipLaunchKernelGGL(hj::lbvh::k_mark_clusters, grid_sub, blk, 0, st, t, m, cmax, idx_mask, cl, sah_clusters);
  }
  uint32_t K = 0;
  HJ_HIP(ctx, hipMemcpyAsync(&K, cl.count, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HJ_HIP(ctx, hipStreamSynchronize(st));
  mark("hierarchy, refit, clusters");
  // the wave-per-cluster re-split starts NOW, into a staging array, and runs while the host builds the top of the tree over
  // the clusters' boxes; k_place_clusters moves the records to their places once those are known
  // (on a stream of its own - a batch slot's, idle outside render calls -, so that the small copies the host needs for the top
  // do not queue behind it)
  hj_bvh_node* d_staged = nullptr;
  struct Events { hipEvent_t a = nullptr, b = nullptr; ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); } } ev;
  hipStream_t side = ctx->slots[0].stream;
  if (sah_wave) {
    HJ_DEVBUF(d_staged, hj_bvh_node, 2 * (size_t)m);
    HJ_HIP(ctx, hipEventCreateWithFlags(&ev.a, hipEventDisableTiming));
    HJ_HIP(ctx, hipEventCreateWithFlags(&ev.b, hipEventDisableTiming));
    HJ_HIP(ctx, hipEventRecord(ev.a, st));
    HJ_HIP(ctx, hipStreamWaitEvent(side, ev.a, 0));
    hipLaunchKernelGGL(hj::lbvh::k_emit_clusters_sah_wave, dim3(K), dim3(64), 0, side, t, K, cl, idx_mask, d_staged,
                       tn.bvh_child_order);
    HJ_HIP(ctx, hipEventRecord(ev.b, side));
  }
  // ---- the top of the tree on the host: binned SAH over the K clusters and the nbig large shapes
  struct Item { float lo[3], hi[3]; uint32_t shape; uint32_t cluster; uint32_t records; float weight; uint32_t first; };
  std::vector<Item> items(K + nbig);
  {
    std::vector<float4> clo(K), chi(K);
    HJ_HIP(ctx, hipMemcpy(clo.data(), cl.lo, sizeof(float4) * K, hipMemcpyDeviceToHost));
    HJ_HIP(ctx, hipMemcpy(chi.data(), cl.hi, sizeof(float4) * K, hipMemcpyDeviceToHost));
    for (uint32_t k = 0; k < K; k++) {
      Item& it = items[k];
      it.lo[0] = clo[k].x; it.lo[1] = clo[k].y; it.lo[2] = clo[k].z;
      it.hi[0] = chi[k].x; it.hi[1] = chi[k].y; it.hi[2] = chi[k].z;
      const uint32_t cnt = __builtin_bit_cast(uint32_t, chi[k].w);
      it.shape = HJ_BVH_INNER; it.cluster = k; it.records = 2 * cnt - 1; it.weight = (float)cnt;
      it.first = __builtin_bit_cast(uint32_t, clo[k].w);
    }
    // the cluster numbers come from an atomic counter: put the list into the order of the sorted leaves, so that the few
    // order-dependent decisions below (equal centroids) do not depend on the run
    std::sort(items.begin(), items.begin() + K, [](const Item& x, const Item& y) { return x.first < y.first; });
    if (nbig != 0) {   // boxes of the large shapes: the ones k_shape_boxes computed (src/shape.rs:13-20,46-54, src/main.rs:74-79)
      // one gather kernel + one copy for all of them (they sort behind the small shapes: keys [m, n))
      float4* d_big = nullptr;
      HJ_DEVBUF(d_big, float4, 2 * (size_t)nbig);
      hipLaunchKernelGGL(hj::lbvh::k_gather_big, dim3((nbig + 63u) / 64u), dim3(64), 0, st, t, m, nbig, idx_mask, d_big);
      std::vector<float4> big(2 * (size_t)nbig);
      HJ_HIP(ctx, hipMemcpyAsync(big.data(), d_big, sizeof(float4) * big.size(), hipMemcpyDeviceToHost, st));
      HJ_HIP(ctx, hipStreamSynchronize(st));
      for (uint32_t k = 0; k < nbig; k++) {
        const float4 lo = big[2 * (size_t)k], hi = big[2 * (size_t)k + 1];
        Item& it = items[K + k];
        it.lo[0] = lo.x; it.lo[1] = lo.y; it.lo[2] = lo.z; it.hi[0] = hi.x; it.hi[1] = hi.y; it.hi[2] = hi.z;
        it.shape = __builtin_bit_cast(uint32_t, lo.w); it.cluster = 0; it.records = 1; it.weight = 1.0f; it.first = 0;
      }
    }
  }
  std::vector<std::pair<uint32_t, hj_bvh_node>> top_records; // (position, record) of the host-built part
  std::vector<uint32_t> cbase(K, 0), cexit(K, 0);
  {
    using Records = std::vector<std::pair<uint32_t, hj_bvh_node>>;
    // A node of the host-built part: a leaf stands for one item (a cluster's subtree or a large shape) or, in the part above
    // the worker tasks, for a whole task subtree (`sub`); an inner node keeps its children's boxes (what the reference's
    // flattened records hold: a node's box is the one its PARENT kept for it, src/main.rs:214-231).
    struct TNode { int32_t left = -1, right = -1, item = -1, sub = -1; float lo[2][3], hi[2][3]; uint32_t records = 0; float weight = 0; };
    struct Tree { std::vector<TNode> n; };
    struct Task { size_t a, b; int depth; };
    struct Builder {
      std::vector<Item>& items;
      std::vector<uint32_t> ids;
      size_t task_items = 0;                  // subtrees of at most this many items are set aside as tasks (0: never)
      int child_order = 3;                    // HJ_BVH_CHILD_ORDER (0: as split)
      int rotate_passes = 8;                  // HJ_BVH_ROTATE
      std::vector<Task> tasks;
      std::vector<Tree> task_trees;
      static float area(const float* lo, const float* hi) {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return (dx >= 0 && dy >= 0 && dz >= 0) ? dx * dy + dy * dz + dz * dx : 0.f;
      }
      static void grow(float* lo, float* hi, const float* alo, const float* ahi) {
        for (int k = 0; k < 3; k++) { lo[k] = std::fmin(lo[k], alo[k]); hi[k] = std::fmax(hi[k], ahi[k]); }
      }
      void bounds(size_t a, size_t b, float* lo, float* hi) const {
        for (int k = 0; k < 3; k++) { lo[k] = INFINITY; hi[k] = -INFINITY; }
        for (size_t i = a; i < b; i++) grow(lo, hi, items[ids[i]].lo, items[ids[i]].hi);
      }
      // binned-SAH tree over ids[a, b) into t; `defer`: subtrees small enough become tasks (leaves with `sub` = task number)
      int32_t build(Tree& t, size_t a, size_t b, int depth, bool defer) {
        const int32_t me = (int32_t)t.n.size();
        t.n.emplace_back();
        if (b - a == 1) {
          const Item& it = items[ids[a]];
          t.n[me].item = (int32_t)ids[a]; t.n[me].records = it.records; t.n[me].weight = it.weight;
          return me;
        }
        if (defer && b - a <= task_items) {
          uint32_t rec = (uint32_t)(b - a) - 1; float w = 0.f;
          for (size_t i = a; i < b; i++) { rec += items[ids[i]].records; w += items[ids[i]].weight; }
          t.n[me].sub = (int32_t)tasks.size(); t.n[me].records = rec; t.n[me].weight = w;
          tasks.push_back(Task{a, b, depth});
          return me;
        }
        float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (size_t i = a; i < b; i++) {
          const Item& it = items[ids[i]];
          for (int k = 0; k < 3; k++) { const float c = it.lo[k] + it.hi[k]; clo[k] = std::fmin(clo[k], c); chi[k] = std::fmax(chi[k], c); }
        }
        // binned SAH, 16 bins per axis, the three axes in one pass over the items: cost = area(L) * weight(L) + area(R) * weight(R)
        constexpr int B = 16;
        float scale[3];
        for (int ax = 0; ax < 3; ax++) { const float ext = chi[ax] - clo[ax]; scale[ax] = ext > 0.f ? (float)B / ext : 0.f; }
        auto bin_of = [&](const Item& it, int ax) {
          const int q = (int)(((it.lo[ax] + it.hi[ax]) - clo[ax]) * scale[ax]);
          return q < 0 ? 0 : q >= B ? B - 1 : q;
        };
        float blo[3][B][3], bhi[3][B][3], bw[3][B];
        for (int ax = 0; ax < 3; ax++)
          for (int q = 0; q < B; q++) { bw[ax][q] = 0.f; for (int k = 0; k < 3; k++) { blo[ax][q][k] = INFINITY; bhi[ax][q][k] = -INFINITY; } }
        for (size_t i = a; i < b; i++) {
          const Item& it = items[ids[i]];
          for (int ax = 0; ax < 3; ax++) {
            if (scale[ax] == 0.f) continue;
            const int q = bin_of(it, ax);
            grow(blo[ax][q], bhi[ax][q], it.lo, it.hi);
            bw[ax][q] += it.weight;
          }
        }
        float best = INFINITY; int best_axis = -1, best_bin = 0;
        for (int ax = 0; ax < 3; ax++) {
          if (scale[ax] == 0.f) continue;
          float rarea[B], rw[B];
          float alo[3] = {INFINITY, INFINITY, INFINITY}, ahi[3] = {-INFINITY, -INFINITY, -INFINITY}, aw = 0.f;
          for (int q = B - 1; q >= 1; q--) {
            grow(alo, ahi, blo[ax][q], bhi[ax][q]);
            aw += bw[ax][q];
            rarea[q] = area(alo, ahi); rw[q] = aw;
          }
          float llo[3] = {INFINITY, INFINITY, INFINITY}, lhi[3] = {-INFINITY, -INFINITY, -INFINITY}, lw = 0.f;
          for (int q = 0; q < B - 1; q++) {
            grow(llo, lhi, blo[ax][q], bhi[ax][q]);
            lw += bw[ax][q];
            if (lw == 0.f || rw[q + 1] == 0.f) continue;
            const float cost = area(llo, lhi) * lw + rarea[q + 1] * rw[q + 1];
            if (cost < best) { best = cost; best_axis = ax; best_bin = q; }
          }
        }
        size_t mid;
        if (best_axis < 0 || depth > 256) {
          mid = a + (b - a) / 2;                                         // all centroids equal (or a degenerate chain): halves in list order
        } else {
          auto left_of = [&](uint32_t id) { return bin_of(items[id], best_axis) <= best_bin; };
          mid = (size_t)(std::stable_partition(ids.begin() + (std::ptrdiff_t)a, ids.begin() + (std::ptrdiff_t)b, left_of) - ids.begin());
          if (mid == a || mid == b) mid = a + (b - a) / 2;
        }
        float lo2[2][3], hi2[2][3];
        bounds(a, mid, lo2[0], hi2[0]);
        bounds(mid, b, lo2[1], hi2[1]);
        const int32_t l = build(t, a, mid, depth + 1, defer), r = build(t, mid, b, depth + 1, defer);
        TNode& nd = t.n[me];
        nd.left = l; nd.right = r;
        for (int c = 0; c < 2; c++) for (int k = 0; k < 3; k++) { nd.lo[c][k] = lo2[c][k]; nd.hi[c][k] = hi2[c][k]; }
        nd.records = 1 + t.n[l].records + t.n[r].records;
        nd.weight = t.n[l].weight + t.n[r].weight;
        return me;
      }
      // Tree rotations (host/scene.cpp Rotator; Kensler 2008): for a node with children A and B, B goes down into A in exchange
      // for one of A's children when that shrinks A's box most, or two grandchildren swap across; bottom-up, pass after pass.
      static float joined_area(const float* alo, const float* ahi, const float* blo, const float* bhi) {
        float lo[3], hi[3];
        for (int k = 0; k < 3; k++) { lo[k] = std::fmin(alo[k], blo[k]); hi[k] = std::fmax(ahi[k], bhi[k]); }
        return area(lo, hi);
      }
      static void refresh(Tree& t, int32_t nd) {                     // sums of an inner node after its children changed
        TNode& n = t.n[nd];
        n.records = 1 + t.n[n.left].records + t.n[n.right].records;
        n.weight = t.n[n.left].weight + t.n[n.right].weight;
      }
      static void set_box(TNode& n, int c, const TNode& child) {     // n's box for child c = union of that child's two boxes
        for (int k = 0; k < 3; k++) { n.lo[c][k] = std::fmin(child.lo[0][k], child.lo[1][k]); n.hi[c][k] = std::fmax(child.hi[0][k], child.hi[1][k]); }
      }
      double rotate(Tree& t, int32_t root) {
        double gain = 0;
        std::vector<int32_t> post, st{root};                        // post-order without recursion (chains can be deep)
        while (!st.empty()) {
          const int32_t i = st.back(); st.pop_back();
          if (t.n[i].left < 0) continue;
          post.push_back(i);
          st.push_back(t.n[i].left); st.push_back(t.n[i].right);
        }
        for (size_t k = post.size(); k-- > 0;) {
          TNode& n = t.n[post[k]];
          int32_t* ch[2] = {&n.left, &n.right};
          float best = 0.f; int bo = -1, bg = -1, xg = -1;
          for (int o = 0; o < 2; o++) {                              // child o is opened, the other child goes down into it
            const TNode& a = t.n[*ch[o]];
            if (a.left < 0) continue;
            for (int g = 0; g < 2; g++) {                            // a's child g comes up, a's child 1 - g stays
              const float delta = joined_area(a.lo[1 - g], a.hi[1 - g], n.lo[1 - o], n.hi[1 - o]) - area(n.lo[o], n.hi[o]);
              if (delta < best) { best = delta; bo = o; bg = g; }
            }
          }
          if (t.n[n.left].left >= 0 && t.n[n.right].left >= 0) {     // grandchildren across: left's child g with right's child 0
            const TNode &a = t.n[n.left], &c = t.n[n.right];
            for (int g = 0; g < 2; g++) {
              const float delta = joined_area(a.lo[1 - g], a.hi[1 - g], c.lo[0], c.hi[0]) + joined_area(c.lo[1], c.hi[1], a.lo[g], a.hi[g])
                                  - area(n.lo[0], n.hi[0]) - area(n.lo[1], n.hi[1]);
              if (delta < best) { best = delta; bo = -1; xg = g; }
            }
          }
          if (xg >= 0) {
            TNode &a = t.n[n.left], &c = t.n[n.right];
            int32_t& ai = xg == 0 ? a.left : a.right;
            std::swap(ai, c.left);
            for (int k2 = 0; k2 < 3; k2++) { std::swap(a.lo[xg][k2], c.lo[0][k2]); std::swap(a.hi[xg][k2], c.hi[0][k2]); }
            refresh(t, n.left); refresh(t, n.right);
            set_box(n, 0, a); set_box(n, 1, c);
            gain -= best;
          } else if (bo >= 0) {
            TNode& a = t.n[*ch[bo]];
            int32_t& up = bg == 0 ? a.left : a.right;
            std::swap(*ch[1 - bo], up);
            for (int k2 = 0; k2 < 3; k2++) { std::swap(n.lo[1 - bo][k2], a.lo[bg][k2]); std::swap(n.hi[1 - bo][k2], a.hi[bg][k2]); }
            refresh(t, *ch[bo]);
            set_box(n, bo, a);
            gain -= best;
          }
        }
        return gain;
      }
      void polish(Tree& t, int32_t root) {
        for (int p = 0; p < rotate_passes && t.n[root].left >= 0; p++)
          if (rotate(t, root) <= 0) break;
      }
      // records in pre-order (src/main.rs:203-231), the child with fewer shapes first (host/scene.cpp order_children)
      void emit(const Tree& t, int32_t nd, const float* lo, const float* hi, uint32_t pos, uint32_t exit, Records& out,
                std::vector<uint32_t>& cbase, std::vector<uint32_t>& cexit, std::vector<std::array<uint32_t, 2>>* task_place) const {
        struct F { int32_t nd; float lo[3], hi[3]; uint32_t pos, exit; };
        std::vector<F> st;
        F f0; f0.nd = nd; f0.pos = pos; f0.exit = exit;
        for (int k = 0; k < 3; k++) { f0.lo[k] = lo[k]; f0.hi[k] = hi[k]; }
        st.push_back(f0);
        while (!st.empty()) {
          const F f = st.back(); st.pop_back();
          const TNode& n = t.n[f.nd];
          if (n.sub >= 0) { (*task_place)[(size_t)n.sub] = {f.pos, f.exit}; continue; }
          if (n.item >= 0) {
            const Item& it = items[(size_t)n.item];
            if (it.shape == HJ_BVH_INNER) { cbase[it.cluster] = f.pos; cexit[it.cluster] = f.exit; continue; }
            hj_bvh_node rec;
            for (int k = 0; k < 3; k++) { rec.aabb_min[k] = it.lo[k]; rec.aabb_max[k] = it.hi[k]; }
            rec.shape_index = it.shape; rec.exit_index = f.exit;
            out.emplace_back(f.pos, rec);
            continue;
          }
          hj_bvh_node rec;
          for (int k = 0; k < 3; k++) { rec.aabb_min[k] = f.lo[k]; rec.aabb_max[k] = f.hi[k]; }
          rec.shape_index = HJ_BVH_INNER; rec.exit_index = f.exit;
          out.emplace_back(f.pos, rec);
          int first = 0;
          if (child_order != 0) {
            const float wl = t.n[n.left].weight, wr = t.n[n.right].weight;
            if (wr < wl || (wr == wl && area(n.lo[1], n.hi[1]) < area(n.lo[0], n.hi[0]))) first = 1;
          }
          const int32_t c0 = first == 0 ? n.left : n.right, c1 = first == 0 ? n.right : n.left;
          const uint32_t right_pos = f.pos + 1 + t.n[c0].records;
          F a, b2;
          a.nd = c0; a.pos = f.pos + 1; a.exit = right_pos;           // exit of a first child = its sibling
          b2.nd = c1; b2.pos = right_pos; b2.exit = f.exit;            // a second child inherits its parent's exit
          for (int k = 0; k < 3; k++) { a.lo[k] = n.lo[first][k]; a.hi[k] = n.hi[first][k]; b2.lo[k] = n.lo[1 - first][k]; b2.hi[k] = n.hi[1 - first][k]; }
          st.push_back(b2); st.push_back(a);
        }
      }
    } builder{items, {}, 0, tn.bvh_child_order, 0, {}, {}};
    // Rotation passes over the top (HJ_LBVH_TOP_ROTATE; -1 = the default rule): they pay where the top IS most of the tree - the
    // 6 k-triangle box: 112 items, frame rate 0.94 -> 0.99 of the host tree's - and cost 1.5 % (and 1.7 ms) on the 1 M-triangle
    // mesh, whose 25 k cluster boxes a binned SAH already arranges well: small tops only.
    {
      const int r = tn.lbvh_top_rotate;
      builder.rotate_passes = r >= 0 ? r : (items.size() < 4096 ? 8 : 0);
    }
    builder.ids.resize(items.size());
    for (size_t k = 0; k < items.size(); k++) builder.ids[k] = (uint32_t)k;
    // The top levels here, the subtrees below them on worker threads (disjoint ranges of ids[], nothing shared but read-only
    // data): build + rotation passes per subtree in parallel, then the rotation passes over the part above them (its leaves
    // are the finished subtrees), then the records.
    const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    builder.task_items = hw > 1 && items.size() >= 4096 ? items.size() / (4 * hw) : 0;
    Tree top;
    const int32_t top_root = builder.build(top, 0, items.size(), 0, builder.task_items != 0);
    builder.task_trees.resize(builder.tasks.size());
    if (!builder.tasks.empty()) {
      std::atomic<size_t> next{0};
      auto work = [&]() {
        for (size_t i; (i = next.fetch_add(1)) < builder.tasks.size();) {
          const Task& tk = builder.tasks[i];
          Tree& tt = builder.task_trees[i];
          const int32_t r = builder.build(tt, tk.a, tk.b, tk.depth, false);
          builder.polish(tt, r);
        }
      };
      std::vector<std::thread> pool;
      try {
        for (unsigned w = 1; w < hw; w++) pool.emplace_back(work);
      } catch (const std::exception&) {}                               // fewer threads than asked for: the rest is done here
      work();
      for (auto& th : pool) th.join();
    }
    builder.polish(top, top_root);
    float rlo[3], rhi[3];
    builder.bounds(0, items.size(), rlo, rhi);
    std::vector<std::array<uint32_t, 2>> place(builder.tasks.size());
    builder.emit(top, top_root, rlo, rhi, 0, root_exit, top_records, cbase, cexit, &place);
    for (size_t i = 0; i < builder.tasks.size(); i++) {
      const Task& tk = builder.tasks[i];
      float lo[3], hi[3];
      builder.bounds(tk.a, tk.b, lo, hi);
      builder.emit(builder.task_trees[i], 0, lo, hi, place[i][0], place[i][1], top_records, cbase, cexit, nullptr);
    }
  }
  mark("host SAH over the clusters");
  HJ_HIP(ctx, hipMemcpyAsync(const_cast<uint32_t*>(cl.base), cbase.data(), sizeof(uint32_t) * K, hipMemcpyHostToDevice, st));
  HJ_HIP(ctx, hipMemcpyAsync(const_cast<uint32_t*>(cl.exit), cexit.data(), sizeof(uint32_t) * K, hipMemcpyHostToDevice, st));
  if (sah_wave) {
    HJ_HIP(ctx, hipStreamWaitEvent(st, ev.b, 0));
    hipLaunchKernelGGL(hj::lbvh::k_place_clusters, dim3(K), dim3(64), 0, st, K, cl, d_staged, d_out);
  }
  else if (sah_clusters)
    hipLaunchKernelGGL(hj::lbvh::k_emit_clusters_sah, dim3((K + hj::lbvh::kSahThreads - 1) / hj::lbvh::kSahThreads),
                       dim3(hj::lbvh::kSahThreads), 0, st, t, K, cl, idx_mask, d_out, tn.bvh_child_order);
  else
    hipLaunchKernelGGL(hj::lbvh::k_emit_clusters, grid_sub, blk, 0, st, t, m, cl, idx_mask, d_out);
  HJ_HIP(ctx, hipGetLastError());
  mark("cluster subtrees");
  // ---- child order voted by a sample of the scene's own rays (kernels/hj_vote.h; HJ_LBVH_VOTE_PATHS camera paths, 0 = the order
  // by shape count the stages above produced): the host-built records join the others on the device first
  const size_t vote_paths = (size_t)tn.lbvh_vote_paths;
  hj_bvh_node* final_tree = nullptr;
  if (vote_paths != 0 && total >= 3) {
    hj_bvh_node* d_voted = nullptr;
    HJ_DEVBUF(d_voted, hj_bvh_node, total);
    rc = put_records(ctx, top_records, d_out);
    if (rc != HJ_OK) return rc;
    const VoteShapes vsh{sh.spheres, sh.quads, sh.triangles, sh.vertices};
    const std::string error_before = get_error(ctx);
    rc = vote_on_device(ctx, s, vsh, d_out, total, vote_paths, d_voted, timing, nullptr);
    if (rc != HJ_OK && rc != HJ_ERR_UNSUPPORTED) return rc;
    if (rc == HJ_ERR_UNSUPPORTED) put_error(ctx, error_before);      // the build succeeds without the vote: no stale message behind HJ_OK
    mark("ray-voted child order");
    // (a tree deeper than the exchange's level loop goes - thousands of shapes in a chain - keeps the order it has)
    final_tree = rc == HJ_OK ? d_voted : d_out;
    rc = HJ_OK;
  } else {
    rc = put_records(ctx, top_records, d_out);               // the host-built top joins the cluster subtrees on the device
    if (rc != HJ_OK) return rc;
    final_tree = d_out;
  }
  if (out_nodes) HJ_HIP(ctx, hipMemcpyAsync(out_nodes, final_tree, sizeof(hj_bvh_node) * total, hipMemcpyDeviceToHost, st));
  HJ_HIP(ctx, hipStreamSynchronize(st));
  mark("records to the host");
  // The tree and the shape arrays it was built over STAY on the device (hj_context::resident): hj_scene_upload with scene->bvh ==
  // NULL derives the kernels' records from them without a trip through the host; hj_bvh_device_read copies the tree out.
  {
    auto take = [&](void* p, DevBuf& into) {
      for (auto& b : bufs) if (b.p == p && p != nullptr) { into = b; b.p = nullptr; b.bytes = 0; return; }
    };
    hj_context::ResidentTree& rt = ctx->resident;
    take(final_tree, rt.nodes); take(keep_sp, rt.spheres); take(keep_qd, rt.quads); take(keep_tr, rt.triangles); take(keep_vx, rt.vertices);
    rt.total = total; rt.ns = s->num_spheres; rt.nq = s->num_quads; rt.nt = s->num_triangles; rt.nv = s->num_vertices;
    rt.shapes_hash = shape_arrays_hash(s);
    rt.valid = rt.nodes.p != nullptr;
  }
#undef HJ_DEVBUF
  if (out_num_nodes) *out_num_nodes = total;
  return HJ_OK;
}

// The tree hj_build_bvh_device left on the device, copied to the host (for a host that wants to keep or inspect it, and for the
// tests' oracle, which walks the same array).
int hj_bvh_device_read(hj_context* ctx, hj_bvh_node* out_nodes, size_t capacity, size_t* out_num_nodes) {
  if (!ctx) return HJ_ERR_INVALID;
  HJ_NOT_BUSY(ctx);
  HJ_NOT_PIPELINED(ctx);
  if (!ctx->resident.valid) return set_error(ctx, HJ_ERR_STATE, "hj_bvh_device_read: no tree on the device (hj_build_bvh_device builds one; hj_scene_upload with scene->bvh == NULL consumes it)");
  if (out_num_nodes) *out_num_nodes = ctx->resident.total;
  if (!out_nodes) return HJ_OK;
  if (capacity < ctx->resident.total) return set_error(ctx, HJ_ERR_INVALID, "node buffer holds %zu records, the tree has %zu", capacity, ctx->resident.total);
  HJ_HIP(ctx, hipSetDevice(ctx->device));
  HJ_HIP(ctx, hipMemcpy(out_nodes, ctx->resident.nodes.p, sizeof(hj_bvh_node) * ctx->resident.total, hipMemcpyDeviceToHost));
  return HJ_OK;
}

}  // extern "C"
