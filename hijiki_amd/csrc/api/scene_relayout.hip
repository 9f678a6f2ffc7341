// The re-layout of hj_scene_upload ON THE DEVICE (large trees): from the reference's skip-link array and triangle / vertex
// arrays, uploaded as they are, to what the kernels read (kernels/hj_device.h) - pre-gathered triangle records, collapsed
// tree, pair nodes, hot-first node order, sibling groups, explicit links.  Same derivation as the host code in
// scene_upload.hip (which stays for small trees, where it takes a millisecond, and for arrays that are not trees), step by step:
//
//   k_rl_triangles                 tri_isect / tri_shade, one thread per triangle
//   k_rl_init, k_rl_links          area, parent and parent count of every node: the device path wants a TREE (every node but
//                                  the root has exactly one parent, children lie behind their parent); anything else -> host path
//   k_rl_collapse_level  x depth   the collapse (an inner node over two inner nodes dropped when its area is > thr x its nearest
//                                  kept ancestor's): the ancestor's area comes down level by level, one launch per level
//   k_rl_pair_flags, scan, k_rl_pair_records    pair nodes numbered in array order, their 96-byte records
//   k_rl_kept + scan, radix sort   the kept nodes in array order; stable sort by area, descending: the first 512 are the hot ones
//   k_rl_preorder_flags + scan     small-tree order of the cold nodes: pre-order    (or)
//   k_rl_group_sizes x depth (bottom-up), k_rl_group_offsets x depth (top-down)    sibling groups: the cold children of a node
//                                  side by side, every group on an even index, the group of the larger child following directly
//                                  (the host's depth-first numbering, with every group padded to an even size so that a
//                                  subtree's size does not depend on where it starts)
//   k_rl_records                   the 32-byte device records with explicit links
//
// The image does not depend on the layout; tests render every scene kind through both paths (HJ_UPLOAD_DEVICE = 0 / 1).
#include "hj_internal.h"
#include "scene_relayout.hpp"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#pragma clang fp contract(off)

using namespace hjapi;

namespace {

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr int kMaxKids = 64;            // children of a kept node after the collapse (a collapsed binary tree: 2^levels); more -> host path

struct RL {
  const hj_bvh_node* bvh;
  uint32_t N, first_tri, nshapes;
  float thr;                            // collapse threshold
  float* sa;
  float* anc;
  uint32_t* parent;
  uint32_t* nparents;
  uint32_t* depth;
  uint8_t* del;                         // 1: no record (collapsed inner node, or a leaf folded into a pair)
  uint8_t* is_hot;
  uint32_t* pair_flag;                  // then (after the scan) the pair index of a pair node
  uint32_t* pair_idx;
  uint32_t* map;                        // device index of a kept node
  uint32_t* tsub;                       // slots of the sibling groups below a node
  uint32_t* goff;                       // where a node's group of children starts
  uint32_t* err;                        // [0] not a tree  [1] too many children  [2] levels that did something (collapse pass)
};

__device__ inline bool rl_inner(const RL& r, uint32_t i) { return r.bvh[i].shape_index == HJ_BVH_INNER; }
__device__ inline uint32_t rl_resolve(const RL& r, uint32_t i) { while (i < r.N && r.del[i]) i++; return i; }   // first kept node of a subtree

__global__ void k_rl_triangles(const hj_triangle* __restrict__ tris, const hj_vertex* __restrict__ verts, uint32_t nt,
                               float4* __restrict__ isect, float4* __restrict__ shade) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nt) return;
  const hj_vertex A = verts[tris[i].v[0]], B = verts[tris[i].v[1]], C = verts[tris[i].v[2]];
  isect[3 * (size_t)i + 0] = make_float4(A.pos[0], A.pos[1], A.pos[2], 0.f);
  isect[3 * (size_t)i + 1] = make_float4(B.pos[0] - A.pos[0], B.pos[1] - A.pos[1], B.pos[2] - A.pos[2], 0.f);
  isect[3 * (size_t)i + 2] = make_float4(C.pos[0] - A.pos[0], C.pos[1] - A.pos[1], C.pos[2] - A.pos[2], 0.f);
  shade[4 * (size_t)i + 0] = make_float4(A.normal[0], A.normal[1], A.normal[2], A.u);
  shade[4 * (size_t)i + 1] = make_float4(B.normal[0], B.normal[1], B.normal[2], B.u);
  shade[4 * (size_t)i + 2] = make_float4(C.normal[0], C.normal[1], C.normal[2], C.u);
  shade[4 * (size_t)i + 3] = make_float4(A.v, B.v, C.v, 0.f);
}

__global__ void k_rl_init(RL r) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= r.N) return;
  const hj_bvh_node nd = r.bvh[i];
  const float dx = nd.aabb_max[0] - nd.aabb_min[0], dy = nd.aabb_max[1] - nd.aabb_min[1], dz = nd.aabb_max[2] - nd.aabb_min[2];
  float a = (dx >= 0 && dy >= 0 && dz >= 0) ? dx * dy + dy * dz + dz * dx : 0.f;
  if (!(a == a)) a = 0.f;
  r.sa[i] = a;
  r.anc[i] = 0.f;
  r.parent[i] = kNone; r.nparents[i] = 0; r.depth[i] = i == 0 ? 0u : kNone;
  r.del[i] = 0; r.is_hot[i] = 0; r.pair_flag[i] = 0; r.map[i] = 0; r.tsub[i] = 0; r.goff[i] = 0;
}

__global__ void k_rl_links(RL r) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= r.N) return;
  if (r.bvh[i].exit_index <= i) { atomicOr(&r.err[0], 1u); return; }
  if (!rl_inner(r, i)) return;
  const uint32_t l = i + 1;
  if (l >= r.N) { atomicOr(&r.err[0], 1u); return; }
  const uint32_t rr = r.bvh[l].exit_index;
  if (rr >= r.N || rr <= l || r.bvh[rr].exit_index != r.bvh[i].exit_index) { atomicOr(&r.err[0], 1u); return; }   // (the right child's exit is its parent's)
  r.parent[l] = i; r.parent[rr] = i;
  atomicAdd(&r.nparents[l], 1u); atomicAdd(&r.nparents[rr], 1u);
}
__global__ void k_rl_check_tree(RL r) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= r.N) return;
  if (r.nparents[i] != (i == 0 ? 0u : 1u)) atomicOr(&r.err[0], 1u);
}

// One level of the collapse (scene_upload.hip: "Collapse"): nodes of depth L decide, and hand their children the area of their
// nearest kept ancestor and depth L + 1.
__global__ void k_rl_collapse_level(RL r, uint32_t L) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= r.N || r.depth[i] != L || !rl_inner(r, i)) return;
  // (branch-free on purpose: with the short-circuit form of this condition AMD clang 22 kept the zero high half of a 64-bit index
  // in a register it re-used for a load inside a narrower exec mask - lanes whose left child is a leaf then formed a wild address)
  const size_t l = (size_t)i + 1;
  const hj_bvh_node ni = r.bvh[i], nl = r.bvh[l];
  const size_t rr = nl.exit_index;
  const hj_bvh_node nr = r.bvh[rr];
  bool in = true;
  for (int k = 0; k < 3; k++)
    in = in & (nl.aabb_min[k] >= ni.aabb_min[k]) & (nl.aabb_max[k] <= ni.aabb_max[k]) & (nr.aabb_min[k] >= ni.aabb_min[k]) & (nr.aabb_max[k] <= ni.aabb_max[k]);
  const float a = r.anc[i], si = r.sa[i];
  const bool d = (i != 0) & (nl.shape_index == HJ_BVH_INNER) & (nr.shape_index == HJ_BVH_INNER) & (a > 0.f) & (si > r.thr * a) & in;
  r.del[i] = d ? 1 : 0;
  const float down = d ? a : si;
  r.anc[l] = down; r.anc[rr] = down;
  r.depth[l] = L + 1; r.depth[rr] = L + 1;
  r.err[2] = L + 1;                                 // (any writer: a level that had an inner node)
}

__global__ void k_rl_pair_flags(RL r) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= r.N) return;
  uint32_t f = 0;
  if (i + 2 < r.N && rl_inner(r, i)) {
    const uint32_t l = i + 1, rr = r.bvh[l].exit_index;
    if (rr == l + 1) {
      const uint32_t sl = r.bvh[l].shape_index, sr = r.bvh[rr].shape_index;
      if (sl != HJ_BVH_INNER && sr != HJ_BVH_INNER && sl >= r.first_tri && sr >= r.first_tri && r.bvh[rr].exit_index == r.bvh[i].exit_index) f = 1;
    }
  }
  r.pair_flag[i] = f;
}
__global__ void k_rl_pair_records(RL r, const float4* __restrict__ isect, float4* __restrict__ pairs) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= r.N || !r.pair_flag[i]) return;
  const uint32_t l = i + 1, rr = l + 1;
  const uint32_t sh[2] = {r.bvh[l].shape_index, r.bvh[rr].shape_index};
  float4* out = pairs + 6 * (size_t)r.pair_idx[i];
  for (int k = 0; k < 2; k++) {
    const size_t t = sh[k] - r.first_tri;
    float4 a = isect[3 * t];
    a.w = __uint_as_float(sh[k]);
    out[3 * k] = a; out[3 * k + 1] = isect[3 * t + 1]; out[3 * k + 2] = isect[3 * t + 2];
  }
  r.del[l] = 1; r.del[rr] = 1;                       // no records for the two leaves
}

// kept nodes, in array order: flags for the scan, then keys (area bits; areas are >= 0, so the bits order like the floats) and
// values (node index) for the stable descending sort
__global__ void k_rl_kept_flags(RL r, uint32_t* __restrict__ flag) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < r.N) flag[i] = r.del[i] ? 0u : 1u;
}
__global__ void k_rl_kept_list(RL r, const uint32_t* __restrict__ flag, const uint32_t* __restrict__ rank, uint32_t* __restrict__ keys,
                               uint32_t* __restrict__ vals) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= r.N || !flag[i]) return;
  keys[rank[i]] = __float_as_uint(r.sa[i]);
  vals[rank[i]] = i;
}
__global__ void k_rl_mark_hot(RL r, const uint32_t* __restrict__ sorted, uint32_t hot) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= hot) return;
  r.is_hot[sorted[k]] = 1;
  r.map[sorted[k]] = k;
}

// small trees: the cold kept nodes keep pre-order behind the hot ones
__global__ void k_rl_cold_flags(RL r, uint32_t* __restrict__ flag) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < r.N) flag[i] = (!r.del[i] && !r.is_hot[i]) ? 1u : 0u;
}
__global__ void k_rl_preorder_map(RL r, const uint32_t* __restrict__ flag, const uint32_t* __restrict__ rank, uint32_t hot) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < r.N && flag[i]) r.map[i] = hot + rank[i];
}

// The kept children of kept node i, in the walk's order (scene_upload.hip "node order").  Returns their number, or -1.
__device__ inline int rl_kids(const RL& r, uint32_t i, uint32_t* kids) {
  const uint32_t e0 = r.bvh[i].exit_index;
  const uint32_t end = e0 < r.N ? rl_resolve(r, e0) : r.N;
  int n = 0;
  for (uint32_t c = rl_resolve(r, i + 1); c < r.N && c != end;) {
    if (n == kMaxKids) return -1;
    kids[n++] = c;
    const uint32_t e = r.bvh[c].exit_index;
    c = e < r.N ? rl_resolve(r, e) : r.N;
  }
  return n;
}
__device__ inline bool rl_has_group(const RL& r, uint32_t i) { return rl_inner(r, i) && !r.pair_flag[i]; }   // a kept node whose children have records

// bottom-up: slots of a node's own group (its cold children, padded to an even number) + of the groups below its children
__global__ void k_rl_group_sizes(RL r, uint32_t L) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= r.N || r.depth[i] != L || r.del[i] || !rl_has_group(r, i)) return;
  uint32_t kids[kMaxKids];
  const int n = rl_kids(r, i, kids);
  if (n < 0) { atomicOr(&r.err[1], 1u); return; }
  uint32_t cold = 0, below = 0;
  for (int k = 0; k < n; k++) {
    cold += r.is_hot[kids[k]] ? 0u : 1u;
    if (rl_has_group(r, kids[k])) below += r.tsub[kids[k]];
  }
  r.tsub[i] = ((cold + 1u) & ~1u) + below;
}
// top-down: a node's cold children take the slots of its group in the walk's order; the groups of its children follow, the
// largest child's first
__global__ void k_rl_group_offsets(RL r, uint32_t L) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= r.N || r.depth[i] != L || r.del[i] || !rl_has_group(r, i)) return;
  uint32_t kids[kMaxKids];
  const int n = rl_kids(r, i, kids);
  if (n < 0) return;
  uint32_t next = r.goff[i], cold = 0;
  for (int k = 0; k < n; k++) if (!r.is_hot[kids[k]]) { r.map[kids[k]] = next + cold; cold++; }
  next += (cold + 1u) & ~1u;
  // stable insertion sort by area, descending
  for (int a = 1; a < n; a++) {
    const uint32_t x = kids[a];
    const float sx = r.sa[x];
    int b = a - 1;
    while (b >= 0 && r.sa[kids[b]] < sx) { kids[b + 1] = kids[b]; b--; }
    kids[b + 1] = x;
  }
  for (int k = 0; k < n; k++) if (rl_has_group(r, kids[k])) { r.goff[kids[k]] = next; next += r.tsub[kids[k]]; }
}

// the second copy of the tree (api/scene_upload.hip, kernels/hj_intersect.h general_position): the reference's own array, record i
// at base + i - nothing collapsed; pair nodes keep their mark, their two leaves' records are never reached
__global__ void k_rl_records2(RL r, float4* __restrict__ dev, uint32_t base) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= r.N) return;
  const hj_bvh_node nd = r.bvh[i];
  uint32_t a;
  if (nd.shape_index != HJ_BVH_INNER) a = nd.shape_index;
  else if (r.pair_flag[i]) a = hj::kInnerFlag | hj::kPairFlag | r.pair_idx[i];
  else a = hj::kInnerFlag | (i + 1 < r.N ? base + i + 1 : hj::kEndOfWalk);
  const uint32_t b = nd.exit_index < r.N ? base + nd.exit_index : hj::kEndOfWalk;
  float4* rec = dev + 2 * ((size_t)base + i);
  rec[0] = make_float4(nd.aabb_min[0], nd.aabb_min[1], nd.aabb_min[2], __uint_as_float(a));
  rec[1] = make_float4(nd.aabb_max[0], nd.aabb_max[1], nd.aabb_max[2], __uint_as_float(b));
}

__global__ void k_rl_records(RL r, float4* __restrict__ dev) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= r.N || r.del[i]) return;
  const hj_bvh_node nd = r.bvh[i];
  uint32_t a;
  if (nd.shape_index != HJ_BVH_INNER) a = nd.shape_index;
  else if (r.pair_flag[i]) a = hj::kInnerFlag | hj::kPairFlag | r.pair_idx[i];
  else {
    const uint32_t l = rl_resolve(r, i + 1);
    a = hj::kInnerFlag | (l < r.N ? r.map[l] : hj::kEndOfWalk);
  }
  const uint32_t e = nd.exit_index < r.N ? rl_resolve(r, nd.exit_index) : r.N;
  const uint32_t b = e < r.N ? r.map[e] : hj::kEndOfWalk;
  float4* rec = dev + 2 * (size_t)r.map[i];
  rec[0] = make_float4(nd.aabb_min[0], nd.aabb_min[1], nd.aabb_min[2], __uint_as_float(a));
  rec[1] = make_float4(nd.aabb_max[0], nd.aabb_max[1], nd.aabb_max[2], __uint_as_float(b));
}

}  // namespace

namespace hjapi {

int relayout_on_device(hj_context* ctx, const hj_scene_desc* s, const hj_triangle* d_tris, const hj_vertex* d_verts, bool pairs_on,
                       int node_order, float collapse_thr, bool timing, RelayoutOut& out, const hj_bvh_node* d_tree) {
  // d_tree: the skip-link array is on the device already (hj_build_bvh_device's tree: hj_context::resident); s->bvh is not read then
  out = RelayoutOut{};
  const size_t N = s->num_bvh_nodes;
  if (N < 3 || N >= 0x3FFFFFFFu) return HJ_ERR_UNSUPPORTED;
  hipStream_t st = ctx->stream;
  auto t_last = std::chrono::steady_clock::now();
  auto mark = [&](const char* what) {
    if (!timing) return;
    (void)hipStreamSynchronize(st);
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "hj_scene_upload (device): %-24s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  std::vector<DevBuf> tmp;                                  // released on return
  struct Release { std::vector<DevBuf>& b; ~Release() { for (auto& x : b) x.release(); } } release{tmp};
  tmp.reserve(32);
  int rc = HJ_OK;
  auto scratch = [&](size_t bytes, void** p) -> int {
    tmp.emplace_back();
    const int rc2 = dev_alloc(ctx, tmp.back(), std::max<size_t>(bytes, 16));
    *p = tmp.back().p;
    return rc2;
  };
#define HJ_TMP(ptr, type, count) do { void* p_ = nullptr; rc = scratch(sizeof(type) * (count), &p_); if (rc != HJ_OK) return rc; ptr = static_cast<type*>(p_); } while (0)
  auto keep = [&](size_t bytes, void** p) -> int {           // buffers that stay with the scene
    ctx->scene_bufs.emplace_back();
    const int rc2 = dev_alloc(ctx, ctx->scene_bufs.back(), std::max<size_t>(bytes, 16));
    *p = ctx->scene_bufs.back().p;
    return rc2;
  };
  const uint32_t n32 = (uint32_t)N, nt = (uint32_t)s->num_triangles;
  const dim3 blk(256), grid((n32 + 255) / 256);

  // triangle records
  float4 *isect = nullptr, *shade = nullptr;
  { void* p = nullptr; rc = keep(sizeof(float4) * 3 * (size_t)nt + 64, &p); if (rc != HJ_OK) return rc; isect = static_cast<float4*>(p); }
  { void* p = nullptr; rc = keep(sizeof(float4) * 4 * (size_t)nt + 64, &p); if (rc != HJ_OK) return rc; shade = static_cast<float4*>(p); }
  if (nt) hipLaunchKernelGGL(k_rl_triangles, dim3((nt + 255) / 256), blk, 0, st, d_tris, d_verts, nt, isect, shade);
  mark("triangle records");

  // the skip-link array as it is
  hj_bvh_node* d_bvh = const_cast<hj_bvh_node*>(d_tree);      // (the kernels only read it)
  if (!d_bvh) {
    HJ_TMP(d_bvh, hj_bvh_node, N);
    HJ_HIP(ctx, hipMemcpyAsync(d_bvh, s->bvh, sizeof(hj_bvh_node) * N, hipMemcpyHostToDevice, st));
  }
  RL r{};
  r.bvh = d_bvh; r.N = n32; r.first_tri = (uint32_t)(s->num_spheres + s->num_quads);
  r.nshapes = (uint32_t)(s->num_spheres + s->num_quads + s->num_triangles);
  r.thr = collapse_thr;
  HJ_TMP(r.sa, float, N); HJ_TMP(r.anc, float, N); HJ_TMP(r.parent, uint32_t, N); HJ_TMP(r.nparents, uint32_t, N);
  HJ_TMP(r.depth, uint32_t, N); HJ_TMP(r.del, uint8_t, N); HJ_TMP(r.is_hot, uint8_t, N); HJ_TMP(r.pair_flag, uint32_t, N);
  HJ_TMP(r.pair_idx, uint32_t, N + 1); HJ_TMP(r.map, uint32_t, N); HJ_TMP(r.tsub, uint32_t, N); HJ_TMP(r.goff, uint32_t, N);
  HJ_TMP(r.err, uint32_t, 4);
  uint32_t *flag = nullptr, *rank = nullptr, *keys = nullptr, *vals = nullptr, *keys2 = nullptr, *vals2 = nullptr;
  HJ_TMP(flag, uint32_t, N); HJ_TMP(rank, uint32_t, N + 1); HJ_TMP(keys, uint32_t, N); HJ_TMP(vals, uint32_t, N);
  HJ_TMP(keys2, uint32_t, N); HJ_TMP(vals2, uint32_t, N);
  HJ_HIP(ctx, hipMemsetAsync(r.err, 0, sizeof(uint32_t) * 4, st));
  size_t scan_bytes = 0, sort_bytes = 0;
  HJ_HIP(ctx, rocprim::exclusive_scan(nullptr, scan_bytes, flag, rank, 0u, N, rocprim::plus<uint32_t>(), st));
  HJ_HIP(ctx, rocprim::radix_sort_pairs_desc(nullptr, sort_bytes, keys, keys2, vals, vals2, N, 0, 32, st));
  void* prim_tmp = nullptr;
  HJ_TMP(prim_tmp, uint8_t, std::max(scan_bytes, sort_bytes));
  size_t prim_bytes = std::max(scan_bytes, sort_bytes);
  uint32_t h_err[4] = {0, 0, 0, 0};
  auto read_err = [&]() -> int {
    HJ_HIP(ctx, hipMemcpyAsync(h_err, r.err, sizeof h_err, hipMemcpyDeviceToHost, st));
    HJ_HIP(ctx, hipStreamSynchronize(st));
    return HJ_OK;
  };
  auto last_of_scan = [&](const uint32_t* fl, const uint32_t* rk, uint32_t& total) -> int {    // exclusive scan: total = rank[N-1] + flag[N-1]
    uint32_t a = 0, b = 0;
    HJ_HIP(ctx, hipMemcpyAsync(&a, rk + (N - 1), 4, hipMemcpyDeviceToHost, st));
    HJ_HIP(ctx, hipMemcpyAsync(&b, fl + (N - 1), 4, hipMemcpyDeviceToHost, st));
    HJ_HIP(ctx, hipStreamSynchronize(st));
    total = a + b;
    return HJ_OK;
  };

  const bool dbg = ctx->tuning.relayout_debug;
  auto check = [&](const char* what) {
    if (!dbg) return;
    const hipError_t e1 = hipStreamSynchronize(st), e2 = hipGetLastError();
    std::fprintf(stderr, "rl debug: %-20s sync %s, last %s\n", what, hipGetErrorString(e1), hipGetErrorString(e2));
  };
  if (dbg) std::fprintf(stderr, "rl debug: N %u bvh %p sa %p anc %p parent %p nparents %p depth %p del %p hot %p pf %p pi %p map %p tsub %p goff %p err %p\n", r.N,
                        (const void*)r.bvh, (void*)r.sa, (void*)r.anc, (void*)r.parent, (void*)r.nparents, (void*)r.depth, (void*)r.del, (void*)r.is_hot,
                        (void*)r.pair_flag, (void*)r.pair_idx, (void*)r.map, (void*)r.tsub, (void*)r.goff, (void*)r.err);
  check("uploads");
  hipLaunchKernelGGL(k_rl_init, grid, blk, 0, st, r);
  check("init");
  hipLaunchKernelGGL(k_rl_links, grid, blk, 0, st, r);
  check("links");
  hipLaunchKernelGGL(k_rl_check_tree, grid, blk, 0, st, r);
  check("check_tree");
  rc = read_err();
  if (rc != HJ_OK) return rc;
  if (h_err[0]) return HJ_ERR_UNSUPPORTED;                   // not a tree: the host path reasons about arbitrary arrays
  mark("node upload + tree check");

  // collapse, level by level (the levels end when one has no inner node)
  uint32_t levels = 0;
  for (uint32_t L = 0; L < 4096; L += 16) {
    for (uint32_t k = 0; k < 16; k++) { hipLaunchKernelGGL(k_rl_collapse_level, grid, blk, 0, st, r, L + k); if (dbg && L + k < 3) check("collapse level"); }
    rc = read_err();
    if (rc != HJ_OK) return rc;
    levels = h_err[2];                                       // deepest level + 1 that held an inner node
    if (levels < L + 16) break;
  }
  mark("collapse");

  // pair nodes
  uint32_t num_pairs = 0;
  float4* pairs = nullptr;
  if (pairs_on) {
    hipLaunchKernelGGL(k_rl_pair_flags, grid, blk, 0, st, r);
    HJ_HIP(ctx, rocprim::exclusive_scan(prim_tmp, prim_bytes, r.pair_flag, r.pair_idx, 0u, N, rocprim::plus<uint32_t>(), st));
    rc = last_of_scan(r.pair_flag, r.pair_idx, num_pairs);
    if (rc != HJ_OK) return rc;
  }
  { void* p = nullptr; rc = keep(sizeof(float4) * 6 * (size_t)num_pairs + 64, &p); if (rc != HJ_OK) return rc; pairs = static_cast<float4*>(p); }
  if (num_pairs) hipLaunchKernelGGL(k_rl_pair_records, grid, blk, 0, st, r, isect, pairs);
  mark("pair nodes");

  // kept nodes, hot-first
  uint32_t M = 0;
  hipLaunchKernelGGL(k_rl_kept_flags, grid, blk, 0, st, r, flag);
  HJ_HIP(ctx, rocprim::exclusive_scan(prim_tmp, prim_bytes, flag, rank, 0u, N, rocprim::plus<uint32_t>(), st));
  rc = last_of_scan(flag, rank, M);
  if (rc != HJ_OK) return rc;
  hipLaunchKernelGGL(k_rl_kept_list, grid, blk, 0, st, r, flag, rank, keys, vals);
  HJ_HIP(ctx, rocprim::radix_sort_pairs_desc(prim_tmp, prim_bytes, keys, keys2, vals, vals2, M, 0, 32, st));   // stable: equal areas keep array order
  const uint32_t hot = std::min<uint32_t>(hj::kHotNodes, M);
  hipLaunchKernelGGL(k_rl_mark_hot, dim3((hot + 255) / 256), blk, 0, st, r, vals2, hot);
  mark("hot-first sort");

  // order of the cold nodes
  uint32_t m_all = 0;
  if (node_order == 0 || (node_order < 0 && num_pairs == 0)) {    // small trees (cache-resident): pre-order; large ones: sibling groups
    uint32_t cold = 0;
    hipLaunchKernelGGL(k_rl_cold_flags, grid, blk, 0, st, r, flag);
    HJ_HIP(ctx, rocprim::exclusive_scan(prim_tmp, prim_bytes, flag, rank, 0u, N, rocprim::plus<uint32_t>(), st));
    rc = last_of_scan(flag, rank, cold);
    if (rc != HJ_OK) return rc;
    hipLaunchKernelGGL(k_rl_preorder_map, grid, blk, 0, st, r, flag, rank, hot);
    m_all = hot + cold;
  } else {
    for (uint32_t L = levels + 1; L-- > 0;) hipLaunchKernelGGL(k_rl_group_sizes, grid, blk, 0, st, r, L);
    // the root: a record of its own behind the hot ones when it is not one of them, then its group on the next even index
    uint8_t root_hot = 0;
    uint32_t t_root = 0;
    HJ_HIP(ctx, hipMemcpyAsync(&root_hot, r.is_hot, 1, hipMemcpyDeviceToHost, st));
    HJ_HIP(ctx, hipMemcpyAsync(&t_root, r.tsub, 4, hipMemcpyDeviceToHost, st));
    rc = read_err();
    if (rc != HJ_OK) return rc;
    if (h_err[1]) return HJ_ERR_UNSUPPORTED;                 // a node with more than kMaxKids kept children
    uint32_t base = (hot + 1u) & ~1u;
    if (!root_hot) { const uint32_t m0 = base; HJ_HIP(ctx, hipMemcpyAsync(r.map, &m0, 4, hipMemcpyHostToDevice, st)); base += 2; }
    HJ_HIP(ctx, hipMemcpyAsync(r.goff, &base, 4, hipMemcpyHostToDevice, st));
    HJ_HIP(ctx, hipStreamSynchronize(st));                   // (the two words above live on this stack frame)
    for (uint32_t L = 0; L <= levels; L++) hipLaunchKernelGGL(k_rl_group_offsets, grid, blk, 0, st, r, L);
    m_all = base + t_root;
  }
  mark("node order");

  // device records (zero-filled padding), placed so that the array does not cross a 4 GiB boundary (kernels/hj_walk.h)
  if ((size_t)m_all + N >= hj::kEndOfWalk) return set_error(ctx, HJ_ERR_UNSUPPORTED, "BVH of %zu records: too large", (size_t)m_all + N);
  const size_t rec_bytes = sizeof(float4) * 2 * ((size_t)m_all + N);      // the two copies of the tree
  const size_t bytes = std::max<size_t>(rec_bytes, 16) + 128;
  if (bytes >= (1ull << 32)) return set_error(ctx, HJ_ERR_UNSUPPORTED, "BVH of %u records: the device node array is limited to 4 GiB", M);
  ctx->scene_bufs.emplace_back();
  {
    DevBuf& b = ctx->scene_bufs.back();
    rc = dev_alloc(ctx, b, bytes);
    if (rc != HJ_OK) return rc;
    uintptr_t start = reinterpret_cast<uintptr_t>(b.p);
    if ((start >> 32) != ((start + bytes - 1) >> 32)) {
      b.release();
      rc = dev_alloc(ctx, b, 2 * bytes);
      if (rc != HJ_OK) return rc;
      start = reinterpret_cast<uintptr_t>(b.p);
      if ((start >> 32) != ((start + bytes - 1) >> 32)) start = ((start >> 32) + 1) << 32;
    }
    float4* dev = reinterpret_cast<float4*>(start);
    HJ_HIP(ctx, hipMemsetAsync(dev, 0, rec_bytes, st));
    hipLaunchKernelGGL(k_rl_records, grid, blk, 0, st, r, dev);
    hipLaunchKernelGGL(k_rl_records2, grid, blk, 0, st, r, dev, m_all);
    out.nodes = dev;
  }
  uint32_t root = 0;
  HJ_HIP(ctx, hipMemcpyAsync(&root, r.map, 4, hipMemcpyDeviceToHost, st));
  HJ_HIP(ctx, hipStreamSynchronize(st));
  HJ_HIP(ctx, hipGetLastError());
  mark("device records");
  out.tri_isect = isect; out.tri_shade = shade; out.tri_pair = pairs;
  out.num_nodes = m_all + n32; out.root = root; out.root2 = m_all; out.num_hot = hot; out.num_pairs = num_pairs; out.kept = M;
#undef HJ_TMP
  return HJ_OK;
}

}  // namespace hjapi
