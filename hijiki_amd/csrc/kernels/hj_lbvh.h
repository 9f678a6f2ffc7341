// Device-side BVH build: a Morton-code LBVH in the reference's node format.
//
// The reference builds its tree on the host with the `bvh` crate and flattens it depth-first with skip links
// (src/main.rs:199-231): one shape per leaf, pre-order numbering (left child = next record), every node stores
// the box its PARENT kept for it (= the bounds of its own subtree), exit = the record that follows its subtree
// (the root and every node on the right spine: 1 000 000).  This file produces the same FORMAT for a different
// topology: shapes sorted along a Morton curve of their centroids (as many bits per axis as a 64-bit key leaves beside
// the shape index: 14 for a million shapes), hierarchy by longest common prefix (Karras 2012, "Maximizing parallelism
// in the construction of BVHs, octrees and k-d trees"), bounds by a bottom-up pass.  LARGE shapes (box area above
// HJ_LBVH_BIG_PCT = 2 % of the area of the SCENE's box, e.g. the walls of the box around a mesh) are kept OUT of the Morton tree: sorted by centroid they
// would sit deep inside it and blow the boxes of all their ancestors up to scene size; they go into a small SAH tree
// that the host builds over them and over the CLUSTERS of the Morton tree (its subtrees of at most 64 leaves; api/lbvh_build.hip),
// so that only the lowest levels keep the Morton splits.  The image does not depend on the
// topology except through epsilon-ties.
//
// Pre-order without a traversal: a subtree over k leaves has 2k - 1 records, so for a node whose subtree covers
// the sorted leaves [first, first + k)
//     position = 2 * first + (number of LEFT turns on the way from the root to the node)
//     exit     = position + 2k - 1
// (the leaves to the left of the node fill 2 * first records minus one per subtree they form, and there is one
// such subtree per RIGHT turn; the node's proper ancestors add one record each).
#pragma once
#include "hj_device.h"

namespace hj {
namespace lbvh {

constexpr uint32_t kLeafBit = 0x80000000u;   // child reference: leaf (sorted position) or internal node index
constexpr uint32_t kNoParent = 0xFFFFFFFFu;

struct Shapes {               // the shape arrays of an hj_scene_desc, on the device
  const float4* spheres;      // hj_sphere
  const float4* quads;        // hj_quad as 3 x float4
  const hj_triangle* triangles;
  const hj_vertex* vertices;
  uint32_t ns, nq, nt;
};

struct Tree {                 // working arrays, n = number of shapes
  float4* leaf_lo;            // [n] bounds of shape i (global shape index), w unused
  float4* leaf_hi;
  int* bounds;                // [12] order-preserving ints: [0..5] bounds of the shapes' CENTROIDS (min xyz, max xyz: the Morton
                              //      grid), [6..11] bounds of the shapes' BOXES (the scene's box: the large-shape threshold)
  unsigned long long* keys;   // [n] large-shape flag (bit 63) | morton << idx_bits | shape index
  uint32_t* child;            // [2 * (n - 1)] left, right of internal node i
  uint32_t* first;            // [n - 1] first sorted leaf of internal node i
  uint32_t* count;            // [n - 1] leaves below internal node i
  uint32_t* parent;           // [2n - 1] parent of internal node i (index i) / of sorted leaf k (index n - 1 + k);
                              //          bit 31 set when the node is the LEFT child
  float4* node_lo;            // [n - 1] bounds of internal node i
  float4* node_hi;
  uint32_t* arrived;          // [n - 1] refit counters
};

// float <-> int that keeps the order (for atomicMin / atomicMax on floats)
HJ_DEV int ordered(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7FFFFFFF; }
HJ_DEV float unordered(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

__global__ void k_init_bounds(int* bounds) {
  if (threadIdx.x < 12) bounds[threadIdx.x] = (threadIdx.x % 6u) < 3u ? 0x7FFFFFFF : (int)0x80000000;
}

// Shape bounds as the host computes them (src/shape.rs:13-20,46-54, src/main.rs:74-79) + bounds of the centroids.
__global__ __launch_bounds__(256) void k_shape_boxes(Shapes s, Tree t, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  const bool valid = i < n;
  if (valid) {
    if (i < s.ns) {
      const float4 sp = s.spheres[i];
      const float c[3] = {sp.x, sp.y, sp.z};
      for (int k = 0; k < 3; k++) { lo[k] = c[k] - sp.w; hi[k] = c[k] + sp.w; }
      for (int k = 0; k < 3; k++) { const float a = f_min(lo[k], hi[k]), b = f_max(lo[k], hi[k]); lo[k] = a; hi[k] = b; }
    } else if (i < s.ns + s.nq) {
      const uint32_t q = i - s.ns;
      const float4 o = s.quads[3 * q], e1 = s.quads[3 * q + 1], e2 = s.quads[3 * q + 2];
      const float O[3] = {o.x, o.y, o.z}, A[3] = {e1.x, e1.y, e1.z}, B[3] = {e2.x, e2.y, e2.z};
      for (int k = 0; k < 3; k++) {
        const float p1 = O[k] + A[k], p2 = O[k] + B[k], p3 = (O[k] + A[k]) + B[k];
        lo[k] = f_min(f_min(O[k], p1), f_min(p2, p3));
        hi[k] = f_max(f_max(O[k], p1), f_max(p2, p3));
      }
    } else {
      const hj_triangle tr = s.triangles[i - s.ns - s.nq];
      const hj_vertex a = s.vertices[tr.v[0]], b = s.vertices[tr.v[1]], c = s.vertices[tr.v[2]];
      for (int k = 0; k < 3; k++) {
        lo[k] = f_min(f_min(a.pos[k], b.pos[k]), c.pos[k]);
        hi[k] = f_max(f_max(a.pos[k], b.pos[k]), c.pos[k]);
      }
    }
    t.leaf_lo[i] = make_float4(lo[0], lo[1], lo[2], 0.f);
    t.leaf_hi[i] = make_float4(hi[0], hi[1], hi[2], 0.f);
  }
  // centroid bounds: wave reduction, then one atomic pair per wave and axis
  for (int k = 0; k < 3; k++) {
    const float c = 0.5f * (lo[k] + hi[k]);
    const bool ok = valid && c == c;                       // NaN centroids do not take part
    int mn = ok ? ordered(c) : 0x7FFFFFFF, mx = ok ? ordered(c) : (int)0x80000000;
    for (int o = 32; o > 0; o >>= 1) {
      const int a = __shfl_xor(mn, o), b = __shfl_xor(mx, o);
      mn = a < mn ? a : mn;
      mx = b > mx ? b : mx;
    }
    if ((threadIdx.x & 63u) == 0) {
      atomicMin(&t.bounds[k], mn);
      atomicMax(&t.bounds[3 + k], mx);
    }
    // the scene's box (NaN or inverted boxes do not take part)
    const bool okb = valid && lo[k] <= hi[k];
    int bmn = okb ? ordered(lo[k]) : 0x7FFFFFFF, bmx = okb ? ordered(hi[k]) : (int)0x80000000;
    for (int o = 32; o > 0; o >>= 1) {
      const int a = __shfl_xor(bmn, o), b = __shfl_xor(bmx, o);
      bmn = a < bmn ? a : bmn;
      bmx = b > bmx ? b : bmx;
    }
    if ((threadIdx.x & 63u) == 0) {
      atomicMin(&t.bounds[6 + k], bmn);
      atomicMax(&t.bounds[9 + k], bmx);
    }
  }
}

HJ_DEV unsigned long long spread21(unsigned long long v) {   // 21 bits -> every third bit
  v &= 0x1FFFFFull;
  v = (v | (v << 32)) & 0x001F00000000FFFFull;
  v = (v | (v << 16)) & 0x001F0000FF0000FFull;
  v = (v | (v << 8)) & 0x100F00F00F00F00Full;
  v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}

constexpr unsigned long long kBigShape = 1ull << 63;

// idx_bits: bits of the shape index; axis_bits: Morton bits per axis (3 * axis_bits + idx_bits <= 63);
// big_frac: shapes whose box area exceeds big_frac x the area of the scene's box are flagged (and counted in *nbig).
__global__ __launch_bounds__(256) void k_morton_keys(Tree t, uint32_t n, uint32_t idx_bits, uint32_t axis_bits, float big_frac,
                                                     uint32_t* nbig) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 lo = t.leaf_lo[i], hi = t.leaf_hi[i];
  const float c[3] = {0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z)};
  unsigned long long q[3];
  const float scale = (float)(1u << axis_bits);
  for (int k = 0; k < 3; k++) {
    const float mn = unordered(t.bounds[k]), mx = unordered(t.bounds[3 + k]);
    const float e = mx - mn;
    float u = e > 0.f ? (c[k] - mn) / e : 0.f;
    u = u == u ? f_min(f_max(u, 0.f), 1.f) : 0.f;
    q[k] = (unsigned long long)f_min(u * scale, scale - 1.0f);
  }
  const unsigned long long code = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
  const float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
  float sx[3];
  for (int k = 0; k < 3; k++) { const float e = unordered(t.bounds[9 + k]) - unordered(t.bounds[6 + k]); sx[k] = e > 0.f ? e : 0.f; }
  const float area = dx * dy + dy * dz + dz * dx, scene = sx[0] * sx[1] + sx[1] * sx[2] + sx[2] * sx[0];
  const bool big = big_frac > 0.f && scene > 0.f && area > big_frac * scene;
  if (big) atomicAdd(nbig, 1u);
  t.keys[i] = (big ? kBigShape : 0ull) | (code << idx_bits) | i;        // the index makes every key unique
}

// Boxes (and, in lo.w, shape indices) of the nbig large shapes, which sort behind the m small ones: out[2k], out[2k + 1].
__global__ void k_gather_big(Tree t, uint32_t m, uint32_t nbig, unsigned long long idx_mask, float4* out) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nbig) return;
  const uint32_t shape = (uint32_t)(t.keys[m + k] & idx_mask);
  float4 lo = t.leaf_lo[shape];
  lo.w = __uint_as_float(shape);
  out[2 * k] = lo;
  out[2 * k + 1] = t.leaf_hi[shape];
}

// Karras 2012, section 4: internal node i of the radix tree over the sorted (unique) keys.
__global__ __launch_bounds__(256) void k_hierarchy(Tree t, uint32_t n) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  const int last_internal = (int)n - 2;
  if (i > last_internal) return;
  const unsigned long long* __restrict__ key = t.keys;
  auto delta = [&](int a, int b) -> int { return (b < 0 || b >= (int)n) ? -1 : __clzll((long long)(key[a] ^ key[b])); };
  const int d = delta(i, i + 1) - delta(i, i - 1) >= 0 ? 1 : -1;
  const int dmin = delta(i, i - d);
  int lmax = 2;
  while (delta(i, i + lmax * d) > dmin) lmax *= 2;
  int l = 0;
  for (int s = lmax / 2; s >= 1; s /= 2)
    if (delta(i, i + (l + s) * d) > dmin) l += s;
  const int j = i + l * d;
  const int dnode = delta(i, j);
  int sp = 0;
  for (int div = 2;; div *= 2) {                          // steps ceil(l/2), ceil(l/4), ... 1
    const int s = (l + div - 1) / div;
    if (delta(i, i + (sp + s) * d) > dnode) sp += s;
    if (s <= 1) break;
  }
  const int gamma = i + sp * d + (d < 0 ? -1 : 0);
  const int lo = i < j ? i : j, hi = i < j ? j : i;
  const uint32_t left = lo == gamma ? (kLeafBit | (uint32_t)gamma) : (uint32_t)gamma;
  const uint32_t right = hi == gamma + 1 ? (kLeafBit | (uint32_t)(gamma + 1)) : (uint32_t)(gamma + 1);
  t.child[2 * i] = left;
  t.child[2 * i + 1] = right;
  t.first[i] = (uint32_t)lo;
  t.count[i] = (uint32_t)(hi - lo + 1);
  t.parent[(left & kLeafBit) ? (n - 1) + (left & ~kLeafBit) : left] = (uint32_t)i | kLeafBit;   // bit 31: "I am the left child"
  t.parent[(right & kLeafBit) ? (n - 1) + (right & ~kLeafBit) : right] = (uint32_t)i;
  if (i == 0) t.parent[0] = kNoParent;
  t.arrived[i] = 0;
}

// A float4 written earlier in the same launch by a thread of another CU: the vector L1 is not coherent between CUs
// (a neighbouring record of the same line may sit there from before the write), so read at agent scope.
HJ_DEV float4 ld_agent(const float4* p) {
  const float* f = reinterpret_cast<const float*>(p);
  return make_float4(__hip_atomic_load(f + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                     __hip_atomic_load(f + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                     __hip_atomic_load(f + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), 0.f);
}

// Bounds bottom-up: the second thread to arrive at a node owns it (no waiting: the first one simply leaves).
__global__ __launch_bounds__(256) void k_refit(Tree t, uint32_t n, unsigned long long idx_mask) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  uint32_t p = t.parent[(n - 1) + k];
  while (p != kNoParent) {
    const uint32_t node = p & ~kLeafBit;
    __threadfence();                                         // my child's bounds are visible before I count myself in
    if (atomicAdd(&t.arrived[node], 1u) == 0u) return;
    __threadfence();
    float4 lo[2], hi[2];
    for (int c = 0; c < 2; c++) {
      const uint32_t ch = t.child[2 * node + c];
      if (ch & kLeafBit) {
        const uint32_t shape = (uint32_t)(t.keys[ch & ~kLeafBit] & idx_mask);
        lo[c] = t.leaf_lo[shape];
        hi[c] = t.leaf_hi[shape];
      } else {
        lo[c] = ld_agent(&t.node_lo[ch]);                      // written by another CU in THIS launch: not through the L1
        hi[c] = ld_agent(&t.node_hi[ch]);
      }
    }
    t.node_lo[node] = make_float4(f_min(lo[0].x, lo[1].x), f_min(lo[0].y, lo[1].y), f_min(lo[0].z, lo[1].z), 0.f);
    t.node_hi[node] = make_float4(f_max(hi[0].x, hi[1].x), f_max(hi[0].y, hi[1].y), f_max(hi[0].z, hi[1].z), 0.f);
    p = t.parent[node];
  }
}

// CLUSTERS: the Morton tree is cut into subtrees of at most `cmax` leaves (node ids: internal node i = i, sorted leaf k =
// n - 1 + k).  The host builds a SAH tree over the clusters' boxes (and the large shapes), so only the lowest levels keep
// the Morton splits (an HLBVH in the sense of Garanzha et al. 2011, with the host doing the top levels).
constexpr uint32_t kNoCluster = 0xFFFFFFFFu;
struct Clusters {
  uint32_t* count;        // [1] number of clusters
  uint32_t* slot_of;      // [2n - 1] cluster number of a node that is a cluster root, kNoCluster otherwise
  uint32_t* node;         // [n] root node id of cluster k
  float4* lo;             // [n] box of cluster k, w = bits of its first sorted leaf
  float4* hi;             //                        w = bits of its leaf count
  const uint32_t* base;   // [K] position of the cluster's first record in the output (from the host)
  const uint32_t* exit;   // [K] exit of the cluster's right spine
};

// `scan_boxes`: the internal nodes have no boxes (k_refit was skipped: k_emit_clusters_sah computes its own), so a cluster
// root unites the boxes of its (at most cmax) leaves itself.
__global__ __launch_bounds__(256) void k_mark_clusters(Tree t, uint32_t n, uint32_t cmax, unsigned long long idx_mask, Clusters c,
                                                       bool scan_boxes) {
  const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= 2 * n - 1) return;
  const bool leaf = id >= n - 1;
  const uint32_t cnt = leaf ? 1u : t.count[id];
  const uint32_t p = t.parent[id];
  const uint32_t pcnt = p == kNoParent ? 0xFFFFFFFFu : t.count[p & ~kLeafBit];
  uint32_t slot = kNoCluster;
  if (cnt <= cmax && pcnt > cmax) {
    slot = atomicAdd(c.count, 1u);
    c.node[slot] = id;
    float4 lo, hi;
    uint32_t first;
    if (leaf) {
      const uint32_t shape = (uint32_t)(t.keys[id - (n - 1)] & idx_mask);
      lo = t.leaf_lo[shape]; hi = t.leaf_hi[shape]; first = id - (n - 1);
    } else if (scan_boxes) {
      first = t.first[id];
      lo = make_float4(INFINITY, INFINITY, INFINITY, 0.f); hi = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.f);
      for (uint32_t i = 0; i < cnt; i++) {
        const uint32_t shape = (uint32_t)(t.keys[first + i] & idx_mask);
        const float4 a = t.leaf_lo[shape], b = t.leaf_hi[shape];
        lo.x = f_min(lo.x, a.x); lo.y = f_min(lo.y, a.y); lo.z = f_min(lo.z, a.z);
        hi.x = f_max(hi.x, b.x); hi.y = f_max(hi.y, b.y); hi.z = f_max(hi.z, b.z);
      }
    } else {
      lo = t.node_lo[id]; hi = t.node_hi[id]; first = t.first[id];
    }
    lo.w = __uint_as_float(first); hi.w = __uint_as_float(cnt);
    c.lo[slot] = lo; c.hi[slot] = hi;
  }
  c.slot_of[id] = slot;
}

// Records of the nodes INSIDE the clusters (the nodes above the cluster roots are replaced by the host's tree):
// position = base of the cluster + the pre-order position inside the cluster's subtree (see the top of this file).
__global__ __launch_bounds__(256) void k_emit_clusters(Tree t, uint32_t n, Clusters c, unsigned long long idx_mask, hj_bvh_node* out) {
  const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= 2 * n - 1) return;
  uint32_t root = id, turns = 0;
  while (c.slot_of[root] == kNoCluster) {
    const uint32_t p = t.parent[root];
    if (p == kNoParent) return;                               // above the clusters
    turns += p >> 31;
    root = p & ~kLeafBit;
  }
  const uint32_t slot = c.slot_of[root];
  const uint32_t rfirst = __float_as_uint(c.lo[slot].w), rcnt = __float_as_uint(c.hi[slot].w);
  const bool leaf = id >= n - 1;
  const uint32_t k = leaf ? id - (n - 1) : 0;
  const uint32_t first = leaf ? k : t.first[id], cnt = leaf ? 1u : t.count[id];
  const uint32_t base = c.base[slot];
  const uint32_t pos = base + 2 * (first - rfirst) + turns, end = pos + 2 * cnt - 1;
  float4 lo, hi;
  uint32_t shape = HJ_BVH_INNER;
  if (leaf) {
    shape = (uint32_t)(t.keys[k] & idx_mask);
    lo = t.leaf_lo[shape]; hi = t.leaf_hi[shape];
  } else {
    lo = t.node_lo[id]; hi = t.node_hi[id];
  }
  hj_bvh_node nd;
  nd.aabb_min[0] = lo.x; nd.aabb_min[1] = lo.y; nd.aabb_min[2] = lo.z;
  nd.shape_index = shape;
  nd.aabb_max[0] = hi.x; nd.aabb_max[1] = hi.y; nd.aabb_max[2] = hi.z;
  nd.exit_index = end >= base + 2 * rcnt - 1 ? c.exit[slot] : end;
  out[pos] = nd;
}

// The records of ONE cluster per thread, with the cluster's leaves re-split top-down by binned SAH (8 bins x 3 axes; the
// Morton splits inside a cluster are what is left of the LBVH's quality gap once the host has built the top by SAH).  The
// subtree is emitted directly in pre-order: a range of m leaves at position p takes records [p, p + 2m - 1), its left part
// starts at p + 1, its right part where the left one ends, exits as in src/main.rs:214-231.  Clusters hold at most
// kClusterMax leaves, so everything fits in per-thread arrays.
constexpr uint32_t kClusterMax = 64;
constexpr uint32_t kSahThreads = 16;                               // clusters per workgroup: 25 KB of LDS, many small groups
__global__ __launch_bounds__(kSahThreads) void k_emit_clusters_sah(Tree t, uint32_t K, Clusters c, unsigned long long idx_mask,
                                                                   hj_bvh_node* out, int child_order) {
  // the cluster's leaf boxes, staged once ([.][leaf][thread]: a thread's walk over its leaves stays in its own banks), and
  // the order of the leaves, permuted in place by the splits
  __shared__ float s_box[6][kClusterMax][kSahThreads];
  __shared__ uint8_t s_ids[kClusterMax][kSahThreads];
  __shared__ float s_bin[8][7][kSahThreads];                       // per bin: box and count (dynamically indexed: not registers)
  const uint32_t k = blockIdx.x * kSahThreads + threadIdx.x, tid = threadIdx.x;
  if (k >= K) return;
  const uint32_t first = __float_as_uint(c.lo[k].w), cnt = __float_as_uint(c.hi[k].w);
  for (uint32_t i = 0; i < cnt; i++) {
    const uint32_t id = (uint32_t)(t.keys[first + i] & idx_mask);
    const float4 a = t.leaf_lo[id], b = t.leaf_hi[id];
    s_box[0][i][tid] = a.x; s_box[1][i][tid] = a.y; s_box[2][i][tid] = a.z;
    s_box[3][i][tid] = b.x; s_box[4][i][tid] = b.y; s_box[5][i][tid] = b.z;
    s_ids[i][tid] = (uint8_t)i;
  }
  struct Range { uint32_t lo, hi, pos, exit; };
  Range stack[kClusterMax];
  uint32_t sp = 0;
  stack[sp++] = Range{0u, cnt, c.base[k], c.exit[k]};
  while (sp != 0) {
    const Range r = stack[--sp];
    const uint32_t m = r.hi - r.lo;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = r.lo; i < r.hi; i++) {
      const uint32_t li = s_ids[i][tid];
      for (int ax = 0; ax < 3; ax++) {
        const float al = s_box[ax][li][tid], bh = s_box[3 + ax][li][tid], cc = al + bh;
        lo[ax] = f_min(lo[ax], al); hi[ax] = f_max(hi[ax], bh);
        clo[ax] = f_min(clo[ax], cc); chi[ax] = f_max(chi[ax], cc);
      }
    }
    hj_bvh_node nd;
    nd.aabb_min[0] = lo[0]; nd.aabb_min[1] = lo[1]; nd.aabb_min[2] = lo[2];
    nd.aabb_max[0] = hi[0]; nd.aabb_max[1] = hi[1]; nd.aabb_max[2] = hi[2];
    nd.exit_index = r.exit;
    if (m == 1) {
      nd.shape_index = (uint32_t)(t.keys[first + s_ids[r.lo][tid]] & idx_mask);
      out[r.pos] = nd;
      continue;
    }
    nd.shape_index = HJ_BVH_INNER;
    out[r.pos] = nd;
    // binned SAH: cost = area(L) * |L| + area(R) * |R|
    constexpr int B = 8;
    float best = INFINITY;
    int best_axis = -1, best_bin = 0;
    for (int ax = 0; ax < 3; ax++) {
      const float ext = chi[ax] - clo[ax];
      if (!(ext > 0.f)) continue;
      for (int q = 0; q < B; q++) { s_bin[q][6][tid] = 0.f; for (int d = 0; d < 3; d++) { s_bin[q][d][tid] = INFINITY; s_bin[q][3 + d][tid] = -INFINITY; } }
      for (uint32_t i = r.lo; i < r.hi; i++) {
        const uint32_t li = s_ids[i][tid];
        int q = (int)(((s_box[ax][li][tid] + s_box[3 + ax][li][tid]) - clo[ax]) / ext * (float)B);
        q = q < 0 ? 0 : q >= B ? B - 1 : q;
        s_bin[q][6][tid] += 1.f;
        for (int d = 0; d < 3; d++) {
          s_bin[q][d][tid] = f_min(s_bin[q][d][tid], s_box[d][li][tid]);
          s_bin[q][3 + d][tid] = f_max(s_bin[q][3 + d][tid], s_box[3 + d][li][tid]);
        }
      }
      float rarea[B];
      uint32_t rn[B];
      {
        float alo[3] = {INFINITY, INFINITY, INFINITY}, ahi[3] = {-INFINITY, -INFINITY, -INFINITY};
        uint32_t an = 0;
        for (int q = B - 1; q >= 1; q--) {
          for (int d = 0; d < 3; d++) { alo[d] = f_min(alo[d], s_bin[q][d][tid]); ahi[d] = f_max(ahi[d], s_bin[q][3 + d][tid]); }
          an += (uint32_t)s_bin[q][6][tid];
          const float dx = ahi[0] - alo[0], dy = ahi[1] - alo[1], dz = ahi[2] - alo[2];
          rarea[q] = an ? dx * dy + dy * dz + dz * dx : 0.f;
          rn[q] = an;
        }
      }
      float llo[3] = {INFINITY, INFINITY, INFINITY}, lhi[3] = {-INFINITY, -INFINITY, -INFINITY};
      uint32_t ln = 0;
      for (int q = 0; q < B - 1; q++) {
        for (int d = 0; d < 3; d++) { llo[d] = f_min(llo[d], s_bin[q][d][tid]); lhi[d] = f_max(lhi[d], s_bin[q][3 + d][tid]); }
        ln += (uint32_t)s_bin[q][6][tid];
        if (ln == 0 || rn[q + 1] == 0) continue;
        const float dx = lhi[0] - llo[0], dy = lhi[1] - llo[1], dz = lhi[2] - llo[2];
        const float cost = (dx * dy + dy * dz + dz * dx) * (float)ln + rarea[q + 1] * (float)rn[q + 1];
        if (cost < best) { best = cost; best_axis = ax; best_bin = q; }
      }
    }
    uint32_t mid = r.lo + m / 2;                                   // all centroids equal: halves in order
    if (best_axis >= 0) {
      const float ext = chi[best_axis] - clo[best_axis];
      uint32_t w = r.lo;                                           // the order inside a side is free
      for (uint32_t i = r.lo; i < r.hi; i++) {
        const uint32_t li = s_ids[i][tid];
        int q = (int)(((s_box[best_axis][li][tid] + s_box[3 + best_axis][li][tid]) - clo[best_axis]) / ext * (float)B);
        q = q < 0 ? 0 : q >= B ? B - 1 : q;
        if (q <= best_bin) { s_ids[i][tid] = s_ids[w][tid]; s_ids[w][tid] = (uint8_t)li; w++; }
      }
      if (w > r.lo && w < r.hi) mid = w;
    }
    if (child_order != 0 && r.hi - mid < mid - r.lo) {
      // the side with fewer leaves first (host/scene.cpp order_children): swap the two blocks of the id list (three reversals)
      auto reverse = [&](uint32_t a, uint32_t b) {
        for (; a + 1 < b; a++, b--) { const uint8_t x = s_ids[a][tid]; s_ids[a][tid] = s_ids[b - 1][tid]; s_ids[b - 1][tid] = x; }
      };
      reverse(r.lo, r.hi); 
      mid = r.lo + (r.hi - mid);
      reverse(r.lo, mid); reverse(mid, r.hi);
    }
    const uint32_t left_pos = r.pos + 1, right_pos = left_pos + 2 * (mid - r.lo) - 1;
    stack[sp++] = Range{mid, r.hi, right_pos, r.exit};             // a right child inherits its parent's exit
    stack[sp++] = Range{r.lo, mid, left_pos, right_pos};           // exit of a left child = its sibling
  }
}

// The same re-split with ONE WAVE per cluster (round 4): the leaf boxes of a cluster of up to kWaveClusterMax leaves sit in
// LDS, the 64 lanes stride over a range's leaves for its bounds and its 3 x 8 bins (LDS atomics on order-preserving ints:
// independent of the order of arrival), lanes 0..20 price the 21 split candidates, a ballot partition (stable) re-orders the
// range's ids.  Same bins, same cost, same tie-break (lowest axis, lowest bin) and same child order as k_emit_clusters_sah;
// larger clusters mean fewer Morton borders inside the tree and fewer items for the host's SAH top (1 M triangles: ~3 k
// clusters of up to 512 leaves instead of 25 k of up to 64).
constexpr uint32_t kWaveClusterMax = 512;
constexpr uint32_t kOpenExit = 0xFFFFFFFFu;
struct SahRange { uint32_t lo, hi, pos, exit; };
__global__ __launch_bounds__(64) void k_emit_clusters_sah_wave(Tree t, uint32_t K, Clusters c, unsigned long long idx_mask,
                                                               hj_bvh_node* out, int child_order) {
  __shared__ float s_box[6][kWaveClusterMax];
  __shared__ uint16_t s_ids[kWaveClusterMax], s_tmp[kWaveClusterMax];
  __shared__ int s_bin[3][8][7];                                     // per axis and bin: ordered(min xyz), ordered(max xyz), count
  __shared__ SahRange s_stack[kWaveClusterMax];
  const uint32_t k = blockIdx.x, lane = threadIdx.x;
  if (k >= K) return;
  const uint32_t first = __float_as_uint(c.lo[k].w), cnt = __float_as_uint(c.hi[k].w);
  for (uint32_t i = lane; i < cnt; i += 64u) {
    const uint32_t id = (uint32_t)(t.keys[first + i] & idx_mask);
    const float4 a = t.leaf_lo[id], b = t.leaf_hi[id];
    s_box[0][i] = a.x; s_box[1][i] = a.y; s_box[2][i] = a.z;
    s_box[3][i] = b.x; s_box[4][i] = b.y; s_box[5][i] = b.z;
    s_ids[i] = (uint16_t)i;
  }
  // `out` is a STAGING array: the cluster's records go to [2 * first, 2 * first + 2 * cnt - 1) with the exits of its right
  // spine left open (kOpenExit) - the cluster's place in the tree is not known yet, the host builds the top of the tree while
  // this kernel runs; k_place_clusters moves the records to their place afterwards.
  if (lane == 0) s_stack[0] = SahRange{0u, cnt, 2u * first, kOpenExit};
  __syncthreads();                                                   // (one wave: a fence for the LDS writes above)
  uint32_t sp = 1;
  auto wave_min = [](float v) { for (int o = 32; o > 0; o >>= 1) v = f_min(v, __shfl_xor(v, o)); return v; };
  auto wave_max = [](float v) { for (int o = 32; o > 0; o >>= 1) v = f_max(v, __shfl_xor(v, o)); return v; };
  auto shape_of = [&](uint32_t pos) { return (uint32_t)(t.keys[first + s_ids[pos]] & idx_mask); };
  auto leaf_record = [&](uint32_t pos, uint32_t at, uint32_t exit) {     // (one lane)
    const uint32_t li = s_ids[pos];
    hj_bvh_node nd;
    nd.aabb_min[0] = s_box[0][li]; nd.aabb_min[1] = s_box[1][li]; nd.aabb_min[2] = s_box[2][li];
    nd.aabb_max[0] = s_box[3][li]; nd.aabb_max[1] = s_box[4][li]; nd.aabb_max[2] = s_box[5][li];
    nd.shape_index = shape_of(pos); nd.exit_index = exit;
    out[at] = nd;
  };
  while (sp != 0) {
    sp--;
    const SahRange r = s_stack[sp];                                  // (every lane reads the same entry: a broadcast)
    const uint32_t m = r.hi - r.lo;
    if (m == 1) {
      if (lane == 0) leaf_record(r.lo, r.pos, r.exit);
      continue;
    }
    // bounds of the range and of its centroids (sums lo + hi, as in the one-thread kernel)
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = r.lo + lane; i < r.hi; i += 64u) {
      const uint32_t li = s_ids[i];
      for (int ax = 0; ax < 3; ax++) {
        const float al = s_box[ax][li], bh = s_box[3 + ax][li], cc = al + bh;
        lo[ax] = f_min(lo[ax], al); hi[ax] = f_max(hi[ax], bh);
        clo[ax] = f_min(clo[ax], cc); chi[ax] = f_max(chi[ax], cc);
      }
    }
    for (int ax = 0; ax < 3; ax++) { lo[ax] = wave_min(lo[ax]); hi[ax] = wave_max(hi[ax]); clo[ax] = wave_min(clo[ax]); chi[ax] = wave_max(chi[ax]); }
    if (lane == 0) {
      hj_bvh_node nd;
      nd.aabb_min[0] = lo[0]; nd.aabb_min[1] = lo[1]; nd.aabb_min[2] = lo[2];
      nd.aabb_max[0] = hi[0]; nd.aabb_max[1] = hi[1]; nd.aabb_max[2] = hi[2];
      nd.shape_index = HJ_BVH_INNER; nd.exit_index = r.exit;
      out[r.pos] = nd;
    }
    if (m == 2) {                                                    // two leaves: nothing to choose (equal counts keep their order)
      if (lane == 0) { leaf_record(r.lo, r.pos + 1, r.pos + 2); leaf_record(r.lo + 1, r.pos + 2, r.exit); }
      continue;
    }
    // bins
    constexpr int B = 8;
    for (uint32_t e = lane; e < 3u * B * 7u; e += 64u) {
      const uint32_t f = e % 7u;
      (&s_bin[0][0][0])[e] = f < 3u ? 0x7FFFFFFF : f < 6u ? (int)0x80000000 : 0;
    }
    __syncthreads();
    float ext[3];
    for (int ax = 0; ax < 3; ax++) ext[ax] = chi[ax] - clo[ax];
    auto bin_of = [&](uint32_t li, int ax) {
      int q = (int)(((s_box[ax][li] + s_box[3 + ax][li]) - clo[ax]) / ext[ax] * (float)B);
      return q < 0 ? 0 : q >= B ? B - 1 : q;
    };
    for (uint32_t i = r.lo + lane; i < r.hi; i += 64u) {
      const uint32_t li = s_ids[i];
      for (int ax = 0; ax < 3; ax++) {
        if (!(ext[ax] > 0.f)) continue;
        const int q = bin_of(li, ax);
        for (int d = 0; d < 3; d++) {
          atomicMin(&s_bin[ax][q][d], ordered(s_box[d][li]));
          atomicMax(&s_bin[ax][q][3 + d], ordered(s_box[3 + d][li]));
        }
        atomicAdd(&s_bin[ax][q][6], 1);
      }
    }
    __syncthreads();
    // the 21 candidates: lane j = 7 * axis + bin prices "bins 0..bin to the left": cost = area(L) * |L| + area(R) * |R|
    float cost = INFINITY;
    if (lane < 21u) {
      const int ax = (int)lane / 7, q = (int)lane % 7;
      if (ext[ax] > 0.f) {
        float llo[3] = {INFINITY, INFINITY, INFINITY}, lhi[3] = {-INFINITY, -INFINITY, -INFINITY};
        float rlo[3] = {INFINITY, INFINITY, INFINITY}, rhi[3] = {-INFINITY, -INFINITY, -INFINITY};
        uint32_t ln = 0, rn = 0;
        for (int b = 0; b < B; b++) {
          const uint32_t nb = (uint32_t)s_bin[ax][b][6];
          if (nb == 0) continue;
          float* tlo = b <= q ? llo : rlo;
          float* thi = b <= q ? lhi : rhi;
          for (int d = 0; d < 3; d++) { tlo[d] = f_min(tlo[d], unordered(s_bin[ax][b][d])); thi[d] = f_max(thi[d], unordered(s_bin[ax][b][3 + d])); }
          if (b <= q) ln += nb; else rn += nb;
        }
        if (ln != 0 && rn != 0) {
          const float dx = lhi[0] - llo[0], dy = lhi[1] - llo[1], dz = lhi[2] - llo[2];
          const float ex = rhi[0] - rlo[0], ey = rhi[1] - rlo[1], ez = rhi[2] - rlo[2];
          cost = (dx * dy + dy * dz + dz * dx) * (float)ln + (ex * ey + ey * ez + ez * ex) * (float)rn;
        }
      }
    }
    // the cheapest candidate, the lowest lane among equals (the one-thread kernel's loop order with its strict <)
    float bc = cost;
    uint32_t bj = lane;
    for (int o = 32; o > 0; o >>= 1) {
      const float oc = __shfl_xor(bc, o);
      const uint32_t oj = (uint32_t)__shfl_xor((int)bj, o);
      if (oc < bc || (oc == bc && oj < bj)) { bc = oc; bj = oj; }
    }
    uint32_t mid = r.lo + m / 2;                                     // all centroids equal: halves in order
    if (bc < INFINITY) {
      const int best_axis = (int)bj / 7, best_bin = (int)bj % 7;
      // stable partition of the ids: left part to s_tmp[r.lo ...], right part behind it
      uint32_t nleft = 0;
      for (uint32_t i0 = r.lo; i0 < r.hi; i0 += 64u) {
        const uint32_t i = i0 + lane;
        const bool left = i < r.hi && bin_of(s_ids[i], best_axis) <= best_bin;
        nleft += (uint32_t)__popcll(__ballot(left));
      }
      uint32_t wl = r.lo, wr = r.lo + nleft;
      for (uint32_t i0 = r.lo; i0 < r.hi; i0 += 64u) {
        const uint32_t i = i0 + lane;
        const bool in = i < r.hi;
        const uint32_t li = in ? s_ids[i] : 0u;
        const bool left = in && bin_of(li, best_axis) <= best_bin, right = in && !left;
        const unsigned long long ml = __ballot(left), mr = __ballot(right), below = (1ull << lane) - 1ull;
        if (left) s_tmp[wl + (uint32_t)__popcll(ml & below)] = (uint16_t)li;
        if (right) s_tmp[wr + (uint32_t)__popcll(mr & below)] = (uint16_t)li;
        wl += (uint32_t)__popcll(ml); wr += (uint32_t)__popcll(mr);
      }
      __syncthreads();
      if (nleft != 0 && nleft != m) {
        mid = r.lo + nleft;
        for (uint32_t i = r.lo + lane; i < r.hi; i += 64u) s_ids[i] = s_tmp[i];
        __syncthreads();
      }
    }
    if (child_order != 0 && r.hi - mid < mid - r.lo) {
      // the side with fewer leaves first (host/scene.cpp order_children): the two blocks of the id list change places
      const uint32_t nl = mid - r.lo, nr = r.hi - mid;
      for (uint32_t i = r.lo + lane; i < r.hi; i += 64u) s_tmp[i] = s_ids[i];
      __syncthreads();
      for (uint32_t i = lane; i < m; i += 64u) s_ids[r.lo + i] = i < nr ? s_tmp[mid + i] : s_tmp[r.lo + (i - nr)];
      __syncthreads();
      mid = r.lo + nr;
      (void)nl;
    }
    const uint32_t left_pos = r.pos + 1, right_pos = left_pos + 2 * (mid - r.lo) - 1;
    if (lane == 0) {
      s_stack[sp] = SahRange{mid, r.hi, right_pos, r.exit};          // a right child inherits its parent's exit
      s_stack[sp + 1] = SahRange{r.lo, mid, left_pos, right_pos};    // exit of a left child = its sibling
    }
    sp += 2;
    __syncthreads();
  }
}

// Cluster k's staged records (k_emit_clusters_sah_wave) to their place: record j of the cluster -> out[base[k] + j], exits
// inside the cluster shifted with it, the open exits of its right spine = exit[k].  One wave per cluster.
__global__ __launch_bounds__(64) void k_place_clusters(uint32_t K, Clusters c, const hj_bvh_node* __restrict__ staged, hj_bvh_node* __restrict__ out) {
  const uint32_t k = blockIdx.x;
  if (k >= K) return;
  const uint32_t first = __float_as_uint(c.lo[k].w), cnt = __float_as_uint(c.hi[k].w);
  const uint32_t from = 2u * first, to = c.base[k], end = c.exit[k];
  for (uint32_t j = threadIdx.x; j < 2u * cnt - 1u; j += 64u) {
    hj_bvh_node nd = staged[from + j];
    nd.exit_index = nd.exit_index == kOpenExit ? end : nd.exit_index - from + to;
    out[to + j] = nd;
  }
}

// One record of the reference's flattened array per tree node (internal nodes: threads [0, n-1), leaves: the rest).
// The tree occupies records [base, base + 2n - 1) of `out`; `end_exit` is the exit of its right spine.
__global__ __launch_bounds__(256) void k_emit(Tree t, uint32_t n, uint32_t base, uint32_t end_exit, unsigned long long idx_mask,
                                              hj_bvh_node* out) {
  const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t total = 2 * n - 1;
  if (id >= total) return;
  const bool leaf = id >= n - 1;
  const uint32_t k = leaf ? id - (n - 1) : 0;
  const uint32_t first = leaf ? k : t.first[id], cnt = leaf ? 1u : t.count[id];
  uint32_t left_turns = 0;
  for (uint32_t p = t.parent[id]; p != kNoParent; p = t.parent[p & ~kLeafBit]) left_turns += p >> 31;
  const uint32_t pos = base + 2 * first + left_turns, end = pos + 2 * cnt - 1;
  float4 lo, hi;
  uint32_t shape = HJ_BVH_INNER;
  if (leaf) {
    shape = (uint32_t)(t.keys[k] & idx_mask);
    lo = t.leaf_lo[shape];
    hi = t.leaf_hi[shape];
  } else {
    lo = t.node_lo[id];
    hi = t.node_hi[id];
  }
  hj_bvh_node nd;
  nd.aabb_min[0] = lo.x; nd.aabb_min[1] = lo.y; nd.aabb_min[2] = lo.z;
  nd.shape_index = shape;
  nd.aabb_max[0] = hi.x; nd.aabb_max[1] = hi.y; nd.aabb_max[2] = hi.z;
  nd.exit_index = end >= base + total ? end_exit : end;     // right spine: the exit of the subtree's root (src/main.rs:214-231)
  out[pos] = nd;
}

}  // namespace lbvh
}  // namespace hj
