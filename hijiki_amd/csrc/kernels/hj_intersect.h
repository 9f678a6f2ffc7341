// Shape tests and one node of the BVH walk (reference shader/shapes/*.glsl, shader/scene.glsl:97-158) + the small
// helpers every stage uses: wave-aggregated LDS appends, (non-temporal) path-state accessors.
#pragma once
#include "hj_device.h"

#pragma clang fp contract(off)

namespace hj {

#ifndef HJ_BLOCK_THREADS
#define HJ_BLOCK_THREADS 256   // path workgroup size (128 and 512 measured: see DESIGN.md)
#endif
constexpr int kBlockThreads = HJ_BLOCK_THREADS;

// ---------------------------------------------------------------- helpers

// Wave-aggregated append to a workgroup-private queue: ballot + one LDS atomic per wave, lane order kept.
// Must be reached by all active lanes of the wave together.
HJ_DEV uint32_t lds_push(uint32_t* lds_counter, bool pred) {
  const unsigned long long mask = __ballot(pred);
  if (mask == 0) return 0xFFFFFFFFu;
  const uint32_t lane = __lane_id();
  const uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(lds_counter, (uint32_t)__popcll(mask));
  base = __shfl(base, (int)leader);
  const uint32_t prefix = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
  return pred ? base + prefix : 0xFFFFFFFFu;
}

// Next 64-entry chunk of the workgroup's segment (dynamic balance between its waves).
HJ_DEV uint32_t lds_fetch_chunk(uint32_t* lds_head) {
  uint32_t c = 0;
  if (__lane_id() == 0) c = atomicAdd(lds_head, 64u);
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
}

// Path-state accessors.  NT marks them non-temporal (streaming) so that the record and sample streams do not displace
// scene data (nodes, triangles) from the caches: measured +4.4 % on the 1 M-triangle scene and -0.5 % / -3 % on the two
// cbox scenes, whose trees stay cache-resident either way - so hj_scene_upload sets it for large trees (DeviceScene::stream_state).
typedef float f4s __attribute__((ext_vector_type(4)));
template <bool NT>
HJ_DEV float4 ldp(const float4* p, uint32_t i) {
  if (NT) {
    const f4s v = __builtin_nontemporal_load(reinterpret_cast<const f4s*>(p + i));
    return make_float4(v.x, v.y, v.z, v.w);
  }
  return p[i];
}
template <bool NT>
HJ_DEV void stp(float4* p, uint32_t i, float4 v) {
  if (NT) {
    f4s w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
    __builtin_nontemporal_store(w, reinterpret_cast<f4s*>(p + i));
  } else {
    p[i] = v;
  }
}
struct Ray { v3 o, d; float tmin, tmax; };
struct RawHit { float t, u, v; int id; };

// reference shader/shapes/sphere.glsl:18-41
HJ_DEV bool intersect_sphere(const Ray& r, float4 sp, RawHit& h) {
  const v3 l = r.o - xyz(sp);
  const float b = 2.0f * dot3(r.d, l);
  const float c = dot3(l, l) - sp.w * sp.w;
  float d = b * b - 4.0f * c;
  if (d < 0.0f) return false;
  d = __builtin_sqrtf(d);
  const float t0 = -0.5f * (b + d);
  if (r.tmin <= t0 && t0 <= r.tmax) { h.t = t0; return true; }
  const float t1 = -0.5f * (b - d);
  if (r.tmin <= t1 && t1 <= r.tmax) { h.t = t1; return true; }
  return false;
}
// reference shader/shapes/triangle.glsl:15-52 on the values of the pre-gathered record (a, b - a, c - a): ONE text for every walk
HJ_DEV bool triangle_test(const Ray& r, float4 A, float4 B, float4 C, RawHit& h) {
  const v3 ab = xyz(B), ac = xyz(C);
  const v3 n = cross3(ab, ac);
  const v3 ro = r.o - xyz(A);
  const v3 q = cross3(ro, r.d);
  const float d = 1.0f / dot3(r.d, n);
  const float u = d * (-dot3(q, ac));
  const float v = d * dot3(q, ab);
  if (u < 0.0f || v < 0.0f || u + v > 1.0f) return false;
  const float t = d * (-dot3(n, ro));
  if (r.tmin <= t && t <= r.tmax) { h.t = t; h.u = u; h.v = v; return true; }
  return false;
}

// reference shader/shapes/quad.glsl:7-25 on record values (origin, edge1, edge2)
HJ_DEV bool quad_test(const Ray& r, float4 O, float4 E1, float4 E2, RawHit& h) {
  const v3 e1 = xyz(E1), e2 = xyz(E2);
  const v3 n = cross3(e1, e2);
  const v3 ro = r.o - xyz(O);
  const v3 q = cross3(ro, r.d);
  const float d = 1.0f / dot3(r.d, n);
  const float u = d * (-dot3(q, e2));
  const float v = d * dot3(q, e1);
  if (u < 0.0f || u > 1.0f || v < 0.0f || v > 1.0f) return false;
  const float t = d * (-dot3(n, ro));
  if (r.tmin <= t && t <= r.tmax) { h.t = t; h.u = u; h.v = v; return true; }
  return false;
}

HJ_DEV bool intersect_shape(const DeviceScene& sc, const Ray& r, uint32_t shape, RawHit& h) {
  if (shape < sc.ns) return intersect_sphere(r, sc.spheres[shape], h);
  if (shape < sc.ns + sc.nq) {
    const float4* __restrict__ rec = sc.quads + 3 * (size_t)(shape - sc.ns);
    return quad_test(r, rec[0], rec[1], rec[2], h);
  }
  const float4* __restrict__ rec = sc.tri_isect + 3 * (size_t)(shape - sc.ns - sc.nq);      // one address, three offsets
  return triangle_test(r, rec[0], rec[1], rec[2], h);
}

// What the walk does when it stands on a leaf: `a` is the first word of the node the lane stopped at.
//   leaf record:  a = shape index                      -> one shape test (scene.glsl:105-119)
//   PAIR record:  a = kInnerFlag | kPairFlag | pair    -> an inner node whose two children are triangle leaves, entered:
//                 the reference now visits the left leaf, tests its triangle, goes to its exit = the right leaf, tests
//                 that one with the tMax the first test left, and goes on to the right leaf's exit = the pair's own
//                 exit.  Both triangles sit side by side in sc.tri_pair (their shape indices in the w lanes), so the two
//                 node fetches and one of the two leaf phases of that sequence are gone; the tests and their order
//                 are the same.
// Returns true when the ray is finished (an any-hit ray that hit).
template <bool PAIRS>
HJ_DEV bool leaf_test(const DeviceScene& sc, Ray& r, uint32_t a, RawHit& h, bool any) {
  if (!PAIRS || (a & kInnerFlag) == 0u) {
    if (intersect_shape(sc, r, a, h)) {
      h.id = (int)a;
      if (any) return true;
      r.tmax = h.t - kEps;
    }
    return false;
  }
  const float4* __restrict__ rec = sc.tri_pair + 6 * (size_t)(a & kIndexMask);
  const float4 A = rec[0], B = rec[1], C = rec[2], D = rec[3], E = rec[4], F = rec[5];
  // (computing both triangles' (u, v, t) side by side without the early returns was measured: 1 % slower on the 1 M-triangle
  // scene - most tests end at the u / v check)
  if (triangle_test(r, A, B, C, h)) {
    h.id = (int)__float_as_uint(A.w);
    if (any) return true;
    r.tmax = h.t - kEps;
  }
  if (triangle_test(r, D, E, F, h)) {
    h.id = (int)__float_as_uint(D.w);
    if (any) return true;
    r.tmax = h.t - kEps;
  }
  return false;
}

// One node of the walk (scene.glsl:103-131).  Both 16-byte halves are consumed and the box test is evaluated
// BEFORE the leaf/inner decision, with selects only (no branch for the compiler to sink the loads behind): one
// memory round trip per node.  For a leaf the box result is ignored (leaf boxes are never tested upstream).
// Returns true when the lane has to stop for shape tests (a leaf: a = shape index; a pair node it enters: a = the
// node's first word); otherwise advances cur to the left child or the exit.
template <bool PAIRS>
HJ_DEV bool node_step(float4 n0, float4 n1, v3 inv, v3 off, const Ray& r, uint32_t& cur, uint32_t& a, uint32_t& ex) {
  const float tnx = fmaf(n0.x, inv.x, off.x), tpx = fmaf(n1.x, inv.x, off.x);
  const float tny = fmaf(n0.y, inv.y, off.y), tpy = fmaf(n1.y, inv.y, off.y);
  const float tnz = fmaf(n0.z, inv.z, off.z), tpz = fmaf(n1.z, inv.z, off.z);
  const float t0 = f_max(f_max(f_min(tnx, tpx), f_min(tny, tpy)), f_min(tnz, tpz));
  const float t1 = f_min(f_min(f_max(tnx, tpx), f_max(tny, tpy)), f_max(tnz, tpz));
  const bool enter = (t0 < t1 + kEps && t0 < r.tmax && t1 > r.tmin);
  a = __float_as_uint(n0.w);
  ex = __float_as_uint(n1.w);
  // the lane stops on a leaf, and on a pair node whose box it enters (leaf_test)
  const bool stop = (a & kInnerFlag) == 0u || (PAIRS && (a & kPairFlag) != 0u && enter);
  const uint32_t nxt = enter ? (a & kIndexMask) : ex;
  cur = stop ? cur : nxt;
  return stop;
}

// Rays in GENERAL POSITION: every 1 / d and every -o / d finite, no 1 / d zero.  For them the slab test (scene.glsl:120-131) is
// monotone in the bounds of the box - "a box inside a box that the ray misses is missed" - which is what the upload's collapse and
// its guard nodes rest on.  A direction with a zero (or denormal, infinite, NaN) component makes the test form
// inf - inf and drop the NaN in its min / max: what it answers then depends on the SIGNS of the bounds, not on their order (a box
// [0, 2] passes where [-1e-4, 2] fails), upstream and here alike - so those rays, one in 10^5 ... 10^7 where surfaces are
// axis-aligned, walk the second copy of the tree, which is the reference's own (DeviceScene::root2).
HJ_DEV bool general_position(v3 inv, v3 off) {
  const float chk = ((inv.x - inv.x) + (inv.y - inv.y)) + ((inv.z - inv.z) + (off.x - off.x)) + ((off.y - off.y) + (off.z - off.z));
  return chk == 0.0f && inv.x != 0.0f && inv.y != 0.0f && inv.z != 0.0f;
}

// reference shader/scene.glsl:97-158.  ANYHIT: stop at the first accepted hit
// (the shadow overload scene.glsl:92-96 only uses the boolean, and the first
// accepted hit in visiting order is the same with or without tMax shrinking).
template <bool USE_BVH, bool ANYHIT>
HJ_DEV bool traverse(const DeviceScene& sc, Ray r, RawHit& h) {
  h.id = -1;
  if (USE_BVH) {
    const v3 inv = V(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
    const v3 off = V(-(r.o.x * inv.x), -(r.o.y * inv.y), -(r.o.z * inv.z));
    const uint32_t nn = sc.num_nodes;
    uint32_t cur = general_position(inv, off) ? sc.root : sc.root2;
    // "while-while": every lane first walks inner nodes until it stands on a leaf (or leaves the tree), then the
    // lanes that reached a leaf run the (much longer) shape test TOGETHER instead of interleaved with box tests.
    // Visiting order per ray is exactly the reference's pre-order skip-link walk.
    for (;;) {
      uint32_t a = 0, ex = 0;
      bool at_leaf = false;
      while (cur < nn && !at_leaf) {
        const float4 n0 = sc.nodes[2 * cur], n1 = sc.nodes[2 * cur + 1];
        at_leaf = node_step<true>(n0, n1, inv, off, r, cur, a, ex);
      }
      if (!at_leaf) break;
      if (leaf_test<true>(sc, r, a, h, ANYHIT)) return true;   // leaf boxes are never tested (scene.glsl:105-119)
      cur = ex;
    }
  } else {
    if (sc.ns > 100 || sc.nq > 100) return false;  // scene.glsl:135-138
    const uint32_t total = sc.ns + sc.nq + sc.nt;
    for (uint32_t s = 0; s < total; s++) {
      if (intersect_shape(sc, r, s, h)) {
        h.id = (int)s;
        if (ANYHIT) return true;
        r.tmax = h.t - kEps;
      }
    }
  }
  return h.id != -1;
}

}  // namespace hj
