// Wavefront path-tracing kernels for gfx950 (wave64).
//
// One batch = up to `capacity` camera paths (many ImageBlocks of many passes).  Every path workgroup owns one
// private segment of every queue (hj_device.h) and advances ITS paths one bounce per round:
//
//   per round:
//     stage_gen_camera     render.glsl:149-162               top-up: seed, camera ray of NEW paths into free path slots
//     stage_trace_merged   scene.glsl:92-133                 skip-link BVH walk of the closest-hit rays (-> hit record)
//                                                            and of the previous round's shadow rays (any-hit, boolean-
//                                                            equivalent to the reference's closest-hit; adds NEE radiance)
//     compact_hits_by_tag                                    hits binned by MATERIAL TAG in queue order (wave ballots)
//     stage_shade          scene.glsl:160-175, render.glsl:102-144, material.glsl
//                                                            populate, emission, NEE sample, BSDF sample,
//                                                            roulette -> next ray queue + shadow queue
//   k_reconstruct                     reconstruction.glsl:22-66
//
// k_path_wavefront runs all stages of a batch in ONE persistent launch (workgroup barriers only): the kernel's own code
// is the walk (trace_persistent: in-wave ray replacement, merged first step, bounded burst), the other three stages are
// CALLED device functions (stage_*_call: register-allocated on their own, so the walk stays free of spills);
// k_gen_camera / k_trace_closest / k_shade / k_trace_shadow launch the stages one by one (diagnostic path).
// No stage uses a global atomic: appends are wave ballot + one LDS atomic.
//
// Every path owns its RNG state, so queue order never changes results; the per-path order of radiance
// additions is the reference's (emission, then NEE, bounce by bounce) because a workgroup's stages are
// separated by barriers (fused) or kernel boundaries (split).
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

// reference shader/shapes/triangle.glsl:15-52 on the pre-gathered record
HJ_DEV bool intersect_triangle(const DeviceScene& sc, const Ray& r, uint32_t ix, RawHit& h) {
  const float4* __restrict__ rec = sc.tri_isect + 3 * (size_t)ix;      // one address, three offsets
  const float4 A = rec[0], B = rec[1], C = rec[2];
  const v3 a = xyz(A), ab = xyz(B), ac = xyz(C);
  const v3 n = cross3(ab, ac);
  const v3 ro = r.o - a;
  const v3 q = cross3(ro, r.d);
  const float d = 1.0f / dot3(r.d, n);
  const float u = d * (-dot3(q, ac));
  const float v = d * dot3(q, ab);
  if (u < 0.0f || v < 0.0f || u + v > 1.0f) return false;
  const float t = d * (-dot3(n, ro));
  if (r.tmin <= t && t <= r.tmax) { h.t = t; h.u = u; h.v = v; return true; }
  return false;
}
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
// reference shader/shapes/quad.glsl:7-25
HJ_DEV bool intersect_quad(const DeviceScene& sc, const Ray& r, uint32_t ix, RawHit& h) {
  const float4* __restrict__ rec = sc.quads + 3 * (size_t)ix;
  const v3 o = xyz(rec[0]), e1 = xyz(rec[1]), e2 = xyz(rec[2]);
  const v3 n = cross3(e1, e2);
  const v3 ro = r.o - o;
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
  if (shape < sc.ns + sc.nq) return intersect_quad(sc, r, shape - sc.ns, h);
  return intersect_triangle(sc, r, shape - sc.ns - sc.nq, h);
}

// triangle.glsl:15-52 on record values (a, b - a, c - a)
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

// quad.glsl:7-25 on record values (origin, edge1, edge2)
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
    uint32_t cur = sc.root;
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

// Persistent "while-while" walk with in-wave ray replacement (BVH mode): a lane whose ray has left the tree
// does not idle until the slowest lane of its wave is done - as soon as kRefillMin lanes are free the wave
// pulls that many new rays from its workgroup's queue segment (one LDS atomic) and the walk continues.
// Measured need: with one ray per lane for the lifetime of a wave, VALU instructions of the bounce-ray
// traversal ran with 9.6 of 64 lanes active (rocprofv3 SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU).
// Each ray still performs exactly the reference's pre-order skip-link walk (scene.glsl:97-133).
//   fetch(i, slot, ray, any, h)  loads queue entry i (a shadow ray: any = true, its pending contribution in h, its sample in slot)
//   finish(done, slot, h, any)   wave-convergent: called when some lanes are done; `done` lanes have a final result
// A round of the loop: service phase (only when enough lanes are free) -> merged first step (leaf lanes fetch their shape
// record, the others their node, in one trip) -> up to inner_burst - 1 plain box steps for the lanes not standing on a leaf.

#ifdef HJ_WALK_STATS
// Diagnostic build only (tools/build_variant.sh stats -DHJ_WALK_STATS): wave-level occupancy of the walk's phases.
// [0] outer iterations [1] inner wave-steps [2] lanes in them [3] leaf phases [4] lanes in them [5] refills
// [6] lanes refilled [7] lanes active at the start of an outer iteration; [29] of g_round_stats: wave time at the barrier behind the walk
__device__ unsigned long long g_walk_stats[16];   // [10..12] wave cycles by phase, [13] total, [14] lane-steps on nodes outside the LDS copy
// rounds of the fused kernel by size bucket b (rays of the round in [64 * 4^b / 4, 64 * 4^b), b = 0..7):
// [b] rounds, [8 + b] rays, [16 + b] wave-cycles (wall clock of the round x waves of the workgroup still alive)
__device__ unsigned long long g_round_stats[32];   // [0..23] rounds by size; [24..28] wall cycles x waves of top-up, walk, hit compaction, shade, the rest of a round
#define HJ_STAT(i, v) do { const long long v_ = (long long)(v); if (__lane_id() == 0) ws[i] += (unsigned long long)v_; } while (0)
#else
#define HJ_STAT(i, v) do { } while (0)
#endif

// MODE 0: closest-hit rays, 1: any-hit (shadow) rays, 2: both kinds in one queue (fetch says which per ray).
// PAIRS: the scene has pair nodes (leaf_test); without them the code for them is not even compiled in (it costs 4 % on
// a scene that has none).
#ifndef HJ_MERGE_LEAF
#define HJ_MERGE_LEAF 2      // 0: separate leaf phase everywhere, 1: merged first step on trees without pair nodes only, 2: everywhere
#endif
#ifndef HJ_SHADOW_CARRY
#define HJ_SHADOW_CARRY 1
#endif
#ifndef HJ_FETCH_SELECT
#define HJ_FETCH_SELECT 1
#endif
template <int MODE, bool PAIRS, class Fetch, class Finish>
HJ_DEV void trace_persistent(const DeviceScene& sc, uint32_t n, uint32_t* s_head, const float4* s_nodes,
                             Fetch fetch, Finish finish) {
  constexpr bool MERGE = HJ_MERGE_LEAF == 2 || (HJ_MERGE_LEAF == 1 && !PAIRS);
  const uint32_t lane = __lane_id();
  const uint32_t nn = sc.num_nodes, nhot = sc.num_hot;
  bool active = false, pending = false, exhausted = false, any = (MODE == 1);
  uint32_t slot = 0, cur = 0;
  Ray r; r.o = V(0, 0, 0); r.d = V(0, 0, 0); r.tmin = 0.f; r.tmax = 0.f;
  v3 inv = V(0, 0, 0), off = V(0, 0, 0);
  // base addresses of the node array and of its LDS copy as opaque VGPR values (see the box-step loop)
  uint32_t nb_glo, nb_ghi, nb_llo, nb_lhi;
  {
    const uint64_t gb = reinterpret_cast<uint64_t>(sc.nodes), lb = reinterpret_cast<uint64_t>(s_nodes);
    asm volatile("v_mov_b32 %0, %1" : "=v"(nb_glo) : "s"((uint32_t)gb));
    asm volatile("v_mov_b32 %0, %1" : "=v"(nb_ghi) : "s"((uint32_t)(gb >> 32)));
    asm volatile("v_mov_b32 %0, %1" : "=v"(nb_llo) : "s"((uint32_t)lb));
    asm volatile("v_mov_b32 %0, %1" : "=v"(nb_lhi) : "s"((uint32_t)(lb >> 32)));
  }
  RawHit h; h.t = 0.f; h.u = 0.f; h.v = 0.f; h.id = -1;
#if defined(HJ_VALU_PROBE) || defined(HJ_LOAD_PROBE) || defined(HJ_LEAF_VALU_PROBE) || defined(HJ_WIDE_PROBE)
  float valu_probe = 1.0f;
#endif
  uint32_t shape = 0, ex = 0;
  bool at_leaf = false;                  // (MERGE: a leaf reached in one round is tested in the first step of the next)
#ifdef HJ_WALK_STATS
  unsigned long long ws[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long t_begin = clock64();
#endif
  // Merged first step of a round (MERGE): a lane that reached a leaf in the previous round fetches its SHAPE record in the
  // same memory trip in which the other lanes fetch their next node (one address select, the same load instructions), then
  // each kind computes its own test.  The leaf tests of a round so cost no memory round trip of their own, and the lane goes
  // on with the box steps of this round.  Per ray the sequence of box tests, shape tests and tMax updates is unchanged
  // (scene.glsl:102-133).  step0_issue only issues the loads, step0_compute consumes them.
  bool go = false, more = false, pair = false;      // the record has a third 16-byte part; a pair record: six
  float4 x0 = make_float4(0.f, 0.f, 0.f, 0.f), x1 = x0, x2 = x0, x3 = x0, x4 = x0, x5 = x0;
  auto step0_issue = [&]() {
    go = active && (at_leaf || cur < nn);
    more = false; pair = false;
    x0 = x1 = x2 = x3 = x4 = x5 = make_float4(0.f, 0.f, 0.f, 0.f);   // (nothing is carried from one round to the next)
    if (go) {
      uint32_t a_lo, a_hi;
      if (at_leaf) {
        uint64_t pa;
        if (PAIRS && (shape & kInnerFlag) != 0u) {
          pa = reinterpret_cast<uint64_t>(sc.tri_pair) + 96ull * (uint64_t)(shape & kIndexMask);
          more = true; pair = true;
        } else if (shape < sc.ns) {
          pa = reinterpret_cast<uint64_t>(sc.spheres) + 16ull * (uint64_t)shape;
        } else if (shape < sc.ns + sc.nq) {
          pa = reinterpret_cast<uint64_t>(sc.quads) + 48ull * (uint64_t)(shape - sc.ns);
          more = true;
        } else {
          pa = reinterpret_cast<uint64_t>(sc.tri_isect) + 48ull * (uint64_t)(shape - sc.ns - sc.nq);
          more = true;
        }
        a_lo = (uint32_t)pa; a_hi = (uint32_t)(pa >> 32);
      } else {
        const bool hot = cur < nhot;
        a_lo = (hot ? nb_llo : nb_glo) + (cur << 5); a_hi = hot ? nb_lhi : nb_ghi;
      }
      const float4* __restrict__ p = reinterpret_cast<const float4*>(((uint64_t)a_hi << 32) | (uint64_t)a_lo);
      x0 = p[0]; x1 = p[1];
      if (more) x2 = p[2];
      if (PAIRS && pair) { x3 = p[3]; x4 = p[4]; x5 = p[5]; }
    }
  };
  auto step0_compute = [&]() {
    if (go) {
      if (at_leaf) {
        const bool anyhit = MODE == 1 || (MODE == 2 && any);
        bool done = false;
        if (PAIRS && pair) {
          if (triangle_test(r, x0, x1, x2, h)) { h.id = (int)__float_as_uint(x0.w); if (anyhit) done = true; else r.tmax = h.t - kEps; }
          if (!done && triangle_test(r, x3, x4, x5, h)) { h.id = (int)__float_as_uint(x3.w); if (anyhit) done = true; else r.tmax = h.t - kEps; }
        } else {
          bool hit;
          if (shape < sc.ns) hit = intersect_sphere(r, x0, h);
          else if (shape < sc.ns + sc.nq) hit = quad_test(r, x0, x1, x2, h);
          else hit = triangle_test(r, x0, x1, x2, h);
          if (hit) { h.id = (int)shape; if (anyhit) done = true; else r.tmax = h.t - kEps; }
        }
#ifdef HJ_LEAF_VALU_PROBE   // diagnostic: extra VALU instructions in the leaf branch of the merged step (a pair test has ~130)
#pragma unroll
        for (int k_ = 0; k_ < HJ_LEAF_VALU_PROBE; k_++) asm volatile("v_add_f32 %0, %0, %0" : "+v"(valu_probe));
#endif
        if (done) { active = false; pending = true; }   // occluded shadow ray (h.id >= 0 tells finish)
        cur = ex; at_leaf = false;
      } else {
        at_leaf = node_step<PAIRS>(x0, x1, inv, off, r, cur, shape, ex);
      }
    }
  };
  for (;;) {
#ifdef HJ_WALK_STATS
    const unsigned long long t_a = clock64();   // [10] service, [11] box steps, [12] leaf tests: wave cycles by phase
#endif
    // Service phase: only when enough lanes are free.  Finished lanes keep their result in registers until
    // then, so that result STORES and new-ray LOADS are issued together, once per phase: vmcnt counts loads and
    // stores in one in-order counter on gfx950, and a store between two node fetches would stall the walk for
    // a full write acknowledgement.
#ifdef HJ_LANE_LIMIT   // diagnostic: only the first HJ_LANE_LIMIT lanes of a wave ever hold a ray - how the cost of a wave-step depends on its active lanes
    const unsigned long long idle = __ballot(!active && lane < (uint32_t)(HJ_LANE_LIMIT));
    const uint32_t nidle = (uint32_t)__popcll(idle);
    const bool service = nidle >= (sc.refill_min * (uint32_t)(HJ_LANE_LIMIT) + 63u) / 64u || nidle == (uint32_t)(HJ_LANE_LIMIT);
#else
    const unsigned long long idle = __ballot(!active);
    const uint32_t nidle = (uint32_t)__popcll(idle);
    const bool service = nidle >= sc.refill_min || nidle == 64u;
#endif
    // The loads of the NEW rays are issued first, the results of the finished ones are written (and, for an unoccluded
    // shadow ray, its sample read, added to and written) after them: both memory round trips are then in flight
    // together, and the wait for the new rays does not include the stores (vmcnt retires in order: only what was
    // issued BEFORE a load has to complete with it).
    bool got = false, any2 = any;
    uint32_t slot2 = 0;
    Ray r2; r2.o = V(0, 0, 0); r2.d = V(0, 0, 0); r2.tmin = 0.f; r2.tmax = 0.f;
    RawHit h2; h2.t = 0.f; h2.u = 0.f; h2.v = 0.f; h2.id = -1;   // (a shadow ray carries its pending contribution in t, u, v)
    if (service && !exhausted) {
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(s_head, nidle);
      base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
#ifdef HJ_LANE_LIMIT
      if (!active && lane < (uint32_t)(HJ_LANE_LIMIT)) {
#else
      if (!active) {
#endif
        const uint32_t my = base + (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
        if (my < n) { fetch(my, slot2, r2, any2, h2); got = true; }
      }
      exhausted = base + nidle >= n;
    }
    if (service) {
      if (__ballot(pending) != 0) finish(pending, slot, h, any);
      pending = false;
      HJ_STAT(5, 1); HJ_STAT(6, __popcll(__ballot(got)));
    }
    if (got) {
      slot = slot2; any = any2; r = r2; h = h2;
      inv = V(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
      off = V(-(r.o.x * inv.x), -(r.o.y * inv.y), -(r.o.z * inv.z));
      cur = sc.root; active = true;
    }
    if (__ballot(active || pending) == 0) break;   // (a lane can finish in the merged first step: its result is written by the next service phase)
#ifdef HJ_LDS_RT_PROBE
    // What ONE re-grouping of the wave's rays through LDS costs at the very least: a queue push (ballot + LDS atomic) and the
    // ray's state (12 dwords here; a design needs 14 or more) written to a slot and read back - here to the lane's own slot
    // (conflict-free; slots picked from a queue would be scattered).  HJ_LDS_RT_PROBE = how many of them per round of the walk loop.
    {
      // (WgShared is declared further down: rt_ctr and rt follow its node copy, which is what s_nodes points to)
      char* rt_base = reinterpret_cast<char*>(const_cast<float4*>(s_nodes)) + 32u * kHotNodes;
      uint32_t* rt_ctr = reinterpret_cast<uint32_t*>(rt_base);
      float4* rt = reinterpret_cast<float4*>(rt_base + 16);
      const uint32_t a0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float4*)(rt + 3u * threadIdx.x);
#pragma unroll
      for (int k_ = 0; k_ < HJ_LDS_RT_PROBE; k_++) {
        const uint32_t qpos = lds_push(&rt_ctr[k_ & 1], active);
        f4s w0, w1, w2;
        w0.x = r.o.x; w0.y = r.o.y; w0.z = r.o.z; w0.w = r.tmax;
        w1.x = r.d.x; w1.y = r.d.y; w1.z = r.d.z; w1.w = r.tmin;
        w2.x = h.t; w2.y = __int_as_float(h.id); w2.z = __uint_as_float(cur); w2.w = __uint_as_float(slot + (qpos & 0u));
        asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:16\n\tds_write_b128 %0, %3 offset:32"
                     :: "v"(a0), "v"(w0), "v"(w1), "v"(w2) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b128 %2, %3 offset:32\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(w0), "=&v"(w1), "=&v"(w2) : "v"(a0) : "memory");
        r.o = V(w0.x, w0.y, w0.z); r.tmax = w0.w; r.d = V(w1.x, w1.y, w1.z); r.tmin = w1.w;
        h.t = w2.x; h.id = __float_as_int(w2.y); cur = __float_as_uint(w2.z); slot = __float_as_uint(w2.w);
      }
    }
#endif
    HJ_STAT(0, 1); HJ_STAT(7, __popcll(__ballot(active)));
#ifdef HJ_WALK_STATS
    const unsigned long long t_b = clock64();
    HJ_STAT(10, t_b - t_a);
#endif
    if (!MERGE) at_leaf = false;
    uint32_t burst = sc.inner_burst;       // lanes standing on a leaf wait at most this many box steps of the others
    if (MERGE) {
#ifdef HJ_WALK_STATS
      // the merged step: its node lanes count as a box step, its leaf lanes as a leaf phase; its wave cycles go to [12]
      { const unsigned long long mn = __ballot(active && !at_leaf && cur < nn), mc = __ballot(active && !at_leaf && cur < nn && cur >= nhot);
        const unsigned long long ml = __ballot(active && at_leaf), mp = __ballot(active && at_leaf && (shape & kInnerFlag) != 0u);
        if (lane == 0) {
          if (mn) { ws[1] += 1; ws[2] += __popcll(mn); ws[14] += __popcll(mc); }
          if (ml) { ws[3] += 1; ws[4] += __popcll(ml); ws[15] += __popcll(ml) + __popcll(mp); }
        } }
#endif
      step0_issue(); step0_compute();
      burst--;
#ifdef HJ_WALK_STATS
      HJ_STAT(12, clock64() - t_b);
#endif
    }
    while (active && cur < nn && !at_leaf && burst != 0) {
      // hot node: LDS copy, same 32-byte record layout as in HBM, so that ONE address select feeds both 16-byte
      // loads (FLAT loads of base + 32*cur and +16; a per-array `if` compiled to two exec-masked address blocks).
      // The two base addresses sit in four VGPRs (nb_*): v_cndmask cannot take a scalar source beside VCC, and the
      // compiler otherwise re-creates them with four v_mov per step.
      // Neither array crosses a 4 GiB boundary (hj_scene_upload places the node array so; the LDS aperture cannot),
      // so the low word never carries into the high one: cmp + 2 cndmask + 1 shift-add instead of ten instructions.
      const bool hot = cur < nhot;
#ifdef HJ_WALK_STATS
      { const unsigned long long m = __ballot(true), mc = __ballot(cur >= nhot);     // [14] lane-steps on nodes outside the LDS copy
        if (lane == (uint32_t)__ffsll((long long)m) - 1u) { ws[1] += 1; ws[2] += __popcll(m); ws[14] += __popcll(mc); } }
#endif
      const uint32_t a_lo = (hot ? nb_llo : nb_glo) + (cur << 5), a_hi = hot ? nb_lhi : nb_ghi;
      const float4* __restrict__ nd = reinterpret_cast<const float4*>(((uint64_t)a_hi << 32) | (uint64_t)a_lo);
      const float4 n0 = nd[0], n1 = nd[1];
#ifdef HJ_LOAD_PROBE   // diagnostic: one more 16-byte load per box step; 1: every lane the same address, 2: the lane's own node again, 3: a global (never LDS) address per lane
      {
        const float4* pp = HJ_LOAD_PROBE == 1 ? sc.nodes : HJ_LOAD_PROBE == 2 ? nd : sc.nodes + 2 * cur;
        float4 pv;
        asm volatile("flat_load_dwordx4 %0, %1" : "=v"(pv) : "v"(pp) : "memory");
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        valu_probe += pv.x * 0.0f;
      }
#endif
#ifdef HJ_WIDE_PROBE   // diagnostic: what a 128-byte node would cost per step - the six other 16-byte parts of the node's 128-byte line
      {
        const uint32_t own = (cur & 3u) * 2u;             // the node's own two parts within its group of four records
        const float4* gp = reinterpret_cast<const float4*>((((uint64_t)a_hi << 32) | (uint64_t)a_lo) & ~127ull);
        float4 pv[6];
#pragma unroll
        for (int k_ = 0; k_ < 6; k_++) {
          const float4* pp = gp + ((own + 2u + (uint32_t)k_) & 7u);
          asm volatile("flat_load_dwordx4 %0, %1" : "=v"(pv[k_]) : "v"(pp) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k_ = 0; k_ < 6; k_++) valu_probe += pv[k_].x * 0.0f;
        // the destinations must stay live until the wait: a register the compiler considers dead is handed to the next
        // address computation while the load that will overwrite it is still in flight (a build without this faulted)
        asm volatile("" :: "v"(valu_probe));
      }
#endif
      at_leaf = node_step<PAIRS>(n0, n1, inv, off, r, cur, shape, ex);
      burst--;
#ifdef HJ_VALU_PROBE   // diagnostic: HJ_VALU_PROBE extra VALU instructions per box step (is the walk VALU-bound?)
#pragma unroll
      for (int k_ = 0; k_ < HJ_VALU_PROBE; k_++) asm volatile("v_add_f32 %0, %0, %0" : "+v"(valu_probe));
#endif
    }
    if (active && !at_leaf && cur >= nn) { active = false; pending = true; }   // walked off the end of the tree
#ifdef HJ_WALK_STATS
    const unsigned long long t_c = clock64();
    HJ_STAT(11, t_c - t_b);      // (MERGE: includes the merged step, also counted in [12])
    if (!MERGE) { const unsigned long long m = __ballot(at_leaf), mp = __ballot(at_leaf && (shape & kInnerFlag) != 0u);   // [15] shape records fetched (a pair: two)
      if (m && lane == (uint32_t)__ffsll((long long)m) - 1u) { ws[3] += 1; ws[4] += __popcll(m); ws[15] += __popcll(m) + __popcll(mp); } }
#endif
    if (!MERGE && at_leaf) {
      if (leaf_test<PAIRS>(sc, r, shape, h, MODE == 1 || (MODE == 2 && any))) { active = false; pending = true; }   // occluded shadow ray (h.id >= 0 tells finish)
      cur = ex;
    }
    if (!MERGE) HJ_STAT(12, clock64() - t_c);
  }
#ifdef HJ_WALK_STATS
  HJ_STAT(13, clock64() - t_begin);
  for (int i = 0; i < 16; i++) {      // ws[] lives in whichever lane did the counting: sum over the wave
    unsigned long long v = ws[i];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if (lane == 0 && v) atomicAdd(&g_walk_stats[i], v);
  }
#endif
}

// ------------------------------------------------------------ populate (its)

struct Its { v3 p, n, ft, fb; float u, v; };   // frame = [ft fb n]

// reference shader/shapes/triangle.glsl:54-78
HJ_DEV void populate_triangle(const DeviceScene& sc, uint32_t ix, float hu, float hv, Its& its) {
  const float4* __restrict__ rec = sc.tri_shade + 4 * (size_t)ix;
  const float4 A = rec[0], B = rec[1], C = rec[2], Vv = rec[3];
  const float l0 = (1.0f - hu) - hv, l1 = hu, l2 = hv;
  const v3 ns = (xyz(A) * l0 + xyz(B) * l1) + xyz(C) * l2;
  its.n = normalize3(ns);
  its.u = (A.w * l0 + B.w * l1) + C.w * l2;
  its.v = (Vv.x * l0 + Vv.y * l1) + Vv.z * l2;
  v3 bt = (__builtin_fabsf(its.n.x) > __builtin_fabsf(its.n.y)) ? V(0.f, 1.f, 0.f) : V(1.f, 0.f, 0.f);
  const v3 t = normalize3(cross3(its.n, bt));
  bt = cross3(its.n, t);
  its.ft = t; its.fb = bt;
}
// reference shader/shapes/sphere.glsl:43-52
HJ_DEV void populate_sphere(float4 sp, Its& its) {
  const v3 n = divs(its.p - xyz(sp), sp.w);
  its.n = n;
  const v3 t = normalize3(V(-n.z, 0.0f, n.x));
  its.ft = t; its.fb = cross3(n, t);
  float ux = 0.5f + hj_atan2(n.z, n.x) * (1.0f / kTwoPi);
  const float uy = 0.5f + hj_asin(f_min(f_max(n.y, -1.0f), 1.0f)) * kInvPi;
  if (ux != ux) ux = 0.0f;
  its.u = ux; its.v = uy;
}
// reference shader/shapes/quad.glsl:27-32 (uv stays the raw hit's)
HJ_DEV void populate_quad(const DeviceScene& sc, uint32_t ix, float hu, float hv, Its& its) {
  const v3 t = normalize3(xyz(sc.quads[3 * ix + 1]));
  const v3 b = normalize3(xyz(sc.quads[3 * ix + 2]));
  its.n = cross3(t, b); its.ft = t; its.fb = b; its.u = hu; its.v = hv;
}

// ------------------------------------------------------------ emitter sampling

struct SRec { v3 p, n; float pdf; };

HJ_DEV v3 ld3(const float* p) { return V(p[0], p[1], p[2]); }

// reference shader/scene.glsl:44-89 + shapes/*: sample*.  Always 3 draws.
HJ_DEV v3 sample_emitter(const DeviceScene& sc, v3 ref, uint32_t& rng, v3& sh_dir, float& sh_tmax) {
  float xi = rng_float(rng);
  if (sc.num_emitters == 0) {   // reference reads emitters[0] out of bounds; defined here as "no light"
    rng_uint(rng); rng_uint(rng);
    sh_dir = V(0, 0, 0); sh_tmax = 0.0f;
    return V(0, 0, 0);
  }
  uint32_t e = 0;
  for (uint32_t i = 0; i < sc.num_emitters; i++) {
    xi -= __uint_as_float(__float_as_uint(sc.emit_rec[kEmitRecF4 * i].x));   // emitters[i].pdf
    if (xi < 0.0f) { e = i; break; }
  }
  // one pre-gathered record per emitter (hj_device.h) instead of emitter -> indices -> 3 vertices -> material word
  // -> material: the values are the ones those arrays hold, the chain of dependent fetches is gone
  const float4* __restrict__ er = sc.emit_rec + (size_t)kEmitRecF4 * e;
  const float4 r0 = er[0], r1 = er[1], r2 = er[2], r3 = er[3];
  const float em_pdf = r0.x;
  const uint32_t kind = __float_as_uint(r0.y);
  const v3 power = V(r1.w, r2.w, r3.w);
  SRec sr;
  if (kind == 0u) {                          // sphere.glsl:54-58
    sr.n = rand_uniform_sphere(rng);
    sr.p = xyz(r1) + sr.n * r0.z;
    sr.pdf = 1.0f / (((r0.z * r0.z) * 4.0f) * kPi);
  } else if (kind == 1u) {                   // quad.glsl:34-45
    const v3 o = xyz(r1), e1 = xyz(r2), e2 = xyz(r3);
    const v3 n = cross3(e1, e2);
    const float area = len3(n);
    sr.n = divs(n, area);
    const float u = rng_float(rng), v = rng_float(rng);
    sr.p = (o + e1 * u) + e2 * v;
    sr.pdf = 1.0f / area;
  } else {                                   // triangle.glsl:81-102
    const float4 r4 = er[4], r5 = er[5], r6 = er[6];
    const v3 a = xyz(r1), b = xyz(r2), c = xyz(r3);
    const v3 n = cross3(b - a, c - a);
    const float area = len3(n) * 0.5f;
    const v3 l = rand_barycentric(rng);
    sr.n = normalize3((xyz(r4) * l.x + xyz(r5) * l.y) + xyz(r6) * l.z);
    sr.p = (a * l.x + b * l.y) + c * l.z;
    sr.pdf = 1.0f / area;
  }
  v3 dir = sr.p - ref;
  const float dist = len3(dir);
  dir = divs(dir, dist);
  sh_dir = dir; sh_tmax = dist - kEps;
  const float cosT = -dot3(dir, sr.n);
  if (cosT < 0.0f) return V(0, 0, 0);
  const float pdf = (((em_pdf * sr.pdf) * dist) * dist) / cosT;
  return divs(power, pdf);
}

// reference shader/materials/diffusecb.glsl:6-13
HJ_DEV v3 checkerboard(const DeviceScene& sc, uint32_t idx, float u, float v) {
  const float4 ca = sc.diffusecb[2 * idx], cb = sc.diffusecb[2 * idx + 1];
  float fu = (0.5f * u) / ca.w, fv = (0.5f * v) / cb.w;
  fu = fu - __builtin_floorf(fu); fv = fv - __builtin_floorf(fv);
  const bool a = fu < 0.5f, b = fv < 0.5f;
  return (a != b) ? xyz(cb) : xyz(ca);
}

// ------------------------------------------------------------------ stages
//
// A path workgroup owns queue segment g in every queue, so the stages of one bounce need only workgroup
// barriers between them.  The same stage functions are used by the split per-stage kernels and by the fused
// persistent kernel k_path_wavefront (one launch per batch: camera rays, then the bounce loop).

// A value every lane of the wave reads from the same LDS word: keep it in a scalar register (an LDS load lands in a VGPR).
HJ_DEV uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

struct WgShared {                 // LDS of a path workgroup (16.5 KB)
  uint32_t head;                  // next unread entry of the merged queue being walked
  uint32_t head_cam;              // next 64-ray packet of the round's new camera rays
  uint32_t cnt_hit[kNumTags];     // hits binned by material tag (this round)
  uint32_t wcnt[kBlockThreads / 64][kNumTags];   // per-wave tag counts of the ordered compaction
  uint32_t n_ray[2];              // paths in the arrays of each parity (continuing paths, written by shade)
  uint32_t n_gen;                 // new camera paths the current top-up has appended behind them
  uint32_t n_shadow;              // shadow records
  uint32_t n_unocc;               // statistics: unoccluded shadow rays of this round
  // IMPLICIT camera paths of the current round (kernels with the packet stage): positions [cam_first, n) of the closest-hit
  // queue are the samples of groups cam_k0, cam_k0 + 1, ... of the workgroup's sequence, 64 positions per group, lane = sample:
  // nothing of them is in the path arrays, every stage rebuilds what it needs from the sample index (camera_ray)
  uint32_t cam_first;             // 0xFFFFFFFF: none (every camera path of the round has explicit records)
  uint32_t cam_k0;
  uint32_t n_cam_dead;            // statistics: positions of those groups that hold no sample (ragged blocks)
  float4 nodes[2 * kHotNodes];    // LDS copy of the hottest BVH nodes (same record layout as DeviceScene::nodes)
#ifdef HJ_LDS_RT_PROBE            // diagnostic: 48 bytes per lane, what a ray's state would occupy if rays were re-grouped through LDS
  uint32_t rt_ctr[4];
  float4 rt[3 * kBlockThreads];
#endif
};

HJ_DEV void load_hot_nodes(const DeviceScene& sc, WgShared& sh) {
  for (uint32_t i = threadIdx.x; i < 2 * sc.num_hot; i += blockDim.x) sh.nodes[i] = sc.nodes[i];
}

// Barrier between two stages of a workgroup.  In the tail of a batch the workgroup is down to ONE wave (the others
// have left the kernel): that wave only has to order its own memory operations and never executes s_barrier again.
HJ_DEV void wg_sync(uint32_t waves) {
  if (waves > 1u) {
    __syncthreads();
  } else {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
}

// Sample groups of workgroup g (a group = 64 consecutive samples of a block row; 256 groups per block).
//   round-robin deal: group k of its sequence is global group g + k * num_wg (every workgroup samples the whole image);
//   XCD deal (st.xcd_deal, needs num_wg == 2048): workgroups are dispatched round-robin over the 8 XCDs, so workgroup g
//   runs on XCD g & 7 (an affinity, used for speed only).  It takes group g >> 3 of every block b with b & 7 == g & 7:
//   an XCD then traces the camera rays (and their shadow rays) of one eighth of the block positions - vertical stripes of
//   the image - and its L2 holds that part of a large scene instead of all of it.
HJ_DEV uint32_t wg_num_groups(const BatchState& st, uint32_t g) {
  if (st.xcd_deal) return st.num_blocks > (g & 7u) ? (st.num_blocks - (g & 7u) + 7u) / 8u : 0u;
  const uint32_t groups = (st.num_blocks * kSlotsPerBlock + 63u) / 64u;
  return groups > g ? (groups - g + st.num_wg - 1u) / st.num_wg : 0u;
}
HJ_DEV uint32_t wg_group(const BatchState& st, uint32_t g, uint32_t k) {
  if (st.xcd_deal) return (8u * k + (g & 7u)) * (kSlotsPerBlock / 64u) + (g >> 3);
  return g + k * st.num_wg;
}

// reference shader/render.glsl:26-36,149-162 for sample `smp` of the batch (block smp / 16384, local pixel from the low bits):
// is the sample inside its block and the image (render.glsl:152 compares the LOCAL id with the image size), its RNG state
// after seedRng(block.seed + lx + ly * dimension.x) and the normalised camera direction (origin = camera.position,
// tMin = eps).  ONE text for the top-up, the packet walk and the shade stage: a camera path that is never written to the
// path arrays (below) is rebuilt from its sample index with exactly these operations.
HJ_DEV bool camera_ray(const BatchState& st, const DeviceScene& sc, uint32_t smp, uint32_t& rng, v3& d) {
  if (smp >= st.num_blocks * kSlotsPerBlock) return false;
  const hj_image_block b = st.blocks[smp / kSlotsPerBlock];
  const uint32_t lx = smp & (HJ_BLOCK_SIZE - 1u);
  const uint32_t ly = (smp / HJ_BLOCK_SIZE) & (HJ_BLOCK_SIZE - 1u);
  if (!(lx < b.dimension[0] && ly < b.dimension[1] && lx < b.original_dimension[0] && ly < b.original_dimension[1])) return false;
  const uint32_t seed = b.seed + lx + ly * b.dimension[0];          // render.glsl:156
  rng = rng_seed(seed);
  const float W = (float)b.original_dimension[0], H = (float)b.original_dimension[1];
  const float px = (float)(lx + b.origin[0]) + b.sample_offset[0];
  const float py = (float)(ly + b.origin[1]) + b.sample_offset[1];
  float x = px - 0.5f * W, y = py - 0.5f * H;
  x = (x * sc.tan_half_fov) / (0.5f * W);
  y = (y * sc.tan_half_fov) / (0.5f * W);
  // quaternionRotate(v, q) = (q (x) (v,0)) (x) conj(q), quaternion.glsl:1-19
  const v3 qv = V(sc.camera.rotation[0], sc.camera.rotation[1], sc.camera.rotation[2]);
  const float qw = sc.camera.rotation[3];
  const v3 vv = V(x, -y, -1.0f);
  const float tw = qw * 0.0f - dot3(qv, vv);
  const v3 c1 = cross3(qv, vv);
  const v3 txyz = V((c1.x + qv.x * 0.0f) + vv.x * qw, (c1.y + qv.y * 0.0f) + vv.y * qw, (c1.z + qv.z * 0.0f) + vv.z * qw);
  const v3 cq = -qv;
  const v3 c2 = cross3(txyz, cq);
  const v3 rot = V((c2.x + txyz.x * qw) + cq.x * tw, (c2.y + txyz.y * qw) + cq.y * tw, (c2.z + txyz.z * qw) + cq.z * tw);
  d = normalize3(rot);
  return true;
}

// EXPLICIT top-up (kernels without the packet stage: linear scan, trees without pair nodes, the split-kernel path): camera
// paths for groups [k0, k0 + ngen) of this workgroup's sample sequence, written to the path arrays of `parity` behind the
// n0 continuing paths (positions n0 + sh.n_gen...; the caller guarantees n0 + 64 * ngen <= pool).
template <bool NT>
HJ_DEV void stage_gen_camera(const BatchState& st, const DeviceScene& sc, uint32_t g, WgShared& sh, uint32_t parity,
                             uint32_t n0, uint32_t k0, uint32_t ngen, uint32_t waves) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t seg = g * st.pool + n0;
  for (uint32_t k = k0 + wave; k < k0 + ngen; k += waves) {
    const uint32_t smp = wg_group(st, g, k) * 64u + lane;
    uint32_t rng = 0;
    v3 d = V(0, 0, 0);
    const bool valid = camera_ray(st, sc, smp, rng, d);
    const uint32_t qi = lds_push(&sh.n_gen, valid);
    if (valid) {
      const uint32_t pos = seg + qi;
      // the sample index rides in origin.w, the RNG state in direction.w
      stp<NT>(st.ray_o[parity], pos, make_float4(sc.camera.position[0], sc.camera.position[1], sc.camera.position[2], __uint_as_float(smp | kCameraFlag)));
      stp<NT>(st.ray_d[parity], pos, make_float4(d.x, d.y, d.z, __uint_as_float(rng)));
      stp<NT>(st.thr[parity], pos, make_float4(1.f, 1.f, 1.f, __uint_as_float(1u)));   // wasDiscrete = true, bounce 0
      if (sc.has_extinction) stp<NT>(st.ext[parity], pos, make_float4(0.f, 0.f, 0.f, 0.f));
      stp<NT>(st.smp_rgb, smp, make_float4(0.f, 0.f, 0.f, 1.f));
      stp<NT>(st.smp_nd, smp, make_float4(0.f, 0.f, 0.f, 0.f));
    }
  }
}

// reference shader/scene.glsl:134-158 with a run-time any-hit switch (linear-scan mode of the merged walk)
HJ_DEV void linear_scan(const DeviceScene& sc, Ray r, RawHit& h, bool any) {
  h.id = -1;
  if (sc.ns > 100 || sc.nq > 100) return;  // scene.glsl:135-138
  const uint32_t total = sc.ns + sc.nq + sc.nt;
  for (uint32_t s = 0; s < total; s++) {
    if (intersect_shape(sc, r, s, h)) {
      h.id = (int)s;
      if (any) return;
      r.tmax = h.t - kEps;
    }
  }
}

// One walk phase for BOTH ray kinds of a round: the n closest-hit rays of the paths in flight (arrays of `parity`)
// and the ns shadow rays that shade produced in the previous round are one queue [0, n + ns).  The two are
// independent (the next bounce ray never waits for the NEE visibility), so tracing them together halves the number
// of walk phases per bounce - each of which ends with the workgroup waiting for its slowest ray - and halves the
// chain of dependent walks of a deep path.  Per path the radiance additions keep the reference's order: NEE of
// bounce k-1 is added during this phase, emission of bounce k in the shade that follows the barrier.  A closest-hit
// ray only records its hit (objectID -1 = miss); an unoccluded shadow ray adds its NEE radiance (render.glsl:122-124).
// Needs sh.head == 0 and the hot nodes loaded (synced).
template <bool USE_BVH, bool PAIRS, bool NT>
HJ_DEV void stage_trace_merged(const BatchState& st, const DeviceScene& sc, uint32_t g, uint32_t parity, uint32_t n,
                               uint32_t ns, WgShared& sh) {
  const uint32_t seg = g * st.pool;
  const float4* __restrict__ ro = st.ray_o[parity] + seg;
  const float4* __restrict__ rd = st.ray_d[parity] + seg;
  uint32_t unocc = 0;                                               // wave-uniform count (statistics)
  // HJ_SHADOW_CARRY: a shadow ray brings its pending NEE contribution along in the registers a closest-hit ray uses for
  // (t, u, v) - an accepted hit ends a shadow ray, so nothing overwrites them while they matter - and its SAMPLE index in
  // `slot`: the finish of an unoccluded shadow ray is then one read-modify-write of the sample instead of two dependent trips.
  // HJ_FETCH_SELECT: loads from selected addresses instead of loads in the two arms of a branch.
  auto fetch = [&](uint32_t i, uint32_t& slot, Ray& r, bool& any, RawHit& h) {
    any = i >= n;
    const uint32_t pos = any ? i - n : i;                           // position in the path / shadow arrays
    slot = pos;
    float4 o, d;
#if HJ_FETCH_SELECT
    o = ldp<NT>(any ? st.sh_o + seg : ro, pos); d = ldp<NT>(any ? st.sh_d + seg : rd, pos);
#if HJ_SHADOW_CARRY
    const float4 cc = ldp<NT>(st.sh_c + seg, any ? pos : 0u);       // (a closest-hit ray's third load is a dummy)
    h.t = any ? cc.x : 0.f; h.u = any ? cc.y : 0.f; h.v = any ? cc.z : 0.f;
    slot = any ? __float_as_uint(cc.w) : pos;
#endif
#else
    if (any) {
      o = ldp<NT>(st.sh_o + seg, pos); d = ldp<NT>(st.sh_d + seg, pos);
#if HJ_SHADOW_CARRY
      const float4 cc = ldp<NT>(st.sh_c + seg, pos);
      h.t = cc.x; h.u = cc.y; h.v = cc.z;
      slot = __float_as_uint(cc.w);
#endif
    } else { o = ldp<NT>(ro, pos); d = ldp<NT>(rd, pos); }
#endif
    h.id = -1;
    r.o = xyz(o); r.d = xyz(d);
    r.tmin = (!any && (__float_as_uint(o.w) & kCameraFlag) != 0u) ? kEps : 2.0f * kEps;   // render.glsl:33,132; scene.glsl:85
    r.tmax = any ? d.w : kInf;
  };
  auto finish = [&](bool done, uint32_t slot, const RawHit& h, bool any) {   // wave-convergent
    if (done && !any) stp<NT>(st.hit + seg, slot, make_float4(h.t, __int_as_float(h.id), h.u, h.v));
    const bool add = done && any && h.id < 0;       // unoccluded shadow ray: render.glsl:123
    if (add) {
#if HJ_SHADOW_CARRY
      float4 s = ldp<NT>(st.smp_rgb, slot);
      s.x += h.t; s.y += h.u; s.z += h.v;
      stp<NT>(st.smp_rgb, slot, s);
#else
      const float4 cc = ldp<NT>(st.sh_c + seg, slot);
      const uint32_t smp = __float_as_uint(cc.w);
      float4 s = ldp<NT>(st.smp_rgb, smp);
      s.x += cc.x; s.y += cc.y; s.z += cc.z;
      stp<NT>(st.smp_rgb, smp, s);
#endif
    }
    unocc += (uint32_t)__popcll(__ballot(add));
  };
  if (USE_BVH) {
    trace_persistent<2, PAIRS>(sc, n + ns, &sh.head, sh.nodes, fetch, finish);
  } else {
    const uint32_t lane = threadIdx.x & 63u;
    for (;;) {
      const uint32_t c = lds_fetch_chunk(&sh.head);
      if (c >= n + ns) break;
      const uint32_t i = c + lane;
      const bool valid = i < n + ns;
      uint32_t slot = 0;
      bool any = false;
      RawHit h; h.t = 0.f; h.u = 0.f; h.v = 0.f; h.id = -1;
      if (valid) {
        Ray r;
        fetch(i, slot, r, any, h);
        linear_scan(sc, r, h, any);
      }
      finish(valid, slot, h, any);
    }
  }
  if ((threadIdx.x & 63u) == 0 && unocc != 0) atomicAdd(&sh.n_unocc, unocc);
}

// PACKET walk of the round's new camera rays: 64 consecutive queue entries (one 64-sample group of a block row when the
// block is full) walk the device tree TOGETHER.  The node index is wave-uniform: a hot node comes from the LDS copy as a
// broadcast read, a cold one and every shape record through the scalar cache into SGPRs - no divergent vector-memory
// instruction at all - and every lane keeps its own state.  A lane whose box test fails at node a notes wake = exit(a) and
// sits out until the wave arrives there: the wave goes down to a's first child when ANY lane entered, to exit(a) otherwise,
// and whichever way it takes through a's subtree it leaves it through exit(a).  Per ray the tested boxes, the tested
// shapes, their order and the tMax of every test are those of the merged walk, i.e. the reference's (scene.glsl:97-133).
// Rays of a packet that point apart only lower the lane fill of the steps, never change a result, so ANY 64 entries may
// form a packet.
//   first / chunks: positions [first, first + 64 * chunks) of the path arrays of `parity`; results = hit records, as the
//   merged walk writes them.  Needs sh.head_cam == 0 and the hot nodes loaded.
#ifndef HJ_CAMERA_PACKETS
#define HJ_CAMERA_PACKETS 1
#endif
typedef const __attribute__((address_space(4))) f4s* ScalarF4;        // constant address space: a uniform index gives an s_load
HJ_DEV float4 lds4(ScalarF4 p, uint32_t i) { const f4s v = p[i]; return make_float4(v.x, v.y, v.z, v.w); }
template <bool NT>
HJ_DEV void stage_camera_packets(const BatchState& st, const DeviceScene& sc, uint32_t g, uint32_t parity, uint32_t first,
                                 uint32_t chunks, uint32_t k0, WgShared& sh) {
  (void)parity;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t seg = g * st.pool;
  const ScalarF4 nodes = (ScalarF4)(uintptr_t)sc.nodes;
  const ScalarF4 tris = (ScalarF4)(uintptr_t)sc.tri_isect;
  const ScalarF4 pairs = (ScalarF4)(uintptr_t)sc.tri_pair;
  const ScalarF4 sphs = (ScalarF4)(uintptr_t)sc.spheres;
  const ScalarF4 quads = (ScalarF4)(uintptr_t)sc.quads;
  const uint32_t nn = sc.num_nodes, nhot = sc.num_hot;
  constexpr uint32_t kAwake = 0xFFFFFFFFu, kNever = 0xFFFFFFFEu;
  uint32_t dead = 0;                         // wave-uniform: positions without a sample
  for (;;) {
    const uint32_t c = lds_fetch_chunk(&sh.head_cam);
    if (c >= 64u * chunks) break;
    const uint32_t pos = seg + first + c + lane;
    // IMPLICIT camera paths: chunk c is group k0 + c / 64 of the workgroup's sample sequence, lane = sample.  The ray is built
    // here (render.glsl:26-36,156-162) and never written: shade rebuilds it for the paths that hit something.  The sample's
    // two layers are initialised here (render.glsl:172-174 writes them whatever the path does).
    const uint32_t smp = wg_group(st, g, k0 + (c >> 6)) * 64u + lane;
    uint32_t rng_unused = 0;
    Ray r;
    r.o = V(sc.camera.position[0], sc.camera.position[1], sc.camera.position[2]);
    r.d = V(0, 0, 0);
    const bool valid = camera_ray(st, sc, smp, rng_unused, r.d);
    dead += 64u - (uint32_t)__popcll(__ballot(valid));
    if (valid) {
      stp<NT>(st.smp_rgb, smp, make_float4(0.f, 0.f, 0.f, 1.f));
      stp<NT>(st.smp_nd, smp, make_float4(0.f, 0.f, 0.f, 0.f));
    }
    r.tmin = kEps;                           // render.glsl:33
    r.tmax = kInf;
    const v3 inv = V(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
    const v3 off = V(-(r.o.x * inv.x), -(r.o.y * inv.y), -(r.o.z * inv.z));
    RawHit h; h.t = 0.f; h.u = 0.f; h.v = 0.f; h.id = -1;
    uint32_t wake = valid ? kAwake : kNever; // the node at which a sleeping lane takes part again (a position without a sample: never)
    uint32_t cur = sc.root;                  // wave-uniform
    while (cur < nn) {
      float4 n0, n1;
      if (cur < nhot) {                      // (uniform address: a broadcast read)
        n0 = sh.nodes[2 * cur]; n1 = sh.nodes[2 * cur + 1];
      } else {
        // (the barrier keeps the compiler from issuing the scalar load ahead of the branch, for hot nodes too - loads from the
        // constant address space may be speculated -, which made every step wait for a trip to the L2)
        asm volatile("" ::: "memory");
        n0 = lds4(nodes, 2 * cur); n1 = lds4(nodes, 2 * cur + 1);
      }
      const uint32_t a = __float_as_uint(n0.w), ex = __float_as_uint(n1.w);
      if (wake == cur) wake = kAwake;
      const bool live = wake == kAwake;
      uint32_t nxt = ex;
      if ((a & kInnerFlag) == 0u) {          // a leaf: its shape is tested by every lane that got here (scene.glsl:105-119)
        if (live) {
          bool hit;
          if (a < sc.ns) {
            hit = intersect_sphere(r, lds4(sphs, a), h);
          } else if (a < sc.ns + sc.nq) {
            const uint32_t q = 3u * (a - sc.ns);
            hit = quad_test(r, lds4(quads, q), lds4(quads, q + 1), lds4(quads, q + 2), h);
          } else {
            const uint32_t t = 3u * (a - sc.ns - sc.nq);
            hit = triangle_test(r, lds4(tris, t), lds4(tris, t + 1), lds4(tris, t + 2), h);
          }
          if (hit) { h.id = (int)a; r.tmax = h.t - kEps; }
        }
      } else {                               // scene.glsl:120-131
        const float tnx = fmaf(n0.x, inv.x, off.x), tpx = fmaf(n1.x, inv.x, off.x);
        const float tny = fmaf(n0.y, inv.y, off.y), tpy = fmaf(n1.y, inv.y, off.y);
        const float tnz = fmaf(n0.z, inv.z, off.z), tpz = fmaf(n1.z, inv.z, off.z);
        const float t0 = f_max(f_max(f_min(tnx, tpx), f_min(tny, tpy)), f_min(tnz, tpz));
        const float t1 = f_min(f_min(f_max(tnx, tpx), f_max(tny, tpy)), f_max(tnz, tpz));
        const bool enter = live && (t0 < t1 + kEps && t0 < r.tmax && t1 > r.tmin);
        const bool any_enter = __ballot(enter) != 0;
        if ((a & kPairFlag) != 0u) {         // a pair node: the lanes that entered test its two triangles, left then right (leaf_test)
          if (any_enter) {
            const uint32_t p = 6u * (a & kIndexMask);
            const float4 A = lds4(pairs, p), B = lds4(pairs, p + 1), C = lds4(pairs, p + 2);
            const float4 D = lds4(pairs, p + 3), E = lds4(pairs, p + 4), F = lds4(pairs, p + 5);
            if (enter) {
              if (triangle_test(r, A, B, C, h)) { h.id = (int)__float_as_uint(A.w); r.tmax = h.t - kEps; }
              if (triangle_test(r, D, E, F, h)) { h.id = (int)__float_as_uint(D.w); r.tmax = h.t - kEps; }
            }
          }
        } else {
          if (live && !enter) wake = ex;     // (asleep until the wave leaves this subtree)
          if (any_enter) nxt = a & kIndexMask;
        }
      }
      cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)nxt);
    }
    stp<NT>(st.hit, pos, make_float4(h.t, __int_as_float(h.id), h.u, h.v));
  }
  if (lane == 0 && dead != 0) atomicAdd(&sh.n_cam_dead, dead);
}

// Ordered compaction of the hits of this workgroup's n closest-hit rays by material tag (divergent-BSDF sort): every
// wave takes a contiguous range of queue rows, counts its hits per tag, then (after a prefix over the waves) writes
// the positions to their final places - queue order, not finishing order, so that the paths a shading wave touches
// stay close together in memory.  Paths whose ray missed are over (render.glsl:94-96): nothing refers to them again.
// Starts with a barrier (all hit records written); needs sh.cnt_hit[] == 0; leaves the tag counts there.
// `waves` = waves of the workgroup that take part.
template <bool NT, uint32_t R>
HJ_DEV void compact_hits_by_tag(const BatchState& st, const DeviceScene& sc, uint32_t g, uint32_t n, WgShared& sh,
                                uint32_t waves) {
  const uint32_t G = st.num_wg;
  const uint32_t lane = threadIdx.x & 63u;
  const float4* __restrict__ hit = st.hit + g * st.pool;
  wg_sync(waves);
  const uint32_t wave = threadIdx.x >> 6;
  const uint32_t rows = (n + 63u) >> 6, rpw = (rows + waves - 1u) / waves;
  const uint32_t r0 = wave * rpw < rows ? wave * rpw : rows, r1 = r0 + rpw < rows ? r0 + rpw : rows;
  // Both passes take R rows per memory trip (hit record, then its material word: two dependent fetches per row, and
  // the ballots keep the compiler from overlapping rows by itself).  R = 4: +1 % (cbox), +1.6 % (spheres).  The
  // pair-node instantiation of the fused kernel keeps R = 1: with more, its register allocation puts two scratch reloads
  // into the walk's leaf phase (-3 % at 1 M triangles; tools/spill_scan.py shows them).
  auto tags4 = [&](uint32_t row, uint32_t tag[R]) {
    int id[R];
#pragma unroll
    for (uint32_t j = 0; j < R; j++) {
      const uint32_t i = (row + j) * 64u + lane;
      id[j] = (row + j < r1 && i < n) ? __float_as_int(ldp<NT>(hit, i).y) : -1;
    }
#pragma unroll
    for (uint32_t j = 0; j < R; j++) tag[j] = id[j] >= 0 ? sc.materials[id[j]] >> HJ_MATERIAL_TAG_SHIFT : 0xFFu;
  };
  uint32_t cnt[kNumTags];
#pragma unroll
  for (uint32_t k = 0; k < kNumTags; k++) cnt[k] = 0;
#pragma unroll 1
  for (uint32_t row = r0; row < r1; row += R) {
    uint32_t tag[R];
    tags4(row, tag);
#pragma unroll
    for (uint32_t j = 0; j < R; j++)
#pragma unroll
      for (uint32_t k = 0; k < kNumTags; k++) cnt[k] += (uint32_t)__popcll(__ballot(tag[j] == k));
  }
  if (lane == 0) {
#pragma unroll
    for (uint32_t k = 0; k < kNumTags; k++) sh.wcnt[wave][k] = cnt[k];
  }
  wg_sync(waves);
  uint32_t base[kNumTags];
#pragma unroll
  for (uint32_t k = 0; k < kNumTags; k++) {
    base[k] = 0;
    for (uint32_t w = 0; w < wave; w++) base[k] += uni(sh.wcnt[w][k]);
  }
#pragma unroll 1
  for (uint32_t row = r0; row < r1; row += R) {
    uint32_t tag[R];
    tags4(row, tag);
#pragma unroll
    for (uint32_t j = 0; j < R; j++) {
      const uint32_t i = (row + j) * 64u + lane;
#pragma unroll
      for (uint32_t k = 0; k < kNumTags; k++) {
        const unsigned long long mask = __ballot(tag[j] == k);
        if (tag[j] == k) st.q_hit[((size_t)k * G + g) * st.pool + base[k] + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = i;
        base[k] += (uint32_t)__popcll(mask);
      }
    }
  }
  if (threadIdx.x < kNumTags) {
    uint32_t total = 0;
    for (uint32_t w = 0; w < waves; w++) total += sh.wcnt[w][threadIdx.x];
    sh.cnt_hit[threadIdx.x] = total;
  }
}

// reference shader/scene.glsl:160-175 (populate), render.glsl:102-144, material.glsl:18-91.
// Shades the hits counted in sh.cnt_hit[] (paths of `parity`); the record of a continuing path is written at the
// next free position of the arrays of parity ^ 1 (sh.n_ray[parity ^ 1]), NEE shadow rays become shadow records
// (sh.n_shadow).
template <bool NT>
HJ_DEV void stage_shade(const BatchState& st, const DeviceScene& sc, uint32_t g, uint32_t parity, uint32_t max_bounces,
                        uint32_t rr_start, WgShared& sh, uint32_t waves) {
  const uint32_t G = st.num_wg;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t seg = g * st.pool;
  const uint32_t np = parity ^ 1u;
  const uint32_t cam_first = uni(sh.cam_first), cam_k0 = uni(sh.cam_k0);
  // one material tag at a time: every wave shades ONE tag (no divergent BSDF switch)
  for (uint32_t tag = 0; tag < kNumTags; tag++) {
    const uint32_t n = uni(sh.cnt_hit[tag]);
    const uint32_t* __restrict__ q = st.q_hit + ((size_t)tag * G + g) * st.pool;
    for (uint32_t base = wave * 64u; base < n; base += waves * 64u) {
    const uint32_t i = base + lane;
    const bool valid = i < n;
    bool alive = false, want_shadow = false;
    v3 T = V(0, 0, 0), wo = V(0, 0, 0), ext = V(0, 0, 0), sdir = V(0, 0, 0), scol = V(0, 0, 0);
    float stmax = 0.f;
    Its its; its.p = V(0, 0, 0);
    uint32_t rng = 0, smp = 0, flags_out = 0;
    if (valid) {
      const uint32_t qpos = q[i];
      const uint32_t slot = seg + qpos;
      const float4 hr = ldp<NT>(st.hit, slot);
      // An IMPLICIT camera path (position >= cam_first: its ray was walked as part of a packet) has no record: origin,
      // direction, RNG state, throughput 1 and "bounce 0, wasDiscrete" follow from its sample index (render.glsl:156-169, 86-90).
      const bool implicit = qpos >= cam_first;
      v3 ro, rd;
      uint32_t flags;
      if (implicit) {
        const uint32_t c = qpos - cam_first;
        smp = wg_group(st, g, cam_k0 + (c >> 6)) * 64u + (c & 63u);
        ro = V(sc.camera.position[0], sc.camera.position[1], sc.camera.position[2]);
        rd = V(0, 0, 0);
        (void)camera_ray(st, sc, smp, rng, rd);
        T = V(1.f, 1.f, 1.f);
        flags = 1u;
      } else {
        const float4 ro4 = ldp<NT>(st.ray_o[parity], slot), rd4 = ldp<NT>(st.ray_d[parity], slot);
        const float4 th4 = ldp<NT>(st.thr[parity], slot);
        ro = xyz(ro4); rd = xyz(rd4);
        T = xyz(th4);
        flags = __float_as_uint(th4.w);
        smp = __float_as_uint(ro4.w) & ~kCameraFlag;
        rng = __float_as_uint(rd4.w);
      }
      const bool was_discrete = (flags & 1u) != 0u;
      const uint32_t bounce = flags >> 1;
      const uint32_t id = (uint32_t)__float_as_int(hr.y);
      its.p = V(fmaf(hr.x, rd.x, ro.x), fmaf(hr.x, rd.y, ro.y), fmaf(hr.x, rd.z, ro.z));   // scene.glsl:164
      if (id < sc.ns) populate_sphere(sc.spheres[id], its);
      else if (id < sc.ns + sc.nq) populate_quad(sc, id - sc.ns, hr.z, hr.w, its);
      else populate_triangle(sc, id - sc.ns - sc.nq, hr.z, hr.w, its);
      if (bounce == 0) stp<NT>(st.smp_nd, smp, make_float4(its.n.x, its.n.y, its.n.z, hr.x));   // render.glsl:102-105
      const uint32_t mat = sc.materials[id];
      const uint32_t midx = mat & HJ_MATERIAL_INDEX_MASK;
      if (sc.has_extinction) {                                                             // render.glsl:111-112
        if (!implicit) ext = xyz(ldp<NT>(st.ext[parity], slot));                           // (a camera path starts with extinction 0)
        const float dist = len3(ro - its.p);
        T = T * V(hj_exp(-ext.x * dist), hj_exp(-ext.y * dist), hj_exp(-ext.z * dist));
      }
      alive = true;
      switch (tag) {
        case HJ_MAT_EMISSIVE: {
          if (was_discrete) {                                                              // render.glsl:114-116
            const v3 e = T * xyz(sc.emissive[midx]);
            float4 s = ldp<NT>(st.smp_rgb, smp);
            s.x += e.x; s.y += e.y; s.z += e.z;
            stp<NT>(st.smp_rgb, smp, s);
          }
          alive = false;   // sampleBSDF weight 0, wo unwritten (material.glsl:88-89)
          break;
        }
        case HJ_MAT_DIFFUSE:
        case HJ_MAT_DIFFUSECBOARD: {
          const v3 color = (tag == HJ_MAT_DIFFUSE) ? xyz(sc.diffuse[midx]) : checkerboard(sc, midx, its.u, its.v);
          const v3 imp = sample_emitter(sc, its.p, rng, sdir, stmax);                      // render.glsl:117-126
          if (len3(imp) > kEps && dot3(sdir, its.n) > 0.0f) {
            const float cs = dot3(its.n, sdir);
            const v3 f = (color * cs) * kInvPi;                                            // material.glsl:18-30
            scol = (T * f) * imp;
            want_shadow = true;
          }
          const v3 l = rand_cos_hemisphere(rng);                                           // material.glsl:37-46
          wo = (its.ft * l.x + its.fb * l.y) + its.n * l.z;
          T = T * color;
          break;
        }
        case HJ_MAT_MIRROR:
          wo = reflect3(rd, its.n);
          break;
        case HJ_MAT_DIELECTRIC: {                                                          // material.glsl:50-87
          const float4 m = sc.dielectric[midx];
          float eta = m.w;
          float etaInv = 1.0f / eta;
          float cosI = -dot3(its.n, rd);
          v3 normal = its.n;
          bool inside = cosI > 0.0f;      // sic
          if (cosI < 0.0f) { eta = etaInv; etaInv = 1.0f / eta; normal = -normal; cosI = -cosI; }
          const float k = 1.0f - (etaInv * etaInv) * (1.0f - cosI * cosI);
          if (k <= 0.0f) {
            wo = reflect3(rd, normal);
          } else {
            const float cosO = __builtin_sqrtf(k);
            const float rpar = (eta * cosI - cosO) / (eta * cosI + cosO);
            const float rorth = (cosI - eta * cosO) / (cosI + eta * cosO);
            const float fr = 0.5f * (rpar * rpar + rorth * rorth);
            if (rng_float(rng) < fr) {
              wo = reflect3(rd, normal);
            } else {
              inside = !inside;
              const v3 par = rd - normal * dot3(rd, normal);
              wo = par * etaInv - normal * cosO;
            }
          }
          if (inside) ext = xyz(m);
          break;
        }
        default:
          alive = false;
          break;
      }
      if (alive) {
        const bool discrete = (tag != HJ_MAT_DIFFUSE && tag != HJ_MAT_DIFFUSECBOARD);     // render.glsl:135
        if (bounce >= rr_start) {                                                          // render.glsl:137-144
          const float qq = f_min(0.99f, f_max(T.x, f_max(T.y, T.z)));
          if (rng_float(rng) > qq) alive = false;
          else T = divs(T, qq);
        }
        if (bounce + 1u >= max_bounces) alive = false;                                     // render.glsl:92
        flags_out = (discrete ? 1u : 0u) | ((bounce + 1u) << 1);
      }
    }
    // the record of a continuing path goes to its position in the next round's arrays (coalesced append)
    const uint32_t qn = lds_push(&sh.n_ray[np], alive);
    if (alive) {
      const uint32_t pos = seg + qn;
      stp<NT>(st.ray_o[np], pos, make_float4(its.p.x, its.p.y, its.p.z, __uint_as_float(smp)));
      stp<NT>(st.ray_d[np], pos, make_float4(wo.x, wo.y, wo.z, __uint_as_float(rng)));
      stp<NT>(st.thr[np], pos, make_float4(T.x, T.y, T.z, __uint_as_float(flags_out)));
      if (sc.has_extinction) stp<NT>(st.ext[np], pos, make_float4(ext.x, ext.y, ext.z, 0.f));
    }
    const uint32_t qs = lds_push(&sh.n_shadow, want_shadow);
    if (want_shadow) {
      const uint32_t pos = seg + qs;
      stp<NT>(st.sh_o, pos, make_float4(its.p.x, its.p.y, its.p.z, 0.f));
      stp<NT>(st.sh_d, pos, make_float4(sdir.x, sdir.y, sdir.z, stmax));
      stp<NT>(st.sh_c, pos, make_float4(scol.x, scol.y, scol.z, __uint_as_float(smp)));
    }
    }
  }
}

// The shade stage as a CALLED function (HJ_SHADE_CALL): shade needs about twice the registers of the walk, and inlined into
// the fused kernel it makes the register allocator of that kernel spill - where, is decided globally, and a single reload
// inside the walk loop costs a memory trip per round of the loop.  As a function of its own it is allocated on its own
// (same register budget: the waves-per-SIMD attribute of the calling kernel is propagated to it), and its spills stay
// inside it.  The batch and scene descriptions are read from the calling kernel's argument segment (every kernel that
// calls this starts with (BatchState, DeviceScene)): scalar loads, as in the kernel itself.
#ifndef HJ_SHADE_CALL
#define HJ_SHADE_CALL 2      // 0: every stage inlined into the fused kernel, 1: shade called, 2: top-up, hit compaction and shade called
#endif
typedef __attribute__((address_space(3))) WgShared* WgSharedLds;
constexpr size_t kSceneArgOffset = (sizeof(BatchState) + alignof(DeviceScene) - 1) / alignof(DeviceScene) * alignof(DeviceScene);
struct KernelArgsHead { BatchState st; DeviceScene sc; };      // how the argument segment of those kernels starts
static_assert(offsetof(KernelArgsHead, sc) == kSceneArgOffset, "DeviceScene's place in the kernel argument segment");
template <bool NT>
__device__ __attribute__((noinline)) void stage_shade_call(uint32_t ka_lo, uint32_t ka_hi, uint32_t g, uint32_t parity, uint32_t max_bounces,
                                                            uint32_t rr_start, uint32_t sh_lds, uint32_t waves) {
  // (the argument-segment pointer comes from the kernel: the intrinsic is null in a called function)
  typedef const __attribute__((address_space(4))) char* KArg;
  KArg ka = (KArg)(((uint64_t)uni(ka_hi) << 32) | (uint64_t)uni(ka_lo));
  const BatchState& st = *(const BatchState*)ka;
  const DeviceScene& sc = *(const DeviceScene*)(ka + kSceneArgOffset);
  WgShared& sh = *(WgShared*)(WgSharedLds)(uintptr_t)uni(sh_lds);
  stage_shade<NT>(st, sc, uni(g), uni(parity), uni(max_bounces), uni(rr_start), sh, uni(waves));
}

template <bool NT>
__device__ __attribute__((noinline)) void stage_gen_camera_call(uint32_t ka_lo, uint32_t ka_hi, uint32_t g, uint32_t sh_lds, uint32_t parity,
                                                                 uint32_t n0, uint32_t k0, uint32_t ngen, uint32_t waves) {
  typedef const __attribute__((address_space(4))) char* KArg;
  KArg ka = (KArg)(((uint64_t)uni(ka_hi) << 32) | (uint64_t)uni(ka_lo));
  const BatchState& st = *(const BatchState*)ka;
  const DeviceScene& sc = *(const DeviceScene*)(ka + kSceneArgOffset);
  WgShared& sh = *(WgShared*)(WgSharedLds)(uintptr_t)uni(sh_lds);
  stage_gen_camera<NT>(st, sc, uni(g), sh, uni(parity), uni(n0), uni(k0), uni(ngen), uni(waves));
}
template <bool NT, uint32_t R>
__device__ __attribute__((noinline)) void compact_hits_call(uint32_t ka_lo, uint32_t ka_hi, uint32_t g, uint32_t n, uint32_t sh_lds, uint32_t waves) {
  typedef const __attribute__((address_space(4))) char* KArg;
  KArg ka = (KArg)(((uint64_t)uni(ka_hi) << 32) | (uint64_t)uni(ka_lo));
  const BatchState& st = *(const BatchState*)ka;
  const DeviceScene& sc = *(const DeviceScene*)(ka + kSceneArgOffset);
  WgShared& sh = *(WgShared*)(WgSharedLds)(uintptr_t)uni(sh_lds);
  compact_hits_by_tag<NT, R>(st, sc, uni(g), uni(n), sh, uni(waves));
}

template <bool NT>
__device__ __attribute__((noinline)) void stage_camera_packets_call(uint32_t ka_lo, uint32_t ka_hi, uint32_t g, uint32_t parity, uint32_t first,
                                                                     uint32_t chunks, uint32_t k0, uint32_t sh_lds) {
  typedef const __attribute__((address_space(4))) char* KArg;
  KArg ka = (KArg)(((uint64_t)uni(ka_hi) << 32) | (uint64_t)uni(ka_lo));
  const BatchState& st = *(const BatchState*)ka;
  const DeviceScene& sc = *(const DeviceScene*)(ka + kSceneArgOffset);
  WgShared& sh = *(WgShared*)(WgSharedLds)(uintptr_t)uni(sh_lds);
  stage_camera_packets<NT>(st, sc, uni(g), uni(parity), uni(first), uni(chunks), uni(k0), sh);
}

// ------------------------------------------------------------------ kernels

// The whole life of a batch in ONE launch.  Every workgroup walks through ITS samples (64-sample groups g, g + G, ...):
// rounds of { top-up: new camera paths behind the continuing ones -> ONE walk phase for the closest-hit rays of the
// paths in flight and the shadow rays of the previous round -> hits compacted by material tag -> shade, which writes
// the continuing paths compacted into the other parity's arrays } on its private segments.  Workgroups never
// exchange data, so there is no grid barrier, no host round trip and no per-stage launch; while one workgroup shades
// (memory bound) its CU neighbours walk the BVH (latency bound).  Path regeneration keeps ~pool paths in flight per
// workgroup until its samples run out; only then do the rounds shrink, and once a round fits one wave the other
// waves leave the kernel (their registers and wave slots start workgroups of the next batch) and wave 0 finishes the
// long paths alone, without barriers.
// Exit condition every wave reaches: no rays, no shadow rays and no samples left (every path ends: a bounce ends it
// with probability >= 1 % from bounce rr_start on, and max_bounces caps it).
#ifndef HJ_TAIL1
#define HJ_TAIL1 128u    // rays of a round at which the workgroup shrinks to one wave (sweep 64..256: within 1 %)
#endif
#ifndef HJ_PATH_WAVES
#define HJ_PATH_WAVES 7   // 72 VGPRs; measured on the compacted-record kernel: 6 waves (80 VGPRs) -6 %, 8 waves (64 VGPRs) -2 %, 5 waves -5 %
#endif
template <bool USE_BVH, bool PAIRS, bool NT>
__global__ __launch_bounds__(kBlockThreads) __attribute__((amdgpu_waves_per_eu(HJ_PATH_WAVES, 8))) void k_path_wavefront(BatchState st, DeviceScene sc, uint32_t max_bounces,
                                                                  uint32_t rr_start) {
  // NT (large trees): the path state is streamed past the caches (ldp / stp)
  __shared__ WgShared sh;
  const uint32_t g = blockIdx.x;
#if HJ_SHADE_CALL
  const uint64_t ka_ = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
  const uint32_t ka_lo = (uint32_t)ka_, ka_hi = (uint32_t)(ka_ >> 32), sh_lds = (uint32_t)(uintptr_t)(WgSharedLds)&sh;
#endif
  // Camera paths without records: kernels that have the packet stage (BVH walk over a tree with pair nodes, stages called)
  constexpr bool IMPLICIT = HJ_CAMERA_PACKETS && HJ_SHADE_CALL >= 2 && USE_BVH && PAIRS;
  uint32_t groups_left = wg_num_groups(st, g);
  uint32_t total_closest = 0, total_shadow = 0, total_hits = 0, total_unocc = 0;   // (thread 0's copies are published)
  if (groups_left != 0) {
    uint32_t k_next = 0;                     // next group of this workgroup's sample sequence
    if (threadIdx.x == 0) { sh.n_ray[0] = 0; sh.n_ray[1] = 0; sh.n_gen = 0; sh.n_shadow = 0; sh.n_unocc = 0; sh.cam_first = 0xFFFFFFFFu; sh.cam_k0 = 0; sh.n_cam_dead = 0; }
    if (USE_BVH) load_hot_nodes(sc, sh);
    uint32_t waves = blockDim.x >> 6;
    wg_sync(waves);
    for (uint32_t parity = 0;; parity ^= 1u) {
      // top-up: new camera paths behind the continuing ones, whole 64-sample groups while they fit.  IMPLICIT (kernels with
      // the packet stage): nothing is written - positions [n0, n0 + 64 * ngen) simply ARE the samples of groups k0 ... of the
      // workgroup's sequence; the packet stage builds their rays and shade rebuilds the paths that hit (camera_ray).
      const uint32_t n0 = uni(sh.n_ray[parity]);
      const uint32_t ngen = min(groups_left, (st.pool - n0) >> 6);
      const uint32_t k0 = k_next;
      if (ngen != 0) {
        if (!IMPLICIT) {
#ifdef HJ_WALK_STATS
          const unsigned long long gen_t0 = wall_clock64();
#endif
#if HJ_SHADE_CALL >= 2
          stage_gen_camera_call<NT>(ka_lo, ka_hi, g, sh_lds, parity, n0, k_next, ngen, waves);
#else
          stage_gen_camera<NT>(st, sc, g, sh, parity, n0, k_next, ngen, waves);
#endif
          wg_sync(waves);
#ifdef HJ_WALK_STATS
          if (threadIdx.x == 0) atomicAdd(&g_round_stats[24], (wall_clock64() - gen_t0) * waves);
#endif
        }
        k_next += ngen;
        groups_left -= ngen;
      }
      const uint32_t n = IMPLICIT ? n0 + 64u * ngen : n0 + uni(sh.n_gen), ns = uni(sh.n_shadow);
      if (n + ns == 0) {
        if (groups_left == 0) break;
        // every sample of the new groups lay outside its block: next groups.  The other parity's path count is the one the
        // round before last left behind (only a round that reaches the reset below clears it): it must not be found again.
        if (threadIdx.x == 0) sh.n_ray[parity ^ 1u] = 0;
        wg_sync(waves);
        continue;
      }
      // Tail of the workgroup: one wave can hold every ray of a round and the counts never grow again.
      if (waves > 1u && groups_left == 0 && n + ns <= HJ_TAIL1) {
        wg_sync(waves);                      // (everyone has read the counts)
        if (threadIdx.x >= 64u) return;
        waves = 1u;
      }
#ifdef HJ_WALK_STATS
      const unsigned long long round_t0 = wall_clock64();
      const uint32_t round_rays = n + ns;
#endif
      wg_sync(waves);                        // everyone has read the counts before they are reset
      if (threadIdx.x == 0) {
        sh.head = 0; sh.head_cam = 0; sh.n_ray[parity ^ 1u] = 0; sh.n_gen = 0; sh.n_shadow = 0; sh.n_unocc = 0;
        sh.cam_first = (IMPLICIT && ngen != 0) ? n0 : 0xFFFFFFFFu; sh.cam_k0 = k0; sh.n_cam_dead = 0;
      }
      if (threadIdx.x < kNumTags) sh.cnt_hit[threadIdx.x] = 0;
      wg_sync(waves);
#ifdef HJ_WALK_STATS
      const unsigned long long st_t0 = wall_clock64();
#endif
      // the round's new camera rays are the LAST entries of the closest-hit queue: they are walked as packets of 64
      // (stage_camera_packets: one group of a block row each), the merged walk takes the continuing paths and the shadow rays
      uint32_t cam = 0;
#if HJ_CAMERA_PACKETS && HJ_SHADE_CALL >= 2
      if (IMPLICIT && ngen != 0) {
        cam = 64u * ngen;
        stage_camera_packets_call<NT>(ka_lo, ka_hi, g, parity, n0, ngen, k0, sh_lds);
      }
#endif
      stage_trace_merged<USE_BVH, PAIRS, NT>(st, sc, g, parity, n - cam, ns, sh);
#ifdef HJ_WALK_STATS
      const unsigned long long st_tw = wall_clock64();       // this wave has no ray left
      wg_sync(waves);                        // (diagnostic build only: the walk ends for all waves before the compaction is timed)
      const unsigned long long st_t1 = wall_clock64();
      if ((threadIdx.x & 63u) == 0) atomicAdd(&g_round_stats[29], st_t1 - st_tw);   // [29] wave time spent waiting for the workgroup's slowest wave
#endif
#if HJ_SHADE_CALL >= 2
      compact_hits_call<NT, 4u>(ka_lo, ka_hi, g, n, sh_lds, waves);
#else
      compact_hits_by_tag<NT, 4u>(st, sc, g, n, sh, waves);
#endif
      wg_sync(waves);
#ifdef HJ_WALK_STATS
      const unsigned long long st_t2 = wall_clock64();
#endif
#if HJ_SHADE_CALL
      if (n != 0) stage_shade_call<NT>(ka_lo, ka_hi, g, parity, max_bounces, rr_start, sh_lds, waves);
#else
      if (n != 0) stage_shade<NT>(st, sc, g, parity, max_bounces, rr_start, sh, waves);
#endif
      total_closest += n - uni(sh.n_cam_dead);   // (positions of ragged blocks' groups that hold no sample are not rays)
      total_shadow += ns;
      for (uint32_t k = 0; k < kNumTags; k++) total_hits += uni(sh.cnt_hit[k]);
      total_unocc += uni(sh.n_unocc);
      wg_sync(waves);
#ifdef HJ_WALK_STATS
      if (threadIdx.x == 0) {
        uint32_t b = 0;
        while (b < 7u && round_rays >= (16u << (2u * b))) b++;      // 16, 64, 256, 1024, 4096, 16384, 65536
        atomicAdd(&g_round_stats[b], 1ull);
        atomicAdd(&g_round_stats[8 + b], (unsigned long long)round_rays);
        const unsigned long long st_t3 = wall_clock64();
        atomicAdd(&g_round_stats[16 + b], (st_t3 - round_t0) * waves);
        atomicAdd(&g_round_stats[25], (st_t1 - st_t0) * waves);
        atomicAdd(&g_round_stats[26], (st_t2 - st_t1) * waves);
        atomicAdd(&g_round_stats[27], (st_t3 - st_t2) * waves);
        atomicAdd(&g_round_stats[28], (st_t0 - round_t0) * waves);
      }
#endif
    }
  }
  if (threadIdx.x == 0) {
    st.acc_closest[g] = total_closest;
    st.acc_shadow[g] = total_shadow;
    st.acc_hits[g] = total_hits;
    st.acc_unoccluded[g] = total_unocc;
  }
}

// ---- split-kernel path (HJ_RENDER_SPLIT_KERNELS): the same stage functions, one launch per stage per bounce, for
// per-stage timing and counters.  No regeneration: the pool holds every sample of the workgroup (api/render.hip sizes
// it so) and k_gen_camera starts them all.

__global__ __launch_bounds__(kBlockThreads) void k_gen_camera(BatchState st, DeviceScene sc) {
  __shared__ WgShared sh;
  const uint32_t g = blockIdx.x;
  if (threadIdx.x == 0) sh.n_gen = 0;
  __syncthreads();
  stage_gen_camera<false>(st, sc, g, sh, 0, 0, 0, wg_num_groups(st, g), blockDim.x >> 6);
  __syncthreads();
  if (threadIdx.x == 0) {
    st.cnt_ray[0][g] = sh.n_gen;
    st.cnt_shadow[g] = 0;
    st.acc_closest[g] = 0;
    st.acc_shadow[g] = 0;
    st.acc_hits[g] = 0;
    st.acc_unoccluded[g] = 0;
  }
}

template <bool USE_BVH>
__global__ __launch_bounds__(kBlockThreads) void k_trace_closest(BatchState st, DeviceScene sc, uint32_t parity) {
  __shared__ WgShared sh;
  const uint32_t g = blockIdx.x;
  const uint32_t n = st.cnt_ray[parity][g];
  if (threadIdx.x == 0) { sh.head = 0; sh.n_unocc = 0; }
  if (threadIdx.x < kNumTags) sh.cnt_hit[threadIdx.x] = 0;
  if (USE_BVH && n != 0) load_hot_nodes(sc, sh);
  __syncthreads();
  stage_trace_merged<USE_BVH, true, false>(st, sc, g, parity, n, 0, sh);
  compact_hits_by_tag<false, 4u>(st, sc, g, n, sh, blockDim.x >> 6);
  __syncthreads();
  if (threadIdx.x < kNumTags) st.cnt_hit[g * kNumTags + threadIdx.x] = sh.cnt_hit[threadIdx.x];
  if (threadIdx.x == 0) {
    st.acc_closest[g] += n;
    uint32_t hits = 0;
    for (uint32_t k = 0; k < kNumTags; k++) hits += sh.cnt_hit[k];
    st.acc_hits[g] += hits;
  }
}

template <bool USE_BVH>
__global__ __launch_bounds__(kBlockThreads) void k_trace_shadow(BatchState st, DeviceScene sc) {
  __shared__ WgShared sh;
  const uint32_t g = blockIdx.x;
  const uint32_t ns = st.cnt_shadow[g];
  if (threadIdx.x == 0) { sh.head = 0; sh.n_unocc = 0; }
  if (USE_BVH && ns != 0) load_hot_nodes(sc, sh);
  __syncthreads();
  stage_trace_merged<USE_BVH, true, false>(st, sc, g, 0, 0, ns, sh);
  __syncthreads();
  if (threadIdx.x == 0) st.acc_unoccluded[g] += sh.n_unocc;
}

__global__ __launch_bounds__(kBlockThreads) void k_shade(BatchState st, DeviceScene sc, uint32_t parity,
                                                         uint32_t max_bounces, uint32_t rr_start) {
  __shared__ WgShared sh;
  const uint32_t g = blockIdx.x;
  if (threadIdx.x == 0) { sh.n_ray[parity ^ 1u] = 0; sh.n_shadow = 0; sh.cam_first = 0xFFFFFFFFu; sh.cam_k0 = 0; }   // (every camera path has records here)
  if (threadIdx.x < kNumTags) sh.cnt_hit[threadIdx.x] = st.cnt_hit[g * kNumTags + threadIdx.x];
  __syncthreads();
  stage_shade<false>(st, sc, g, parity, max_bounces, rr_start, sh, blockDim.x >> 6);
  __syncthreads();
  if (threadIdx.x == 0) {
    st.cnt_ray[parity ^ 1u][g] = sh.n_ray[parity ^ 1u];
    st.cnt_shadow[g] = sh.n_shadow;
    st.acc_shadow[g] += sh.n_shadow;
  }
}

// Probe kernel behind hj_debug_trace: arbitrary rays -> raw hit records.
template <bool USE_BVH, bool ANYHIT>
__global__ __launch_bounds__(kBlockThreads) void k_debug_trace(DeviceScene sc, const float* __restrict__ rays, uint32_t n,
                                                               float4* __restrict__ hits) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* r8 = rays + (size_t)i * 8;
  Ray r; r.o = V(r8[0], r8[1], r8[2]); r.d = V(r8[3], r8[4], r8[5]); r.tmin = r8[6]; r.tmax = r8[7];
  RawHit h; h.t = 0.f; h.u = 0.f; h.v = 0.f;
  const bool hit = traverse<USE_BVH, ANYHIT>(sc, r, h);
  hits[i] = make_float4(__int_as_float(hit ? h.id : -1), hit ? h.t : 0.f, hit ? h.u : 0.f, hit ? h.v : 0.f);
}

// ------------------------------------------------------------ reconstruction

// One thread per output pixel; gathers, IN BLOCK ORDER, what every block of
// the batch splats onto it.  Per-pixel addition order == the reference's
// serial per-block dispatch order (reconstruction.glsl:22-66, main.rs:1316-1355).
// tile_off / tile_blk: for every 16x16 pixel tile the batch's blocks (ascending = list order) whose 2-pixel-extended
// rectangle touches the tile, built on the host while the path kernel runs (CSR layout).
// The 25 Gaussian tap weights of a block (uniform over the block because the sub-pixel offset is per block:
// reconstruction.glsl:27-28,43-44) are formed per (tile, block) in LDS by the first 25 threads.
__global__ __launch_bounds__(256) void k_reconstruct(BatchState st, float stddev,
                                                     const uint32_t* __restrict__ tile_off,
                                                     const uint32_t* __restrict__ tile_blk,
                                                     float4* __restrict__ accum, uint32_t W, uint32_t H) {
  const int tx0 = (int)(blockIdx.x * 16u), ty0 = (int)(blockIdx.y * 16u);
  const int x = tx0 + (int)(threadIdx.x & 15u), y = ty0 + (int)(threadIdx.x >> 4);
  const bool inimg = x < (int)W && y < (int)H;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  bool touched = false;     // pixels no block of this batch reaches are neither read nor written
  const uint32_t tile = blockIdx.y * gridDim.x + blockIdx.x;
  const uint32_t i0 = tile_off[tile], i1 = tile_off[tile + 1];
  // The samples a tile needs from one block (the tile + 2 pixels all round, 20 x 20) are staged in LDS once and
  // the 25 taps of its 256 pixels read them there instead of 50 global fetches per pixel.
  constexpr int TS = 20;
  __shared__ float4 s_rgb[TS * TS], s_nd[TS * TS];
  __shared__ float s_w[25];
  const int px = (int)(threadIdx.x & 15u), py = (int)(threadIdx.x >> 4);
  const float gq = -1.0f / ((2.0f * stddev) * stddev);
  const float c0 = hj_exp(gq * 4.0f);
  for (uint32_t idx = i0; idx < i1; idx++) {
    const uint32_t bi = tile_blk[idx];
    const hj_image_block b = st.blocks[bi];
    const int ox = (int)b.origin[0], oy = (int)b.origin[1], Dx = (int)b.dimension[0], Dy = (int)b.dimension[1];
    const uint32_t sbase = bi * kSlotsPerBlock;
    const int bx0 = tx0 - 2 - ox, by0 = ty0 - 2 - oy;        // block-local coordinates of LDS entry (0, 0)
    __syncthreads();                                          // previous block's taps are done with the LDS tile
    if (threadIdx.x < 25u) {
      const int dx = (int)(threadIdx.x / 5u) - 2, dy = (int)(threadIdx.x % 5u) - 2;
      const float sx = ((float)dx + b.sample_offset[0]) - 0.5f;
      const float sy = ((float)dy + b.sample_offset[1]) - 0.5f;
      s_w[threadIdx.x] = hj_exp(gq * (sx * sx + sy * sy)) - c0;
    }
    for (int e = (int)threadIdx.x; e < TS * TS; e += 256) {
      const int ex = bx0 + e % TS, ey = by0 + e / TS;
      if (ex >= 0 && ex < Dx && ey >= 0 && ey < Dy) {
        const uint32_t sp = sbase + (uint32_t)ey * HJ_BLOCK_SIZE + (uint32_t)ex;
        s_rgb[e] = st.smp_rgb[sp];
        s_nd[e] = st.smp_nd[sp];
      }
    }
    __syncthreads();
    const int lx = x - ox, ly = y - oy;
    if (!inimg || lx < -2 || lx >= Dx + 2 || ly < -2 || ly >= Dy + 2) continue;
    if (!touched) { acc = accum[(size_t)y * W + x]; touched = true; }   // reconstruction.glsl:26
    v3 nc = V(0, 0, 0);
    if (lx >= 0 && lx < Dx && ly >= 0 && ly < Dy) nc = xyz(s_nd[(py + 2) * TS + (px + 2)]);
    for (int dx = -2; dx <= 2; dx++) {
      if (lx + dx < 0 || lx + dx >= Dx) continue;
      for (int dy = -2; dy <= 2; dy++) {
        if (ly + dy < 0 || ly + dy >= Dy) continue;
        float w = s_w[(dx + 2) * 5 + (dy + 2)];
        if (w < 0.0f) continue;
        const int e = (py + 2 + dy) * TS + (px + 2 + dx);
        const float4 nd = s_nd[e];
        const v3 no = xyz(nd) - nc;
        const float dn = dot3(no, no) * 2.0f;
        if (dn != 0.0f) w *= hj_exp(-dn);     // equal normals (flat walls: most taps): hj_exp(-0) == 1 exactly, the product is w
        const float4 c = s_rgb[e];
        const float v0 = w * c.x, v1 = w * c.y, v2 = w * c.z, v3_ = w * c.w;
        if (v0 != v0 || v1 != v1 || v2 != v2 || v3_ != v3_) continue;
        acc.x += v0; acc.y += v1; acc.z += v2; acc.w += v3_;
      }
    }
  }
  if (touched) accum[(size_t)y * W + x] = acc;
}

}  // namespace hj
