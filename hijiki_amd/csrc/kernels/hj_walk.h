// The persistent BVH walk of the path kernel: in-wave ray replacement, merged first step, bounded burst
// (reference shader/scene.glsl:97-133 per ray).  Its diagnostic probes live in hj_walk_probe.h behind one
// hook struct (`pb`): in the shipped build every hook is an empty inline function.
#pragma once
#include "hj_intersect.h"
#include "hj_walk_probe.h"

#pragma clang fp contract(off)

namespace hj {

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

// MODE 0: closest-hit rays, 1: any-hit (shadow) rays, 2: both kinds in one queue (fetch says which per ray).
// PAIRS: the scene has pair nodes; without them the code for them is not even compiled in (it costs 4 % on a scene that has none).
// (A separate leaf phase behind the box steps instead of the merged first step was the walk of rounds 1-2: -4 % ... -8 %,
// profiles/NOTES.md; the kernels that need no queue - hj_debug_trace - still walk that way: traverse() in hj_intersect.h.)
template <int MODE, bool PAIRS, class Fetch, class Finish>
HJ_DEV void trace_persistent(const DeviceScene& sc, uint32_t n, uint32_t* s_head, const float4* s_nodes,
                             Fetch fetch, Finish finish) {
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
  uint32_t shape = 0, ex = 0;
  bool at_leaf = false;                  // (a leaf reached in one round is tested in the first step of the next)
  WalkProbe pb;
  pb.begin();
  // Merged first step of a round: a lane that reached a leaf in the previous round fetches its SHAPE record in the
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
        if (done) { active = false; pending = true; }   // occluded shadow ray (h.id >= 0 tells finish)
        cur = ex; at_leaf = false;
      } else {
        at_leaf = node_step<PAIRS>(x0, x1, inv, off, r, cur, shape, ex);
      }
    }
  };
  for (;;) {
    pb.round_begin();
    // Service phase: only when enough lanes are free.  Finished lanes keep their result in registers until
    // then, so that result STORES and new-ray LOADS are issued together, once per phase: vmcnt counts loads and
    // stores in one in-order counter on gfx950, and a store between two node fetches would stall the walk for
    // a full write acknowledgement.
    // (WalkProbe::kLanes: 64; fewer only in the lane-limit diagnostic build - constant expressions, folded by the front end)
    const unsigned long long idle = __ballot(!active && (WalkProbe::kLanes == 64u || lane < WalkProbe::kLanes));
    const uint32_t nidle = (uint32_t)__popcll(idle);
    const bool service = nidle >= (WalkProbe::kLanes == 64u ? sc.refill_min : (sc.refill_min * WalkProbe::kLanes + 63u) / 64u) || nidle == WalkProbe::kLanes;
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
      if (!active && (WalkProbe::kLanes == 64u || lane < WalkProbe::kLanes)) {
        const uint32_t my = base + (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
        if (my < n) { fetch(my, slot2, r2, any2, h2); got = true; }
      }
      exhausted = base + nidle >= n;
    }
    if (service) {
      if (__ballot(pending) != 0) finish(pending, slot, h, any);
      pending = false;
      pb.refilled(got);
    }
    if (got) {
      slot = slot2; any = any2; r = r2; h = h2;
      inv = V(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
      off = V(-(r.o.x * inv.x), -(r.o.y * inv.y), -(r.o.z * inv.z));
      cur = general_position(inv, off) ? sc.root : sc.root2; active = true;
    }
    if (__ballot(active || pending) == 0) break;   // (a lane can finish in the merged first step: its result is written by the next service phase)
    pb.regroup(s_nodes, r, h, cur, slot, active);
    pb.service_end(active);
    uint32_t burst = sc.inner_burst;       // lanes standing on a leaf wait at most this many box steps of the others
    pb.merged_begin(active, at_leaf, cur, nn, nhot, shape);
    step0_issue(); step0_compute();
    burst--;
    pb.merged_end();
    while (active && cur < nn && !at_leaf && burst != 0) {
      // hot node: LDS copy, same 32-byte record layout as in HBM, so that ONE address select feeds both 16-byte
      // loads (FLAT loads of base + 32*cur and +16; a per-array `if` compiled to two exec-masked address blocks).
      // The two base addresses sit in four VGPRs (nb_*): v_cndmask cannot take a scalar source beside VCC, and the
      // compiler otherwise re-creates them with four v_mov per step.
      // Neither array crosses a 4 GiB boundary (hj_scene_upload places the node array so; the LDS aperture cannot),
      // so the low word never carries into the high one: cmp + 2 cndmask + 1 shift-add instead of ten instructions.
      const bool hot = cur < nhot;
      pb.box_step(cur, nhot);
      const uint32_t a_lo = (hot ? nb_llo : nb_glo) + (cur << 5), a_hi = hot ? nb_lhi : nb_ghi;
      const float4* __restrict__ nd = reinterpret_cast<const float4*>(((uint64_t)a_hi << 32) | (uint64_t)a_lo);
      const float4 n0 = nd[0], n1 = nd[1];
      pb.box_loads(sc, nd, cur, a_lo, a_hi);
      at_leaf = node_step<PAIRS>(n0, n1, inv, off, r, cur, shape, ex);
      burst--;
      pb.box_valu();
    }
    if (active && !at_leaf && cur >= nn) { active = false; pending = true; }   // walked off the end of the tree
    pb.steps_end();
  }
  pb.end();
}

}  // namespace hj
