// The persistent BVH walk of the path kernel: in-wave ray replacement, merged first step, bounded burst
// (reference shader/scene.glsl:97-133 per ray), and its diagnostic probes.
#pragma once
#include "hj_intersect.h"

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
      cur = general_position(inv, off) ? sc.root : sc.root2; active = true;
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

}  // namespace hj
