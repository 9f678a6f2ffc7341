// The walk's diagnostic probes, behind ONE hook struct (kernels/hj_walk.h calls `pb.<hook>()` at fixed places).  The shipped
// build uses WalkNoProbe: every hook is an empty inline function, the walk's ISA is what it is without this file
// (tools/isa_diff.py checks that against a git revision).  The diagnostic builds (tools/build_variant.sh NAME -D...) select WalkDiagProbe:
//   HJ_WALK_STATS        wave-level occupancy of the walk's phases (tools/walk_stats.py, profiles/rNN_<c>_walk_stats.txt)
//   HJ_VALU_PROBE=N      N more VALU instructions in every plain box step (tools/valu_probe.sh: is the walk VALU-bound?)
//   HJ_LEAF_VALU_PROBE=N N more VALU instructions in the leaf branch of the merged step
//   HJ_LOAD_PROBE=K      one more 16-byte load per box step (1: one address for the wave, 2: the lane's own node again, 3: a global gather)
//   HJ_WIDE_PROBE        the six other 16-byte parts of the node's 128-byte line: what a 128-byte node would cost per step
//   HJ_LANE_LIMIT=L      only the first L lanes of a wave ever hold a ray: the cost of a wave-step by its active lanes
//   HJ_LDS_RT_PROBE=N    N round trips of a ray's state through LDS per round of the walk loop: what a re-grouping costs at least
// The measurements these gave are in profiles/NOTES.md (rounds 2-4) and DESIGN.md section 6.
#pragma once
#include "hj_intersect.h"

namespace hj {

struct WalkNoProbe {
  static constexpr uint32_t kLanes = 64u;                   // lanes of a wave that may hold a ray
  HJ_DEV void begin() {}
  HJ_DEV void round_begin() {}
  HJ_DEV void refilled(bool) {}
  HJ_DEV void regroup(const float4*, Ray&, RawHit&, uint32_t&, uint32_t&, bool) {}
  HJ_DEV void service_end(bool) {}
  HJ_DEV void merged_begin(bool, bool, uint32_t, uint32_t, uint32_t, uint32_t) {}
  HJ_DEV void merged_end() {}
  HJ_DEV void box_step(uint32_t, uint32_t) {}
  HJ_DEV void box_loads(const DeviceScene&, const float4*, uint32_t, uint32_t, uint32_t) {}
  HJ_DEV void box_valu() {}
  HJ_DEV void steps_end() {}
  HJ_DEV void end() {}
};

// Round timing of the fused kernel (k_path_wavefront) and the camera packets' step counters: empty in the shipped build.
struct RoundNoProbe {
  HJ_DEV void gen_begin() {}
  HJ_DEV void gen_end(uint32_t) {}
  HJ_DEV void round_begin(uint32_t) {}
  HJ_DEV void walk_begin() {}
  HJ_DEV void walk_end(uint32_t) {}
  HJ_DEV void compact_end() {}
  HJ_DEV void round_end(uint32_t) {}
};
struct PacketNoProbe {
  HJ_DEV void step(uint32_t, uint32_t, const DeviceScene&) {}
  HJ_DEV void end() {}
};

#if defined(HJ_WALK_STATS) || defined(HJ_VALU_PROBE) || defined(HJ_LOAD_PROBE) || defined(HJ_LEAF_VALU_PROBE) || defined(HJ_WIDE_PROBE) || \
    defined(HJ_LANE_LIMIT) || defined(HJ_LDS_RT_PROBE)
#define HJ_WALK_DIAG 1

#ifdef HJ_WALK_STATS
// [0] outer iterations [1] inner wave-steps [2] lanes in them [3] leaf phases [4] lanes in them [5] refills
// [6] lanes refilled [7] lanes active at the start of an outer iteration; [29] of g_round_stats: wave time at the barrier behind the walk
__device__ unsigned long long g_walk_stats[16];   // [10..12] wave cycles by phase, [13] total, [14] lane-steps on nodes outside the LDS copy, [15] shape records fetched
// rounds of the fused kernel by size bucket b (rays of the round in [64 * 4^b / 4, 64 * 4^b), b = 0..7):
// [b] rounds, [8 + b] rays, [16 + b] wave-cycles (wall clock of the round x waves of the workgroup still alive)
__device__ unsigned long long g_round_stats[32];   // [0..23] rounds by size; [24..28] wall cycles x waves of top-up, walk, hit compaction, shade, the rest of a round
#endif

struct WalkDiagProbe {
#ifdef HJ_LANE_LIMIT
  static constexpr uint32_t kLanes = (uint32_t)(HJ_LANE_LIMIT);
#else
  static constexpr uint32_t kLanes = 64u;
#endif
  float valu = 1.0f;                                        // the extra instructions' operand (kept live)
  bool leaf_lane = false;                                   // this lane runs a shape test in the merged step
#ifdef HJ_WALK_STATS
  unsigned long long ws[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_begin = 0, t_a = 0, t_b = 0;
  HJ_DEV void stat(int i, long long v) { if (__lane_id() == 0) ws[i] += (unsigned long long)v; }
#else
  HJ_DEV void stat(int, long long) {}
#endif
  HJ_DEV void begin() {
#ifdef HJ_WALK_STATS
    t_begin = clock64();
#endif
  }
  HJ_DEV void round_begin() {
#ifdef HJ_WALK_STATS
    t_a = clock64();                                         // [10] service, [11] box steps, [12] leaf tests: wave cycles by phase
#endif
  }
  HJ_DEV void refilled(bool got) { stat(5, 1); stat(6, __popcll(__ballot(got))); }
  // What ONE re-grouping of the wave's rays through LDS costs at the very least: a queue push (ballot + LDS atomic) and the
  // ray's state (12 dwords here; a design needs 14 or more) written to a slot and read back - here to the lane's own slot
  // (conflict-free; slots picked from a queue would be scattered).  HJ_LDS_RT_PROBE = how many of them per round of the walk loop.
  HJ_DEV void regroup(const float4* s_nodes, Ray& r, RawHit& h, uint32_t& cur, uint32_t& slot, bool active) {
#ifdef HJ_LDS_RT_PROBE
    // (WgShared: rt_ctr and rt follow its node copy, which is what s_nodes points to)
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
#else
    (void)s_nodes; (void)r; (void)h; (void)cur; (void)slot; (void)active;
#endif
  }
  HJ_DEV void service_end(bool active) {
    stat(0, 1); stat(7, __popcll(__ballot(active)));
#ifdef HJ_WALK_STATS
    t_b = clock64();
    stat(10, (long long)(t_b - t_a));
#endif
  }
  // the merged step: its node lanes count as a box step, its leaf lanes as a leaf phase; its wave cycles go to [12]
  HJ_DEV void merged_begin(bool active, bool at_leaf, uint32_t cur, uint32_t nn, uint32_t nhot, uint32_t shape) {
    leaf_lane = active && at_leaf;
#ifdef HJ_WALK_STATS
    const unsigned long long mn = __ballot(active && !at_leaf && cur < nn), mc = __ballot(active && !at_leaf && cur < nn && cur >= nhot);
    const unsigned long long ml = __ballot(active && at_leaf), mp = __ballot(active && at_leaf && (shape & kInnerFlag) != 0u);
    if (__lane_id() == 0) {
      if (mn) { ws[1] += 1; ws[2] += __popcll(mn); ws[14] += __popcll(mc); }
      if (ml) { ws[3] += 1; ws[4] += __popcll(ml); ws[15] += __popcll(ml) + __popcll(mp); }
    }
#else
    (void)active; (void)at_leaf; (void)cur; (void)nn; (void)nhot; (void)shape;
#endif
  }
  HJ_DEV void merged_end() {
#ifdef HJ_LEAF_VALU_PROBE   // extra VALU instructions for the lanes that ran a shape test in the merged step (a pair test has ~130)
    if (leaf_lane) {
#pragma unroll
      for (int k_ = 0; k_ < HJ_LEAF_VALU_PROBE; k_++) asm volatile("v_add_f32 %0, %0, %0" : "+v"(valu));
    }
#endif
#ifdef HJ_WALK_STATS
    stat(12, (long long)(clock64() - t_b));
#endif
  }
  HJ_DEV void box_step(uint32_t cur, uint32_t nhot) {        // (called by the lanes that take the step)
#ifdef HJ_WALK_STATS
    const unsigned long long m = __ballot(true), mc = __ballot(cur >= nhot);     // [14] lane-steps on nodes outside the LDS copy
    if (__lane_id() == (uint32_t)__ffsll((long long)m) - 1u) { ws[1] += 1; ws[2] += __popcll(m); ws[14] += __popcll(mc); }
#else
    (void)cur; (void)nhot;
#endif
  }
  HJ_DEV void box_loads(const DeviceScene& sc, const float4* nd, uint32_t cur, uint32_t a_lo, uint32_t a_hi) {
#ifdef HJ_LOAD_PROBE   // one more 16-byte load per box step; 1: every lane the same address, 2: the lane's own node again, 3: a global (never LDS) address per lane
    {
      const float4* pp = HJ_LOAD_PROBE == 1 ? sc.nodes : HJ_LOAD_PROBE == 2 ? nd : sc.nodes + 2 * cur;
      float4 pv;
      asm volatile("flat_load_dwordx4 %0, %1" : "=v"(pv) : "v"(pp) : "memory");
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      valu += pv.x * 0.0f;
    }
#endif
#ifdef HJ_WIDE_PROBE   // what a 128-byte node would cost per step - the six other 16-byte parts of the node's 128-byte line
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
      for (int k_ = 0; k_ < 6; k_++) valu += pv[k_].x * 0.0f;
      // the destinations must stay live until the wait: a register the compiler considers dead is handed to the next
      // address computation while the load that will overwrite it is still in flight (a build without this faulted)
      asm volatile("" :: "v"(valu));
    }
#endif
    (void)sc; (void)nd; (void)cur; (void)a_lo; (void)a_hi;
  }
  HJ_DEV void box_valu() {
#ifdef HJ_VALU_PROBE
#pragma unroll
    for (int k_ = 0; k_ < HJ_VALU_PROBE; k_++) asm volatile("v_add_f32 %0, %0, %0" : "+v"(valu));
#endif
  }
  HJ_DEV void steps_end() {
#ifdef HJ_WALK_STATS
    stat(11, (long long)(clock64() - t_b));      // (includes the merged step, also counted in [12])
#endif
  }
  HJ_DEV void end() {
#ifdef HJ_WALK_STATS
    stat(13, (long long)(clock64() - t_begin));
    for (int i = 0; i < 16; i++) {      // ws[] lives in whichever lane did the counting: sum over the wave
      unsigned long long v = ws[i];
      for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
      if (__lane_id() == 0 && v) atomicAdd(&g_walk_stats[i], v);
    }
#endif
  }
};
using WalkProbe = WalkDiagProbe;
#else
using WalkProbe = WalkNoProbe;
#endif

#ifdef HJ_WALK_STATS
HJ_DEV unsigned long long probe_wall_clock64() { return __builtin_readcyclecounter(); }
HJ_DEV void probe_wg_sync(uint32_t waves) { if (waves > 1u) __syncthreads(); }
struct RoundStatsProbe {
  unsigned long long gen_t0 = 0, round_t0 = 0, st_t0 = 0, st_t1 = 0, st_t2 = 0;
  uint32_t round_rays = 0;
  HJ_DEV void gen_begin() { gen_t0 = probe_wall_clock64(); }
  HJ_DEV void gen_end(uint32_t waves) { if (threadIdx.x == 0) atomicAdd(&g_round_stats[24], (probe_wall_clock64() - gen_t0) * waves); }
  HJ_DEV void round_begin(uint32_t rays) { round_t0 = probe_wall_clock64(); round_rays = rays; }
  HJ_DEV void walk_begin() { st_t0 = probe_wall_clock64(); }
  HJ_DEV void walk_end(uint32_t waves) {
    const unsigned long long st_tw = probe_wall_clock64();       // this wave has no ray left
    probe_wg_sync(waves);                    // (diagnostic build only: the walk ends for all waves before the compaction is timed)
    st_t1 = probe_wall_clock64();
    if ((threadIdx.x & 63u) == 0) atomicAdd(&g_round_stats[29], st_t1 - st_tw);   // [29] wave time spent waiting for the workgroup's slowest wave
  }
  HJ_DEV void compact_end() { st_t2 = probe_wall_clock64(); }
  HJ_DEV void round_end(uint32_t waves) {
    if (threadIdx.x == 0) {
      uint32_t b = 0;
      while (b < 7u && round_rays >= (16u << (2u * b))) b++;      // 16, 64, 256, 1024, 4096, 16384, 65536
      atomicAdd(&g_round_stats[b], 1ull);
      atomicAdd(&g_round_stats[8 + b], (unsigned long long)round_rays);
      const unsigned long long st_t3 = probe_wall_clock64();
      atomicAdd(&g_round_stats[16 + b], (st_t3 - round_t0) * waves);
      atomicAdd(&g_round_stats[25], (st_t1 - st_t0) * waves);
      atomicAdd(&g_round_stats[26], (st_t2 - st_t1) * waves);
      atomicAdd(&g_round_stats[27], (st_t3 - st_t2) * waves);
      atomicAdd(&g_round_stats[28], (st_t0 - round_t0) * waves);
    }
  }
};
// [8] wave-steps of the camera packets, [9] live lanes in them (beside [1], [2] of the merged walk); their cold steps are scalar
// loads, not lane fetches: counted per wave in g_round_stats[30]
struct PacketStatsProbe {
  unsigned long long pk_steps = 0, pk_lanes = 0, pk_cold = 0;   // (wave-uniform) node steps of the packets, live lanes in them, steps on nodes outside the LDS copy
  HJ_DEV void step(uint32_t wake, uint32_t cur, const DeviceScene& sc) { pk_steps += 1; pk_lanes += (unsigned long long)__popcll(__ballot(wake == 0xFFFFFFFFu));   // (a lane is live when it is awake)
    pk_cold += cur >= sc.num_hot ? 1u : 0u; }
  HJ_DEV void end() {
    if (__lane_id() == 0 && pk_steps != 0) { atomicAdd(&g_walk_stats[8], pk_steps); atomicAdd(&g_walk_stats[9], pk_lanes); atomicAdd(&g_round_stats[30], pk_cold); }
  }
};
using RoundProbe = RoundStatsProbe;
using PacketProbe = PacketStatsProbe;
#else
using RoundProbe = RoundNoProbe;
using PacketProbe = PacketNoProbe;
#endif

}  // namespace hj
