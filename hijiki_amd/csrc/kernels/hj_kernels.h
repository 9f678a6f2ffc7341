// Wavefront path-tracing kernels for gfx950 (wave64).
//
// One batch = up to `capacity` camera paths (many ImageBlocks of many passes).  Every path workgroup owns one
// private segment of every queue (hj_device.h) and advances ITS paths one bounce per round:
//
//   per round (hj_stages.h):
//     stage_camera_packets render.glsl:26-36,149-162         NEW camera paths: 64-ray packets built from the sample index (kernels
//                                                            without the packet stage: stage_gen_camera writes explicit records)
//     stage_trace_merged   scene.glsl:92-133                 skip-link BVH walk (hj_walk.h) of the closest-hit rays (-> hit record)
//                                                            and of the previous round's shadow rays (any-hit, boolean-
//                                                            equivalent to the reference's closest-hit; adds NEE radiance)
//     compact_hits_by_tag                                    hits binned by MATERIAL TAG in queue order (wave ballots)
//     stage_shade          scene.glsl:160-175, render.glsl:102-144, material.glsl
//                                                            populate, emission, NEE sample, BSDF sample (hj_shade.h),
//                                                            roulette -> next ray queue + shadow queue
//   k_reconstruct (hj_reconstruct.h)  reconstruction.glsl:22-66
//
// Files: hj_device.h (scene / batch views) <- hj_intersect.h (shape tests, node step) <- hj_walk.h (persistent walk),
// hj_shade.h (populate, emitters) <- hj_stages.h (the stages + their call wrappers) <- this file (the kernels).
//
// k_path_wavefront runs all stages of a batch in ONE persistent launch (workgroup barriers only): the kernel's own code
// is the walk (trace_persistent: in-wave ray replacement, merged first step, bounded burst), the other stages are
// CALLED device functions (stage_*_call: register-allocated on their own, so the walk stays free of spills; inlined they
// made the walk spill - rounds 2-3, profiles/NOTES.md); `rp` = the round timing of the statistics build (hj_walk_probe.h);
// k_gen_camera / k_trace_closest / k_shade / k_trace_shadow launch the stages one by one (diagnostic path).
// No stage uses a global atomic: appends are wave ballot + one LDS atomic.
//
// Every path owns its RNG state, so queue order never changes results; the per-path order of radiance
// additions is the reference's (emission, then NEE, bounce by bounce) because a workgroup's stages are
// separated by barriers (fused) or kernel boundaries (split).
#pragma once
#include "hj_stages.h"
#include "hj_reconstruct.h"

#pragma clang fp contract(off)

namespace hj {

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
  // (the called stages read the batch and scene descriptions from this kernel's argument segment and reach `sh` through its LDS address)
  const uint64_t ka_ = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
  const uint32_t ka_lo = (uint32_t)ka_, ka_hi = (uint32_t)(ka_ >> 32), sh_lds = (uint32_t)(uintptr_t)(WgSharedLds)&sh;
  // Camera paths without records: kernels that have the packet stage (BVH walk over a tree with pair nodes)
  constexpr bool IMPLICIT = USE_BVH && PAIRS;
  RoundProbe rp;
  uint32_t groups_left = wg_num_groups(st, g);
  uint32_t total_closest = 0, total_shadow = 0, total_hits = 0, total_unocc = 0, total_direct = 0;   // (thread 0's copies are published)
  if (groups_left != 0) {
    uint32_t k_next = 0;                     // next group of this workgroup's sample sequence
    if (threadIdx.x == 0) { sh.n_ray[0] = 0; sh.n_ray[1] = 0; sh.n_gen = 0; sh.n_shadow = 0; sh.n_unocc = 0; sh.n_direct = 0; sh.cam_first = 0xFFFFFFFFu; sh.cam_k0 = 0; sh.n_cam_dead = 0; }
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
          rp.gen_begin();
          stage_gen_camera_call<NT>(ka_lo, ka_hi, g, sh_lds, parity, n0, k_next, ngen, waves);
          wg_sync(waves);
          rp.gen_end(waves);
        }
        k_next += ngen;
        groups_left -= ngen;
      }
      const uint32_t n = IMPLICIT ? n0 + 64u * ngen : n0 + uni(sh.n_gen), ns = uni(sh.n_shadow);
      // next-event samples of the previous round's shade that the light-shaft grid answered (intersectScene(shadowRay) == false
      // without a walk): shadow rays of the statistics all the same
      { const uint32_t nd = uni(sh.n_direct); total_shadow += nd; total_unocc += nd; total_direct += nd; }
      if (n + ns == 0) {
        if (groups_left == 0) break;
        // every sample of the new groups lay outside its block: next groups.  The other parity's path count is the one the
        // round before last left behind (only a round that reaches the reset below clears it): it must not be found again.
        if (threadIdx.x == 0) { sh.n_ray[parity ^ 1u] = 0; sh.n_direct = 0; }   // (n_direct: counted above, by thread 0, whose totals are the ones published)
        wg_sync(waves);
        continue;
      }
      // Tail of the workgroup: one wave can hold every ray of a round and the counts never grow again.
      if (waves > 1u && groups_left == 0 && n + ns <= HJ_TAIL1) {
        wg_sync(waves);                      // (everyone has read the counts)
        if (threadIdx.x >= 64u) return;
        waves = 1u;
      }
      rp.round_begin(n + ns);
      wg_sync(waves);                        // everyone has read the counts before they are reset
      if (threadIdx.x == 0) {
        sh.head = 0; sh.head_cam = 0; sh.n_ray[parity ^ 1u] = 0; sh.n_gen = 0; sh.n_shadow = 0; sh.n_unocc = 0; sh.n_direct = 0;
        sh.cam_first = (IMPLICIT && ngen != 0) ? n0 : 0xFFFFFFFFu; sh.cam_k0 = k0; sh.n_cam_dead = 0;
      }
      if (threadIdx.x < kNumTags) sh.cnt_hit[threadIdx.x] = 0;
      wg_sync(waves);
      rp.walk_begin();
      // the round's new camera rays are the LAST entries of the closest-hit queue: they are walked as packets of 64
      // (stage_camera_packets: one group of a block row each), the merged walk takes the continuing paths and the shadow rays
      uint32_t cam = 0;
      if (IMPLICIT && ngen != 0) {
        cam = 64u * ngen;
        stage_camera_packets_call<NT>(ka_lo, ka_hi, g, parity, n0, ngen, k0, sh_lds);
      }
      stage_trace_merged<USE_BVH, PAIRS, NT>(st, sc, g, parity, n - cam, ns, sh);
      rp.walk_end(waves);
      compact_hits_call<NT, 4u>(ka_lo, ka_hi, g, n, sh_lds, waves);
      wg_sync(waves);
      rp.compact_end();
      if (n != 0) stage_shade_call<NT>(ka_lo, ka_hi, g, parity, max_bounces, rr_start, sh_lds, waves);
      total_closest += n - uni(sh.n_cam_dead);   // (positions of ragged blocks' groups that hold no sample are not rays)
      total_shadow += ns;
      for (uint32_t k = 0; k < kNumTags; k++) total_hits += uni(sh.cnt_hit[k]);
      total_unocc += uni(sh.n_unocc);
      wg_sync(waves);
      rp.round_end(waves);
    }
  }
  if (threadIdx.x == 0) {
    st.acc_closest[g] = total_closest;
    st.acc_shadow[g] = total_shadow;
    st.acc_hits[g] = total_hits;
    st.acc_unoccluded[g] = total_unocc;
    st.acc_direct[g] = total_direct;
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
    st.acc_direct[g] = 0;
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
  if (threadIdx.x == 0) { sh.n_ray[parity ^ 1u] = 0; sh.n_shadow = 0; sh.n_direct = 0; sh.cam_first = 0xFFFFFFFFu; sh.cam_k0 = 0; }   // (every camera path has records here)
  if (threadIdx.x < kNumTags) sh.cnt_hit[threadIdx.x] = st.cnt_hit[g * kNumTags + threadIdx.x];
  __syncthreads();
  stage_shade<false>(st, sc, g, parity, max_bounces, rr_start, sh, blockDim.x >> 6);
  __syncthreads();
  if (threadIdx.x == 0) {
    st.cnt_ray[parity ^ 1u][g] = sh.n_ray[parity ^ 1u];
    st.cnt_shadow[g] = sh.n_shadow;
    st.acc_shadow[g] += sh.n_shadow + sh.n_direct;
    st.acc_unoccluded[g] += sh.n_direct;
    st.acc_direct[g] += sh.n_direct;
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

}  // namespace hj
