// The stages of one bounce round of a path workgroup (render.glsl:81-175 cut into wavefront stages): camera rays (explicit
// top-up or 64-ray packets built from the sample index), the merged walk of closest-hit and shadow rays, hit compaction
// by material tag, shade - and their noinline call wrappers (own register allocation per stage).
#pragma once
#include "hj_walk.h"
#include "hj_shade.h"

#pragma clang fp contract(off)

namespace hj {

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
  uint32_t n_direct;              // next-event samples of this round's shade that the light-shaft grid proved unoccluded: added at once, no ray
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

// Sample `lane` of global group `grp`.  A group is 64 samples of one ImageBlock (256 groups per block): 64 consecutive samples of
// a block row, or (tile != 0: DeviceScene::group_tile, large trees) an 8 x 8 pixel tile of the block (tile (grp % 16, grp / 16 % 16),
// lane = 8 * row + column).  The 64 camera rays of a group walk the tree TOGETHER as a packet (stage_camera_packets): over a
// dense mesh an 8 x 8 tile's union of visited nodes is smaller than a 64 x 1 row's (1 M triangles: +4 %), in the box scenes
// the rows' contiguous sample accesses win (c2 / c3: tiles -2.5 / -3 %).  Which samples share a group changes no sample's value.
HJ_DEV uint32_t group_sample(uint32_t grp, uint32_t lane, uint32_t tile) {
  if (tile != 0u) {
    const uint32_t block = grp >> 8, t = grp & 255u;
    const uint32_t lx = ((t & 15u) << 3) | (lane & 7u), ly = ((t >> 4) << 3) | (lane >> 3);
    return block * kSlotsPerBlock + ly * HJ_BLOCK_SIZE + lx;
  }
  return grp * 64u + lane;
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
    const uint32_t smp = group_sample(wg_group(st, g, k), lane, sc.group_tile);
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
  // A shadow ray brings its pending NEE contribution along in the registers a closest-hit ray uses for (t, u, v) - an accepted
  // hit ends a shadow ray, so nothing overwrites them while they matter - and its SAMPLE index in `slot`: the finish of an
  // unoccluded shadow ray is then one read-modify-write of the sample instead of two dependent trips.  The loads come from
  // SELECTED addresses instead of the two arms of a branch (both measured in round 2: profiles/NOTES.md).
  auto fetch = [&](uint32_t i, uint32_t& slot, Ray& r, bool& any, RawHit& h) {
    any = i >= n;
    const uint32_t pos = any ? i - n : i;                           // position in the path / shadow arrays
    slot = pos;
    float4 o, d;
    o = ldp<NT>(any ? st.sh_o + seg : ro, pos); d = ldp<NT>(any ? st.sh_d + seg : rd, pos);
    const float4 cc = ldp<NT>(st.sh_c + seg, any ? pos : 0u);       // (a closest-hit ray's third load is a dummy)
    h.t = any ? cc.x : 0.f; h.u = any ? cc.y : 0.f; h.v = any ? cc.z : 0.f;
    slot = any ? __float_as_uint(cc.w) : pos;
    h.id = -1;
    r.o = xyz(o); r.d = xyz(d);
    r.tmin = (!any && (__float_as_uint(o.w) & kCameraFlag) != 0u) ? kEps : 2.0f * kEps;   // render.glsl:33,132; scene.glsl:85
    r.tmax = any ? d.w : kInf;
  };
  auto finish = [&](bool done, uint32_t slot, const RawHit& h, bool any) {   // wave-convergent
    if (done && !any) stp<NT>(st.hit + seg, slot, make_float4(h.t, __int_as_float(h.id), h.u, h.v));
    const bool add = done && any && h.id < 0;       // unoccluded shadow ray: render.glsl:123
    if (add) {
      float4 s = ldp<NT>(st.smp_rgb, slot);
      s.x += h.t; s.y += h.u; s.z += h.v;
      stp<NT>(st.smp_rgb, slot, s);
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
  PacketProbe pp;
  for (;;) {
    const uint32_t c = lds_fetch_chunk(&sh.head_cam);
    if (c >= 64u * chunks) break;
    const uint32_t pos = seg + first + c + lane;
    // IMPLICIT camera paths: chunk c is group k0 + c / 64 of the workgroup's sample sequence, lane = sample.  The ray is built
    // here (render.glsl:26-36,156-162) and never written: shade rebuilds it for the paths that hit something.  The sample's
    // two layers are initialised here (render.glsl:172-174 writes them whatever the path does).
    const uint32_t smp = group_sample(wg_group(st, g, k0 + (c >> 6)), lane, sc.group_tile);
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
    // (one lane not in general position - hj_intersect.h - and the whole packet walks the reference's own tree: exact for every ray)
    uint32_t cur = __ballot(valid && !general_position(inv, off)) == 0ull ? sc.root : sc.root2;   // wave-uniform
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
      pp.step(wake, cur, sc);
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
  pp.end();
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
  // second pass: the tags the first pass left in hit_tag (a byte per ray: 64 B per row instead of 1 KB + the material gather)
  uint8_t* __restrict__ htag = st.hit_tag + g * st.pool;
  auto tags4_again = [&](uint32_t row, uint32_t tag[R]) {
#pragma unroll
    for (uint32_t j = 0; j < R; j++) {
      const uint32_t i = (row + j) * 64u + lane;
      tag[j] = (row + j < r1 && i < n) ? (uint32_t)htag[i] : 0xFFu;
    }
  };
  uint32_t cnt[kNumTags];
#pragma unroll
  for (uint32_t k = 0; k < kNumTags; k++) cnt[k] = 0;
#pragma unroll 1
  for (uint32_t row = r0; row < r1; row += R) {
    uint32_t tag[R];
    tags4(row, tag);
#pragma unroll
    for (uint32_t j = 0; j < R; j++) {
      const uint32_t i = (row + j) * 64u + lane;
      if (row + j < r1 && i < n) htag[i] = (uint8_t)tag[j];
#pragma unroll
      for (uint32_t k = 0; k < kNumTags; k++) cnt[k] += (uint32_t)__popcll(__ballot(tag[j] == k));
    }
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
    tags4_again(row, tag);
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
    bool alive = false, want_shadow = false, add_now = false;
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
        smp = group_sample(wg_group(st, g, cam_k0 + (c >> 6)), c & 63u, sc.group_tile);
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
      const bool on_its_shape = hit_point_on_its_shape(sc, id, its.p, rd, hr.z, hr.w);      // (for the light-shaft grid, below)
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
          uint32_t em = 0;
          const v3 imp = sample_emitter(sc, its.p, rng, sdir, stmax, em);                  // render.glsl:117-126
          if (len3(imp) > kEps && dot3(sdir, its.n) > 0.0f) {
            const float cs = dot3(its.n, sdir);
            const v3 f = (color * cs) * kInvPi;                                            // material.glsl:18-30
            scol = (T * f) * imp;
            // intersectScene(shadowRay) is known to be false for this cell and emitter (api/light_grid.cpp): the sample is
            // added here, where render.glsl:122-124 adds it, instead of after a walk in the next round
            add_now = shadow_ray_proven_free(sc, its.p, em, on_its_shape);
#ifdef HJ_PROBE_ALL_SHADOW_FREE   // measurement only (WRONG image): no shadow ray is ever walked - the ceiling of everything a visibility structure could prove
            add_now = true;
#endif
            want_shadow = !add_now;
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
    {
      const unsigned long long direct = __ballot(add_now);
      if (direct != 0) {
        if (add_now) {
          float4 sv = ldp<NT>(st.smp_rgb, smp);
          sv.x += scol.x; sv.y += scol.y; sv.z += scol.z;
          stp<NT>(st.smp_rgb, smp, sv);
        }
        if (lane == 0) atomicAdd(&sh.n_direct, (uint32_t)__popcll(direct));
      }
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

// The stages as CALLED functions: shade needs about twice the registers of the walk, and inlined into
// the fused kernel it makes the register allocator of that kernel spill - where, is decided globally, and a single reload
// inside the walk loop costs a memory trip per round of the loop.  As a function of its own it is allocated on its own
// (same register budget: the waves-per-SIMD attribute of the calling kernel is propagated to it), and its spills stay
// inside it.  The batch and scene descriptions are read from the calling kernel's argument segment (every kernel that
// calls this starts with (BatchState, DeviceScene)): scalar loads, as in the kernel itself.
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

}  // namespace hj
