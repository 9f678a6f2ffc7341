// libhijiki_hip.so — C ABI (include/hijiki_hip.h) over the gfx950 kernels.
//
// Replaces, for the hot path only, what the reference's Renderer does through
// wgpu (reference src/main.rs:1143-1424): resource creation, scene upload,
// the per-block dispatch loop and the read-back.  No CPU fallback exists: every
// entry point that computes needs a HIP device and fails with HJ_ERR_DEVICE
// otherwise.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/hijiki_hip.h"
#include "host/blockgen.hpp"
#include "kernels/hj_kernels.h"
#include "kernels/hj_lbvh.h"

#include <rocprim/device/device_radix_sort.hpp>

#pragma clang fp contract(off)

namespace {

thread_local std::string g_create_error;

std::atomic<size_t> g_dev_bytes{0};   // device memory held through DevBuf by every context of the process

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  void release() {
    if (p) {
      (void)hipFree(p);
      g_dev_bytes.fetch_sub(bytes, std::memory_order_relaxed);
    }
    p = nullptr;
    bytes = 0;
  }
};

struct EventPair { hipEvent_t a, b; int kind; };

}  // namespace

constexpr uint32_t kMaxSlots = 4;

struct hj_context {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string error;
  int num_cus = 256;

  // scene
  bool have_scene = false;
  hj::DeviceScene scene{};
  std::vector<DevBuf> scene_bufs;

  // framebuffer
  float4* accum = nullptr;
  bool accum_owned = false;
  uint32_t width = 0, height = 0;

  // Batch slots: batch k runs on slot k mod num_slots (own state arrays, own stream), so the latency-bound tail of
  // one batch (a few long paths) overlaps the throughput phase of the next ones.
  struct BatchSlot {
    hj::BatchState st{};
    std::vector<DevBuf> bufs, sample_bufs;   // path-state arrays + queues; per-sample buffers
    DevBuf d_blocks, d_tiles;
    uint32_t* h_tiles = nullptr;          // pinned staging of the per-tile block lists
    size_t h_tiles_cap = 0;
    hipStream_t stream = nullptr;
    hipStream_t rstream = nullptr;        // the reconstruction's stream (high priority: see hj_context_create)
    hipEvent_t ev_path = nullptr;         // this slot's path kernel has finished (the reconstruction stream waits for it)
    hj_image_block* h_blocks = nullptr;   // pinned staging of the block list
    uint32_t h_blocks_cap = 0;
    uint32_t* h_counts = nullptr;         // pinned read-back: 2 (split-path ray counts) + 4 (statistics) arrays of num_wg words
    hipEvent_t ev_count[2] = {nullptr, nullptr};
    hipEvent_t ev_recon = nullptr;        // this slot's reconstruction has run (orders framebuffer updates)
    hipEvent_t ev_done = nullptr;         // batch complete, statistics copied back
    bool pending = false, recon_recorded = false;
    uint32_t nb_in_flight = 0;            // ImageBlocks of the batch in flight (progress reporting)
    uint32_t g_in_flight = 0;             // workgroups of the batch in flight (statistics read-back)
    size_t alloc_positions = 0;           // record positions the path-state arrays hold (workgroups x pool)
  } slots[kMaxSlots];
  uint32_t num_slots = 3;
  uint32_t slots_eff = 3;                // ... the current render call rotates through (1 when device memory is very short)
  uint32_t num_wg = 2048;                // grid size of the path kernels of a large render call (= queue segments), and the most a call uses
  uint32_t num_wg_small = 1536;          // ... of a small one (run_begin)
  uint32_t num_wg_eff = 2048;            // ... of the current call
  uint32_t pool = 65536;                 // path slots per workgroup of the fused kernel (HJ_POOL)
  uint32_t pool_eff = 65536;             // ... as the current render call uses it (lowered when device memory is short)

  // timing
  std::vector<EventPair> events;
  size_t events_used = 0;

  // progress (hj_set_progress_callback): called from the thread that drives the render, when a batch has completed
  hj_progress_fn progress = nullptr;
  void* progress_user = nullptr;
  uint32_t progress_interval = 128;
  uint64_t blocks_total = 0, blocks_done = 0, blocks_reported = 0;

  // hj_render_frame_async: ONE persistent worker thread per context (started by the first asynchronous frame) runs the
  // blocking render; hj_sync waits for it.  `busy` is what every other entry point checks (HJ_ERR_STATE while a frame is
  // in flight); the last frame's result stays retrievable (hj_sync) until the next asynchronous frame starts.
  std::thread worker;
  std::mutex job_mu;
  std::condition_variable job_cv;
  struct AsyncJob { uint32_t spp, pass_begin, pass_end, rank, world; uint64_t master_seed; hj_render_opts opts; } job{};
  bool job_posted = false, worker_exit = false;
  std::atomic<bool> busy{false};
  bool async_valid = false;               // async_rc / async_stats hold a finished frame's result
  int async_rc = HJ_OK;
  hj_render_stats async_stats{};

  // hj_last_error: the worker thread writes `error` while the caller's thread may read it
  std::mutex err_mu;
};

namespace {

int env_int(const char* name, int dflt, int lo, int hi) {
  const char* v = std::getenv(name);
  if (!v || !*v) return dflt;
  return std::min(hi, std::max(lo, std::atoi(v)));
}

int set_error(hj_context* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx) {
    std::lock_guard<std::mutex> lock(ctx->err_mu);
    ctx->error = buf;
  } else {
    g_create_error = buf;
  }
  return code;
}

std::string get_error(hj_context* ctx) {
  std::lock_guard<std::mutex> lock(ctx->err_mu);
  return ctx->error;
}

// Entry points that touch the context's device state refuse to run while an asynchronous frame is in flight on it
// (the worker thread owns the slots, the streams and the framebuffer until hj_sync).
#define HJ_NOT_BUSY(ctx)                                                                                          \
  do {                                                                                                            \
    if ((ctx)->busy.load(std::memory_order_acquire))                                                              \
      return set_error(ctx, HJ_ERR_STATE, "%s: an asynchronous frame is in flight on this context: call hj_sync first", __func__); \
  } while (0)

#define HJ_HIP(ctx, call)                                                                         \
  do {                                                                                            \
    hipError_t e_ = (call);                                                                       \
    if (e_ != hipSuccess)                                                                         \
      return set_error(ctx, e_ == hipErrorOutOfMemory ? HJ_ERR_NOMEM : HJ_ERR_DEVICE, "%s: %s", #call, \
                       hipGetErrorString(e_));                                                    \
  } while (0)

int dev_alloc(hj_context* ctx, DevBuf& b, size_t bytes) {
  if (b.bytes >= bytes && b.p) return HJ_OK;
  b.release();
  if (bytes == 0) bytes = 16;
  // HJ_ALLOC_LIMIT_MB (test rig): the process's contexts together may hold no more than this; an allocation beyond it fails
  // the way hipMalloc does on a full device - how the out-of-memory paths run on a 288 GB card.
  static const size_t limit = (size_t)env_int("HJ_ALLOC_LIMIT_MB", 0, 0, 1 << 30) << 20;
  if (limit != 0 && g_dev_bytes.load(std::memory_order_relaxed) + bytes > limit)
    return set_error(ctx, HJ_ERR_NOMEM, "hipMalloc(%zu bytes): out of memory (HJ_ALLOC_LIMIT_MB)", bytes);
  HJ_HIP(ctx, hipMalloc(&b.p, bytes));
  b.bytes = bytes;
  g_dev_bytes.fetch_add(bytes, std::memory_order_relaxed);
  return HJ_OK;
}

template <class T>
int upload(hj_context* ctx, const T* src, size_t count, const T** out) {
  ctx->scene_bufs.emplace_back();
  DevBuf& b = ctx->scene_bufs.back();
  // 64 bytes of slack: the walk's merged step reads two 16-byte parts of every shape record, a sphere has one
  int rc = dev_alloc(ctx, b, count * sizeof(T) + 64);
  if (rc != HJ_OK) return rc;
  if (count) HJ_HIP(ctx, hipMemcpy(b.p, src, count * sizeof(T), hipMemcpyHostToDevice));
  *out = static_cast<const T*>(b.p);
  return HJ_OK;
}

void release_scene(hj_context* ctx) {
  for (auto& b : ctx->scene_bufs) b.release();
  ctx->scene_bufs.clear();
  ctx->have_scene = false;
}

void release_slot(hj_context::BatchSlot& sl) {
  for (auto& b : sl.bufs) b.release();
  sl.bufs.clear();
  for (auto& b : sl.sample_bufs) b.release();
  sl.sample_bufs.clear();
  sl.alloc_positions = 0;
  sl.st = hj::BatchState{};
}
void release_batch(hj_context* ctx) {
  for (auto& sl : ctx->slots) release_slot(sl);
}

// Same invariants the reference asserts while packing (src/main.rs:562-565)
// plus every index range a kernel dereferences, and the monotonic-exit
// property that makes the skip-link walk terminate on any input.
int validate_scene(hj_context* ctx, const hj_scene_desc* s) {
  const size_t shapes = s->num_spheres + s->num_quads + s->num_triangles;
  if (s->num_materials != shapes)
    return set_error(ctx, HJ_ERR_INVALID, "materials (%zu) != spheres+quads+triangles (%zu) (assert src/main.rs:562-565)",
                     s->num_materials, shapes);
  if (shapes >= 0x7FFFFFFFu || s->num_bvh_nodes >= 0x7FFFFFFFu) return set_error(ctx, HJ_ERR_INVALID, "scene too large");
  auto need = [&](const void* p, size_t n) { return n == 0 || p != nullptr; };
  if (!need(s->bvh, s->num_bvh_nodes) || !need(s->spheres, s->num_spheres) || !need(s->quads, s->num_quads) ||
      !need(s->triangles, s->num_triangles) || !need(s->vertices, s->num_vertices) ||
      !need(s->materials, s->num_materials) || !need(s->emitters, s->num_emitters) ||
      !need(s->diffuse, s->num_diffuse) || !need(s->diffusecb, s->num_diffusecb) ||
      !need(s->dielectric, s->num_dielectric) || !need(s->emissive, s->num_emissive))
    return set_error(ctx, HJ_ERR_INVALID, "null array with non-zero count");
  for (size_t i = 0; i < s->num_bvh_nodes; i++) {
    const hj_bvh_node& n = s->bvh[i];
    if (n.exit_index <= i) return set_error(ctx, HJ_ERR_INVALID, "bvh node %zu: exit index %u does not move forward", i, n.exit_index);
    if (n.shape_index != HJ_BVH_INNER && n.shape_index >= shapes)
      return set_error(ctx, HJ_ERR_INVALID, "bvh node %zu: shape index %u out of range", i, n.shape_index);
  }
  for (size_t i = 0; i < s->num_triangles; i++)
    for (int k = 0; k < 3; k++)
      if (s->triangles[i].v[k] >= s->num_vertices)
        return set_error(ctx, HJ_ERR_INVALID, "triangle %zu refers to vertex %u of %zu", i, s->triangles[i].v[k], s->num_vertices);
  for (size_t i = 0; i < s->num_materials; i++) {
    const uint32_t tag = s->materials[i] >> HJ_MATERIAL_TAG_SHIFT, idx = s->materials[i] & HJ_MATERIAL_INDEX_MASK;
    size_t lim = 0;
    switch (tag) {
      case HJ_MAT_DIFFUSE: lim = s->num_diffuse; break;
      case HJ_MAT_DIFFUSECBOARD: lim = s->num_diffusecb; break;
      case HJ_MAT_MIRROR: lim = 1; break;
      case HJ_MAT_DIELECTRIC: lim = s->num_dielectric; break;
      case HJ_MAT_EMISSIVE: lim = s->num_emissive; break;
      default: return set_error(ctx, HJ_ERR_INVALID, "shape %zu: unknown material tag %u", i, tag);
    }
    if (idx >= lim) return set_error(ctx, HJ_ERR_INVALID, "shape %zu: material index %u out of range for tag %u", i, idx, tag);
  }
  for (size_t i = 0; i < s->num_emitters; i++) {
    const uint32_t sh = s->emitters[i].shape;
    if (sh >= shapes) return set_error(ctx, HJ_ERR_INVALID, "emitter %zu: shape %u out of range", i, sh);
    if ((s->materials[sh] >> HJ_MATERIAL_TAG_SHIFT) != HJ_MAT_EMISSIVE)
      return set_error(ctx, HJ_ERR_INVALID, "emitter %zu: shape %u is not emissive", i, sh);
  }
  return HJ_OK;
}

// Sample buffers for `num_blocks` ImageBlocks and path-state arrays + queues of `pool` slots per workgroup.
// pool: the fused kernel regenerates paths, so a few thousand slots per workgroup keep it busy whatever the batch
// size (ctx->pool, HJ_POOL); the split-kernel path starts every sample of the batch at once and needs them all.
int ensure_batch(hj_context* ctx, hj_context::BatchSlot& sl, uint32_t num_blocks, bool all_in_flight) {
  const uint32_t cap = num_blocks * hj::kSlotsPerBlock;
  const uint32_t G = ctx->num_wg_eff, Gmax = ctx->num_wg;
  const uint32_t per_wg = (((cap + 63u) / 64u + G - 1u) / G) * 64u;     // samples of the busiest workgroup
  const uint32_t pool = all_in_flight ? per_wg : std::min(per_wg, ctx->pool_eff);
  hj::BatchState& st = sl.st;
  int rc = HJ_OK;
  auto alloc = [&](std::vector<DevBuf>& bufs, size_t bytes, void** out) -> int {
    bufs.emplace_back();
    int rc2 = dev_alloc(ctx, bufs.back(), bytes);
    *out = bufs.back().p;
    return rc2;
  };
#define HJ_ALLOC(bufs, field, type, count)                                 \
  if (rc == HJ_OK) {                                                       \
    void* p_ = nullptr;                                                    \
    rc = alloc(bufs, sizeof(type) * (count), &p_);                         \
    st.field = static_cast<type*>(p_);                                     \
  }
  if (st.capacity < cap) {
    for (auto& b : sl.sample_bufs) b.release();
    sl.sample_bufs.clear();
    st.capacity = 0;
    HJ_ALLOC(sl.sample_bufs, smp_rgb, float4, (size_t)cap)
    HJ_ALLOC(sl.sample_bufs, smp_nd, float4, (size_t)cap)
    if (rc == HJ_OK) rc = dev_alloc(ctx, sl.d_blocks, sizeof(hj_image_block) * num_blocks);
    if (rc == HJ_OK && sl.h_blocks_cap < num_blocks) {
      if (sl.h_blocks) (void)hipHostFree(sl.h_blocks);
      sl.h_blocks = nullptr;
      sl.h_blocks_cap = 0;
      if (hipHostMalloc((void**)&sl.h_blocks, sizeof(hj_image_block) * num_blocks, hipHostMallocDefault) != hipSuccess)
        rc = set_error(ctx, HJ_ERR_NOMEM, "pinned block staging allocation failed");
      else sl.h_blocks_cap = num_blocks;
    }
    if (rc == HJ_OK) st.capacity = cap;
  }
  if (rc == HJ_OK && (sl.alloc_positions < (size_t)G * pool || (ctx->scene.has_extinction && !st.ext[0]))) {
    for (auto& b : sl.bufs) b.release();
    sl.bufs.clear();
    sl.alloc_positions = 0;
    const size_t n = (size_t)G * pool;
    for (int par = 0; par < 2; par++) {
      HJ_ALLOC(sl.bufs, ray_o[par], float4, n)
      HJ_ALLOC(sl.bufs, ray_d[par], float4, n)
      HJ_ALLOC(sl.bufs, thr[par], float4, n)
      if (ctx->scene.has_extinction) { HJ_ALLOC(sl.bufs, ext[par], float4, n) }   // (only tinted dielectrics read it)
      else st.ext[par] = nullptr;
    }
    HJ_ALLOC(sl.bufs, hit, float4, n)
    HJ_ALLOC(sl.bufs, q_hit, uint32_t, n * hj::kNumTags)
    HJ_ALLOC(sl.bufs, sh_o, float4, n)
    HJ_ALLOC(sl.bufs, sh_d, float4, n)
    HJ_ALLOC(sl.bufs, sh_c, float4, n)
    HJ_ALLOC(sl.bufs, cnt_ray[0], uint32_t, Gmax)               // (per-workgroup arrays: for the most workgroups a call may use)
    HJ_ALLOC(sl.bufs, cnt_ray[1], uint32_t, Gmax)
    HJ_ALLOC(sl.bufs, cnt_hit, uint32_t, (size_t)Gmax * hj::kNumTags)
    HJ_ALLOC(sl.bufs, cnt_shadow, uint32_t, Gmax)
    HJ_ALLOC(sl.bufs, acc_closest, uint32_t, (size_t)4 * Gmax)  // closest | shadow | hits | unoccluded, one read-back
    if (rc == HJ_OK) sl.alloc_positions = n;
  }
#undef HJ_ALLOC
  if (rc == HJ_OK) {
    st.acc_shadow = st.acc_closest + G;
    st.acc_hits = st.acc_closest + 2 * (size_t)G;
    st.acc_unoccluded = st.acc_closest + 3 * (size_t)G;
    st.pool = pool;
  }
  st.num_wg = G;
  if (rc != HJ_OK) release_slot(sl);
  return rc;
}

enum { EV_CLOSEST = 0, EV_SHADOW = 1, EV_SHADE = 2, EV_RECON = 3, EV_PATH = 4, EV_KINDS = 5 };

struct Timer {
  hj_context* ctx;
  bool on;
  int begin(int kind, hipStream_t s) {
    if (!on) return -1;
    if (ctx->events_used == ctx->events.size()) {
      EventPair ep{};
      if (hipEventCreate(&ep.a) != hipSuccess || hipEventCreate(&ep.b) != hipSuccess) { on = false; return -1; }
      ctx->events.push_back(ep);
    }
    EventPair& ep = ctx->events[ctx->events_used];
    ep.kind = kind;
    (void)hipEventRecord(ep.a, s);
    return (int)ctx->events_used++;
  }
  void end(int idx, hipStream_t s) {
    if (idx >= 0) (void)hipEventRecord(ctx->events[idx].b, s);
  }
};

// Reconstruction of one slot's batch, ordered after the previous batch's (framebuffer sums are defined by block order).
int enqueue_reconstruct(hj_context* ctx, hj_context::BatchSlot& sl, hj_context::BatchSlot& other, const hj::BatchState& st,
                        uint32_t nb, const hj_render_opts& o, Timer& tm) {
  // On its own stream of HIGH priority when there is one: behind the path kernel of its batch (ev_path) and behind the
  // previous batch's reconstruction (ev_recon), but its few hundred short workgroups are dispatched ahead of the waiting
  // workgroups of the other slots' persistent kernels, which otherwise take every wave slot that frees up.
  hipStream_t s = sl.rstream ? sl.rstream : sl.stream;
  if (sl.rstream) {
    HJ_HIP(ctx, hipEventRecord(sl.ev_path, sl.stream));
    HJ_HIP(ctx, hipStreamWaitEvent(s, sl.ev_path, 0));
  }
  // Per 16x16 pixel tile, the blocks of this batch whose extended rectangle touches it, in list order (CSR).
  // Built here on the host, which is idle while the path kernel of this batch runs.
  const uint32_t tw = (ctx->width + 15) / 16, th = (ctx->height + 15) / 16, ntiles = tw * th;
  const int R = 2;
  auto tile_range = [&](const hj_image_block& b, uint32_t& x0, uint32_t& x1, uint32_t& y0, uint32_t& y1) -> bool {
    const long px0 = std::max<long>(0, (long)b.origin[0] - R), py0 = std::max<long>(0, (long)b.origin[1] - R);
    const long px1 = std::min<long>(ctx->width, (long)b.origin[0] + b.dimension[0] + R);
    const long py1 = std::min<long>(ctx->height, (long)b.origin[1] + b.dimension[1] + R);
    if (px0 >= px1 || py0 >= py1) return false;
    x0 = (uint32_t)(px0 / 16); x1 = (uint32_t)((px1 - 1) / 16); y0 = (uint32_t)(py0 / 16); y1 = (uint32_t)((py1 - 1) / 16);
    return true;
  };
  size_t entries = 0;
  for (uint32_t bi = 0; bi < nb; bi++) {
    uint32_t x0, x1, y0, y1;
    if (tile_range(sl.h_blocks[bi], x0, x1, y0, y1)) entries += (size_t)(x1 - x0 + 1) * (y1 - y0 + 1);
  }
  const size_t words = (size_t)ntiles + 1 + entries;
  if (sl.h_tiles_cap < words) {
    if (sl.h_tiles) (void)hipHostFree(sl.h_tiles);
    sl.h_tiles = nullptr;
    sl.h_tiles_cap = 0;
    HJ_HIP(ctx, hipHostMalloc((void**)&sl.h_tiles, sizeof(uint32_t) * words * 2, hipHostMallocDefault));
    sl.h_tiles_cap = words * 2;
  }
  {
    const int rc = dev_alloc(ctx, sl.d_tiles, sizeof(uint32_t) * words);
    if (rc != HJ_OK) return rc;
  }
  uint32_t* off = sl.h_tiles;
  uint32_t* blk = sl.h_tiles + ntiles + 1;
  std::memset(off, 0, sizeof(uint32_t) * (ntiles + 1));
  for (uint32_t bi = 0; bi < nb; bi++) {
    uint32_t x0, x1, y0, y1;
    if (!tile_range(sl.h_blocks[bi], x0, x1, y0, y1)) continue;
    for (uint32_t ty = y0; ty <= y1; ty++)
      for (uint32_t tx = x0; tx <= x1; tx++) off[ty * tw + tx + 1]++;
  }
  for (uint32_t t = 0; t < ntiles; t++) off[t + 1] += off[t];
  {
    std::vector<uint32_t> cur(off, off + ntiles);
    for (uint32_t bi = 0; bi < nb; bi++) {       // ascending bi per tile = the order the reference accumulates in
      uint32_t x0, x1, y0, y1;
      if (!tile_range(sl.h_blocks[bi], x0, x1, y0, y1)) continue;
      for (uint32_t ty = y0; ty <= y1; ty++)
        for (uint32_t tx = x0; tx <= x1; tx++) blk[cur[ty * tw + tx]++] = bi;
    }
  }
  HJ_HIP(ctx, hipMemcpyAsync(sl.d_tiles.p, sl.h_tiles, sizeof(uint32_t) * words, hipMemcpyHostToDevice, s));
  if (other.recon_recorded) HJ_HIP(ctx, hipStreamWaitEvent(s, other.ev_recon, 0));
  const int ev = tm.begin(EV_RECON, s);
  const uint32_t* d_off = static_cast<const uint32_t*>(sl.d_tiles.p);
  hipLaunchKernelGGL(hj::k_reconstruct, dim3(tw, th), dim3(256), 0, s, st, o.recon_stddev, d_off,
                     d_off + ntiles + 1, ctx->accum, ctx->width, ctx->height);
  tm.end(ev, s);
  HJ_HIP(ctx, hipEventRecord(sl.ev_recon, s));
  sl.recon_recorded = true;
  if (sl.rstream) HJ_HIP(ctx, hipStreamWaitEvent(sl.stream, sl.ev_recon, 0));   // the slot's batch ends with its reconstruction
  return HJ_OK;
}

// Wait for a slot's batch and fold its per-workgroup ray counters into the statistics.
int harvest(hj_context* ctx, hj_context::BatchSlot& sl, hj_render_stats* stats, bool count_progress = true) {
  if (!sl.pending) return HJ_OK;
  HJ_HIP(ctx, hipEventSynchronize(sl.ev_done));
  sl.pending = false;
  if (count_progress) ctx->blocks_done += sl.nb_in_flight;   // (hj_debug_samples' batches are not part of a frame)
  sl.nb_in_flight = 0;
  if (count_progress && ctx->progress && ctx->blocks_done - ctx->blocks_reported >= ctx->progress_interval) {   // src/main.rs:1335-1340
    ctx->blocks_reported = ctx->blocks_done;
    ctx->progress(ctx->progress_user, ctx->blocks_done, std::max(ctx->blocks_total, ctx->blocks_done));
  }
  if (stats) {
    const uint32_t G = sl.g_in_flight;
    const uint32_t* h_acc = sl.h_counts + (size_t)2 * G;
    for (uint32_t i = 0; i < G; i++) {
      stats->closest_rays += h_acc[i];
      stats->shadow_rays += h_acc[G + i];
      stats->hits += h_acc[2 * (size_t)G + i];
      stats->unoccluded_shadow_rays += h_acc[3 * (size_t)G + i];
    }
    stats->batches += 1;
  }
  return HJ_OK;
}

int stage_blocks(hj_context* ctx, hj_context::BatchSlot& sl, const hj_image_block* blocks, uint32_t nb, hj::BatchState& st,
                 bool all_in_flight) {
  int rc = ensure_batch(ctx, sl, std::max<uint32_t>(nb, 1), all_in_flight);
  if (rc != HJ_OK) return rc;
  st = sl.st;
  st.blocks = static_cast<const hj_image_block*>(sl.d_blocks.p);
  st.num_blocks = nb;
  static const int xcd_env = env_int("HJ_XCD_DEAL", 0, 0, 1);
  st.xcd_deal = (xcd_env && !all_in_flight && st.num_wg == 2048u && hj::kSlotsPerBlock / 64u == 256u) ? 1u : 0u;
  std::memcpy(sl.h_blocks, blocks, sizeof(hj_image_block) * nb);
  HJ_HIP(ctx, hipMemcpyAsync(sl.d_blocks.p, sl.h_blocks, sizeof(hj_image_block) * nb, hipMemcpyHostToDevice, sl.stream));
  return HJ_OK;
}

int finish_batch(hj_context* ctx, hj_context::BatchSlot& sl, const hj::BatchState& st) {
  const uint32_t G = st.num_wg;
  uint32_t* h_acc = sl.h_counts + (size_t)2 * G;
  HJ_HIP(ctx, hipMemcpyAsync(h_acc, st.acc_closest, sizeof(uint32_t) * 4 * G, hipMemcpyDeviceToHost, sl.stream));
  HJ_HIP(ctx, hipEventRecord(sl.ev_done, sl.stream));
  sl.pending = true;
  sl.nb_in_flight = st.num_blocks;
  sl.g_in_flight = G;
  return HJ_OK;
}

// Default path: ONE persistent launch per batch (k_path_wavefront), asynchronous; the caller keeps num_slots batches in flight.
int enqueue_batch_fused(hj_context* ctx, hj_context::BatchSlot& sl, hj_context::BatchSlot& other, const hj_image_block* blocks,
                        uint32_t nb, const hj_render_opts& o, Timer& tm, hj_render_stats* stats, bool reconstruct) {
  hj::BatchState st;
  int rc = stage_blocks(ctx, sl, blocks, nb, st, false);
  if (rc != HJ_OK) return rc;
  const dim3 blk(hj::kBlockThreads), grid(st.num_wg);
  const int ev = tm.begin(EV_PATH, sl.stream);
  // HJ_LDS_PAD_KB (diagnostic): unused dynamic LDS that lowers the number of resident workgroups per CU without
  // touching the code, to measure how the frame rate scales with occupancy.
  static const size_t lds_pad = (size_t)env_int("HJ_LDS_PAD_KB", 0, 0, 64) * 1024;
  const bool pairs = ctx->scene.has_pairs != 0, nt = ctx->scene.stream_state != 0;
  if (!o.use_bvh) hipLaunchKernelGGL((hj::k_path_wavefront<false, false, false>), grid, blk, lds_pad, sl.stream, st, ctx->scene, o.max_bounces, o.rr_start);
  else if (pairs && nt) hipLaunchKernelGGL((hj::k_path_wavefront<true, true, true>), grid, blk, lds_pad, sl.stream, st, ctx->scene, o.max_bounces, o.rr_start);
  else if (pairs) hipLaunchKernelGGL((hj::k_path_wavefront<true, true, false>), grid, blk, lds_pad, sl.stream, st, ctx->scene, o.max_bounces, o.rr_start);
  else if (nt) hipLaunchKernelGGL((hj::k_path_wavefront<true, false, true>), grid, blk, lds_pad, sl.stream, st, ctx->scene, o.max_bounces, o.rr_start);
  else hipLaunchKernelGGL((hj::k_path_wavefront<true, false, false>), grid, blk, lds_pad, sl.stream, st, ctx->scene, o.max_bounces, o.rr_start);
  tm.end(ev, sl.stream);
  if (reconstruct) {
    rc = enqueue_reconstruct(ctx, sl, other, st, nb, o, tm);
    if (rc != HJ_OK) return rc;
  }
  if (stats) stats->bounce_rounds += 1;
  return finish_batch(ctx, sl, st);
}

// Diagnostic path (HJ_RENDER_SPLIT_KERNELS): one launch per stage per bounce, so that each stage can be timed and
// profiled on its own.  The host learns "all queues empty" from counts copied back one bounce late.
int render_batch_split(hj_context* ctx, hj_context::BatchSlot& sl, hj_context::BatchSlot& other, const hj_image_block* blocks,
                       uint32_t nb, const hj_render_opts& o, Timer& tm, hj_render_stats* stats, bool reconstruct) {
  hj::BatchState st;
  int rc = stage_blocks(ctx, sl, blocks, nb, st, true);
  if (rc != HJ_OK) return rc;
  hipStream_t s = sl.stream;
  const uint32_t G = st.num_wg;
  const dim3 blk(hj::kBlockThreads), grid(G);
  const bool bvh = o.use_bvh != 0;
  hipLaunchKernelGGL(hj::k_gen_camera, grid, blk, 0, s, st, ctx->scene);
  uint64_t rounds = 0;
  auto alive_after = [&](uint32_t b) -> uint64_t {
    const uint32_t* c = sl.h_counts + (size_t)(b & 1u) * G;
    uint64_t sum = 0;
    for (uint32_t i = 0; i < G; i++) sum += c[i];
    return sum;
  };
  static const bool trace_bounces = std::getenv("HJ_TRACE_BOUNCES") != nullptr;   // debugging aid: per-bounce table
  for (uint32_t bounce = 0; bounce < o.max_bounces; bounce++) {
    const uint32_t parity = bounce & 1u;
    const size_t ev0 = ctx->events_used;
    int ev = tm.begin(EV_CLOSEST, s);
    if (bvh) hipLaunchKernelGGL(hj::k_trace_closest<true>, grid, blk, 0, s, st, ctx->scene, parity);
    else hipLaunchKernelGGL(hj::k_trace_closest<false>, grid, blk, 0, s, st, ctx->scene, parity);
    tm.end(ev, s);
    ev = tm.begin(EV_SHADE, s);
    hipLaunchKernelGGL(hj::k_shade, grid, blk, 0, s, st, ctx->scene, parity, o.max_bounces, o.rr_start);
    tm.end(ev, s);
    ev = tm.begin(EV_SHADOW, s);
    if (bvh) hipLaunchKernelGGL(hj::k_trace_shadow<true>, grid, blk, 0, s, st, ctx->scene);
    else hipLaunchKernelGGL(hj::k_trace_shadow<false>, grid, blk, 0, s, st, ctx->scene);
    tm.end(ev, s);
    rounds++;
    HJ_HIP(ctx, hipMemcpyAsync(sl.h_counts + (size_t)parity * G, st.cnt_ray[parity ^ 1u], sizeof(uint32_t) * G,
                               hipMemcpyDeviceToHost, s));
    HJ_HIP(ctx, hipEventRecord(sl.ev_count[parity], s));
    if (trace_bounces && tm.on) {
      HJ_HIP(ctx, hipStreamSynchronize(s));
      std::vector<uint32_t> cur(G), sh(G);
      HJ_HIP(ctx, hipMemcpy(cur.data(), st.cnt_ray[parity], sizeof(uint32_t) * G, hipMemcpyDeviceToHost));
      HJ_HIP(ctx, hipMemcpy(sh.data(), st.cnt_shadow, sizeof(uint32_t) * G, hipMemcpyDeviceToHost));
      uint64_t nc = 0, ns = 0, mx = 0;
      for (uint32_t i = 0; i < G; i++) { nc += cur[i]; ns += sh[i]; mx = std::max<uint64_t>(mx, cur[i]); }
      float t[3] = {0, 0, 0};
      for (int k = 0; k < 3; k++) (void)hipEventElapsedTime(&t[k], ctx->events[ev0 + k].a, ctx->events[ev0 + k].b);
      std::fprintf(stderr, "[bounce %3u] rays %9llu (max/wg %5llu) shadow %9llu | closest %8.1f us  shade %7.1f us  shadow %7.1f us\n",
                   bounce, (unsigned long long)nc, (unsigned long long)mx, (unsigned long long)ns, t[0] * 1e3f, t[1] * 1e3f, t[2] * 1e3f);
    }
    if (bounce >= 1) {
      HJ_HIP(ctx, hipEventSynchronize(sl.ev_count[parity ^ 1u]));
      if (alive_after(bounce - 1) == 0) break;
    }
  }
  if (reconstruct) {
    rc = enqueue_reconstruct(ctx, sl, other, st, nb, o, tm);
    if (rc != HJ_OK) return rc;
  }
  if (stats) stats->bounce_rounds += rounds;
  rc = finish_batch(ctx, sl, st);
  if (rc != HJ_OK) return rc;
  return harvest(ctx, sl, stats);
}

int sync_all(hj_context* ctx) {
  HJ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (auto& sl : ctx->slots) {
    HJ_HIP(ctx, hipStreamSynchronize(sl.stream));
    if (sl.rstream) HJ_HIP(ctx, hipStreamSynchronize(sl.rstream));
    sl.pending = false;
    sl.recon_recorded = false;
  }
  return HJ_OK;
}

int check_opts(hj_context* ctx, const hj_render_opts& o) {
  if (o.recon_radius != 2) return set_error(ctx, HJ_ERR_UNSUPPORTED, "only reconstruction radius 2 (the reference's value) is supported");
  if (!(o.recon_stddev > 0.0f)) return set_error(ctx, HJ_ERR_INVALID, "recon_stddev must be > 0");
  if (o.max_bounces == 0) return set_error(ctx, HJ_ERR_INVALID, "max_bounces must be >= 1");
  return HJ_OK;
}

}  // namespace

static void hj_drop_cached_comms(hj_context* ctx);

extern "C" {

uint32_t hj_version(void) { return (0u << 16) | (2u << 8) | 0u; }

void hj_default_render_opts(hj_render_opts* o) {
  if (!o) return;
  std::memset(o, 0, sizeof *o);
  o->use_bvh = 1;
  o->recon_radius = 2;
  o->recon_stddev = 0.5f;
  o->max_bounces = 1000;
  o->rr_start = 4;
  o->batch_blocks = 0;
}

// The text is copied into a buffer of the CALLING thread (valid until that thread's next hj_last_error call): the
// context's own string may be rewritten by its worker thread at any time.
const char* hj_last_error(const hj_context* ctx) {
  if (!ctx) return g_create_error.c_str();
  thread_local std::string copy;
  copy = get_error(const_cast<hj_context*>(ctx));
  return copy.c_str();
}

// Three batch streams + the context stream want their own hardware queues; the HIP runtime's default is 4 queues
// for the whole process and streams that share one serialise (measured: frames 9 % slower when another HIP user
// of the process had taken queues first).  The runtime reads GPU_MAX_HW_QUEUES when it initialises, i.e. at the
// first HIP call of the process, so this only helps when the library is loaded before that; hosts that initialise
// HIP earlier set the variable themselves.  An existing value is respected.
__attribute__((constructor)) static void hj_default_hw_queues() { (void)setenv("GPU_MAX_HW_QUEUES", "8", 0); }

int hj_context_create(int device, hj_context** out) {
  if (!out) return set_error(nullptr, HJ_ERR_INVALID, "null out pointer");
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0)
    return set_error(nullptr, HJ_ERR_DEVICE, "no HIP device available (%s); this library has no CPU fallback",
                     e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if (device < 0 || device >= count) return set_error(nullptr, HJ_ERR_INVALID, "device ordinal %d out of range [0,%d)", device, count);
  hj_context* ctx = new (std::nothrow) hj_context();
  if (!ctx) return set_error(nullptr, HJ_ERR_NOMEM, "out of host memory");
  ctx->device = device;
  auto fail = [&](hipError_t err, const char* what) {
    set_error(nullptr, HJ_ERR_DEVICE, "%s: %s", what, hipGetErrorString(err));
    hj_context_destroy(ctx);
    return (int)HJ_ERR_DEVICE;
  };
  if ((e = hipSetDevice(device)) != hipSuccess) return fail(e, "hipSetDevice");
  hipDeviceProp_t prop;
  if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return fail(e, "hipGetDeviceProperties");
  ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate");
  // Tuning knobs (environment overrides exist only for sweeps; the defaults are the measured optima).
  ctx->num_wg = (uint32_t)ctx->num_cus * (uint32_t)env_int("HJ_WG_PER_CU", 8, 1, 32);   // 8 x 4 waves = the 32-wave CU limit
  // Small render calls (a rank's share of a frame on many GPUs) run 6 workgroups per CU: all of a kernel's workgroups are then
  // resident at once (7 x 4 waves fit a CU at 72 registers; with 8 per CU the last eighth of a batch's workgroups start when the
  // first ones end, a thin second wave that nothing covers at the end of a short frame) - an 8-rank share of the c2 frame 24.6 ->
  // 23.4 ms; large calls keep 8 (the 32768-block frame: 161 against 170 ms).  HJ_WG_SMALL / HJ_WG_SMALL_BLOCKS.
  ctx->num_wg_small = std::min(ctx->num_wg, (uint32_t)ctx->num_cus * (uint32_t)env_int("HJ_WG_SMALL", 6, 1, 32));
  ctx->num_wg_eff = ctx->num_wg;
  ctx->num_slots = (uint32_t)env_int("HJ_SLOTS", 3, 1, (int)kMaxSlots);
  // positions per workgroup: 65536 = every sample of a workgroup's share of an 8192-block batch in flight at once (the walk
  // phases of a round are long, their ramp-down costs once per round: c2 +6 %, c3 +4 % over 8192 positions with path
  // regeneration; 32768: +4.5 %; 24.7 GB of path state per batch slot, lowered by run_begin when the device is short of memory)
  ctx->pool = (uint32_t)env_int("HJ_POOL", 65536, 64, 1 << 20) / 64u * 64u;
  for (auto& sl : ctx->slots) {
    if ((e = hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate");
    if ((e = hipHostMalloc((void**)&sl.h_counts, sizeof(uint32_t) * 6 * ctx->num_wg, hipHostMallocDefault)) != hipSuccess) return fail(e, "hipHostMalloc");
    for (hipEvent_t* ev : {&sl.ev_count[0], &sl.ev_count[1], &sl.ev_recon, &sl.ev_done, &sl.ev_path})
      if ((e = hipEventCreateWithFlags(ev, hipEventDisableTiming)) != hipSuccess) return fail(e, "hipEventCreate");
  }
  // HJ_RECON_PRIORITY (default 1): the reconstructions run on one stream per slot of the device's highest priority.
  if (env_int("HJ_RECON_PRIORITY", 1, 0, 1) != 0) {
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least)
      for (auto& sl : ctx->slots)
        if (hipStreamCreateWithPriority(&sl.rstream, hipStreamNonBlocking, greatest) != hipSuccess) sl.rstream = nullptr;
  }
  *out = ctx;
  return HJ_OK;
}

void hj_context_destroy(hj_context* ctx) {
  if (!ctx) return;
  if (ctx->worker.joinable()) {              // a frame still in flight finishes first (the worker drains its slots)
    {
      std::lock_guard<std::mutex> lock(ctx->job_mu);
      ctx->worker_exit = true;
    }
    ctx->job_cv.notify_all();
    ctx->worker.join();
  }
  hj_drop_cached_comms(ctx);
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  for (auto& sl : ctx->slots) {
    if (sl.stream) (void)hipStreamSynchronize(sl.stream);
    if (sl.rstream) (void)hipStreamSynchronize(sl.rstream);
  }
  release_scene(ctx);
  release_batch(ctx);
  for (auto& sl : ctx->slots) {
    sl.d_blocks.release();
    sl.d_tiles.release();
    if (sl.h_tiles) (void)hipHostFree(sl.h_tiles);
    if (sl.h_blocks) (void)hipHostFree(sl.h_blocks);
    if (sl.h_counts) (void)hipHostFree(sl.h_counts);
    for (hipEvent_t ev : {sl.ev_count[0], sl.ev_count[1], sl.ev_recon, sl.ev_done, sl.ev_path})
      if (ev) (void)hipEventDestroy(ev);
    if (sl.rstream) (void)hipStreamDestroy(sl.rstream);
    if (sl.stream) (void)hipStreamDestroy(sl.stream);
  }
  if (ctx->accum && ctx->accum_owned) (void)hipFree(ctx->accum);
  for (auto& ep : ctx->events) {
    (void)hipEventDestroy(ep.a);
    (void)hipEventDestroy(ep.b);
  }
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int hj_scene_upload(hj_context* ctx, const hj_scene_desc* s) {
  if (!ctx) return HJ_ERR_INVALID;
  HJ_NOT_BUSY(ctx);
  if (!s) return set_error(ctx, HJ_ERR_INVALID, "null scene");
  int rc = validate_scene(ctx, s);
  if (rc != HJ_OK) return rc;
  HJ_HIP(ctx, hipSetDevice(ctx->device));
  rc = sync_all(ctx);
  if (rc != HJ_OK) return rc;
  release_scene(ctx);

  hj::DeviceScene d{};
  d.camera = s->camera;
  d.tan_half_fov = (float)std::tan((double)(0.5f * s->camera.fov) * (3.14159265358979323846 / 180.0));
  d.ns = (uint32_t)s->num_spheres;
  d.nq = (uint32_t)s->num_quads;
  d.nt = (uint32_t)s->num_triangles;
  d.num_emitters = (uint32_t)s->num_emitters;
  d.num_nodes = (uint32_t)s->num_bvh_nodes;
  d.has_extinction = 0;
  for (size_t i = 0; i < s->num_dielectric; i++)
    if (s->dielectric[i].extinction[0] != 0.f || s->dielectric[i].extinction[1] != 0.f || s->dielectric[i].extinction[2] != 0.f)
      d.has_extinction = 1;

  // pre-gathered triangle records (see kernels/hj_device.h)
  std::vector<float4> isect, shade;
  try {
    isect.resize(3 * s->num_triangles);
    shade.resize(4 * s->num_triangles);
  } catch (const std::bad_alloc&) {
    return set_error(ctx, HJ_ERR_NOMEM, "out of host memory");
  }
  for (size_t i = 0; i < s->num_triangles; i++) {
    const hj_vertex& A = s->vertices[s->triangles[i].v[0]];
    const hj_vertex& B = s->vertices[s->triangles[i].v[1]];
    const hj_vertex& C = s->vertices[s->triangles[i].v[2]];
    isect[3 * i + 0] = make_float4(A.pos[0], A.pos[1], A.pos[2], 0.f);
    isect[3 * i + 1] = make_float4(B.pos[0] - A.pos[0], B.pos[1] - A.pos[1], B.pos[2] - A.pos[2], 0.f);
    isect[3 * i + 2] = make_float4(C.pos[0] - A.pos[0], C.pos[1] - A.pos[1], C.pos[2] - A.pos[2], 0.f);
    shade[4 * i + 0] = make_float4(A.normal[0], A.normal[1], A.normal[2], A.u);
    shade[4 * i + 1] = make_float4(B.normal[0], B.normal[1], B.normal[2], B.u);
    shade[4 * i + 2] = make_float4(C.normal[0], C.normal[1], C.normal[2], C.u);
    shade[4 * i + 3] = make_float4(A.v, B.v, C.v, 0.f);
  }
  static_assert(sizeof(hj_bvh_node) == 2 * sizeof(float4), "node = 2 x float4");
  static_assert(sizeof(hj_quad) == 3 * sizeof(float4) && sizeof(hj_sphere) == sizeof(float4), "shape records");
  static_assert(sizeof(hj_diffuse_cb) == 2 * sizeof(float4), "checkerboard record");
#define HJ_UP(expr) do { rc = (expr); if (rc != HJ_OK) { release_scene(ctx); return rc; } } while (0)
  // device node array (kernels/hj_device.h): redundant inner nodes dropped, hottest (largest surface area) nodes
  // first, explicit left/exit links.
  {
    const size_t N = s->num_bvh_nodes;
    std::vector<float> sa(N);
    for (size_t i = 0; i < N; i++) {
      const float dx = s->bvh[i].aabb_max[0] - s->bvh[i].aabb_min[0], dy = s->bvh[i].aabb_max[1] - s->bvh[i].aabb_min[1],
                  dz = s->bvh[i].aabb_max[2] - s->bvh[i].aabb_min[2];
      sa[i] = (dx >= 0 && dy >= 0 && dz >= 0) ? dx * dy + dy * dz + dz * dx : 0.f;
    }
    // Collapse: an inner node P whose two children are inner nodes can be removed from the walk without changing
    // which leaves are tested, in which order, with which tMax: a child box lies inside P's box and every term of
    // the slab test is monotone in the bounds, so "child passes => P passes" and "P fails => both children fail";
    // the children keep their own box tests and exits.  (Not for leaf children: a leaf's box is never tested, so
    // P's test is the only guard in front of its shape test.)  It pays when P usually passes: with pass
    // probability p ~ area(P) / area(nearest kept ancestor), testing P costs 1 + 2p box tests against 2 without.
    std::vector<char> del(N, 0);
    {
      const float thr = (float)env_int("HJ_COLLAPSE_PCT", 50, 0, 1000) / 100.0f;
      auto inner = [&](size_t i) { return s->bvh[i].shape_index == HJ_BVH_INNER; };
      auto inside = [&](size_t c, size_t p) {   // false for NaN bounds
        bool ok = true;
        for (int k = 0; k < 3; k++)
          ok = ok && s->bvh[c].aabb_min[k] >= s->bvh[p].aabb_min[k] && s->bvh[c].aabb_max[k] <= s->bvh[p].aabb_max[k];
        return ok;
      };
      std::vector<float> anc(N, 0.f);   // area of the nearest kept ancestor
      for (size_t i = 0; i < N; i++) {  // pre-order: ancestors come first
        if (!inner(i) || i + 1 >= N) continue;
        const size_t l = i + 1, r = s->bvh[l].exit_index;
        if (r >= N || r <= l) continue;                       // not a well-formed pre-order pair: leave it alone
        // (an uploaded tree whose child boxes stick out of P's box keeps P: the argument above needs containment)
        if (i != 0 && inner(l) && inner(r) && anc[i] > 0.f && sa[i] > thr * anc[i] && inside(l, i) && inside(r, i)) del[i] = 1;
        anc[l] = anc[r] = del[i] ? anc[i] : sa[i];
      }
    }
    // Pair nodes (kernels/hj_kernels.h leaf_test): an inner node whose two children are triangle leaves keeps its
    // record, the two leaves lose theirs (nothing but the pair's own walk ever reaches them: the left one is the
    // node's first child, the right one the left one's exit) and their triangles go side by side into `pairs`.
    std::vector<uint32_t> pair_of(N, 0xFFFFFFFFu);
    std::vector<float4> pairs;
    // A fifth fewer dependent fetch rounds per ray.  Before the walk's merged first step (hj_kernels.h) the longer leaf phase
    // - two tests while the rest of the wave waits - cost more than the rounds saved on cache-resident scenes (-3 % on the
    // 6 k-triangle box against +12 % at 1 M triangles); with the shape fetch riding along with the other lanes' node fetch
    // they pay everywhere: 6 k triangles +3 %, with the spheres +5 %, 60 k +7 %, 200 k +8 %.  HJ_PAIR_LEAVES = 0 / 1 forces;
    // default: trees of >= HJ_PAIR_MIN_NODES records (0: all).
    const int pair_env = env_int("HJ_PAIR_LEAVES", -1, -1, 1);
    if (pair_env == 1 || (pair_env < 0 && N >= (size_t)env_int("HJ_PAIR_MIN_NODES", 0, 0, 1 << 30))) {
      const size_t first_tri = s->num_spheres + s->num_quads;
      for (size_t i = 0; i + 2 < N; i++) {
        if (s->bvh[i].shape_index != HJ_BVH_INNER) continue;
        const size_t l = i + 1, r = s->bvh[l].exit_index;
        if (r >= N || r != l + 1) continue;
        const uint32_t sl = s->bvh[l].shape_index, sr = s->bvh[r].shape_index;
        if (sl == HJ_BVH_INNER || sr == HJ_BVH_INNER || sl < first_tri || sr < first_tri) continue;
        if (s->bvh[r].exit_index != s->bvh[i].exit_index) continue;      // (a well-formed tree: the right child's exit is its parent's)
        pair_of[i] = (uint32_t)(pairs.size() / 6);
        for (uint32_t sh : {sl, sr}) {
          const size_t t = sh - first_tri;
          float4 a = isect[3 * t];
          a.w = __builtin_bit_cast(float, sh);
          pairs.push_back(a); pairs.push_back(isect[3 * t + 1]); pairs.push_back(isect[3 * t + 2]);
        }
        del[l] = del[r] = 1;                                              // no records for the two leaves
      }
    }
    auto resolve = [&](size_t i) { while (i < N && del[i]) i++; return i; };   // first kept node of a subtree
    std::vector<uint32_t> order, map(N, 0);
    for (size_t i = 0; i < N; i++) if (!del[i]) order.push_back((uint32_t)i);
    const size_t M = order.size();
    const uint32_t hot = (uint32_t)std::min<size_t>(hj::kHotNodes, M);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return sa[a] > sa[b]; });
    std::vector<char> is_hot(N, 0);
    for (size_t k = 0; k < hot; k++) { map[order[k]] = (uint32_t)k; is_hot[order[k]] = 1; }
    // (a treelet-blocked order - a node and its largest descendants per 128-byte line - was measured on the 1 M-triangle
    // scene before: within 1 % at 4, 8 and 16 records per treelet)
    uint32_t next = hot;
    const int node_order = env_int("HJ_NODE_ORDER", -1, -1, 1);                      // -1: by tree size (with the pair nodes)
    if (node_order == 0 || (node_order < 0 && pairs.empty())) {
      for (size_t i = 0; i < N; i++) if (!del[i] && !is_hot[i]) map[i] = next++;   // small trees (cache-resident): pre-order
    } else {
      // Large trees: the children of a node side by side, a group of two or more starting on a 64-byte sector - the walk
      // always goes from a child to its sibling (the child's exit), so the sibling's record comes with the child's; the
      // group of the larger child (the likelier visit) follows directly.  Unused slots (padding) are never referenced.
      // 1 M triangles: +2.4 %, 200 k: +1 %; cbox (forced): -0.6 %.
      std::vector<uint32_t> stack, kids;
      if (N && !is_hot[0]) map[0] = next++;
      if (N) stack.push_back(0);
      while (!stack.empty()) {
        const uint32_t i = stack.back();
        stack.pop_back();
        const hj_bvh_node& nd = s->bvh[i];
        if (nd.shape_index != HJ_BVH_INNER || pair_of[i] != 0xFFFFFFFFu) continue;
        const size_t end = nd.exit_index < N ? resolve(nd.exit_index) : N;
        kids.clear();
        for (size_t c = resolve((size_t)i + 1); c < N && c != end;) {
          kids.push_back((uint32_t)c);
          const uint32_t e = s->bvh[c].exit_index;
          c = e < N ? resolve(e) : N;
        }
        uint32_t cold = 0;
        for (uint32_t c : kids) cold += is_hot[c] ? 0u : 1u;
        if (cold >= 2 && (next & 1u)) next++;
        for (uint32_t c : kids) if (!is_hot[c]) map[c] = next++;
        std::stable_sort(kids.begin(), kids.end(), [&](uint32_t x, uint32_t y) { return sa[x] > sa[y]; });
        for (size_t k = kids.size(); k-- > 0;) stack.push_back(kids[k]);
      }
    }
    const size_t M_all = next;                                                       // records incl. padding
    std::vector<float4> dev(2 * M_all, make_float4(0.f, 0.f, 0.f, 0.f));
    for (size_t i = 0; i < N; i++) {
      if (del[i]) continue;
      const hj_bvh_node& nd = s->bvh[i];
      uint32_t a;
      if (nd.shape_index != HJ_BVH_INNER) a = nd.shape_index;
      else if (pair_of[i] != 0xFFFFFFFFu) a = hj::kInnerFlag | hj::kPairFlag | pair_of[i];
      else {
        const size_t l = resolve(i + 1);                                             // left child = next pre-order record
        a = hj::kInnerFlag | (l < N ? map[l] : (uint32_t)M_all);
      }
      const size_t e = nd.exit_index < N ? resolve(nd.exit_index) : N;
      const uint32_t b = e < N ? map[e] : (uint32_t)M_all;                           // >= the record count ends the walk
      float4* rec = &dev[2 * (size_t)map[i]];
      rec[0] = make_float4(nd.aabb_min[0], nd.aabb_min[1], nd.aabb_min[2], __builtin_bit_cast(float, a));
      rec[1] = make_float4(nd.aabb_max[0], nd.aabb_max[1], nd.aabb_max[2], __builtin_bit_cast(float, b));
    }
    HJ_UP(upload(ctx, pairs.data(), pairs.size(), &d.tri_pair));
    d.has_pairs = pairs.empty() ? 0u : 1u;
    // Large trees (their nodes and triangles do not fit the caches): the path-state streams bypass the caches so that they
    // do not evict scene data (1 M triangles +4.4 %; cache-resident scenes lose 0.5 ... 3 % with it).  HJ_STREAM_STATE = 0 / 1 forces.
    {
      const int nt_env = env_int("HJ_STREAM_STATE", -1, -1, 1);
      d.stream_state = (nt_env == 1 || (nt_env < 0 && N >= (size_t)env_int("HJ_STREAM_MIN_NODES", 300000, 0, 1 << 30))) ? 1u : 0u;
    }
    d.num_nodes = (uint32_t)M_all;
    d.root = N ? map[0] : 0u;
    d.num_hot = hot;
    // steps per round of the walk loop (the first one is the merged step that also runs the leaf tests) and free lanes at
    // which a wave fetches new rays: without pair nodes 4 / 32 (6: -0.4 %, 8: -4 % on cbox); with them 7 / 24 on small trees
    // (the rays of the rotated, child-ordered trees are shorter: 5 / 32, the optimum before those passes, is 3 % slower on
    // cbox now; 6 or 8 steps, 20 or 28 lanes: -1 ... -2 %), 8 / 24 up to 600 000 records and 8 / 32 beyond (1 M triangles:
    // 6 .. 10 steps the same, 24 lanes -1 %)
    const bool small_tree = M < 50000;
    d.inner_burst = (uint32_t)env_int("HJ_INNER_BURST", pairs.empty() ? 4 : (small_tree ? 7 : 8), 1, 1 << 20);   // >= 1, or the walk would never advance
    d.refill_min = (uint32_t)env_int("HJ_REFILL_MIN", !pairs.empty() && M < 600000 ? 24 : (int)hj::kRefillMin, 1, 64);   // (20 k / 60 k / 200 k triangles: 24 lanes +2 / +3 / +1 %)
    // The walk adds 32 * index to the low word of the array's address without a carry (kernels/hj_kernels.h): the
    // array must not cross a 4 GiB boundary.  Allocate twice the size and start at the boundary if it would.
    {
      const size_t bytes = std::max<size_t>(dev.size() * sizeof(float4), 16) + 128;   // (slack: a whole 128-byte line may be read around the last record)
      if (bytes >= (1ull << 32)) { release_scene(ctx); return set_error(ctx, HJ_ERR_UNSUPPORTED, "BVH of %zu records: the device node array is limited to 4 GiB", M); }
      ctx->scene_bufs.emplace_back();
      DevBuf& b = ctx->scene_bufs.back();
      HJ_UP(dev_alloc(ctx, b, bytes));
      uintptr_t start = reinterpret_cast<uintptr_t>(b.p);
      if ((start >> 32) != ((start + bytes - 1) >> 32)) {
        b.release();
        HJ_UP(dev_alloc(ctx, b, 2 * bytes));
        start = reinterpret_cast<uintptr_t>(b.p);
        if ((start >> 32) != ((start + bytes - 1) >> 32)) start = ((start >> 32) + 1) << 32;
      }
      if (hipMemcpy(reinterpret_cast<void*>(start), dev.data(), dev.size() * sizeof(float4), hipMemcpyHostToDevice) != hipSuccess) {
        release_scene(ctx);
        return set_error(ctx, HJ_ERR_DEVICE, "node upload failed");
      }
      d.nodes = reinterpret_cast<const float4*>(start);
    }
  }
  HJ_UP(upload(ctx, isect.data(), isect.size(), &d.tri_isect));
  HJ_UP(upload(ctx, shade.data(), shade.size(), &d.tri_shade));
  HJ_UP(upload(ctx, reinterpret_cast<const float4*>(s->spheres), s->num_spheres, &d.spheres));
  HJ_UP(upload(ctx, reinterpret_cast<const float4*>(s->quads), 3 * s->num_quads, &d.quads));
  HJ_UP(upload(ctx, s->triangles, s->num_triangles, &d.triangles));
  HJ_UP(upload(ctx, s->vertices, s->num_vertices, &d.vertices));
  HJ_UP(upload(ctx, s->materials, s->num_materials, &d.materials));
  HJ_UP(upload(ctx, s->emitters, s->num_emitters, &d.emitters));
  {
    std::vector<float4> rec((size_t)hj::kEmitRecF4 * s->num_emitters, make_float4(0.f, 0.f, 0.f, 0.f));
    for (size_t i = 0; i < s->num_emitters; i++) {
      float4* r = &rec[(size_t)hj::kEmitRecF4 * i];
      const uint32_t shape = s->emitters[i].shape;
      const hj_emissive& em = s->emissive[s->materials[shape] & HJ_MATERIAL_INDEX_MASK];
      uint32_t kind;
      if (shape < s->num_spheres) {
        kind = 0;
        const hj_sphere& sp = s->spheres[shape];
        r[1] = make_float4(sp.center[0], sp.center[1], sp.center[2], 0.f);
        r[0].z = sp.radius;
      } else if (shape < s->num_spheres + s->num_quads) {
        kind = 1;
        const hj_quad& q = s->quads[shape - s->num_spheres];
        r[1] = make_float4(q.origin[0], q.origin[1], q.origin[2], 0.f);
        r[2] = make_float4(q.edge1[0], q.edge1[1], q.edge1[2], 0.f);
        r[3] = make_float4(q.edge2[0], q.edge2[1], q.edge2[2], 0.f);
      } else {
        kind = 2;
        const hj_triangle& t = s->triangles[shape - s->num_spheres - s->num_quads];
        for (int k = 0; k < 3; k++) {
          const hj_vertex& v = s->vertices[t.v[k]];
          r[1 + k] = make_float4(v.pos[0], v.pos[1], v.pos[2], 0.f);
          r[4 + k] = make_float4(v.normal[0], v.normal[1], v.normal[2], 0.f);
        }
      }
      r[0].x = s->emitters[i].pdf;
      r[0].y = __builtin_bit_cast(float, kind);
      r[1].w = em.power[0]; r[2].w = em.power[1]; r[3].w = em.power[2];
    }
    HJ_UP(upload(ctx, rec.data(), rec.size(), &d.emit_rec));
  }
  HJ_UP(upload(ctx, reinterpret_cast<const float4*>(s->diffuse), s->num_diffuse, &d.diffuse));
  HJ_UP(upload(ctx, reinterpret_cast<const float4*>(s->diffusecb), 2 * s->num_diffusecb, &d.diffusecb));
  HJ_UP(upload(ctx, reinterpret_cast<const float4*>(s->dielectric), s->num_dielectric, &d.dielectric));
  HJ_UP(upload(ctx, reinterpret_cast<const float4*>(s->emissive), s->num_emissive, &d.emissive));
#undef HJ_UP
  ctx->scene = d;
  ctx->have_scene = true;
  return HJ_OK;
}

int hj_framebuffer_create(hj_context* ctx, uint32_t width, uint32_t height, void* external) {
  if (!ctx) return HJ_ERR_INVALID;
  HJ_NOT_BUSY(ctx);
  if (width == 0 || height == 0 || width > 65536 || height > 65536) return set_error(ctx, HJ_ERR_INVALID, "bad framebuffer size %ux%u", width, height);
  HJ_HIP(ctx, hipSetDevice(ctx->device));
  {
    const int rcs = sync_all(ctx);
    if (rcs != HJ_OK) return rcs;
  }
  if (ctx->accum && ctx->accum_owned) (void)hipFree(ctx->accum);
  ctx->accum = nullptr;
  ctx->accum_owned = false;
  const size_t bytes = (size_t)width * height * sizeof(float4);
  if (external) {
    if ((reinterpret_cast<uintptr_t>(external) & 15u) != 0) return set_error(ctx, HJ_ERR_INVALID, "external framebuffer must be 16-byte aligned");
    ctx->accum = static_cast<float4*>(external);
  } else {
    void* p = nullptr;
    HJ_HIP(ctx, hipMalloc(&p, bytes));
    ctx->accum = static_cast<float4*>(p);
    ctx->accum_owned = true;
  }
  ctx->width = width;
  ctx->height = height;
  return hj_framebuffer_clear(ctx);
}

int hj_framebuffer_clear(hj_context* ctx) {
  if (!ctx) return HJ_ERR_INVALID;
  HJ_NOT_BUSY(ctx);
  if (!ctx->accum) return set_error(ctx, HJ_ERR_STATE, "no framebuffer");
  HJ_HIP(ctx, hipSetDevice(ctx->device));
  HJ_HIP(ctx, hipMemsetAsync(ctx->accum, 0, (size_t)ctx->width * ctx->height * sizeof(float4), ctx->stream));
  HJ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return HJ_OK;
}

void* hj_framebuffer_device_ptr(hj_context* ctx) { return ctx ? ctx->accum : nullptr; }

int hj_framebuffer_read(hj_context* ctx, float* host_rgba) {
  if (!ctx) return HJ_ERR_INVALID;
  HJ_NOT_BUSY(ctx);
  if (!ctx->accum) return set_error(ctx, HJ_ERR_STATE, "no framebuffer");
  if (!host_rgba) return set_error(ctx, HJ_ERR_INVALID, "null destination");
  HJ_HIP(ctx, hipSetDevice(ctx->device));
  HJ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  HJ_HIP(ctx, hipMemcpy(host_rgba, ctx->accum, (size_t)ctx->width * ctx->height * sizeof(float4), hipMemcpyDeviceToHost));
  return HJ_OK;
}

int hj_framebuffer_resolve(hj_context* ctx, float* host_rgb) {
  if (!ctx) return HJ_ERR_INVALID;
  if (!host_rgb) return set_error(ctx, HJ_ERR_INVALID, "null destination");
  std::vector<float> tmp;
  try {
    tmp.resize((size_t)ctx->width * ctx->height * 4);
  } catch (const std::bad_alloc&) {
    return set_error(ctx, HJ_ERR_NOMEM, "out of host memory");
  }
  int rc = hj_framebuffer_read(ctx, tmp.data());
  if (rc != HJ_OK) return rc;
  const size_t n = (size_t)ctx->width * ctx->height;
  for (size_t i = 0; i < n; i++) {   // [r/n, g/n, b/n], src/main.rs:1399
    const float w = tmp[4 * i + 3];
    host_rgb[3 * i + 0] = tmp[4 * i + 0] / w;
    host_rgb[3 * i + 1] = tmp[4 * i + 1] / w;
    host_rgb[3 * i + 2] = tmp[4 * i + 2] / w;
  }
  return HJ_OK;
}

// One render call = begin / submit ... / end, so that hj_render_frame can stream its block list through the batch
// pipeline chunk by chunk without draining the slots between chunks (each drain exposes the tail of the last
// kernels: ~3.5 ms on cbox).
namespace {
struct RenderRun {
  hj_render_opts o{};
  Timer tm{nullptr, false};
  bool split = false;
  hj_render_stats local{};
  hj_render_stats* st = nullptr;
  size_t k = 0;            // batches enqueued so far (slot rotation)
  uint64_t paths = 0;
  uint32_t batch = 0;
  uint32_t shrunk = 0;     // times run_submit lowered the pool or the batch after an allocation failed
  std::chrono::steady_clock::time_point wall0;
};

int run_begin(hj_context* ctx, RenderRun& run, const hj_render_opts* opts, hj_render_stats* stats, size_t total_blocks) {
  if (!ctx->have_scene) return set_error(ctx, HJ_ERR_STATE, "render before hj_scene_upload");
  if (!ctx->accum) return set_error(ctx, HJ_ERR_STATE, "render before hj_framebuffer_create");
  if (opts) run.o = *opts;
  else hj_default_render_opts(&run.o);
  int rc = check_opts(ctx, run.o);
  if (rc != HJ_OK) return rc;
  HJ_HIP(ctx, hipSetDevice(ctx->device));
  run.st = stats ? stats : &run.local;
  std::memset(run.st, 0, sizeof *run.st);
  ctx->events_used = 0;
  run.tm = Timer{ctx, (run.o.flags & HJ_RENDER_TIME_KERNELS) != 0};
  run.split = (run.o.flags & HJ_RENDER_SPLIT_KERNELS) != 0;
  // Default batch: large batches amortise the latency-bound tail of a batch (measured: cbox+mirror+glass 700 ->
  // 960 Mpaths/s from 512 to 2048 blocks, +2-3 % more at 4096), but at least four batches should exist so that the three
  // slots can overlap (tools/batch_probe.py, rank 0's share of the cbox frame at 8 / 4 / 2 / 1 ranks: a quarter of the blocks
  // per batch beats an eighth by 3.4 / 1.6 / 0.8 / 0.8 %, a half loses 3-5 %).  Path state does not grow with the batch
  // (pool), only the sample buffers do (0.5 GB per 1024 blocks).
  const size_t n = total_blocks;
  static const size_t batch_cap = (size_t)env_int("HJ_BATCH_CAP", 8192, 64, 32768);
  run.batch = run.o.batch_blocks ? run.o.batch_blocks
                                 : (uint32_t)std::min<size_t>(batch_cap, std::max<size_t>(256, ((n + 3) / 4 + 63) / 64 * 64));
  run.batch = std::min<uint32_t>(run.batch, run.split ? 2048u : 32768u);   // (a sample index has 31 bits: 131 072 blocks at most)   // the split path keeps every sample of a batch in flight
  // Footprint: per batch slot 512 KB of samples per ImageBlock of the batch + num_wg x pool positions of
  // path state (184 B each, 216 B with tinted dielectrics).  DEFAULTS that do not fit the device's free memory (other
  // contexts on the GPU, the host application) shrink until they do: first the pool (down to 8192 positions), then the
  // batch; an explicit hj_render_opts::batch_blocks is taken as given and fails with HJ_ERR_NOMEM if it does not fit.
  static const size_t small_blocks = (size_t)env_int("HJ_WG_SMALL_BLOCKS", 12288, 0, 1 << 30);
  ctx->num_wg_eff = (!run.split && n < small_blocks) ? ctx->num_wg_small : ctx->num_wg;
  ctx->pool_eff = ctx->pool;
  ctx->slots_eff = ctx->num_slots;
  if (!run.split) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
      // HJ_MEM_LIMIT_MB (test rig): pretend that no more than this is free
      const int limit_mb = env_int("HJ_MEM_LIMIT_MB", 0, 0, 1 << 30);
      if (limit_mb > 0) free_b = std::min<size_t>(free_b, (size_t)limit_mb << 20);
      size_t held = 0;
      for (auto& sl : ctx->slots) {
        for (auto& b : sl.bufs) held += b.bytes;
        for (auto& b : sl.sample_bufs) held += b.bytes;
      }
      auto need = [&](uint32_t batch, uint32_t pool_cap) {
        const size_t slots_needed = std::max<size_t>(1, std::min<size_t>(ctx->num_slots, (n + batch - 1) / std::max<uint32_t>(batch, 1u)));
        const size_t per_wg = ((((size_t)batch * hj::kSlotsPerBlock + 63) / 64 + ctx->num_wg_eff - 1) / ctx->num_wg_eff) * 64;
        const size_t pool = std::min<size_t>(per_wg, pool_cap);
        const size_t state = (size_t)ctx->num_wg_eff * pool * (ctx->scene.has_extinction ? 196u + 20u : 164u + 20u);
        return slots_needed * (state + (size_t)batch * hj::kSlotsPerBlock * 32u);
      };
      const size_t margin = (size_t)512 << 20;
      const size_t avail = held + (free_b > margin ? free_b - margin : 0);
      while (need(run.batch, ctx->pool_eff) > avail) {
        if (ctx->pool_eff > 8192) ctx->pool_eff = std::max(8192u, ctx->pool_eff / 2 / 64 * 64);
        else if (!run.o.batch_blocks && run.batch > 64) run.batch = std::max(64u, run.batch / 2 / 64 * 64);
        else break;
      }
    }
  }
  rc = sync_all(ctx);
  if (rc != HJ_OK) return rc;
  ctx->blocks_total = total_blocks;
  ctx->blocks_done = ctx->blocks_reported = 0;
  run.wall0 = std::chrono::steady_clock::now();
  return HJ_OK;
}

// One step of the out-of-memory fallback (run_submit, hj_reserve): positions per workgroup down to 1024, then the batch down to
// 64 ImageBlocks (unless the caller fixed it), then ONE batch slot instead of three, then 256 positions.  False: nothing left.
bool shrink_footprint(hj_context* ctx, RenderRun& run) {
  if (ctx->pool_eff > 1024u) ctx->pool_eff = std::max(1024u, ctx->pool_eff / 2u / 64u * 64u);
  else if (!run.o.batch_blocks && run.batch > 64u) run.batch = std::max(64u, run.batch / 2u / 64u * 64u);
  else if (ctx->slots_eff > 1u) ctx->slots_eff = 1u;
  else if (ctx->pool_eff > 256u) ctx->pool_eff = std::max(256u, ctx->pool_eff / 2u / 64u * 64u);
  else return false;
  run.shrunk++;
  return true;
}

// Enqueues the batches of `blocks` (copied into the slots' pinned staging before this returns).
int run_submit(hj_context* ctx, RenderRun& run, const hj_image_block* blocks, size_t n) {
  for (size_t i = 0; i < n; i++) {
    const hj_image_block& b = blocks[i];
    if (b.dimension[0] == 0 || b.dimension[1] == 0 || b.dimension[0] > HJ_BLOCK_SIZE || b.dimension[1] > HJ_BLOCK_SIZE)
      return set_error(ctx, HJ_ERR_INVALID, "block %zu: dimension %ux%u outside (0,128]", i, b.dimension[0], b.dimension[1]);
    if (b.original_dimension[0] != ctx->width || b.original_dimension[1] != ctx->height)
      return set_error(ctx, HJ_ERR_INVALID, "block %zu: original_dimension %ux%u != framebuffer %ux%u", i,
                       b.original_dimension[0], b.original_dimension[1], ctx->width, ctx->height);
    run.paths += (uint64_t)std::min(b.dimension[0], b.original_dimension[0]) * std::min(b.dimension[1], b.original_dimension[1]);
  }
  int rc = HJ_OK;
  // (Shrinking the last batches of a run - each 1/2 .. 1/6 of what is left - was measured: no change; the ~3.5 ms a
  // frame loses to pipeline fill and drain does not depend on the size of the last kernels.)
  for (size_t begin = 0; begin < n && rc == HJ_OK; run.k++) {
    const uint32_t nb = (uint32_t)std::min<size_t>(run.batch, n - begin);
    hj_context::BatchSlot& sl = ctx->slots[run.k % ctx->slots_eff];
    hj_context::BatchSlot& other = ctx->slots[(run.k + ctx->slots_eff - 1) % ctx->slots_eff];   // the previous batch's slot
    rc = harvest(ctx, sl, run.st);          // an older batch used this slot: its state arrays are free again
    if (rc != HJ_OK) break;
    rc = run.split ? render_batch_split(ctx, sl, other, blocks + begin, nb, run.o, run.tm, run.st, true)
                   : enqueue_batch_fused(ctx, sl, other, blocks + begin, nb, run.o, run.tm, run.st, true);
    if (rc == HJ_ERR_NOMEM && !run.split) {
      // The device ran out of memory AT the allocation (run_begin's estimate from hipMemGetInfo was taken before another
      // context or the host application grew): nothing of this batch has been enqueued.  Wait for the batches in flight,
      // give back every slot's arrays, shrink - the pool first (down to 1024 positions), then the batch unless the caller
      // fixed it - and try this batch again; HJ_ERR_NOMEM only when nothing is left to shrink.
      const std::string first_error = get_error(ctx);
      int rc2 = HJ_OK;
      for (auto& s2 : ctx->slots) { const int r3 = harvest(ctx, s2, run.st); if (rc2 == HJ_OK) rc2 = r3; }
      if (rc2 == HJ_OK) rc2 = sync_all(ctx);
      if (rc2 != HJ_OK) { rc = rc2; break; }
      release_batch(ctx);
      if (!shrink_footprint(ctx, run)) { set_error(ctx, HJ_ERR_NOMEM, "%s (pool, batch and slots are at their minimum)", first_error.c_str()); break; }
      rc = HJ_OK;
      run.k--;                               // (the loop's increment: this batch has not been enqueued)
      continue;
    }
    begin += nb;
  }
  return rc;
}

// Drains the slots (also after an error, so that nothing of this run is still in flight) and closes the statistics.
int run_end(hj_context* ctx, RenderRun& run, int rc) {
  const std::string first_error = get_error(ctx);
  for (auto& sl : ctx->slots) {
    const int rc2 = harvest(ctx, sl, run.st);
    if (rc == HJ_OK) rc = rc2;
  }
  {  // ALWAYS: after an error, too, nothing of this run may still be writing to the (possibly caller-owned) framebuffer
    const int rc2 = sync_all(ctx);
    if (rc == HJ_OK) rc = rc2;
    else set_error(ctx, rc, "%s", first_error.c_str());   // keep the message of the error that ended the run
  }
  if (rc == HJ_OK && ctx->progress && ctx->blocks_done != ctx->blocks_reported)
    ctx->progress(ctx->progress_user, ctx->blocks_done, std::max(ctx->blocks_total, ctx->blocks_done));
  if (rc == HJ_OK) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) rc = set_error(ctx, HJ_ERR_DEVICE, "kernel launch: %s", hipGetErrorString(e));
  }
  if (rc == HJ_OK) {
    hj_render_stats* st_out = run.st;
    st_out->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - run.wall0).count();
    st_out->paths = run.paths;
    // exclusive time of the dominant kernel: the union of the launches' intervals (launches of different batch slots
    // overlap, so the sum of their durations exceeds the wall clock)
    {
      std::vector<std::pair<float, float>> iv;
      for (size_t i = 0; i < ctx->events_used; i++) {
        const int kind = ctx->events[i].kind;
        if (kind != EV_PATH && kind != EV_CLOSEST && kind != EV_SHADE && kind != EV_SHADOW) continue;
        float a = 0.f, b = 0.f;
        if (hipEventElapsedTime(&a, ctx->events[0].a, ctx->events[i].a) != hipSuccess) continue;
        if (hipEventElapsedTime(&b, ctx->events[0].a, ctx->events[i].b) != hipSuccess) continue;
        iv.emplace_back(a, b);
      }
      std::sort(iv.begin(), iv.end());
      float busy = 0.f, cur_a = 0.f, cur_b = -1.f;
      for (auto& x : iv) {
        if (cur_b < cur_a || x.first > cur_b) { if (cur_b > cur_a) busy += cur_b - cur_a; cur_a = x.first; cur_b = x.second; }
        else cur_b = std::max(cur_b, x.second);
      }
      if (cur_b > cur_a) busy += cur_b - cur_a;
      st_out->path_busy_ms = busy;
    }
    for (size_t i = 0; i < ctx->events_used; i++) {
      float e = 0.f;
      if (hipEventElapsedTime(&e, ctx->events[i].a, ctx->events[i].b) != hipSuccess) continue;
      switch (ctx->events[i].kind) {
        case EV_CLOSEST: st_out->trace_closest_ms += e; st_out->closest_launches++; break;
        case EV_SHADOW: st_out->trace_shadow_ms += e; break;
        case EV_SHADE: st_out->shade_ms += e; break;
        case EV_RECON: st_out->reconstruct_ms += e; break;
        case EV_PATH: st_out->path_ms += e; st_out->path_launches++; break;
      }
    }
  }
  return rc;
}
}  // namespace

int hj_reserve(hj_context* ctx, size_t total_blocks, const hj_render_opts* opts) {
  if (!ctx) return HJ_ERR_INVALID;
  HJ_NOT_BUSY(ctx);
  if (total_blocks == 0) return HJ_OK;
  RenderRun run;
  int rc = run_begin(ctx, run, opts, nullptr, total_blocks);        // (the call's batch size, pool and workgroup count)
  if (rc != HJ_OK || run.split) return rc;
  for (;;) {
    size_t left = total_blocks;
    rc = HJ_OK;
    for (uint32_t k = 0; k < ctx->slots_eff && left != 0 && rc == HJ_OK; k++) {
      const uint32_t nb = (uint32_t)std::min<size_t>(run.batch, left);
      rc = ensure_batch(ctx, ctx->slots[k], nb, false);
      left -= nb;
    }
    if (rc != HJ_ERR_NOMEM) return rc;
    // as in a render call (run_submit): give everything back, shrink the pool, then the batch, and try again
    release_batch(ctx);
    if (!shrink_footprint(ctx, run)) return rc;
  }
}

int hj_render_blocks(hj_context* ctx, const hj_image_block* blocks, size_t n, const hj_render_opts* opts,
                     hj_render_stats* stats) {
  if (!ctx) return HJ_ERR_INVALID;
  HJ_NOT_BUSY(ctx);
  if (n && !blocks) return set_error(ctx, HJ_ERR_INVALID, "null block list");
  RenderRun run;
  int rc = run_begin(ctx, run, opts, stats, n);
  if (rc != HJ_OK) return rc;
  rc = run_submit(ctx, run, blocks, n);
  return run_end(ctx, run, rc);
}

namespace {
int render_frame_impl(hj_context* ctx, uint32_t spp, uint64_t master_seed, uint32_t pass_begin, uint32_t pass_end,
                      uint32_t rank, uint32_t world, const hj_render_opts* opts, hj_render_stats* stats);
}
int hj_render_frame(hj_context* ctx, uint32_t spp, uint64_t master_seed, uint32_t pass_begin, uint32_t pass_end,
                    uint32_t rank, uint32_t world, const hj_render_opts* opts, hj_render_stats* stats) {
  if (!ctx) return HJ_ERR_INVALID;
  HJ_NOT_BUSY(ctx);
  return render_frame_impl(ctx, spp, master_seed, pass_begin, pass_end, rank, world, opts, stats);
}
namespace {
int render_frame_impl(hj_context* ctx, uint32_t spp, uint64_t master_seed, uint32_t pass_begin, uint32_t pass_end,
                      uint32_t rank, uint32_t world, const hj_render_opts* opts, hj_render_stats* stats) {
  if (!ctx->accum) return set_error(ctx, HJ_ERR_STATE, "render before hj_framebuffer_create");
  if (world == 0 || rank >= world) return set_error(ctx, HJ_ERR_INVALID, "bad rank %u / world %u", rank, world);
  if (pass_end > spp || pass_begin > pass_end) return set_error(ctx, HJ_ERR_INVALID, "bad pass range [%u,%u) of %u", pass_begin, pass_end, spp);
  hijiki::BlockGrid grid(ctx->width, ctx->height, HJ_BLOCK_SIZE);
  // Tile sharding: block j of pass p belongs to rank grid.owner(p, j, world) (a diagonal deal that rotates with the
  // pass; with HJ_RENDER_STATIC_DEAL all passes of one block stay on one GPU and accumulate there in pass order,
  // SURVEY.md §8e).  The list is generated in chunks
  // (4096^2 x 4096 spp would be 4.2 M blocks = 168 MB if materialised at once).
  std::vector<hj_image_block> chunk;
  const bool static_deal = opts && (opts->flags & HJ_RENDER_STATIC_DEAL);
  const uint32_t per_pass = grid.per_pass();
  const uint32_t passes_per_chunk = std::max<uint32_t>(1u, 32768u / std::max<uint32_t>(1u, (per_pass + world - 1) / world));
  // blocks this rank will render (for the batch-size rule): every rank owns per_pass / world of each pass, +-1
  const size_t mine_estimate = (size_t)(pass_end - pass_begin) * ((per_pass + world - 1) / world);
  RenderRun run;
  int rc = run_begin(ctx, run, opts, stats, mine_estimate);
  if (rc != HJ_OK) return rc;
  for (uint32_t p0 = pass_begin; p0 < pass_end && rc == HJ_OK; p0 += passes_per_chunk) {
    const uint32_t p1 = std::min(pass_end, p0 + passes_per_chunk);
    for (uint32_t p = p0; p < p1; p++)
      for (uint32_t j = 0; j < per_pass; j++)
        if (grid.owner(static_deal ? 0u : p, j, world) == rank) chunk.push_back(grid.make(master_seed, p, j));
    // whole batches now (the slots keep running while the next chunk is generated); the remainder joins the next chunk
    const size_t full = p1 == pass_end ? chunk.size() : chunk.size() / run.batch * run.batch;
    rc = run_submit(ctx, run, chunk.data(), full);
    chunk.erase(chunk.begin(), chunk.begin() + (std::ptrdiff_t)full);
  }
  return run_end(ctx, run, rc);
}

// The context's worker: sleeps until a frame is posted, renders it (blocking, on this thread), publishes the result.
void worker_main(hj_context* ctx) {
  for (;;) {
    hj_context::AsyncJob j;
    {
      std::unique_lock<std::mutex> lock(ctx->job_mu);
      ctx->job_cv.wait(lock, [&] { return ctx->job_posted || ctx->worker_exit; });
      if (!ctx->job_posted) return;          // (exit is honoured only between frames)
      j = ctx->job;
      ctx->job_posted = false;
    }
    hj_render_stats st{};
    const int rc = render_frame_impl(ctx, j.spp, j.master_seed, j.pass_begin, j.pass_end, j.rank, j.world, &j.opts, &st);
    {
      std::lock_guard<std::mutex> lock(ctx->job_mu);
      ctx->async_rc = rc;
      ctx->async_stats = st;
      ctx->async_valid = true;
      ctx->busy.store(false, std::memory_order_release);
    }
    ctx->job_cv.notify_all();
  }
}
}  // namespace

// ---- asynchronous frame: the blocking render on the context's worker thread, so that ONE host thread can keep several
// GPUs (contexts) rendering at the same time and overlap one context's drain with work on the others.

int hj_render_frame_async(hj_context* ctx, uint32_t spp, uint64_t master_seed, uint32_t pass_begin, uint32_t pass_end,
                          uint32_t rank, uint32_t world, const hj_render_opts* opts) {
  if (!ctx) return HJ_ERR_INVALID;
  HJ_NOT_BUSY(ctx);
  hj_context::AsyncJob j{};
  j.spp = spp; j.master_seed = master_seed; j.pass_begin = pass_begin; j.pass_end = pass_end; j.rank = rank; j.world = world;
  if (opts) j.opts = *opts;
  else hj_default_render_opts(&j.opts);
  if (!ctx->worker.joinable()) {
    try {
      ctx->worker = std::thread(worker_main, ctx);
    } catch (const std::exception& e) {
      return set_error(ctx, HJ_ERR_NOMEM, "could not start the render thread: %s", e.what());
    }
  }
  {
    std::lock_guard<std::mutex> lock(ctx->job_mu);
    ctx->job = j;
    ctx->job_posted = true;
    ctx->async_valid = false;
    ctx->busy.store(true, std::memory_order_release);
  }
  ctx->job_cv.notify_all();
  return HJ_OK;
}

// Waits for the frame in flight (if any) and returns its status and statistics; the result of the LAST asynchronous
// frame stays available until the next one starts, so hj_sync after hj_comm_reduce_framebuffers (which joins every
// frame itself) still yields the statistics.  With no asynchronous frame ever started: HJ_OK, *stats untouched.
int hj_sync(hj_context* ctx, hj_render_stats* stats) {
  if (!ctx) return HJ_ERR_INVALID;
  std::unique_lock<std::mutex> lock(ctx->job_mu);
  ctx->job_cv.wait(lock, [&] { return !ctx->busy.load(std::memory_order_acquire); });
  if (!ctx->async_valid) return HJ_OK;
  if (stats) *stats = ctx->async_stats;
  return ctx->async_rc;                      // the worker's error text is in hj_last_error(ctx)
}

int hj_device_count(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

void hj_set_progress_callback(hj_context* ctx, hj_progress_fn fn, void* user, uint32_t interval_blocks) {
  if (!ctx) return;
  ctx->progress = fn;
  ctx->progress_user = user;
  ctx->progress_interval = interval_blocks ? interval_blocks : 1u;
}

// ---- RCCL through dlopen: the library itself has no link-time dependency on librccl.  A copy that the process has
// already mapped (PyTorch bundles one) is reused (RTLD_NOLOAD) instead of mapping a second one beside it.
namespace {
struct Rccl {
  void* lib = nullptr;
  int (*CommInitAll)(void**, int, const int*) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Reduce)(const void*, void*, size_t, int, int, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool load() {
    if (lib) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* name : names)
      if ((lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD)) != nullptr) break;
    if (!lib)
      for (const char* name : names)
        if ((lib = dlopen(name, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
    if (!lib) return false;
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(lib, "ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
    Reduce = reinterpret_cast<decltype(Reduce)>(dlsym(lib, "ncclReduce"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    return CommInitAll && CommDestroy && GroupStart && GroupEnd && Reduce && GetErrorString;
  }
};
Rccl g_rccl;
std::mutex g_rccl_mutex;
constexpr int kNcclFloat32 = 7, kNcclSum = 0;   // ncclFloat / ncclSum of rccl.h
}  // namespace

// The contexts of one process (one per GPU) and their RCCL communicators, created ONCE (ncclCommInitAll costs hundreds
// of milliseconds) and reused by every frame's reduce.
struct hj_comm {
  std::vector<hj_context*> ctxs;
  std::vector<void*> comms;                  // empty for n == 1 and for a shared-GPU test rig
  bool shared_gpu = false;                   // HJ_COMM_SHARED_GPU=1: contexts on ONE GPU, summed by a kernel instead of RCCL
};

namespace hj {
__global__ void k_add_framebuffer(float4* __restrict__ dst, const float4* __restrict__ src, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const float4 a = dst[i], b = src[i]; dst[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
}
}  // namespace hj

namespace {
std::vector<hj_comm*> g_cached_comms;        // communicators made on behalf of hj_reduce_framebuffers

// Test rigs without several GPUs (HJ_COMM_SHARED_GPU=1): the contexts of a communicator may live on one GPU; their
// framebuffers are then summed by a kernel, in context order, instead of by RCCL.  Everything else of the multi-context
// path (worker threads, frames in flight on all contexts, the joins inside the reduce) is the real thing.
bool shared_gpu_allowed() { return env_int("HJ_COMM_SHARED_GPU", 0, 0, 1) != 0; }

int check_reduce_args(hj_context* const* ctxs, int n, int root) {
  if (!ctxs || n < 1 || root < 0 || root >= n || !ctxs[root]) return HJ_ERR_INVALID;
  hj_context* r = ctxs[root];
  for (int i = 0; i < n; i++) {
    if (!ctxs[i] || !ctxs[i]->accum) return set_error(r, HJ_ERR_STATE, "context %d has no framebuffer", i);
    if (ctxs[i]->width != r->width || ctxs[i]->height != r->height) return set_error(r, HJ_ERR_INVALID, "framebuffer sizes differ");
    for (int j = 0; j < i; j++) {
      if (ctxs[j] == ctxs[i]) return set_error(r, HJ_ERR_INVALID, "context %d appears twice", i);
      if (ctxs[j]->device == ctxs[i]->device && !shared_gpu_allowed())
        return set_error(r, HJ_ERR_INVALID, "contexts %d and %d share GPU %d", j, i, ctxs[i]->device);
    }
  }
  return HJ_OK;
}

void comm_release(hj_comm* c) {
  for (void* x : c->comms)
    if (x) (void)g_rccl.CommDestroy(x);
  delete c;
}
}  // namespace

int hj_comm_create(hj_context* const* ctxs, int n, hj_comm** out) {
  if (!out) return HJ_ERR_INVALID;
  *out = nullptr;
  if (!ctxs || n < 1 || !ctxs[0]) return HJ_ERR_INVALID;
  hj_context* r = ctxs[0];
  bool shared = false;
  for (int i = 0; i < n; i++) {
    if (!ctxs[i]) return set_error(r, HJ_ERR_INVALID, "null context %d", i);
    for (int j = 0; j < i; j++) {
      if (ctxs[j] == ctxs[i]) return set_error(r, HJ_ERR_INVALID, "context %d appears twice", i);
      if (ctxs[j]->device == ctxs[i]->device) {
        if (!shared_gpu_allowed()) return set_error(r, HJ_ERR_INVALID, "contexts %d and %d share GPU %d", j, i, ctxs[i]->device);
        shared = true;
      }
    }
  }
  hj_comm* c = new (std::nothrow) hj_comm();
  if (!c) return set_error(r, HJ_ERR_NOMEM, "out of host memory");
  c->ctxs.assign(ctxs, ctxs + n);
  c->shared_gpu = shared;
  // HJ_COMM_FORCE_RCCL=1 (test rigs with one GPU): a single context also gets a communicator and its reduce goes through
  // ncclReduce (one rank, in place), so that the loader, the entry points and the stream handling run before a second GPU exists.
  if ((n > 1 || env_int("HJ_COMM_FORCE_RCCL", 0, 0, 1) != 0) && !shared) {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (!g_rccl.load()) {
      delete c;
      return set_error(r, HJ_ERR_UNSUPPORTED, "librccl.so could not be loaded: %s", dlerror());
    }
    c->comms.assign((size_t)n, nullptr);
    std::vector<int> devs((size_t)n);
    for (int i = 0; i < n; i++) devs[(size_t)i] = ctxs[i]->device;
    const int nrc = g_rccl.CommInitAll(c->comms.data(), n, devs.data());
    if (nrc != 0) {
      comm_release(c);
      return set_error(r, HJ_ERR_DEVICE, "ncclCommInitAll: %s", g_rccl.GetErrorString(nrc));
    }
  }
  *out = c;
  return HJ_OK;
}

void hj_comm_destroy(hj_comm* c) {
  if (!c) return;
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  comm_release(c);
}

int hj_comm_reduce_framebuffers(hj_comm* c, int root) {
  if (!c) return HJ_ERR_INVALID;
  const int n = (int)c->ctxs.size();
  int rc = check_reduce_args(c->ctxs.data(), n, root);
  if (rc != HJ_OK) return rc;
  hj_context* r = c->ctxs[(size_t)root];
  // every context's frame must be complete: join asynchronous renders, then drain the streams
  for (int i = 0; i < n; i++) {
    hj_context* x = c->ctxs[(size_t)i];
    const int rs = hj_sync(x, nullptr);
    if (rs != HJ_OK) return set_error(r, rs, "context %d: render failed: %s", i, get_error(x).c_str());
    if (hipSetDevice(x->device) != hipSuccess || sync_all(x) != HJ_OK)
      return set_error(r, HJ_ERR_DEVICE, "context %d: stream synchronisation failed: %s", i, get_error(x).c_str());
  }
  if (n == 1 && c->comms.empty()) return HJ_OK;
  if (c->shared_gpu) {
    const size_t px = (size_t)r->width * r->height;
    HJ_HIP(r, hipSetDevice(r->device));
    for (int i = 0; i < n; i++)
      if (i != root)
        hipLaunchKernelGGL(hj::k_add_framebuffer, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, r->stream, r->accum,
                           c->ctxs[(size_t)i]->accum, px);
    HJ_HIP(r, hipStreamSynchronize(r->stream));
    HJ_HIP(r, hipGetLastError());
    return HJ_OK;
  }
  const size_t count = (size_t)r->width * r->height * 4;
  int nrc = g_rccl.GroupStart();
  for (int i = 0; i < n && nrc == 0; i++) {
    hj_context* x = c->ctxs[(size_t)i];
    (void)hipSetDevice(x->device);
    nrc = g_rccl.Reduce(x->accum, x->accum, count, kNcclFloat32, kNcclSum, root, c->comms[(size_t)i], x->stream);
  }
  const int erc = g_rccl.GroupEnd();
  if (nrc == 0) nrc = erc;
  rc = HJ_OK;
  for (int i = 0; i < n; i++) {
    (void)hipSetDevice(c->ctxs[(size_t)i]->device);
    if (hipStreamSynchronize(c->ctxs[(size_t)i]->stream) != hipSuccess) rc = HJ_ERR_DEVICE;
  }
  if (nrc != 0) return set_error(r, HJ_ERR_DEVICE, "ncclReduce: %s", g_rccl.GetErrorString(nrc));
  if (rc != HJ_OK) return set_error(r, rc, "stream synchronisation after the reduce failed");
  return HJ_OK;
}

// Convenience form without a communicator object: the communicators of a context list are created on first use and
// kept (keyed by the list) until one of the contexts is destroyed.
int hj_reduce_framebuffers(hj_context* const* ctxs, int n, int root) {
  int rc = check_reduce_args(ctxs, n, root);
  if (rc != HJ_OK) return rc;
  hj_comm* c = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    for (hj_comm* x : g_cached_comms)
      if ((int)x->ctxs.size() == n && std::equal(x->ctxs.begin(), x->ctxs.end(), ctxs)) c = x;
  }
  if (!c) {
    rc = hj_comm_create(ctxs, n, &c);
    if (rc != HJ_OK) {
      if (ctxs[root] != ctxs[0]) set_error(ctxs[root], rc, "%s", get_error(ctxs[0]).c_str());
      return rc;
    }
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    g_cached_comms.push_back(c);
  }
  return hj_comm_reduce_framebuffers(c, root);
}

}  // extern "C"
static void hj_drop_cached_comms(hj_context* ctx) {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  for (size_t i = 0; i < g_cached_comms.size();) {
    hj_comm* c = g_cached_comms[i];
    if (std::find(c->ctxs.begin(), c->ctxs.end(), ctx) != c->ctxs.end()) {
      comm_release(c);
      g_cached_comms.erase(g_cached_comms.begin() + (std::ptrdiff_t)i);
    } else {
      i++;
    }
  }
}
extern "C" {

// the tree of Scene::compile (src/main.rs:199-231), built on the device (kernels/hj_lbvh.h).
int hj_build_bvh_device(hj_context* ctx, const hj_scene_desc* s, hj_bvh_node* out_nodes, size_t capacity, size_t* out_num_nodes) {
  if (!ctx) return HJ_ERR_INVALID;
  HJ_NOT_BUSY(ctx);
  if (!s || !out_nodes) return set_error(ctx, HJ_ERR_INVALID, "null argument");
  const size_t n = s->num_spheres + s->num_quads + s->num_triangles;
  if (n < 2) return set_error(ctx, HJ_ERR_INVALID, "scene needs at least 2 shapes (reference panics: root would be a leaf, src/main.rs:230)");
  if (n >= hj::kInnerFlag / 4) return set_error(ctx, HJ_ERR_INVALID, "scene too large");
  const size_t total = 2 * n - 1;
  if (capacity < total) return set_error(ctx, HJ_ERR_INVALID, "node buffer holds %zu records, the tree has %zu", capacity, total);
  if ((s->num_spheres && !s->spheres) || (s->num_quads && !s->quads) || (s->num_triangles && (!s->triangles || !s->vertices)))
    return set_error(ctx, HJ_ERR_INVALID, "null shape array");
  for (size_t i = 0; i < s->num_triangles; i++)
    for (int k = 0; k < 3; k++)
      if (s->triangles[i].v[k] >= s->num_vertices) return set_error(ctx, HJ_ERR_INVALID, "triangle %zu refers to unknown vertex", i);
  HJ_HIP(ctx, hipSetDevice(ctx->device));
  // HJ_LBVH_TIMING=1: wall time of the build's stages on stderr (the stream is drained at every mark)
  const bool timing = env_int("HJ_LBVH_TIMING", 0, 0, 1) != 0;
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
  float big_frac = (float)env_int("HJ_LBVH_BIG_PCT", 2, 0, 100) / 100.0f;
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
  const uint32_t cmax_env = (uint32_t)env_int("HJ_LBVH_CLUSTER", 64, 0, 1 << 20);
  const uint32_t cmax = cmax_env == 0 ? m : cmax_env;
  // inside the clusters: SAH re-split (one thread per cluster, which needs no boxes of the Morton tree's internal nodes) or the
  // Morton topology as it is (HJ_LBVH_SAH=0, or clusters larger than the kernel's arrays: bottom-up refit first)
  const bool sah_clusters = cmax <= hj::lbvh::kClusterMax && env_int("HJ_LBVH_SAH", 1, 0, 1) != 0;
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
    } builder{items, {}, 0, env_int("HJ_BVH_CHILD_ORDER", 3, 0, 9), 0, {}, {}};
    // Rotation passes over the top (HJ_LBVH_TOP_ROTATE; -1 = the default rule): they pay where the top IS most of the tree - the
    // 6 k-triangle box: 112 items, frame rate 0.94 -> 0.99 of the host tree's - and cost 1.5 % (and 1.7 ms) on the 1 M-triangle
    // mesh, whose 25 k cluster boxes a binned SAH already arranges well: small tops only.
    {
      const int r = env_int("HJ_LBVH_TOP_ROTATE", -1, -1, 64);
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
  if (sah_clusters)
    hipLaunchKernelGGL(hj::lbvh::k_emit_clusters_sah, dim3((K + hj::lbvh::kSahThreads - 1) / hj::lbvh::kSahThreads),
                       dim3(hj::lbvh::kSahThreads), 0, st, t, K, cl, idx_mask, d_out, env_int("HJ_BVH_CHILD_ORDER", 3, 0, 9));
  else
    hipLaunchKernelGGL(hj::lbvh::k_emit_clusters, grid_sub, blk, 0, st, t, m, cl, idx_mask, d_out);
  HJ_HIP(ctx, hipGetLastError());
  mark("cluster subtrees");
  HJ_HIP(ctx, hipMemcpyAsync(out_nodes, d_out, sizeof(hj_bvh_node) * total, hipMemcpyDeviceToHost, st));
  HJ_HIP(ctx, hipStreamSynchronize(st));
  for (const auto& pr : top_records) out_nodes[pr.first] = pr.second;
  mark("records to the host");
#undef HJ_DEVBUF
  if (out_num_nodes) *out_num_nodes = total;
  return HJ_OK;
}

int hj_debug_trace(hj_context* ctx, const float* rays, size_t n, uint32_t use_bvh, uint32_t any_hit, float* hits) {
  if (!ctx) return HJ_ERR_INVALID;
  HJ_NOT_BUSY(ctx);
  if (!ctx->have_scene) return set_error(ctx, HJ_ERR_STATE, "trace before hj_scene_upload");
  if (n == 0) return HJ_OK;
  if (!rays || !hits) return set_error(ctx, HJ_ERR_INVALID, "null argument");
  if (n > 0x7FFFFFFFu) return set_error(ctx, HJ_ERR_INVALID, "too many rays");
  HJ_HIP(ctx, hipSetDevice(ctx->device));
  DevBuf d_rays, d_hits;
  int rc = dev_alloc(ctx, d_rays, n * 8 * sizeof(float));
  if (rc == HJ_OK) rc = dev_alloc(ctx, d_hits, n * sizeof(float4));
  if (rc == HJ_OK) {
    hipError_t e = hipMemcpyAsync(d_rays.p, rays, n * 8 * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    const dim3 grid((unsigned)((n + hj::kBlockThreads - 1) / hj::kBlockThreads)), blk(hj::kBlockThreads);
    const float* r = static_cast<const float*>(d_rays.p);
    float4* h = static_cast<float4*>(d_hits.p);
    const uint32_t cnt = (uint32_t)n;
    if (e == hipSuccess) {
      if (use_bvh && any_hit) hipLaunchKernelGGL((hj::k_debug_trace<true, true>), grid, blk, 0, ctx->stream, ctx->scene, r, cnt, h);
      else if (use_bvh) hipLaunchKernelGGL((hj::k_debug_trace<true, false>), grid, blk, 0, ctx->stream, ctx->scene, r, cnt, h);
      else if (any_hit) hipLaunchKernelGGL((hj::k_debug_trace<false, true>), grid, blk, 0, ctx->stream, ctx->scene, r, cnt, h);
      else hipLaunchKernelGGL((hj::k_debug_trace<false, false>), grid, blk, 0, ctx->stream, ctx->scene, r, cnt, h);
      e = hipMemcpyAsync(hits, d_hits.p, n * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) rc = set_error(ctx, HJ_ERR_DEVICE, "hj_debug_trace: %s", hipGetErrorString(e));
  }
  d_rays.release();
  d_hits.release();
  return rc;
}

int hj_debug_samples(hj_context* ctx, const hj_image_block* block, const hj_render_opts* opts, float* samples) {
  if (!ctx) return HJ_ERR_INVALID;
  HJ_NOT_BUSY(ctx);
  if (!ctx->have_scene) return set_error(ctx, HJ_ERR_STATE, "render before hj_scene_upload");
  if (!block || !samples) return set_error(ctx, HJ_ERR_INVALID, "null argument");
  if (block->dimension[0] == 0 || block->dimension[1] == 0 || block->dimension[0] > HJ_BLOCK_SIZE || block->dimension[1] > HJ_BLOCK_SIZE)
    return set_error(ctx, HJ_ERR_INVALID, "block dimension outside (0,128]");
  hj_render_opts o;
  if (opts) o = *opts;
  else hj_default_render_opts(&o);
  int rc = check_opts(ctx, o);
  if (rc != HJ_OK) return rc;
  HJ_HIP(ctx, hipSetDevice(ctx->device));
  ctx->events_used = 0;
  Timer tm{ctx, false};
  rc = sync_all(ctx);
  if (rc != HJ_OK) return rc;
  hj_context::BatchSlot& sl = ctx->slots[0];
  rc = (o.flags & HJ_RENDER_SPLIT_KERNELS) ? render_batch_split(ctx, sl, ctx->slots[1], block, 1, o, tm, nullptr, /*reconstruct=*/false)
                                           : enqueue_batch_fused(ctx, sl, ctx->slots[1], block, 1, o, tm, nullptr, /*reconstruct=*/false);
  if (rc == HJ_OK) rc = harvest(ctx, sl, nullptr, /*count_progress=*/false);
  if (rc != HJ_OK) return rc;
  std::vector<float4> rgb(hj::kSlotsPerBlock), nd(hj::kSlotsPerBlock);
  HJ_HIP(ctx, hipMemcpy(rgb.data(), sl.st.smp_rgb, sizeof(float4) * hj::kSlotsPerBlock, hipMemcpyDeviceToHost));
  HJ_HIP(ctx, hipMemcpy(nd.data(), sl.st.smp_nd, sizeof(float4) * hj::kSlotsPerBlock, hipMemcpyDeviceToHost));
  for (uint32_t y = 0; y < block->dimension[1]; y++)
    for (uint32_t x = 0; x < block->dimension[0]; x++) {
      float* out = samples + ((size_t)y * block->dimension[0] + x) * 8;
      const float4 a = rgb[y * HJ_BLOCK_SIZE + x], b = nd[y * HJ_BLOCK_SIZE + x];
      out[0] = a.x; out[1] = a.y; out[2] = a.z; out[3] = a.w; out[4] = b.x; out[5] = b.y; out[6] = b.z; out[7] = b.w;
    }
  return HJ_OK;
}

#ifdef HJ_WALK_STATS
extern "C" __attribute__((visibility("default"))) int hj_debug_round_stats(unsigned long long out[32], int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(hj::g_round_stats), 32 * sizeof(unsigned long long)) != hipSuccess) return HJ_ERR_DEVICE;
  if (reset) {
    unsigned long long z[32] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(hj::g_round_stats), z, sizeof z) != hipSuccess) return HJ_ERR_DEVICE;
  }
  return HJ_OK;
}
extern "C" __attribute__((visibility("default"))) int hj_debug_walk_stats(unsigned long long out[16], int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(hj::g_walk_stats), 16 * sizeof(unsigned long long)) != hipSuccess) return HJ_ERR_DEVICE;
  if (reset) {
    unsigned long long z[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(hj::g_walk_stats), z, sizeof z) != hipSuccess) return HJ_ERR_DEVICE;
  }
  return HJ_OK;
}
#endif

uint32_t hj_block_seed(uint64_t master, uint32_t pass, uint32_t j) { return hijiki::block_seed(master, pass, j); }
uint32_t hj_block_owner(uint32_t width, uint32_t height, uint32_t pass, uint32_t j, uint32_t world) {
  if (!width || !height || !world) return 0;
  return hijiki::BlockGrid(width, height, HJ_BLOCK_SIZE).owner(pass, j, world);
}
void hj_pass_offset(uint64_t master, uint32_t k, float out[2]) { hijiki::pass_offset(master, k, out); }

}  // extern "C"
