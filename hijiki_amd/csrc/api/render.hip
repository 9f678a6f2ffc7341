// Render calls: batch slots, the launches of the kernels (kernels/hj_kernels.h), hj_render_blocks / hj_render_frame
// (+ asynchronous form), hj_reserve, and the probe entry points.  What Renderer::render does upstream
// (src/main.rs:1316-1355), many ImageBlocks at a time.
#include "hj_internal.h"
#include "../kernels/hj_kernels.h"

#pragma clang fp contract(off)

using namespace hjapi;

namespace {

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
    HJ_ALLOC(sl.bufs, hit_tag, uint8_t, n)
    HJ_ALLOC(sl.bufs, q_hit, uint32_t, n * hj::kNumTags)
    HJ_ALLOC(sl.bufs, sh_o, float4, n)
    HJ_ALLOC(sl.bufs, sh_d, float4, n)
    HJ_ALLOC(sl.bufs, sh_c, float4, n)
    HJ_ALLOC(sl.bufs, cnt_ray[0], uint32_t, Gmax)               // (per-workgroup arrays: for the most workgroups a call may use)
    HJ_ALLOC(sl.bufs, cnt_ray[1], uint32_t, Gmax)
    HJ_ALLOC(sl.bufs, cnt_hit, uint32_t, (size_t)Gmax * hj::kNumTags)
    HJ_ALLOC(sl.bufs, cnt_shadow, uint32_t, Gmax)
    HJ_ALLOC(sl.bufs, acc_closest, uint32_t, (size_t)5 * Gmax)  // closest | shadow | hits | unoccluded | proven free, one read-back
    if (rc == HJ_OK) sl.alloc_positions = n;
  }
#undef HJ_ALLOC
  if (rc == HJ_OK) {
    st.acc_shadow = st.acc_closest + G;
    st.acc_hits = st.acc_closest + 2 * (size_t)G;
    st.acc_unoccluded = st.acc_closest + 3 * (size_t)G;
    st.acc_direct = st.acc_closest + 4 * (size_t)G;
    st.pool = pool;
  }
  st.num_wg = G;
  if (rc != HJ_OK) release_slot(sl);
  return rc;
}

// The scene as a render call's kernels see it: the light-shaft grid (api/light_grid.cpp) answers "no shape of the TREE lies
// between this cell and that emitter", so it is taken away from a linear-scan render (scene.glsl:134-158 tests every shape of
// the arrays, in the tree or not) and from a call that asks for every shadow ray to be walked (HJ_RENDER_NO_LIGHT_GRID).
hj::DeviceScene scene_for(const hj_context* ctx, const hj_render_opts& o) {
  hj::DeviceScene sc = ctx->scene;
  if (!o.use_bvh || (o.flags & HJ_RENDER_NO_LIGHT_GRID)) sc.light_grid = nullptr;
  return sc;
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
      stats->shadow_rays_proven_free += h_acc[4 * (size_t)G + i];
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
  st.xcd_deal = (ctx->tuning.xcd_deal != 0 && !all_in_flight && st.num_wg == 2048u && hj::kSlotsPerBlock / 64u == 256u) ? 1u : 0u;
  std::memcpy(sl.h_blocks, blocks, sizeof(hj_image_block) * nb);
  HJ_HIP(ctx, hipMemcpyAsync(sl.d_blocks.p, sl.h_blocks, sizeof(hj_image_block) * nb, hipMemcpyHostToDevice, sl.stream));
  return HJ_OK;
}

int finish_batch(hj_context* ctx, hj_context::BatchSlot& sl, const hj::BatchState& st) {
  const uint32_t G = st.num_wg;
  uint32_t* h_acc = sl.h_counts + (size_t)2 * G;
  HJ_HIP(ctx, hipMemcpyAsync(h_acc, st.acc_closest, sizeof(uint32_t) * 5 * G, hipMemcpyDeviceToHost, sl.stream));
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
  const size_t lds_pad = (size_t)ctx->tuning.lds_pad_kb * 1024;
  const bool pairs = ctx->scene.has_pairs != 0, nt = ctx->scene.stream_state != 0;
  const hj::DeviceScene scn = scene_for(ctx, o);
  if (!o.use_bvh) hipLaunchKernelGGL((hj::k_path_wavefront<false, false, false>), grid, blk, lds_pad, sl.stream, st, scn, o.max_bounces, o.rr_start);
  else if (pairs && nt) hipLaunchKernelGGL((hj::k_path_wavefront<true, true, true>), grid, blk, lds_pad, sl.stream, st, scn, o.max_bounces, o.rr_start);
  else if (pairs) hipLaunchKernelGGL((hj::k_path_wavefront<true, true, false>), grid, blk, lds_pad, sl.stream, st, scn, o.max_bounces, o.rr_start);
  else if (nt) hipLaunchKernelGGL((hj::k_path_wavefront<true, false, true>), grid, blk, lds_pad, sl.stream, st, scn, o.max_bounces, o.rr_start);
  else hipLaunchKernelGGL((hj::k_path_wavefront<true, false, false>), grid, blk, lds_pad, sl.stream, st, scn, o.max_bounces, o.rr_start);
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
  const bool trace_bounces = ctx->tuning.trace_bounces;   // debugging aid: per-bounce table
  for (uint32_t bounce = 0; bounce < o.max_bounces; bounce++) {
    const uint32_t parity = bounce & 1u;
    const size_t ev0 = ctx->events_used;
    int ev = tm.begin(EV_CLOSEST, s);
    if (bvh) hipLaunchKernelGGL(hj::k_trace_closest<true>, grid, blk, 0, s, st, ctx->scene, parity);
    else hipLaunchKernelGGL(hj::k_trace_closest<false>, grid, blk, 0, s, st, ctx->scene, parity);
    tm.end(ev, s);
    ev = tm.begin(EV_SHADE, s);
    hipLaunchKernelGGL(hj::k_shade, grid, blk, 0, s, st, scene_for(ctx, o), parity, o.max_bounces, o.rr_start);
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

int check_opts(hj_context* ctx, const hj_render_opts& o) {
  if (o.recon_radius != 2) return set_error(ctx, HJ_ERR_UNSUPPORTED, "only reconstruction radius 2 (the reference's value) is supported");
  if (!(o.recon_stddev > 0.0f)) return set_error(ctx, HJ_ERR_INVALID, "recon_stddev must be > 0");
  if (o.max_bounces == 0) return set_error(ctx, HJ_ERR_INVALID, "max_bounces must be >= 1");
  return HJ_OK;
}

}  // namespace

extern "C" {

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
  bool no_drain = false;   // HJ_RENDER_NO_DRAIN: a frame of a back-to-back sequence (run_end does not wait)
  std::chrono::steady_clock::time_point wall0;
};

int run_begin(hj_context* ctx, RenderRun& run, const hj_render_opts* opts, hj_render_stats* stats, size_t total_blocks) {
  if (!ctx->have_scene) return set_error(ctx, HJ_ERR_STATE, "render before hj_scene_upload");
  if (!ctx->accum) return set_error(ctx, HJ_ERR_STATE, "render before hj_framebuffer_create");
  ctx->tuning = Tuning::from_env();            // (every render call and hj_reserve start here)
  if (opts) run.o = *opts;
  else hj_default_render_opts(&run.o);
  int rc = check_opts(ctx, run.o);
  if (rc != HJ_OK) return rc;
  HJ_HIP(ctx, hipSetDevice(ctx->device));
  run.split = (run.o.flags & HJ_RENDER_SPLIT_KERNELS) != 0;
  run.no_drain = (run.o.flags & HJ_RENDER_NO_DRAIN) != 0 && !run.split;
  if (run.no_drain) {
    // a frame of a back-to-back sequence: statistics and timing events accumulate in the context until hj_pipeline_wait(ctx, 0)
    run.st = &ctx->pipe_stats;
    if (!ctx->pipe_active) {
      std::memset(run.st, 0, sizeof *run.st);
      ctx->events_used = 0;
      ctx->pipe_k = 0;
      ctx->pipe_wall0 = std::chrono::steady_clock::now();
    }
    run.k = ctx->pipe_k;                     // (the slot rotation goes on where the frame before stopped)
  } else {
    run.st = stats ? stats : &run.local;
    std::memset(run.st, 0, sizeof *run.st);
    ctx->events_used = 0;
  }
  run.tm = Timer{ctx, (run.o.flags & HJ_RENDER_TIME_KERNELS) != 0};
  // Default batch: large batches amortise the latency-bound tail of a batch (measured: cbox+mirror+glass 700 ->
  // 960 Mpaths/s from 512 to 2048 blocks, +2-3 % more at 4096), but at least four batches should exist so that the three
  // slots can overlap (tools/batch_probe.py, rank 0's share of the cbox frame at 8 / 4 / 2 / 1 ranks: a quarter of the blocks
  // per batch beats an eighth by 3.4 / 1.6 / 0.8 / 0.8 %, a half loses 3-5 %).  Path state does not grow with the batch
  // (pool), only the sample buffers do (0.5 GB per 1024 blocks).
  const size_t n = total_blocks;
  const size_t batch_cap = (size_t)ctx->tuning.batch_cap;
  run.batch = run.o.batch_blocks ? run.o.batch_blocks
                                 : (uint32_t)std::min<size_t>(batch_cap, std::max<size_t>(256, ((n + 3) / 4 + 63) / 64 * 64));
  // Frames back to back (HJ_RENDER_NO_DRAIN): the slots overlap ACROSS frames, so a frame need not be cut into four batches for
  // them - half a frame per batch, a whole small one (fewer, larger batches spend less of their time in tails): rank 0's share of
  // the c2 frame at 8 / 4 ranks 21.2 -> 20.3 ms / 40.7 -> 40.2 ms (tools/frames_probe.py); the one-rank frame is at the cap already.
  if (run.no_drain && !run.o.batch_blocks)
    run.batch = (uint32_t)std::min<size_t>(batch_cap, std::max<size_t>(256, (std::max<size_t>((n + 1) / 2, std::min<size_t>(n, 4096)) + 63) / 64 * 64));
  run.batch = std::min<uint32_t>(run.batch, run.split ? 2048u : 32768u);   // (a sample index has 31 bits: 131 072 blocks at most)   // the split path keeps every sample of a batch in flight
  // Footprint: per batch slot 512 KB of samples per ImageBlock of the batch + num_wg x pool positions of
  // path state (185 B each, 217 B with tinted dielectrics).  DEFAULTS that do not fit the device's free memory (other
  // contexts on the GPU, the host application) shrink until they do: first the pool (down to 8192 positions), then the
  // batch; an explicit hj_render_opts::batch_blocks is taken as given and fails with HJ_ERR_NOMEM if it does not fit.
  const size_t small_blocks = (size_t)ctx->tuning.wg_small_blocks;
  ctx->num_wg_eff = (!run.split && n < small_blocks) ? ctx->num_wg_small : ctx->num_wg;
  ctx->pool_eff = ctx->pool;
  ctx->slots_eff = ctx->num_slots;
  if (!run.split) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
      // HJ_MEM_LIMIT_MB (test rig): pretend that no more than this is free
      const int limit_mb = ctx->tuning.mem_limit_mb;
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
        const size_t state = (size_t)ctx->num_wg_eff * pool * (ctx->scene.has_extinction ? 196u + 21u : 164u + 21u);
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
  if (!(run.no_drain && ctx->pipe_active)) {   // (a continuing sequence: the slots are in use, run_submit waits for them one by one)
    rc = sync_all(ctx);
    if (rc != HJ_OK) return rc;
  }
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
  // The contexts of one process (one per GPU, or several on one) enqueue their batches - and allocate a slot's arrays where they
  // are missing - under one lock.  A context that runs out of memory keeps the lock from the moment it gives its slots back until
  // its retry has its arrays: without it another context's larger attempt took what was just released and the first one found not
  // even its smallest configuration (two contexts under HJ_ALLOC_LIMIT_MB starting frames at once: HJ_ERR_NOMEM once in ten runs).
  // An enqueue does not wait for the GPU; the out-of-memory path does, for this context's own batches only.
  std::unique_lock<std::mutex> alloc_lock(alloc_mutex(), std::defer_lock);
  // (Shrinking the last batches of a run - each 1/2 .. 1/6 of what is left - was measured: no change; the ~3.5 ms a
  // frame loses to pipeline fill and drain does not depend on the size of the last kernels.)
  for (size_t begin = 0; begin < n && rc == HJ_OK; run.k++) {
    const uint32_t nb = (uint32_t)std::min<size_t>(run.batch, n - begin);
    hj_context::BatchSlot& sl = ctx->slots[run.k % ctx->slots_eff];
    hj_context::BatchSlot& other = ctx->slots[(run.k + ctx->slots_eff - 1) % ctx->slots_eff];   // the previous batch's slot
    rc = harvest(ctx, sl, run.st);          // an older batch used this slot: its state arrays are free again
    if (rc != HJ_OK) break;
    if (!run.split && !alloc_lock.owns_lock()) alloc_lock.lock();   // (the split path, a diagnostic, waits for its batch inside)
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
      continue;                              // (with the lock)
    }
    if (alloc_lock.owns_lock()) alloc_lock.unlock();
    begin += nb;
  }
  return rc;
}

// Drains the slots (also after an error, so that nothing of this run is still in flight) and closes the statistics.
// Kernel times of the events recorded since they were last reset, into *st_out (HJ_RENDER_TIME_KERNELS).
void collect_timing(hj_context* ctx, hj_render_stats* st_out) {
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

int run_end(hj_context* ctx, RenderRun& run, int rc) {
  if (run.no_drain && rc == HJ_OK) {
    // a frame of a back-to-back sequence: nothing is waited for; one event behind the frame's last batch (its slot's stream
    // waits for its reconstruction, and the reconstructions are chained in order: everything of the frame precedes it)
    hipEvent_t ev = nullptr;
    if (!ctx->frame_event_pool.empty()) { ev = ctx->frame_event_pool.back(); ctx->frame_event_pool.pop_back(); }
    else HJ_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipStream_t last = run.k != ctx->pipe_k ? ctx->slots[(run.k - 1) % ctx->slots_eff].stream : ctx->stream;   // (no batch: nothing to wait for)
    if (hipEventRecord(ev, last) != hipSuccess) { ctx->frame_event_pool.push_back(ev); return set_error(ctx, HJ_ERR_DEVICE, "hipEventRecord failed"); }
    ctx->frame_events.push_back(ev);
    ctx->pipe_active = true;
    ctx->pipe_k = run.k;
    run.st->paths += run.paths;
    return HJ_OK;
  }
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
    st_out->paths = run.no_drain ? st_out->paths + run.paths : run.paths;
    collect_timing(ctx, st_out);
  }
  if (run.no_drain) {                          // (an error inside a sequence: everything has been drained above)
    for (hipEvent_t e : ctx->frame_events) ctx->frame_event_pool.push_back(e);
    ctx->frame_events.clear();
    ctx->pipe_active = false;
  }
  return rc;
}
}  // namespace

int hj_reserve(hj_context* ctx, size_t total_blocks, const hj_render_opts* opts) {
  if (!ctx) return HJ_ERR_INVALID;
  HJ_NOT_BUSY(ctx);
  HJ_NOT_PIPELINED(ctx);
  if (total_blocks == 0) return HJ_OK;
  RenderRun run;
  int rc = run_begin(ctx, run, opts, nullptr, total_blocks);        // (the call's batch size, pool and workgroup count)
  if (rc != HJ_OK || run.split) return rc;
  std::lock_guard<std::mutex> alloc_lock(alloc_mutex());           // (run_submit: one context at a time sizes its slots)
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
  HJ_NOT_PIPELINED(ctx);
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
  if (!(opts && (opts->flags & HJ_RENDER_NO_DRAIN) && !(opts->flags & HJ_RENDER_SPLIT_KERNELS))) HJ_NOT_PIPELINED(ctx);
  return render_frame_impl(ctx, spp, master_seed, pass_begin, pass_end, rank, world, opts, stats);
}

// Waits until at most `keep` of the frames submitted with HJ_RENDER_NO_DRAIN are in flight; keep = 0: drains the pipeline
// and closes the sequence's statistics (include/hijiki_hip.h).
int hj_pipeline_wait(hj_context* ctx, uint32_t keep, hj_render_stats* totals) {
  if (!ctx) return HJ_ERR_INVALID;
  HJ_NOT_BUSY(ctx);
  HJ_HIP(ctx, hipSetDevice(ctx->device));
  while (ctx->frame_events.size() > keep) {
    hipEvent_t ev = ctx->frame_events.front();
    HJ_HIP(ctx, hipEventSynchronize(ev));
    ctx->frame_events.erase(ctx->frame_events.begin());
    ctx->frame_event_pool.push_back(ev);
  }
  if (keep != 0) return HJ_OK;
  if (!ctx->pipe_active) {                    // nothing in flight: the totals of the last sequence again
    if (totals) *totals = ctx->pipe_stats;
    return HJ_OK;
  }
  int rc = HJ_OK;
  for (auto& sl : ctx->slots) {
    const int rc2 = harvest(ctx, sl, &ctx->pipe_stats);
    if (rc == HJ_OK) rc = rc2;
  }
  {
    const int rc2 = sync_all(ctx);
    if (rc == HJ_OK) rc = rc2;
  }
  ctx->pipe_active = false;
  if (rc == HJ_OK) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) rc = set_error(ctx, HJ_ERR_DEVICE, "kernel launch: %s", hipGetErrorString(e));
  }
  if (rc == HJ_OK) {
    ctx->pipe_stats.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ctx->pipe_wall0).count();
    collect_timing(ctx, &ctx->pipe_stats);
    if (totals) *totals = ctx->pipe_stats;
  }
  return rc;
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
  HJ_NOT_PIPELINED(ctx);
  if (opts && (opts->flags & HJ_RENDER_NO_DRAIN))
    return set_error(ctx, HJ_ERR_INVALID, "HJ_RENDER_NO_DRAIN belongs to hj_render_frame (it returns at once by itself), not to the asynchronous form");
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


int hj_debug_trace(hj_context* ctx, const float* rays, size_t n, uint32_t use_bvh, uint32_t any_hit, float* hits) {
  if (!ctx) return HJ_ERR_INVALID;
  HJ_NOT_BUSY(ctx);
  HJ_NOT_PIPELINED(ctx);
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
  HJ_NOT_PIPELINED(ctx);
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

}  // extern "C"
