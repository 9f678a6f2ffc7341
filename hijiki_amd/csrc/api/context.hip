// libhijiki_hip.so - C ABI (include/hijiki_hip.h) over the gfx950 kernels: context, framebuffer, errors.
//
// Replaces, for the hot path only, what the reference's Renderer does through
// wgpu (reference src/main.rs:1143-1424): resource creation, scene upload,
// the per-block dispatch loop and the read-back.  No CPU fallback exists: every
// entry point that computes needs a HIP device and fails with HJ_ERR_DEVICE
// otherwise.
#include "hj_internal.h"

#pragma clang fp contract(off)

using namespace hjapi;

namespace {
thread_local std::string g_create_error;
}

namespace hjapi {

std::atomic<size_t> g_dev_bytes{0};
std::mutex& alloc_mutex() { static std::mutex mu; return mu; }

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

uint64_t shape_arrays_hash(const hj_scene_desc* s) {
  uint64_t h = 0xcbf29ce484222325ull;
  auto mix = [&](uint64_t v) { h = (h ^ v) * 0x100000001b3ull; h ^= h >> 29; };
  auto array = [&](const void* p, size_t count, size_t elem) {
    mix(count);
    const size_t bytes = count * elem;
    if (!p || bytes < 16) { if (p) for (size_t i = 0; i < bytes; i++) mix(static_cast<const unsigned char*>(p)[i]); return; }
    const size_t pieces = std::min<size_t>(4096, bytes / 16), step = (bytes - 16) / std::max<size_t>(1, pieces - 1);
    for (size_t k = 0; k < pieces; k++) {
      uint64_t w[2];
      std::memcpy(w, static_cast<const char*>(p) + std::min(bytes - 16, k * step), 16);
      mix(w[0]); mix(w[1]);
    }
  };
  array(s->spheres, s->num_spheres, sizeof(hj_sphere));
  array(s->quads, s->num_quads, sizeof(hj_quad));
  array(s->triangles, s->num_triangles, sizeof(hj_triangle));
  array(s->vertices, s->num_vertices, sizeof(hj_vertex));
  return h;
}

std::string get_error(hj_context* ctx) {
  std::lock_guard<std::mutex> lock(ctx->err_mu);
  return ctx->error;
}

// (for a call that tried something optional, failed at it and went on: hj_last_error must not show the attempt's message after HJ_OK)
void put_error(hj_context* ctx, const std::string& text) {
  std::lock_guard<std::mutex> lock(ctx->err_mu);
  ctx->error = text;
}

int dev_alloc(hj_context* ctx, DevBuf& b, size_t bytes) {
  if (b.bytes >= bytes && b.p) return HJ_OK;
  b.release();
  if (bytes == 0) bytes = 16;
  // HJ_ALLOC_LIMIT_MB (test rig): the process's contexts together may hold no more than this; an allocation beyond it fails
  // the way hipMalloc does on a full device - how the out-of-memory paths run on a 288 GB card.
  static const size_t limit = (size_t)Tuning::from_env().alloc_limit_mb << 20;
  if (limit != 0 && g_dev_bytes.load(std::memory_order_relaxed) + bytes > limit)
    return set_error(ctx, HJ_ERR_NOMEM, "hipMalloc(%zu bytes): out of memory (HJ_ALLOC_LIMIT_MB)", bytes);
  HJ_HIP(ctx, hipMalloc(&b.p, bytes));
  b.bytes = bytes;
  g_dev_bytes.fetch_add(bytes, std::memory_order_relaxed);
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

}  // namespace hjapi

extern "C" {

uint32_t hj_version(void) { return (0u << 16) | (3u << 8) | 0u; }   // 0.3.0: hj_render_stats grew (shadow_rays_proven_free), hj_scene_upload refuses non-tree link arrays

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
  // Tuning knobs (environment overrides exist only for sweeps; the defaults are the measured optima: DESIGN.md 4, profiles/NOTES.md).
  const Tuning tn = ctx->tuning = Tuning::from_env();
  ctx->num_wg = (uint32_t)ctx->num_cus * (uint32_t)tn.wg_per_cu;   // 8 x 4 waves = the 32-wave CU limit
  // Small render calls (a rank's share of a frame on many GPUs) run 6 workgroups per CU: all of a kernel's workgroups are then
  // resident at once (7 x 4 waves fit a CU at 72 registers; with 8 per CU the last eighth of a batch's workgroups start when the
  // first ones end, a thin second wave that nothing covers at the end of a short frame) - an 8-rank share of the c2 frame 24.6 ->
  // 23.4 ms; large calls keep 8 (the 32768-block frame: 161 against 170 ms).  HJ_WG_SMALL / HJ_WG_SMALL_BLOCKS.
  ctx->num_wg_small = std::min(ctx->num_wg, (uint32_t)ctx->num_cus * (uint32_t)tn.wg_small);
  ctx->num_wg_eff = ctx->num_wg;
  ctx->num_slots = (uint32_t)std::min<int>(tn.slots, (int)kMaxSlots);
  // positions per workgroup: the more paths a workgroup has in flight, the longer the walk phases of its rounds and the less
  // their ramp-down weighs.  One blocking frame after the other: 32768 is +4.5 % over 8192 on c2 / c3, 65536 (every sample
  // of a workgroup's share of an 8192-block batch in flight at once: round 3's default, 24.7 GB of path state per slot) +6 %.
  // Frames back to back: 16384 ... 65536 are the same within a per cent (c2 3433-3441 / 3404-3412 / 3391-3411, c3 2769 / 2773-2791 /
  // 2779-2788, c4 1166-1171 / 1160-1172 / 1160 Mrays/s at 16384 / 32768 / 65536; 8192: -2.5 %): the tails that a large pool
  // shortens are covered by the next frame there.  32768 = 12.4 GB per slot, 50 GB per context instead of 86.
  ctx->pool = (uint32_t)tn.pool / 64u * 64u;
  for (auto& sl : ctx->slots) {
    if ((e = hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate");
    if ((e = hipHostMalloc((void**)&sl.h_counts, sizeof(uint32_t) * 7 * ctx->num_wg, hipHostMallocDefault)) != hipSuccess) return fail(e, "hipHostMalloc");
    for (hipEvent_t* ev : {&sl.ev_count[0], &sl.ev_count[1], &sl.ev_recon, &sl.ev_done, &sl.ev_path})
      if ((e = hipEventCreateWithFlags(ev, hipEventDisableTiming)) != hipSuccess) return fail(e, "hipEventCreate");
  }
  // HJ_RECON_PRIORITY (default 1): the reconstructions run on one stream per slot of the device's highest priority.
  if (tn.recon_priority != 0) {
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
  drop_cached_comms(ctx);
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  for (auto& sl : ctx->slots) {
    if (sl.stream) (void)hipStreamSynchronize(sl.stream);
    if (sl.rstream) (void)hipStreamSynchronize(sl.rstream);
  }
  release_scene(ctx);
  ctx->resident.release();
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
  for (hipEvent_t e : ctx->frame_events) (void)hipEventDestroy(e);
  for (hipEvent_t e : ctx->frame_event_pool) (void)hipEventDestroy(e);
  for (auto& ep : ctx->events) {
    (void)hipEventDestroy(ep.a);
    (void)hipEventDestroy(ep.b);
  }
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int hj_framebuffer_create(hj_context* ctx, uint32_t width, uint32_t height, void* external) {
  if (!ctx) return HJ_ERR_INVALID;
  HJ_NOT_BUSY(ctx);
  HJ_NOT_PIPELINED(ctx);
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
  HJ_NOT_PIPELINED(ctx);
  if (!ctx->accum) return set_error(ctx, HJ_ERR_STATE, "no framebuffer");
  HJ_HIP(ctx, hipSetDevice(ctx->device));
  HJ_HIP(ctx, hipMemsetAsync(ctx->accum, 0, (size_t)ctx->width * ctx->height * sizeof(float4), ctx->stream));
  HJ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return HJ_OK;
}

void* hj_framebuffer_device_ptr(hj_context* ctx) { return ctx ? ctx->accum : nullptr; }

// Frames submitted from now on accumulate into `external` (same size as the framebuffer hj_framebuffer_create made or was
// given); no synchronisation: frames already submitted (HJ_RENDER_NO_DRAIN) keep the buffer they were submitted with.
int hj_framebuffer_bind(hj_context* ctx, void* external) {
  if (!ctx) return HJ_ERR_INVALID;
  HJ_NOT_BUSY(ctx);
  if (!ctx->accum) return set_error(ctx, HJ_ERR_STATE, "no framebuffer: hj_framebuffer_create first (it fixes the size)");
  if (!external || (reinterpret_cast<uintptr_t>(external) & 15u) != 0) return set_error(ctx, HJ_ERR_INVALID, "external framebuffer must be non-null and 16-byte aligned");
  if (ctx->accum_owned) return set_error(ctx, HJ_ERR_STATE, "the context owns its framebuffer: create it with an external buffer to bind others");
  ctx->accum = static_cast<float4*>(external);
  return HJ_OK;
}

int hj_framebuffer_read(hj_context* ctx, float* host_rgba) {
  if (!ctx) return HJ_ERR_INVALID;
  HJ_NOT_BUSY(ctx);
  HJ_NOT_PIPELINED(ctx);
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

uint32_t hj_block_seed(uint64_t master, uint32_t pass, uint32_t j) { return hijiki::block_seed(master, pass, j); }
uint32_t hj_block_owner(uint32_t width, uint32_t height, uint32_t pass, uint32_t j, uint32_t world) {
  if (!width || !height || !world) return 0;
  return hijiki::BlockGrid(width, height, HJ_BLOCK_SIZE).owner(pass, j, world);
}
void hj_pass_offset(uint64_t master, uint32_t k, float out[2]) { hijiki::pass_offset(master, k, out); }

}  // extern "C"
