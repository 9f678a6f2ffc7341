// Internals shared by the translation units of libhijiki_hip.so (api/*.hip): the context object, device buffers and the
// small helpers every entry point uses.  Nothing here is part of the C ABI (include/hijiki_hip.h).
//
//   api/context.hip       context create / destroy, framebuffer, errors, environment switches            (no kernels)
//   api/scene_upload.hip  hj_scene_upload: validation and the re-layout of the reference's scene arrays  (no kernels)
//   api/render.hip        batch slots, the launches of kernels/hj_kernels.h, render calls, probes
//   api/comm.hip          RCCL (dlopen), hj_comm_*, hj_reduce_framebuffers
//   api/lbvh_build.hip    hj_build_bvh_device: host half of kernels/hj_lbvh.h
//   api/tree_vote.hip     hj_tune_bvh_device: host half of kernels/hj_vote.h (child order voted by sampled rays)
#pragma once
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

#include "../../../include/hijiki_hip.h"
#include "hj_tuning.h"
#include "../host/blockgen.hpp"
#include "../kernels/hj_device.h"

namespace hjapi {

extern std::atomic<size_t> g_dev_bytes;   // device memory held through DevBuf by every context of the process

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

}  // namespace hjapi

using hjapi::DevBuf;
using hjapi::EventPair;

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
    uint32_t* h_counts = nullptr;         // pinned read-back: 2 (split-path ray counts) + 5 (statistics) arrays of num_wg words
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
  uint32_t pool = 32768;                 // path slots per workgroup of the fused kernel (HJ_POOL)
  uint32_t pool_eff = 32768;             // ... as the current render call uses it (lowered when device memory is short)

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

  // Frames back to back WITHOUT draining the batch pipeline between them (HJ_RENDER_NO_DRAIN, hj_pipeline_wait): one event per
  // frame submitted (recorded behind its last batch), the statistics accumulated since the last full drain, the slot rotation
  // carried from frame to frame.
  std::vector<hipEvent_t> frame_events;   // oldest first: frames submitted, not yet waited for
  std::vector<hipEvent_t> frame_event_pool;
  bool pipe_active = false;               // something submitted with HJ_RENDER_NO_DRAIN has not been drained yet
  hj_render_stats pipe_stats{};
  size_t pipe_k = 0;
  std::chrono::steady_clock::time_point pipe_wall0;

  // The tree hj_build_bvh_device built last, still on the device together with the shape arrays it was built over: hj_scene_upload
  // with scene->bvh == NULL takes it over (no trip through the host), hj_bvh_device_read copies it out.  Released by the next
  // build, by the upload that consumes it, and with the context.
  struct ResidentTree {
    hjapi::DevBuf nodes, spheres, quads, triangles, vertices;
    size_t total = 0, ns = 0, nq = 0, nt = 0, nv = 0;
    uint64_t shapes_hash = 0;                    // hjapi::shape_arrays_hash of the arrays the tree was built over
    bool valid = false;
    void release() { nodes.release(); spheres.release(); quads.release(); triangles.release(); vertices.release(); total = 0; valid = false; }
  } resident;

  // the library's environment switches (api/hj_tuning.h) as the entry point in progress read them: hj_context_create, then every
  // hj_scene_upload / render call / BVH build refreshes the copy at its start; nothing below an entry point reads the environment
  hjapi::Tuning tuning;

  // hj_last_error: the worker thread writes `error` while the caller's thread may read it
  std::mutex err_mu;
};

namespace hjapi {

int set_error(hj_context* ctx, int code, const char* fmt, ...) __attribute__((format(printf, 3, 4)));
std::string get_error(hj_context* ctx);
// A fingerprint of a scene's shape arrays (counts + 4096 evenly spaced 16-byte pieces of each): guards the hand-over of a tree that
// stayed on the device against an upload of OTHER geometry with the same counts (an accident, not an adversary).
uint64_t shape_arrays_hash(const hj_scene_desc* s);
void put_error(hj_context* ctx, const std::string& text);
int dev_alloc(hj_context* ctx, DevBuf& b, size_t bytes);
std::mutex& alloc_mutex();                           // process-wide: a context sizing its batch slots (api/render.hip run_submit)
int validate_scene(hj_context* ctx, const hj_scene_desc* s);   // api/scene_upload.hip: every invariant an upload checks
void release_scene(hj_context* ctx);
void release_slot(hj_context::BatchSlot& sl);
void release_batch(hj_context* ctx);
int sync_all(hj_context* ctx);                       // drains the context's streams (api/context.hip)
void drop_cached_comms(hj_context* ctx);             // api/comm.hip: the communicators hj_reduce_framebuffers made for ctx

}  // namespace hjapi

// Entry points that touch the context's device state refuse to run while an asynchronous frame is in flight on it
// (the worker thread owns the slots, the streams and the framebuffer until hj_sync).
#define HJ_NOT_BUSY(ctx)                                                                                          \
  do {                                                                                                            \
    if ((ctx)->busy.load(std::memory_order_acquire))                                                              \
      return set_error(ctx, HJ_ERR_STATE, "%s: an asynchronous frame is in flight on this context: call hj_sync first", __func__); \
  } while (0)

// ... and while frames submitted with HJ_RENDER_NO_DRAIN are still in flight (hj_pipeline_wait(ctx, 0, ...) drains them).
#define HJ_NOT_PIPELINED(ctx)                                                                                     \
  do {                                                                                                            \
    if ((ctx)->pipe_active)                                                                                       \
      return set_error(ctx, HJ_ERR_STATE, "%s: frames submitted with HJ_RENDER_NO_DRAIN are in flight: call hj_pipeline_wait(ctx, 0, ...) first", __func__); \
  } while (0)

#define HJ_HIP(ctx, call)                                                                         \
  do {                                                                                            \
    hipError_t e_ = (call);                                                                       \
    if (e_ != hipSuccess)                                                                         \
      return set_error(ctx, e_ == hipErrorOutOfMemory ? HJ_ERR_NOMEM : HJ_ERR_DEVICE, "%s: %s", #call, \
                       hipGetErrorString(e_));                                                    \
  } while (0)

namespace hjapi {

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

}  // namespace hjapi
