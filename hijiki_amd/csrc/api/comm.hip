// The exchange step of the multi-GPU path inside ONE process: RCCL through dlopen (no link-time dependency), the
// communicator object over the contexts of a process, one sum-reduce of the framebuffers over xGMI.
#include <dlfcn.h>
#include "hj_internal.h"

#pragma clang fp contract(off)

using namespace hjapi;

extern "C" {

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
bool shared_gpu_allowed() { return Tuning::from_env().comm_shared_gpu != 0; }

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
  if ((n > 1 || Tuning::from_env().comm_force_rccl != 0) && !shared) {
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
    if (x->pipe_active) return set_error(r, HJ_ERR_STATE, "context %d: frames submitted with HJ_RENDER_NO_DRAIN are in flight: hj_pipeline_wait first", i);
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

namespace hjapi {
void drop_cached_comms(hj_context* ctx) {
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
}  // namespace hjapi
