// hj_tune_bvh_device and the host half of kernels/hj_vote.h: the child order of a flattened tree voted by sampled rays, on the device
// (host counterpart: host/tree_opt.cpp order_children_by_rays, hjh_compiled_tune_bvh).
#include "hj_internal.h"
#include "tree_vote.hpp"
#include "../kernels/hj_vote.h"

#pragma clang fp contract(off)

using namespace hjapi;

namespace hjapi {

namespace {
struct Scratch {
  hj_context* ctx;
  std::vector<DevBuf> bufs;
  explicit Scratch(hj_context* c) : ctx(c) { bufs.reserve(24); }
  ~Scratch() { for (auto& b : bufs) b.release(); }
  int get(size_t bytes, void** p) {
    bufs.emplace_back();
    const int rc = dev_alloc(ctx, bufs.back(), std::max<size_t>(bytes, 16));
    *p = bufs.back().p;
    return rc;
  }
};
}  // namespace

#define HJ_SCRATCH(sc, ptr, type, count) do { void* p_ = nullptr; const int rc_ = (sc).get(sizeof(type) * (count), &p_); if (rc_ != HJ_OK) return rc_; ptr = static_cast<type*>(p_); } while (0)

int vote_on_device(hj_context* ctx, const hj_scene_desc* s, const VoteShapes& shapes, const hj_bvh_node* d_nodes, size_t N, size_t paths,
                   hj_bvh_node* d_out, bool timing, VoteResult* result) {
  hipStream_t st = ctx->stream;
  if (result) *result = VoteResult{};
  if (N < 3 || paths == 0) {
    HJ_HIP(ctx, hipMemcpyAsync(d_out, d_nodes, sizeof(hj_bvh_node) * N, hipMemcpyDeviceToDevice, st));
    return HJ_OK;
  }
  if (N >= 0x7FFFFFFFu) return set_error(ctx, HJ_ERR_UNSUPPORTED, "tree of %zu records: too large for the device vote", N);
  paths = std::min<size_t>(paths, (size_t)1 << 24);
  auto t_last = std::chrono::steady_clock::now();
  auto mark = [&](const char* what) {
    if (!timing) return;
    (void)hipStreamSynchronize(st);
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "ray-voted child order (device): %-20s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  Scratch sc(ctx);
  const size_t nshapes = s->num_spheres + s->num_quads + s->num_triangles;
  hj::vote::Scene vs{};
  vs.spheres = shapes.spheres; vs.quads = shapes.quads; vs.triangles = shapes.triangles; vs.vertices = shapes.vertices;
  vs.ns = (uint32_t)s->num_spheres; vs.nq = (uint32_t)s->num_quads; vs.nt = (uint32_t)s->num_triangles;
  vs.cam = s->camera;
  vs.nodes = reinterpret_cast<const float4*>(d_nodes);
  vs.N = (uint32_t)N;
  vs.w_shadow = (uint32_t)Tuning::pick(ctx->tuning.bvh_vote_shadow, N >= (size_t)ctx->tuning.stream_min_nodes ? 4 : 1);
  if (s->materials && s->num_materials == nshapes && nshapes) {
    uint32_t* m = nullptr;
    HJ_SCRATCH(sc, m, uint32_t, nshapes);
    HJ_HIP(ctx, hipMemcpyAsync(m, s->materials, sizeof(uint32_t) * nshapes, hipMemcpyHostToDevice, st));
    vs.materials = m;
  }
  if (s->emitters && s->num_emitters) {
    hj_emitter* e = nullptr;
    HJ_SCRATCH(sc, e, hj_emitter, s->num_emitters);
    HJ_HIP(ctx, hipMemcpyAsync(e, s->emitters, sizeof(hj_emitter) * s->num_emitters, hipMemcpyHostToDevice, st));
    vs.emitters = e; vs.ne = (uint32_t)std::min<size_t>(s->num_emitters, 0x7FFFFFFFu);
  }
  if (s->dielectric && s->num_dielectric) {
    hj_dielectric* d = nullptr;
    HJ_SCRATCH(sc, d, hj_dielectric, s->num_dielectric);
    HJ_HIP(ctx, hipMemcpyAsync(d, s->dielectric, sizeof(hj_dielectric) * s->num_dielectric, hipMemcpyHostToDevice, st));
    vs.dielectric = d; vs.ndielectric = (uint32_t)std::min<size_t>(s->num_dielectric, 0x7FFFFFFFu);
  }
  HJ_SCRATCH(sc, vs.gain_l, unsigned long long, N);
  HJ_SCRATCH(sc, vs.gain_r, unsigned long long, N);
  HJ_HIP(ctx, hipMemsetAsync(vs.gain_l, 0, sizeof(unsigned long long) * N, st));
  HJ_HIP(ctx, hipMemsetAsync(vs.gain_r, 0, sizeof(unsigned long long) * N, st));
  mark("uploads");
  const uint32_t np = (uint32_t)paths;
  hipLaunchKernelGGL(hj::vote::k_vote_paths, dim3((np + 63u) / 64u), dim3(64), 0, st, vs, np);
  mark("sampled paths");

  hj::vote::Reorder r{};
  r.in = vs.nodes; r.out = reinterpret_cast<float4*>(d_out); r.N = vs.N; r.gain_l = vs.gain_l; r.gain_r = vs.gain_r;
  HJ_SCRATCH(sc, r.depth, uint32_t, N);
  HJ_SCRATCH(sc, r.npos, uint32_t, N);
  HJ_SCRATCH(sc, r.nexit, uint32_t, N);
  HJ_SCRATCH(sc, r.info, uint32_t, 4);
  HJ_HIP(ctx, hipMemsetAsync(r.info, 0, sizeof(uint32_t) * 4, st));
  const dim3 blk(256), grid((vs.N + 255u) / 256u);
  hipLaunchKernelGGL(hj::vote::k_ro_init, grid, blk, 0, st, r);
  uint32_t info[4] = {0, 0, 0, 0};
  for (uint32_t L = 0; L < 8192; L += 16) {                       // (the levels end when one has no inner node)
    for (uint32_t k = 0; k < 16; k++) hipLaunchKernelGGL(hj::vote::k_ro_level, grid, blk, 0, st, r, L + k);
    HJ_HIP(ctx, hipMemcpyAsync(info, r.info, sizeof info, hipMemcpyDeviceToHost, st));
    HJ_HIP(ctx, hipStreamSynchronize(st));
    if (info[0] < L + 16) break;
  }
  hipLaunchKernelGGL(hj::vote::k_ro_scatter, grid, blk, 0, st, r);
  HJ_HIP(ctx, hipMemcpyAsync(info, r.info, sizeof info, hipMemcpyDeviceToHost, st));
  HJ_HIP(ctx, hipStreamSynchronize(st));
  HJ_HIP(ctx, hipGetLastError());
  mark("exchange");
  if (info[2] & 1u) return set_error(ctx, HJ_ERR_INVALID, "ray-voted child order: the array is not a pre-order skip-link tree");
  if (info[2] & 2u) return set_error(ctx, HJ_ERR_UNSUPPORTED, "ray-voted child order: the tree is deeper than 8192 levels");
  if (result) { result->exchanged = info[1]; result->levels = info[0]; }
  if (timing) std::fprintf(stderr, "ray-voted child order (device): %zu paths, %u of %zu inner nodes exchanged, %u levels\n", paths, info[1], N / 2, info[0]);
  return HJ_OK;
}

int put_records(hj_context* ctx, const std::vector<std::pair<uint32_t, hj_bvh_node>>& records, hj_bvh_node* d_array) {
  if (records.empty()) return HJ_OK;
  hipStream_t st = ctx->stream;
  Scratch sc(ctx);
  std::vector<uint32_t> pos(records.size());
  std::vector<hj_bvh_node> rec(records.size());
  for (size_t k = 0; k < records.size(); k++) { pos[k] = records[k].first; rec[k] = records[k].second; }
  uint32_t* d_pos = nullptr;
  hj_bvh_node* d_rec = nullptr;
  HJ_SCRATCH(sc, d_pos, uint32_t, pos.size());
  HJ_SCRATCH(sc, d_rec, hj_bvh_node, rec.size());
  HJ_HIP(ctx, hipMemcpyAsync(d_pos, pos.data(), sizeof(uint32_t) * pos.size(), hipMemcpyHostToDevice, st));
  HJ_HIP(ctx, hipMemcpyAsync(d_rec, rec.data(), sizeof(hj_bvh_node) * rec.size(), hipMemcpyHostToDevice, st));
  const uint32_t n = (uint32_t)records.size();
  hipLaunchKernelGGL(hj::vote::k_put_records, dim3((n + 255u) / 256u), dim3(256), 0, st, d_pos, reinterpret_cast<const float4*>(d_rec), n,
                     reinterpret_cast<float4*>(d_array));
  HJ_HIP(ctx, hipStreamSynchronize(st));                         // (the staging vectors and buffers live on this frame)
  return HJ_OK;
}

}  // namespace hjapi

extern "C" {

// No counterpart upstream (the reference walks the tree the `bvh` crate hands it, src/main.rs:199-231); host form: hjh_compiled_tune_bvh.
int hj_tune_bvh_device(hj_context* ctx, const hj_scene_desc* s, hj_bvh_node* out_nodes, size_t capacity, size_t vote_paths) {
  if (!ctx) return HJ_ERR_INVALID;
  HJ_NOT_BUSY(ctx);
  HJ_NOT_PIPELINED(ctx);
  if (!s || !out_nodes) return set_error(ctx, HJ_ERR_INVALID, "null argument");
  const size_t N = s->num_bvh_nodes;
  if (capacity < N) return set_error(ctx, HJ_ERR_INVALID, "node buffer holds %zu records, the tree has %zu", capacity, N);
  int rc = validate_scene(ctx, s);
  if (rc != HJ_OK) return rc;
  if (N == 0) return HJ_OK;
  HJ_HIP(ctx, hipSetDevice(ctx->device));
  const Tuning tn = ctx->tuning = Tuning::from_env();
  const bool timing = tn.lbvh_timing != 0;
  hipStream_t st = ctx->stream;
  Scratch sc(ctx);
  float4 *sp = nullptr, *qd = nullptr;
  hj_triangle* tr = nullptr;
  hj_vertex* vx = nullptr;
  hj_bvh_node *d_in = nullptr, *d_out = nullptr;
  HJ_SCRATCH(sc, sp, float4, s->num_spheres);
  HJ_SCRATCH(sc, qd, float4, 3 * s->num_quads);
  HJ_SCRATCH(sc, tr, hj_triangle, s->num_triangles);
  HJ_SCRATCH(sc, vx, hj_vertex, s->num_vertices);
  HJ_SCRATCH(sc, d_in, hj_bvh_node, N);
  HJ_SCRATCH(sc, d_out, hj_bvh_node, N);
  if (s->num_spheres) HJ_HIP(ctx, hipMemcpyAsync(sp, s->spheres, sizeof(float4) * s->num_spheres, hipMemcpyHostToDevice, st));
  if (s->num_quads) HJ_HIP(ctx, hipMemcpyAsync(qd, s->quads, sizeof(float4) * 3 * s->num_quads, hipMemcpyHostToDevice, st));
  if (s->num_triangles) HJ_HIP(ctx, hipMemcpyAsync(tr, s->triangles, sizeof(hj_triangle) * s->num_triangles, hipMemcpyHostToDevice, st));
  if (s->num_vertices) HJ_HIP(ctx, hipMemcpyAsync(vx, s->vertices, sizeof(hj_vertex) * s->num_vertices, hipMemcpyHostToDevice, st));
  HJ_HIP(ctx, hipMemcpyAsync(d_in, s->bvh, sizeof(hj_bvh_node) * N, hipMemcpyHostToDevice, st));
  const VoteShapes shapes{sp, qd, tr, vx};
  rc = vote_on_device(ctx, s, shapes, d_in, N, vote_paths, d_out, timing, nullptr);
  if (rc != HJ_OK) return rc;
  HJ_HIP(ctx, hipMemcpyAsync(out_nodes, d_out, sizeof(hj_bvh_node) * N, hipMemcpyDeviceToHost, st));
  HJ_HIP(ctx, hipStreamSynchronize(st));
  return HJ_OK;
}

}  // extern "C"
