// Test infrastructure (CPU): the K-wide tree of hijiki_amd/csrc/wide_tree.h visits, per ray, exactly the shapes the
// reference's binary skip-link walk visits (reference shader/scene.glsl:97-133), in the same order, each with the same
// tMax.  Both walks below use the SAME float32 box test (the kernels' formula, hj_kernels.h node_step) and the same shape
// tests, so comparing the two event sequences checks the tree transformation and the wide walk's rules, nothing else.
//   g++ -O2 -std=c++17 -ffp-contract=off -mfma -shared -fPIC -o wide_check.so wide_check.cpp
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../hijiki_amd/csrc/wide_tree.h"

namespace {
constexpr float kEps = 1e-4f;
struct Ray { float o[3], d[3], tmin, tmax; };
struct Scene {
  const hj_bvh_node* bvh; size_t N;
  const float* spheres; size_t ns;      // 4 floats
  const float* quads; size_t nq;        // 12 floats
  const uint32_t* tris; size_t nt;      // 3 indices
  const float* verts;                   // 8 floats per vertex (pos @0)
};
struct Event { uint32_t shape; uint32_t tmax_bits; };

float dot3(const float* a, const float* b) { return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])); }
void cross3(const float* a, const float* b, float* c) {
  c[0] = fmaf(a[1], b[2], -(a[2] * b[1])); c[1] = fmaf(a[2], b[0], -(a[0] * b[2])); c[2] = fmaf(a[0], b[1], -(a[1] * b[0]));
}
bool tri_like(const float* a, const float* e1, const float* e2, const Ray& r, bool quad, float& t) {
  float n[3], ro[3], q[3];
  cross3(e1, e2, n);
  for (int k = 0; k < 3; k++) ro[k] = r.o[k] - a[k];
  cross3(ro, r.d, q);
  const float d = 1.0f / dot3(r.d, n);
  const float u = d * (-dot3(q, e2)), v = d * dot3(q, e1);
  if (quad ? (u < 0.f || u > 1.f || v < 0.f || v > 1.f) : (u < 0.f || v < 0.f || u + v > 1.f)) return false;
  t = d * (-dot3(n, ro));
  return r.tmin <= t && t <= r.tmax;
}
bool shape_test(const Scene& s, uint32_t shape, const Ray& r, float& t) {
  if (shape < s.ns) {
    const float* sp = s.spheres + 4 * shape;
    float l[3];
    for (int k = 0; k < 3; k++) l[k] = r.o[k] - sp[k];
    const float b = 2.0f * dot3(r.d, l), c = dot3(l, l) - sp[3] * sp[3];
    float disc = b * b - 4.0f * c;
    if (disc < 0.f) return false;
    disc = sqrtf(disc);
    const float t0 = -0.5f * (b + disc), t1 = -0.5f * (b - disc);
    if (r.tmin <= t0 && t0 <= r.tmax) { t = t0; return true; }
    if (r.tmin <= t1 && t1 <= r.tmax) { t = t1; return true; }
    return false;
  }
  if (shape < s.ns + s.nq) {
    const float* q = s.quads + 12 * (shape - s.ns);
    return tri_like(q, q + 4, q + 8, r, true, t);
  }
  const uint32_t* ix = s.tris + 3 * (size_t)(shape - s.ns - s.nq);
  const float *A = s.verts + 8 * (size_t)ix[0], *B = s.verts + 8 * (size_t)ix[1], *C = s.verts + 8 * (size_t)ix[2];
  float e1[3], e2[3];
  for (int k = 0; k < 3; k++) { e1[k] = B[k] - A[k]; e2[k] = C[k] - A[k]; }
  return tri_like(A, e1, e2, r, false, t);
}
bool box_test(const float* lo, const float* hi, const float* inv, const float* off, float tmin, float tmax) {
  const float tnx = fmaf(lo[0], inv[0], off[0]), tpx = fmaf(hi[0], inv[0], off[0]);
  const float tny = fmaf(lo[1], inv[1], off[1]), tpy = fmaf(hi[1], inv[1], off[1]);
  const float tnz = fmaf(lo[2], inv[2], off[2]), tpz = fmaf(hi[2], inv[2], off[2]);
  const float t0 = fmaxf(fmaxf(fminf(tnx, tpx), fminf(tny, tpy)), fminf(tnz, tpz));
  const float t1 = fminf(fminf(fmaxf(tnx, tpx), fmaxf(tny, tpy)), fmaxf(tnz, tpz));
  return t0 < t1 + kEps && t0 < tmax && t1 > tmin;
}
void visit(const Scene& s, uint32_t shape, Ray& r, bool anyhit, bool& done, std::vector<Event>& ev) {
  uint32_t bits;
  std::memcpy(&bits, &r.tmax, 4);
  ev.push_back({shape, bits});
  float t;
  if (shape_test(s, shape, r, t)) {
    if (anyhit) done = true;
    else r.tmax = t - kEps;
  }
}
void walk_reference(const Scene& s, Ray r, bool anyhit, std::vector<Event>& ev) {
  float inv[3], off[3];
  for (int k = 0; k < 3; k++) { inv[k] = 1.0f / r.d[k]; off[k] = -(r.o[k] * inv[k]); }
  bool done = false;
  for (size_t cur = 0; cur < s.N && !done;) {
    const hj_bvh_node& nd = s.bvh[cur];
    if (nd.shape_index != HJ_BVH_INNER) {
      visit(s, nd.shape_index, r, anyhit, done, ev);
      cur = nd.exit_index;
    } else {
      cur = box_test(nd.aabb_min, nd.aabb_max, inv, off, r.tmin, r.tmax) ? cur + 1 : nd.exit_index;
    }
  }
}
// steps: wide nodes fetched
void walk_wide(const Scene& s, const hj_wide::Tree& t, uint32_t K, const std::vector<uint32_t>& pair_shapes, Ray r, bool anyhit,
               std::vector<Event>& ev, uint64_t& steps) {
  float inv[3], off[3];
  for (int k = 0; k < 3; k++) { inv[k] = 1.0f / r.d[k]; off[k] = -(r.o[k] * inv[k]); }
  bool done = false;
  uint32_t cur = 0;
  while ((cur & hj_wide::kIndex) < t.num_nodes && !done) {
    steps++;
    const float* nd = &t.rec[(size_t)(cur & hj_wide::kIndex) * K * 8];
    const uint32_t s0 = cur >> 30;
    uint32_t link = 0, next = 0;
    bool anyp = false;
    std::memcpy(&next, nd + (K - 1) * 8 + 7, 4);
    for (uint32_t sl = K; sl-- > 0;) {                       // as the kernel: every slot is evaluated, the first passing one wins
      uint32_t l, nx;
      std::memcpy(&l, nd + sl * 8 + 3, 4);
      std::memcpy(&nx, nd + sl * 8 + 7, 4);
      bool pass = box_test(nd + sl * 8, nd + sl * 8 + 4, inv, off, r.tmin, r.tmax) || (int32_t)l >= (int32_t)hj_wide::kUnguarded;
      pass = pass && sl >= s0;
      if (pass) { link = l; next = nx; anyp = true; }
    }
    if (!anyp) { cur = next; continue; }
    if ((link & (hj_wide::kInner | hj_wide::kPair)) == hj_wide::kInner) { cur = link & hj_wide::kIndex; continue; }
    cur = next;
    if ((int32_t)link < 0) {                                 // pair: left triangle, then the right one
      const uint32_t p = link & hj_wide::kIndex;
      visit(s, pair_shapes[2 * p], r, anyhit, done, ev);
      if (!done) visit(s, pair_shapes[2 * p + 1], r, anyhit, done, ev);
    } else {
      visit(s, link & hj_wide::kIndex, r, anyhit, done, ev);
    }
  }
}
}  // namespace

// Returns the number of rays whose event sequences differ (0 = the wide walk is the reference's), -1 when the tree is
// not a well-formed binary tree (wide_tree.h refuses it); out[0] = wide nodes, out[1] = wide steps, out[2] = events.
extern "C" long wide_check(const hj_bvh_node* bvh, size_t N, const float* spheres, size_t ns, const float* quads, size_t nq,
                           const uint32_t* tris, size_t nt, const float* verts, const float* rays, size_t nrays, int anyhit,
                           uint32_t K, uint64_t* out) {
  Scene s{bvh, N, spheres, ns, quads, nq, tris, nt, verts};
  // pair nodes as hj_scene_upload finds them
  std::vector<uint32_t> pair_of(N, 0xFFFFFFFFu), pair_shapes;
  const size_t first_tri = ns + nq;
  for (size_t i = 0; i + 2 < N; i++) {
    if (bvh[i].shape_index != HJ_BVH_INNER) continue;
    const size_t l = i + 1, r = bvh[l].exit_index;
    if (r >= N || r != l + 1) continue;
    const uint32_t sl = bvh[l].shape_index, sr = bvh[r].shape_index;
    if (sl == HJ_BVH_INNER || sr == HJ_BVH_INNER || sl < first_tri || sr < first_tri) continue;
    if (bvh[r].exit_index != bvh[i].exit_index) continue;
    pair_of[i] = (uint32_t)(pair_shapes.size() / 2);
    pair_shapes.push_back(sl); pair_shapes.push_back(sr);
  }
  hj_wide::Tree t;
  if (!hj_wide::build(bvh, N, pair_of, K, t)) return -1;
  long bad = 0;
  uint64_t steps = 0, events = 0;
  std::vector<Event> a, b;
  for (size_t i = 0; i < nrays; i++) {
    Ray r;
    std::memcpy(&r, rays + 8 * i, sizeof r);
    a.clear(); b.clear();
    walk_reference(s, r, anyhit != 0, a);
    walk_wide(s, t, K, pair_shapes, r, anyhit != 0, b, steps);
    events += a.size();
    if (a.size() != b.size() || (a.size() && std::memcmp(a.data(), b.data(), a.size() * sizeof(Event)) != 0)) bad++;
  }
  if (out) { out[0] = t.num_nodes; out[1] = steps; out[2] = events; }
  return bad;
}
