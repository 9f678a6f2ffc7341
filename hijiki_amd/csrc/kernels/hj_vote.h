// Child order of a flattened tree voted by a sample of the renderer's own rays - on the device.
//
// The walk of the reference (shader/scene.glsl:97-133) visits the children of a node in array order, so WHICH child of a node is
// the left one decides how much of the other child a ray still has to enter once it has its hit.  host/tree_opt.cpp
// (order_children_by_rays) samples camera paths on the host, lets every ray vote at the ancestors of the leaf it hits and
// exchanges children where the sample says so; this file does the same on the flattened array where it lies on the device
// (a 1 M-triangle tree: milliseconds instead of a third of a second), for trees the host never held as a linked structure
// (hj_build_bvh_device) and for any other (hj_tune_bvh_device).  It is a HEURISTIC's sample: plain float arithmetic, its own
// random numbers, nothing of the numeric contract - the image does not depend on the tree except through epsilon ties
// (DESIGN.md section 5), and the oracle walks whatever tree comes out.
//
//   k_vote_paths     one thread per camera path (render.glsl:81-147 in outline: closest hit, next-event shadow ray at diffuse
//                    surfaces, cosine / mirror / dielectric bounce, at most 12 bounces).  Every ray that hits a leaf walks
//                    down from the root to that leaf; at every ancestor it counts the nodes of the OTHER child it enters
//                    (ci) and those it enters in front of its hit (ct): with the hit's child first a closest-hit ray is
//                    spared ci - ct of them (tMax = t_hit culls what lies behind), an any-hit shadow ray all ci.
//   k_ro_level       top-down, one launch per level of the tree: new position and exit of the two children of every node of
//                    the level (exchanged where gain_right > gain_left; a subtree keeps its size, so positions need no scan)
//   k_ro_scatter     the records at their new positions
//
// Integer votes added with atomics: the result does not depend on the order the threads run in.
#pragma once
#include "hj_device.h"

#pragma clang fp contract(off)

namespace hj {
namespace vote {

constexpr uint32_t kNone = 0xFFFFFFFFu;

struct Scene {
  const float4* spheres;            // hj_sphere
  const float4* quads;              // hj_quad as 3 x float4
  const hj_triangle* triangles;
  const hj_vertex* vertices;
  uint32_t ns, nq, nt;
  const uint32_t* materials;        // one tag word per shape, or null (everything diffuse)
  const hj_emitter* emitters;
  uint32_t ne;
  const hj_dielectric* dielectric;
  uint32_t ndielectric;
  hj_camera cam;
  const float4* nodes;              // hj_bvh_node as 2 x float4: (min, shape index) (max, exit index)
  uint32_t N;
  unsigned long long* gain_l;       // [N] node visits the sample saves with the left / the right child first
  unsigned long long* gain_r;
  uint32_t w_shadow;                // weight of a shadow ray's vote in quarters of a closest-hit ray's (host/tree_opt.cpp Voter::w_shadow)
};

struct Rng {                        // SplitMix64, as the host's sample
  unsigned long long s;
  HJ_DEV unsigned long long next() {
    unsigned long long z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  HJ_DEV float uni() { return (float)(next() >> 40) * (1.0f / 16777216.0f); }
};

struct Ray { v3 o, d; float tmin, tmax; };
struct Prep { v3 inv, off; };
struct Hit { int shape; uint32_t pos; float t, u, v; };

HJ_DEV v3 vnorm(v3 a) { const float l = len3(a); return l > 0.f ? a * (1.0f / l) : a; }

HJ_DEV Prep prep(const Ray& r) {
  Prep p;
  p.inv = V(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
  p.off = V(-(r.o.x * p.inv.x), -(r.o.y * p.inv.y), -(r.o.z * p.inv.z));
  return p;
}
// the reference's slab test (scene.glsl:120-131): entry distance, or +inf when the box is not entered
HJ_DEV float entry(float4 lo, float4 hi, const Prep& p, float tmin, float tmax) {
  const float tnx = lo.x * p.inv.x + p.off.x, tpx = hi.x * p.inv.x + p.off.x;
  const float tny = lo.y * p.inv.y + p.off.y, tpy = hi.y * p.inv.y + p.off.y;
  const float tnz = lo.z * p.inv.z + p.off.z, tpz = hi.z * p.inv.z + p.off.z;
  const float t0 = f_max(f_max(f_min(tnx, tpx), f_min(tny, tpy)), f_min(tnz, tpz));
  const float t1 = f_min(f_min(f_max(tnx, tpx), f_max(tny, tpy)), f_max(tnz, tpz));
  return (t0 < t1 + kEps && t0 < tmax && t1 > tmin) ? t0 : kInf;
}

// shape tests (shapes/*.glsl) in plain float
HJ_DEV bool hit_shape(const Scene& s, uint32_t obj, const Ray& r, float tmax, float& t, float& u, float& v) {
  if (obj < s.ns) {
    const float4 sp = s.spheres[obj];
    const v3 oc = r.o - xyz(sp);
    const float b = dot3(oc, r.d), c = dot3(oc, oc) - sp.w * sp.w, disc = b * b - c;
    if (disc < 0.f) return false;
    const float sq = __builtin_sqrtf(disc);
    float tt = -b - sq;
    if (tt < r.tmin || tt > tmax) tt = -b + sq;
    if (tt < r.tmin || tt > tmax) return false;
    t = tt; u = 0.f; v = 0.f;
    return true;
  }
  v3 a, ab, ac;
  const bool quad = obj < s.ns + s.nq;
  if (quad) {
    const uint32_t q = obj - s.ns;
    a = xyz(s.quads[3 * q]); ab = xyz(s.quads[3 * q + 1]); ac = xyz(s.quads[3 * q + 2]);
  } else {
    const hj_triangle tr = s.triangles[obj - s.ns - s.nq];
    const hj_vertex A = s.vertices[tr.v[0]], B = s.vertices[tr.v[1]], C = s.vertices[tr.v[2]];
    a = V(A.pos[0], A.pos[1], A.pos[2]);
    ab = V(B.pos[0], B.pos[1], B.pos[2]) - a;
    ac = V(C.pos[0], C.pos[1], C.pos[2]) - a;
  }
  const v3 n = cross3(ab, ac), ro = r.o - a, q = cross3(ro, r.d);
  const float inv = 1.0f / dot3(r.d, n);
  const float uu = inv * -dot3(q, ac), vv = inv * dot3(q, ab), tt = inv * -dot3(n, ro);
  if (!(uu >= 0.f && vv >= 0.f && (quad ? (uu <= 1.f && vv <= 1.f) : uu + vv <= 1.f) && tt >= r.tmin && tt <= tmax)) return false;
  t = tt; u = uu; v = vv;
  return true;
}

// closest hit by the reference's walk (array order, tMax shrinking); `pos` = the array index of the leaf that holds the hit
HJ_DEV Hit closest(const Scene& s, const Ray& r) {
  const Prep p = prep(r);
  float tmax = r.tmax;
  Hit h{-1, 0u, 0.f, 0.f, 0.f};
  uint32_t i = 0;
  while (i < s.N) {
    const float4 lo = s.nodes[2 * (size_t)i], hi = s.nodes[2 * (size_t)i + 1];
    const uint32_t shape = __float_as_uint(lo.w), ex = __float_as_uint(hi.w);
    if (entry(lo, hi, p, r.tmin, tmax) < kInf) {
      if (shape != HJ_BVH_INNER) {
        float t, u, v;
        if (hit_shape(s, shape, r, tmax, t, u, v)) { tmax = t; h.shape = (int)shape; h.pos = i; h.t = t; h.u = u; h.v = v; }
        i = ex;
      } else {
        i = i + 1;
      }
    } else {
      i = ex;
    }
  }
  return h;
}

// One ray's votes: down the ancestors of the leaf it hit
HJ_DEV void cast(const Scene& s, const Ray& r, const Hit& h, bool any) {
  if (h.shape < 0) return;
  const Prep p = prep(r);
  uint32_t i = 0, end_i = s.N;
  for (int guard = 0; guard < 4096 && i != h.pos; guard++) {
    const uint32_t l = i + 1;
    if (l >= s.N) return;
    const uint32_t rr = __float_as_uint(s.nodes[2 * (size_t)l + 1].w);          // the left child's exit = the right child
    if (rr >= end_i || rr <= l) return;                                          // (not a tree: no vote)
    const bool in_left = h.pos < rr;
    const uint32_t ob = in_left ? rr : l, oe = in_left ? end_i : rr;
    uint32_t ci = 0, ct = 0;
    for (uint32_t j = ob; j < oe;) {
      const float4 lo = s.nodes[2 * (size_t)j], hi = s.nodes[2 * (size_t)j + 1];
      const float e = entry(lo, hi, p, r.tmin, r.tmax);
      const bool inner = __float_as_uint(lo.w) == HJ_BVH_INNER;
      if (e < kInf) { ci++; ct += e < h.t ? 1u : 0u; }
      const uint32_t nx = (e < kInf && inner) ? j + 1 : __float_as_uint(hi.w);
      if (nx <= j) return;
      j = nx;
    }
    const uint32_t g = any ? ci * s.w_shadow : (ci - ct) * 4u;
    if (g) atomicAdd(in_left ? &s.gain_l[i] : &s.gain_r[i], (unsigned long long)g);
    if (in_left) { i = l; end_i = rr; } else { i = rr; }
  }
}

HJ_DEV uint32_t material_tag(const Scene& s, uint32_t obj) { return s.materials ? s.materials[obj] >> HJ_MATERIAL_TAG_SHIFT : (uint32_t)HJ_MAT_DIFFUSE; }

// surface point and shading normal of a hit (populate*, scene.glsl:160-175, without the tangent frames)
HJ_DEV void surface(const Scene& s, uint32_t obj, const Ray& r, const Hit& h, v3& pos, v3& n) {
  pos = r.o + r.d * h.t;
  if (obj < s.ns) {
    const float4 sp = s.spheres[obj];
    n = (pos - xyz(sp)) * (1.0f / sp.w);
  } else if (obj < s.ns + s.nq) {
    const uint32_t q = obj - s.ns;
    n = vnorm(cross3(vnorm(xyz(s.quads[3 * q + 1])), vnorm(xyz(s.quads[3 * q + 2]))));
  } else {
    const hj_triangle tr = s.triangles[obj - s.ns - s.nq];
    const hj_vertex A = s.vertices[tr.v[0]], B = s.vertices[tr.v[1]], C = s.vertices[tr.v[2]];
    n = vnorm(V(A.normal[0], A.normal[1], A.normal[2]) * (1.f - h.u - h.v) + V(B.normal[0], B.normal[1], B.normal[2]) * h.u +
              V(C.normal[0], C.normal[1], C.normal[2]) * h.v);
  }
}
HJ_DEV v3 point_on(const Scene& s, uint32_t obj, Rng& g) {
  const float a = g.uni(), b = g.uni();
  if (obj < s.ns) {
    const float4 sp = s.spheres[obj];
    const float z = 2.f * a - 1.f, ph = kTwoPi * b, rr = __builtin_sqrtf(f_max(0.f, 1.f - z * z));
    return xyz(sp) + V(rr * __cosf(ph), rr * __sinf(ph), z) * sp.w;
  }
  if (obj < s.ns + s.nq) {
    const uint32_t q = obj - s.ns;
    return xyz(s.quads[3 * q]) + xyz(s.quads[3 * q + 1]) * a + xyz(s.quads[3 * q + 2]) * b;
  }
  float uu = a, vv = b;
  if (uu + vv > 1.f) { uu = 1.f - uu; vv = 1.f - vv; }
  const hj_triangle tr = s.triangles[obj - s.ns - s.nq];
  const hj_vertex A = s.vertices[tr.v[0]], B = s.vertices[tr.v[1]], C = s.vertices[tr.v[2]];
  const v3 p0 = V(A.pos[0], A.pos[1], A.pos[2]);
  return p0 + (V(B.pos[0], B.pos[1], B.pos[2]) - p0) * uu + (V(C.pos[0], C.pos[1], C.pos[2]) - p0) * vv;
}

__global__ __launch_bounds__(64) void k_vote_paths(Scene s, uint32_t num_paths) {
  const uint32_t index = blockIdx.x * blockDim.x + threadIdx.x;
  if (index >= num_paths) return;
  Rng g{0x48494A494B49ull ^ ((unsigned long long)index * 0xD1342543DE82EF95ull)};
  const uint32_t nshapes = s.ns + s.nq + s.nt;
  const float th = __tanf(0.5f * s.cam.fov * 0.017453292f);
  const float x = (2.f * g.uni() - 1.f) * th, y = (2.f * g.uni() - 1.f) * th;
  const v3 qv = V(s.cam.rotation[0], s.cam.rotation[1], s.cam.rotation[2]);
  const float qw = s.cam.rotation[3];
  const v3 vv = V(x, -y, -1.f);
  const v3 tq = cross3(qv, vv) * 2.0f;                                     // v + w t + q x t, t = 2 q x v
  Ray r;
  r.o = V(s.cam.position[0], s.cam.position[1], s.cam.position[2]);
  r.d = vnorm(vv + tq * qw + cross3(qv, tq));
  r.tmin = kEps; r.tmax = kInf;
  for (int bounce = 0; bounce < 12; bounce++) {
    const Hit h = closest(s, r);
    cast(s, r, h, false);
    if (h.shape < 0) return;
    const uint32_t obj = (uint32_t)h.shape;
    const uint32_t tag = material_tag(s, obj);
    if (tag == HJ_MAT_EMISSIVE) return;
    v3 pos, n;
    surface(s, obj, r, h, pos, n);
    v3 wo;
    if (tag == HJ_MAT_DIFFUSE || tag == HJ_MAT_DIFFUSECBOARD) {
      if (s.ne) {                                                          // scene.glsl:54-89
        uint32_t k = (uint32_t)(g.uni() * (float)s.ne);
        if (k >= s.ne) k = s.ne - 1;
        const uint32_t e = s.emitters[k].shape;
        if (e < nshapes) {
          const v3 lp = point_on(s, e, g);
          v3 d = lp - pos;
          const float dist = len3(d);
          d = d * (1.0f / dist);
          if (dist > 3.f * kEps && dot3(d, n) > 0.f) {
            Ray sr;
            sr.o = pos; sr.d = d; sr.tmin = 2.f * kEps; sr.tmax = dist - kEps;
            const Hit sh = closest(s, sr);
            cast(s, sr, sh, true);
          }
        }
      }
      const float a = g.uni(), b = g.uni(), rr = __builtin_sqrtf(a), ph = kTwoPi * b;
      const v3 bt = __builtin_fabsf(n.x) > __builtin_fabsf(n.y) ? V(0.f, 1.f, 0.f) : V(1.f, 0.f, 0.f);
      const v3 tx = vnorm(cross3(n, bt)), ty = cross3(n, tx);
      wo = tx * (rr * __cosf(ph)) + ty * (rr * __sinf(ph)) + n * __builtin_sqrtf(f_max(0.f, 1.f - a));
    } else if (tag == HJ_MAT_MIRROR) {
      wo = r.d - n * (2.f * dot3(n, r.d));
    } else {                                                               // dielectric: material.glsl:50-87
      const uint32_t mi = s.materials ? (s.materials[obj] & HJ_MATERIAL_INDEX_MASK) : 0u;
      float eta = mi < s.ndielectric ? s.dielectric[mi].eta : 1.5f, cos_i = -dot3(n, r.d);
      v3 nn = n;
      float eta_inv = 1.0f / eta;
      if (cos_i < 0.f) { eta = eta_inv; eta_inv = 1.0f / eta; nn = n * -1.0f; cos_i = -cos_i; }
      const float k = 1.f - eta_inv * eta_inv * (1.f - cos_i * cos_i);
      bool reflect = k <= 0.f;
      if (!reflect) {
        const float cos_o = __builtin_sqrtf(k);
        const float rp = (eta * cos_i - cos_o) / (eta * cos_i + cos_o), ro = (cos_i - eta * cos_o) / (cos_i + eta * cos_o);
        reflect = g.uni() < 0.5f * (rp * rp + ro * ro);
        if (!reflect) wo = (r.d - nn * dot3(r.d, nn)) * eta_inv - nn * cos_o;
      }
      if (reflect) wo = r.d - nn * (2.f * dot3(nn, r.d));
    }
    if (bounce > 3 && g.uni() > 0.75f) return;                             // (roulette at about the renderer's survival rate)
    r.o = pos; r.d = vnorm(wo); r.tmin = 2.f * kEps; r.tmax = kInf;
  }
}

// ---- the exchange: new positions top-down, level by level

struct Reorder {
  const float4* in;                 // the tree as it is
  float4* out;                      // ... re-ordered
  uint32_t N;
  const unsigned long long* gain_l;
  const unsigned long long* gain_r;
  uint32_t* depth;                  // [N]
  uint32_t* npos;                   // [N] new position
  uint32_t* nexit;                  // [N] new exit
  uint32_t* info;                   // [0] deepest level + 1 that held an inner node  [1] nodes whose children were exchanged  [2] link errors
};

__global__ void k_ro_init(Reorder r) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= r.N) return;
  r.depth[i] = i == 0 ? 0u : kNone;
  if (i == 0) { r.npos[0] = 0; r.nexit[0] = __float_as_uint(r.in[1].w); }
}
__global__ void k_ro_level(Reorder r, uint32_t L) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= r.N || r.depth[i] != L) return;
  if (__float_as_uint(r.in[2 * (size_t)i].w) != HJ_BVH_INNER) return;
  const size_t l = (size_t)i + 1;
  const uint32_t ex = __float_as_uint(r.in[2 * (size_t)i + 1].w);
  const uint32_t end_i = ex < r.N ? ex : r.N;
  if (l >= r.N) { atomicOr(&r.info[2], 1u); return; }
  const uint32_t rr = __float_as_uint(r.in[2 * l + 1].w);
  if (rr <= l || rr >= end_i) { atomicOr(&r.info[2], 1u); return; }
  const uint32_t size_l = rr - (uint32_t)l, size_r = end_i - rr;
  const bool swap = r.gain_r[i] > r.gain_l[i];                        // (ties and nodes no ray voted on keep their order)
  const uint32_t first = swap ? rr : (uint32_t)l, second = swap ? (uint32_t)l : rr, size_first = swap ? size_r : size_l;
  const uint32_t p = r.npos[i] + 1;
  r.npos[first] = p; r.npos[second] = p + size_first;
  r.nexit[first] = p + size_first; r.nexit[second] = r.nexit[i];
  r.depth[l] = L + 1; r.depth[rr] = L + 1;
  r.info[0] = L + 1;
  if (swap) atomicAdd(&r.info[1], 1u);
}
__global__ void k_ro_scatter(Reorder r) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= r.N) return;
  if (r.depth[i] == kNone) { atomicOr(&r.info[2], 2u); return; }     // a record the root does not reach
  const float4 lo = r.in[2 * (size_t)i];
  float4 hi = r.in[2 * (size_t)i + 1];
  hi.w = __uint_as_float(r.nexit[i]);
  const size_t o = r.npos[i];
  r.out[2 * o] = lo; r.out[2 * o + 1] = hi;
}

// (position, record) pairs written into an array: the host-built top of hj_build_bvh_device's tree
__global__ void k_put_records(const uint32_t* __restrict__ pos, const float4* __restrict__ rec, uint32_t n, float4* __restrict__ out) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const size_t o = pos[k];
  out[2 * o] = rec[2 * (size_t)k]; out[2 * o + 1] = rec[2 * (size_t)k + 1];
}

}  // namespace vote
}  // namespace hj
