// What a hit needs: populate* (reference shader/shapes/*.glsl), emitter sampling (shader/scene.glsl:44-89), the checkerboard
// texture (shader/materials/diffusecb.glsl) - used by the shade stage (hj_stages.h).
#pragma once
#include "hj_intersect.h"
#include "hj_light_grid_const.h"

#pragma clang fp contract(off)

namespace hj {

// ------------------------------------------------------------ populate (its)

struct Its { v3 p, n, ft, fb; float u, v; };   // frame = [ft fb n]

// reference shader/shapes/triangle.glsl:54-78
HJ_DEV void populate_triangle(const DeviceScene& sc, uint32_t ix, float hu, float hv, Its& its) {
  const float4* __restrict__ rec = sc.tri_shade + 4 * (size_t)ix;
  const float4 A = rec[0], B = rec[1], C = rec[2], Vv = rec[3];
  const float l0 = (1.0f - hu) - hv, l1 = hu, l2 = hv;
  const v3 ns = (xyz(A) * l0 + xyz(B) * l1) + xyz(C) * l2;
  its.n = normalize3(ns);
  its.u = (A.w * l0 + B.w * l1) + C.w * l2;
  its.v = (Vv.x * l0 + Vv.y * l1) + Vv.z * l2;
  v3 bt = (__builtin_fabsf(its.n.x) > __builtin_fabsf(its.n.y)) ? V(0.f, 1.f, 0.f) : V(1.f, 0.f, 0.f);
  const v3 t = normalize3(cross3(its.n, bt));
  bt = cross3(its.n, t);
  its.ft = t; its.fb = bt;
}
// reference shader/shapes/sphere.glsl:43-52
HJ_DEV void populate_sphere(float4 sp, Its& its) {
  const v3 n = divs(its.p - xyz(sp), sp.w);
  its.n = n;
  const v3 t = normalize3(V(-n.z, 0.0f, n.x));
  its.ft = t; its.fb = cross3(n, t);
  float ux = 0.5f + hj_atan2(n.z, n.x) * (1.0f / kTwoPi);
  const float uy = 0.5f + hj_asin(f_min(f_max(n.y, -1.0f), 1.0f)) * kInvPi;
  if (ux != ux) ux = 0.0f;
  its.u = ux; its.v = uy;
}
// reference shader/shapes/quad.glsl:27-32 (uv stays the raw hit's)
HJ_DEV void populate_quad(const DeviceScene& sc, uint32_t ix, float hu, float hv, Its& its) {
  const v3 t = normalize3(xyz(sc.quads[3 * ix + 1]));
  const v3 b = normalize3(xyz(sc.quads[3 * ix + 2]));
  its.n = cross3(t, b); its.ft = t; its.fb = b; its.u = hu; its.v = hv;
}

// ------------------------------------------------------------ emitter sampling

struct SRec { v3 p, n; float pdf; };

HJ_DEV v3 ld3(const float* p) { return V(p[0], p[1], p[2]); }

// reference shader/scene.glsl:44-89 + shapes/*: sample*.  Always 3 draws.
// `e` = the emitter that was sampled.
HJ_DEV v3 sample_emitter(const DeviceScene& sc, v3 ref, uint32_t& rng, v3& sh_dir, float& sh_tmax, uint32_t& e) {
  float xi = rng_float(rng);
  e = 0;
  if (sc.num_emitters == 0) {   // reference reads emitters[0] out of bounds; defined here as "no light"
    rng_uint(rng); rng_uint(rng);
    sh_dir = V(0, 0, 0); sh_tmax = 0.0f;
    return V(0, 0, 0);
  }
  for (uint32_t i = 0; i < sc.num_emitters; i++) {
    xi -= __uint_as_float(__float_as_uint(sc.emit_rec[kEmitRecF4 * i].x));   // emitters[i].pdf
    if (xi < 0.0f) { e = i; break; }
  }
  // one pre-gathered record per emitter (hj_device.h) instead of emitter -> indices -> 3 vertices -> material word
  // -> material: the values are the ones those arrays hold, the chain of dependent fetches is gone
  const float4* __restrict__ er = sc.emit_rec + (size_t)kEmitRecF4 * e;
  const float4 r0 = er[0], r1 = er[1], r2 = er[2], r3 = er[3];
  const float em_pdf = r0.x;
  const uint32_t kind = __float_as_uint(r0.y);
  const v3 power = V(r1.w, r2.w, r3.w);
  SRec sr;
  if (kind == 0u) {                          // sphere.glsl:54-58
    sr.n = rand_uniform_sphere(rng);
    sr.p = xyz(r1) + sr.n * r0.z;
    sr.pdf = 1.0f / (((r0.z * r0.z) * 4.0f) * kPi);
  } else if (kind == 1u) {                   // quad.glsl:34-45
    const v3 o = xyz(r1), e1 = xyz(r2), e2 = xyz(r3);
    const v3 n = cross3(e1, e2);
    const float area = len3(n);
    sr.n = divs(n, area);
    const float u = rng_float(rng), v = rng_float(rng);
    sr.p = (o + e1 * u) + e2 * v;
    sr.pdf = 1.0f / area;
  } else {                                   // triangle.glsl:81-102
    const float4 r4 = er[4], r5 = er[5], r6 = er[6];
    const v3 a = xyz(r1), b = xyz(r2), c = xyz(r3);
    const v3 n = cross3(b - a, c - a);
    const float area = len3(n) * 0.5f;
    const v3 l = rand_barycentric(rng);
    sr.n = normalize3((xyz(r4) * l.x + xyz(r5) * l.y) + xyz(r6) * l.z);
    sr.p = (a * l.x + b * l.y) + c * l.z;
    sr.pdf = 1.0f / area;
  }
  v3 dir = sr.p - ref;
  const float dist = len3(dir);
  dir = divs(dir, dist);
  sh_dir = dir; sh_tmax = dist - kEps;
  const float cosT = -dot3(dir, sr.n);
  if (cosT < 0.0f) return V(0, 0, 0);
  const float pdf = (((em_pdf * sr.pdf) * dist) * dist) / cosT;
  return divs(power, pdf);
}

// Light-shaft visibility grid (api/light_grid.cpp).  A cell on a mesh or in a corner is proven for hit points that lie ON their shape,
// and the reference's hit point need not (hj_light_grid_const.h): is this one - p, of the ray with direction rd that hit shape id at
// (hu, hv) - where such a proof assumes it?  Depends on the hit alone, so the shade stage asks at its top, where the two records
// travel with the loads populate needs anyway (asked at the grid lookup it was one more dependent round trip per round: -6 % on c2).
HJ_DEV bool hit_point_on_its_shape(const DeviceScene& sc, uint32_t id, v3 p, v3 rd, float hu, float hv) {
  if (sc.light_grid == nullptr || (sc.lg_res & kLightGridHasRecords) == 0u) return false;          // (uniform)
  const uint32_t res = sc.lg_res & kLightGridResMask;
  const uint32_t cells_bytes = (res * res * res * (uint32_t)sizeof(uint16_t) + 15u) & ~15u;
  const float4* rec = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(sc.light_grid) + cells_bytes) + 2u * (id >= sc.ns ? id - sc.ns : 0u);
  const float4 r0 = rec[0], r1 = rec[1];                                         // n, delta; a, kind
  const v3 n = xyz(r0), pa = p - xyz(r1);
  // |n.(p - a)| and what its float evaluation can lose: 5 ulp of |p - a| (the difference, three products, two sums, n's own rounding)
  const float f = __builtin_fabsf(dot3(n, pa)) + kLightGridUlp5 * ((__builtin_fabsf(pa.x) + __builtin_fabsf(pa.y)) + __builtin_fabsf(pa.z));
  const float dn = dot3(rd, n), dd = dot3(rd, rd);
  const float third = (r1.w != 0.0f) ? f_min(1.0f - hu, 1.0f - hv) : (1.0f - hu) - hv;
  const float inside = f_min(f_min(hu, hv), third);
  // (every comparison is false for a NaN and for the all-zero record of a shape nothing is claimed about)
  return id >= sc.ns && inside >= r0.w && dn * dn >= (kLightGridSinIn * kLightGridSinIn) * dd && dd > 0.0f &&
         (f * f) * dd <= (kLightGridSlide * kLightGridSlide) * (dn * dn);
}

// True when the grid PROVES that the shadow ray from hit point p to the sampled point of emitter e is unoccluded (whatever tree is
// walked): no ray needs to be traced for this sample.  Low byte of the cell: for every hit point in it; high byte: for one on its shape.
HJ_DEV bool shadow_ray_proven_free(const DeviceScene& sc, v3 p, uint32_t e, bool on_its_shape) {
  if (sc.light_grid == nullptr || e >= 8u) return false;
  const float fx = (p.x - sc.lg_lo[0]) * sc.lg_inv[0], fy = (p.y - sc.lg_lo[1]) * sc.lg_inv[1], fz = (p.z - sc.lg_lo[2]) * sc.lg_inv[2];
  const uint32_t res = sc.lg_res & kLightGridResMask;
  const float r = (float)res;
  if (!(fx >= 0.0f && fy >= 0.0f && fz >= 0.0f && fx < r && fy < r && fz < r)) return false;   // (NaN: false)
  const uint32_t cell = ((uint32_t)fz * res + (uint32_t)fy) * res + (uint32_t)fx;
  const uint32_t bits = sc.light_grid[cell];
  return ((bits >> e) & 1u) != 0u || (on_its_shape && ((bits >> (8u + e)) & 1u) != 0u);
}

// reference shader/materials/diffusecb.glsl:6-13
HJ_DEV v3 checkerboard(const DeviceScene& sc, uint32_t idx, float u, float v) {
  const float4 ca = sc.diffusecb[2 * idx], cb = sc.diffusecb[2 * idx + 1];
  float fu = (0.5f * u) / ca.w, fv = (0.5f * v) / cb.w;
  fu = fu - __builtin_floorf(fu); fv = fv - __builtin_floorf(fv);
  const bool a = fu < 0.5f, b = fv < 0.5f;
  return (a != b) ? xyz(cb) : xyz(ca);
}

}  // namespace hj
