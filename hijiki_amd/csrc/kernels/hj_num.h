// Device numeric primitives — numeric contract HJ-NUM-1.
//
// IEEE binary32, round-to-nearest-even, NO contraction except the explicit
// fmaf() calls below (the translation unit is compiled with -ffp-contract=off
// and the pragma underneath), correctly rounded division and sqrt
// (-fhip-fp32-correctly-rounded-divide-sqrt), IEEE minNum/maxNum, and own
// polynomial exp / sincos(2*pi*v) / atan2 / asin instead of the ocml versions,
// so that every branch decision of a path is reproducible on any IEEE machine.
// The reference's GLSL leaves all of this to the driver's shader compiler
// (shaderc row), so any choice here is within its tolerance.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace hj {

#define HJ_DEV __device__ __forceinline__

constexpr float kEps = 1e-4f;                    // M_EPS, reference shader/math.glsl:2
constexpr float kPi = 3.14159265358979323846f;   // M_PI,  reference shader/math.glsl:1
constexpr float kInvPi = 1.0f / kPi;
constexpr float kTwoPi = 6.28318530717958647692f;
constexpr float kInf = __builtin_huge_valf();

struct v3 { float x, y, z; };

HJ_DEV v3 V(float x, float y, float z) { return v3{x, y, z}; }
HJ_DEV v3 operator+(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
HJ_DEV v3 operator-(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
HJ_DEV v3 operator*(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
HJ_DEV v3 operator*(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
HJ_DEV v3 operator-(v3 a) { return V(-a.x, -a.y, -a.z); }
HJ_DEV v3 xyz(float4 a) { return V(a.x, a.y, a.z); }

HJ_DEV float f_min(float a, float b) { return __builtin_fminf(a, b); }   // minNum
HJ_DEV float f_max(float a, float b) { return __builtin_fmaxf(a, b); }   // maxNum

HJ_DEV float dot3(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
HJ_DEV v3 cross3(v3 a, v3 b) {
  return V(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
HJ_DEV float len3(v3 a) { return __builtin_sqrtf(dot3(a, a)); }
HJ_DEV v3 divs(v3 a, float s) { float r = 1.0f / s; return a * r; }      // vector / scalar
HJ_DEV v3 normalize3(v3 a) { return divs(a, len3(a)); }
HJ_DEV v3 reflect3(v3 I, v3 N) { float s = 2.0f * dot3(N, I); return I - N * s; }

HJ_DEV float hj_exp(float x) {
  if (x != x) return x;
  if (!(x > -87.0f)) return 0.0f;
  if (x > 88.0f) return kInf;
  float n = fmaf(x, 1.44269504088896341f, 12582912.0f) - 12582912.0f;
  float r = fmaf(n, -0.693359375f, x);
  r = fmaf(n, 2.12194440e-4f, r);
  float z = r * r;
  float p = 1.9875691500e-4f;
  p = fmaf(p, r, 1.3981999507e-3f);
  p = fmaf(p, r, 8.3334519073e-3f);
  p = fmaf(p, r, 4.1665795894e-2f);
  p = fmaf(p, r, 1.6666665459e-1f);
  p = fmaf(p, r, 5.0000001201e-1f);
  p = fmaf(p, z, r);
  p = p + 1.0f;
  int ni = (int)n;
  return p * __uint_as_float((uint32_t)(ni + 127) << 23);
}

HJ_DEV void hj_sincos2pi(float v, float& S, float& C) {
  float k = fmaf(v, 4.0f, 12582912.0f) - 12582912.0f;
  float r = fmaf(k, -0.25f, v);
  float x = r * kTwoPi;
  float z = x * x;
  float sp = -1.9515295891e-4f;
  sp = fmaf(sp, z, 8.3321608736e-3f);
  sp = fmaf(sp, z, -1.6666654611e-1f);
  float s = fmaf(sp * z, x, x);
  float cp = 2.443315711809948e-5f;
  cp = fmaf(cp, z, -1.388731625493765e-3f);
  cp = fmaf(cp, z, 4.166664568298827e-2f);
  float c = fmaf(cp, z * z, fmaf(-0.5f, z, 1.0f));
  int q = ((int)k) & 3;
  S = s; C = c;
  if (q == 1) { S = c; C = -s; }
  else if (q == 2) { S = -s; C = -c; }
  else if (q == 3) { S = -c; C = s; }
}

HJ_DEV float hj_atan_pos(float x) {
  float y = 0.0f;
  if (x > 2.414213562373095f) { y = 1.5707963267948966f; x = -(1.0f / x); }
  else if (x > 0.4142135623730950f) { y = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
  float z = x * x;
  float p = 8.05374449538e-2f;
  p = fmaf(p, z, -1.38776856032e-1f);
  p = fmaf(p, z, 1.99777106478e-1f);
  p = fmaf(p, z, -3.33329491539e-1f);
  return y + fmaf(p * z, x, x);
}
HJ_DEV float hj_atan2(float y, float x) {
  if (x != x || y != y) return __builtin_nanf("");
  if (x == 0.0f && y == 0.0f) return __builtin_nanf("");
  float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
  float a;
  if (ax == 0.0f) a = 1.5707963267948966f;
  else a = hj_atan_pos(ay / ax);
  if (x < 0.0f) a = kPi - a;
  return (y < 0.0f) ? -a : a;
}
HJ_DEV float hj_asin(float x) {
  float a = __builtin_fabsf(x);
  float z, w;
  bool big = a > 0.5f;
  if (big) { z = 0.5f * (1.0f - a); w = __builtin_sqrtf(z); }
  else { w = a; z = a * a; }
  float p = 4.2163199048e-2f;
  p = fmaf(p, z, 2.4181311049e-2f);
  p = fmaf(p, z, 4.5470025998e-2f);
  p = fmaf(p, z, 7.4953002686e-2f);
  p = fmaf(p, z, 1.6666752422e-1f);
  float r = fmaf(p * z, w, w);
  if (big) r = 1.5707963267948966f - (r + r);
  return (x < 0.0f) ? -r : r;
}

// ------------------------------------------------------------------- RNG
// reference shader/rand.glsl:2-20
HJ_DEV uint32_t rng_seed(uint32_t seed) {
  seed = (seed ^ 61u) ^ (seed >> 16);
  seed *= 9u;
  seed = seed ^ (seed >> 4);
  seed *= 0x27d4eb2du;
  seed = seed ^ (seed >> 15);
  return seed;
}
HJ_DEV uint32_t rng_uint(uint32_t& s) {
  s ^= s << 13; s ^= s >> 17; s ^= s << 5;
  return s;
}
HJ_DEV float rng_float(uint32_t& s) { return (float)rng_uint(s) * (1.0f / 4294967296.0f); }

// reference shader/rand.glsl:22-30
HJ_DEV v3 rand_cos_hemisphere(uint32_t& s) {
  float u = rng_float(s), v = rng_float(s);
  float r = __builtin_sqrtf(u);
  float sn, cs; hj_sincos2pi(v, sn, cs);
  return V(r * cs, r * sn, __builtin_sqrtf(f_max(0.0f, 1.0f - u)));
}
// reference shader/rand.glsl:32-40
HJ_DEV v3 rand_uniform_sphere(uint32_t& s) {
  float u = rng_float(s), v = rng_float(s);
  float z = 2.0f * u - 1.0f;
  float sn, cs; hj_sincos2pi(v, sn, cs);
  float r = __builtin_sqrtf(1.0f - z * z);
  return V(r * cs, r * sn, z);
}
// reference shader/rand.glsl:42-50 (with its overwrite bug)
HJ_DEV v3 rand_barycentric(uint32_t& s) {
  float u = rng_float(s), v = rng_float(s);
  if (u + v > 1.0f) { u = 1.0f - v; v = 1.0f - u; }
  return V(u, v, (1.0f - u) - v);
}

}  // namespace hj
