/*
 * hj_oracle.c — CPU restatement of the Hijiki integrator + reconstruction.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP hot
 * path and the "Nori-style" CPU baseline timed by bench.py.  Nothing in the
 * product (hijiki_amd/) links, imports or calls it; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * PARITY STATUS: **parity unpinned** for the rendered image.  The reference
 * (/root/reference) has no tests, golden images or known-answer vectors for
 * this path, and it cannot be built or run here (Rust + wgpu + shaderc +
 * Vulkan are absent), so no output of the reference itself exists anywhere in
 * this project.  What IS pinned: the integer RNG sequence (SURVEY.md Appendix
 * D, derived from shader/rand.glsl), the reconstruction tap weights (Appendix
 * B-7), analytic intersection / Fresnel identities, BVH-vs-linear-scan
 * equivalence and a closed-form furnace test (tests/test_oracle_*.py).
 *
 * Each function cites the reference lines it restates (paths relative to
 * /root/reference).  The structure follows the GLSL (one path at a time, full
 * bounce loop, closest-hit everywhere) — deliberately NOT the wavefront
 * structure of the HIP kernels — so the two are independent statements of the
 * same arithmetic.
 *
 * Numeric contract "HJ-NUM-1" (shared with the kernels by specification, not
 * by code): IEEE binary32, round-to-nearest-even, no
 * contraction except the explicit fmaf() calls written here, correctly
 * rounded / and sqrt, IEEE-754 minNum/maxNum for min/max, vector/scalar =
 * vector * (1/scalar), normalize(v) = v * (1/sqrt(dot(v,v))), and the
 * polynomial exp / sincos(2*pi*v) / atan2 / asin below instead of libm.
 * Compile with -ffp-contract=off and without -ffast-math.
 */
#define _GNU_SOURCE
#include "../include/hijiki_hip.h"

#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#if defined(__GNUC__)
#pragma GCC optimize("no-fast-math")
#endif

#define HJO_EXPORT __attribute__((visibility("default")))

#define M_EPSF 1e-4f                     /* shader/math.glsl:2 */
#define M_PIF 3.14159265358979323846f    /* shader/math.glsl:1 */
#define INV_PIF (1.0f / M_PIF)
#define TWO_PIF 6.28318530717958647692f

/* ------------------------------------------------------------ primitives */

typedef struct { float x, y, z; } v3;

static inline float bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
/* IEEE negation of a value an fmaf just produced.  gcc folds -fmaf(a, b, c) into ONE vfnmsub instruction, which computes
 * -(a*b) - c: the same number, but +0 where the negated +0 sum is -0 (ab = -0, c = +0, or exact cancellation of non-zero terms).
 * HJ-NUM-1 says negation flips the sign bit, as the kernels' source modifiers do; a zero's sign decides the
 * sign of 1 / d for a direction component, i.e. what the slab test answers.  Found by tests/test_gpu_fuzz.py (round 6): rays that
 * start exactly in a quad's plane gave t = +0 here and -0 on the GPU.  The empty asm hides the producer from the fold. */
static inline float neg_of(float x) { __asm__("" : "+x"(x)); return -x; }

/* IEEE-754-2008 minNum / maxNum (a quiet NaN loses against a number). */
static inline float f_min(float a, float b) {
  float m = (a < b) ? a : b;
  if (b != b) m = a;
  return m;
}
static inline float f_max(float a, float b) {
  float m = (a > b) ? a : b;
  if (b != b) m = a;
  return m;
}

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v_add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v_sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v_mul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v_scale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 v_neg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline float dot3(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline v3 cross3(v3 a, v3 b) {
  return V(fmaf(a.y, b.z, -(a.z * b.y)),
           fmaf(a.z, b.x, -(a.x * b.z)),
           fmaf(a.x, b.y, -(a.y * b.x)));
}
static inline float len3(v3 a) { return sqrtf(dot3(a, a)); }
static inline v3 v_divs(v3 a, float s) { float r = 1.0f / s; return v_scale(a, r); }
static inline v3 normalize3(v3 a) { return v_divs(a, len3(a)); }

/* exp(x): Cody-Waite reduction + degree-6 polynomial.  Results below 2^-126
 * flush to 0 (x <= -87), overflow above 88. */
static inline float hj_exp(float x) {
  if (x != x) return x;
  if (!(x > -87.0f)) return 0.0f;
  if (x > 88.0f) return INFINITY;
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
  return p * bits2f((uint32_t)(ni + 127) << 23);
}

/* (sin, cos)(2*pi*v) for v in [0,1]: quadrant by rint(4v), then polynomials
 * on [-pi/4, pi/4]. */
static inline void hj_sincos2pi(float v, float* s_out, float* c_out) {
  float k = fmaf(v, 4.0f, 12582912.0f) - 12582912.0f;
  float r = fmaf(k, -0.25f, v);
  float x = r * TWO_PIF;
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
  float S = s, C = c;
  if (q == 1) { S = c; C = neg_of(s); }
  else if (q == 2) { S = neg_of(s); C = neg_of(c); }
  else if (q == 3) { S = neg_of(c); C = s; }
  *s_out = S; *c_out = C;
}

/* atan(x) for x >= 0. */
static inline float hj_atan_pos(float x) {
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
/* atan2(y, x); (0,0) -> NaN so that the reference's isnan guard
 * (shader/shapes/sphere.glsl:49-51) fires. */
static inline float hj_atan2(float y, float x) {
  if (x != x || y != y) return NAN;
  if (x == 0.0f && y == 0.0f) return NAN;
  float ax = fabsf(x), ay = fabsf(y);
  float a;
  if (ax == 0.0f) a = 1.5707963267948966f;
  else a = hj_atan_pos(ay / ax);
  if (x < 0.0f) a = M_PIF - a;
  return (y < 0.0f) ? -a : a;
}
/* asin(x), |x| <= 1. */
static inline float hj_asin(float x) {
  float a = fabsf(x);
  float z, w;
  int big = a > 0.5f;
  if (big) { z = 0.5f * (1.0f - a); w = sqrtf(z); }
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

/* -------------------------------------------------------------------- RNG */

/* shader/rand.glsl:9-16 (Wang hash) */
static inline uint32_t rng_seed(uint32_t seed) {
  seed = (seed ^ 61u) ^ (seed >> 16);
  seed *= 9u;
  seed = seed ^ (seed >> 4);
  seed *= 0x27d4eb2du;
  seed = seed ^ (seed >> 15);
  return seed;
}
/* shader/rand.glsl:2-7 (xorshift32) */
static inline uint32_t rng_uint(uint32_t* s) {
  uint32_t x = *s;
  x ^= x << 13; x ^= x >> 17; x ^= x << 5;
  *s = x;
  return x;
}
/* shader/rand.glsl:18-20; uint->float is round-to-nearest-even (can be 1.0) */
static inline float rng_float(uint32_t* s) { return (float)rng_uint(s) * (1.0f / 4294967296.0f); }

/* shader/rand.glsl:22-30 */
static inline v3 rand_cos_hemisphere(uint32_t* s) {
  float u = rng_float(s), v = rng_float(s);
  float r = sqrtf(u);
  float sn, cs; hj_sincos2pi(v, &sn, &cs);
  return V(r * cs, r * sn, sqrtf(f_max(0.0f, 1.0f - u)));
}
/* shader/rand.glsl:32-40 */
static inline v3 rand_uniform_sphere(uint32_t* s) {
  float u = rng_float(s), v = rng_float(s);
  float z = 2.0f * u - 1.0f;
  float sn, cs; hj_sincos2pi(v, &sn, &cs);
  float r = sqrtf(1.0f - z * z);
  return V(r * cs, r * sn, z);
}
/* shader/rand.glsl:42-50 — including the u/v overwrite bug (Appendix C-1) */
static inline v3 rand_barycentric(uint32_t* s) {
  float u = rng_float(s), v = rng_float(s);
  if (u + v > 1.0f) { u = 1.0f - v; v = 1.0f - u; }
  return V(u, v, (1.0f - u) - v);
}

/* ------------------------------------------------------------------ types */

typedef struct { v3 o, d; float tmin, tmax; } ray_t;
/* shader/render.glsl:39-46 */
typedef struct { int id; float t; v3 p, n; float u, v; v3 ft, fb, fn; float raw_u, raw_v; } its_t;
/* shader/render.glsl:48-52 */
typedef struct { v3 p, n; float pdf; } srec_t;

/* Work counters of the REFERENCE algorithm (closest-hit walks for camera/bounce rays and for shadow
 * rays alike); they define the algorithmic bytes of SURVEY.md §8(d).  nodes/tri/sphere/quad count the
 * closest (camera + bounce) calls, the shadow_* ones the intersectScene(shadowRay) calls. */
typedef struct hjo_counters {
  uint64_t paths, closest_calls, shadow_calls, nodes, tri_tests, sphere_tests, quad_tests,
      hits, nee_evals, shadow_nodes, shadow_tri_tests, shadow_sphere_tests, shadow_quad_tests, shadow_hits;
} hjo_counters;

typedef struct {
  const hj_scene_desc* sc;
  uint32_t ns, nq, nt;
  int use_bvh;
  hjo_counters* ctr;
} scene_t;

typedef struct { uint64_t *nodes, *tri, *sphere, *quad; uint32_t* hist; int anyhit; } walk_ctr;
/* CPU-BASELINE switch (bench.py's cpu_baseline leg only; parity runs never set it): shadow rays stop at their first
 * accepted hit, as a production CPU renderer (Nori) does, instead of the reference's full closest-hit walk
 * (scene.glsl:92-96 "TODO: optimize").  The shadow overload only uses the boolean, and the first accepted hit in visiting
 * order is the same with or without tMax shrinking, so the image is identical (tests/test_oracle_shading.py checks). */
static int g_shadow_anyhit = 0;
HJO_EXPORT void hjo_set_shadow_anyhit(int on) { g_shadow_anyhit = on; }
static inline walk_ctr closest_ctr(hjo_counters* c) { walk_ctr w = {&c->nodes, &c->tri_tests, &c->sphere_tests, &c->quad_tests, NULL, 0}; return w; }
static inline walk_ctr shadow_ctr(hjo_counters* c) {
  walk_ctr w = {&c->shadow_nodes, &c->shadow_tri_tests, &c->shadow_sphere_tests, &c->shadow_quad_tests, NULL, g_shadow_anyhit}; return w;
}

/* DIRECTIONAL TREES (no counterpart upstream: scene.glsl:97-133 walks ONE array): K link orderings of the scene's tree, one per
 * direction class of the rays (include/hijiki_hip.h: hj_ray_direction_class); a ray walks the array of its class from node 0 to
 * the end.  Same boxes, same leaves, same tests: results differ from the one-array walk only at epsilon ties (SURVEY C-11).
 * arrays = K consecutive arrays of sc->num_bvh_nodes records; mode 0 / NULL switches back to sc->bvh. */
static int g_dir_mode = 0;
static const hj_bvh_node* g_dir_arrays = NULL;
HJO_EXPORT void hjo_set_directional_bvh(int mode, const hj_bvh_node* arrays) {
  g_dir_mode = (arrays || mode < 0) ? mode : 0;
  g_dir_arrays = arrays;
}

static inline v3 ld3(const float* p) { return V(p[0], p[1], p[2]); }

/* ------------------------------------------------------------ intersection */

/* shader/shapes/triangle.glsl:15-52 */
static inline int intersect_triangle(const scene_t* S, const ray_t* r, uint32_t ix, its_t* its) {
  const hj_triangle* T = &S->sc->triangles[ix];
  const hj_vertex* A = &S->sc->vertices[T->v[0]];
  const hj_vertex* B = &S->sc->vertices[T->v[1]];
  const hj_vertex* C = &S->sc->vertices[T->v[2]];
  v3 a = ld3(A->pos);
  v3 ab = v_sub(ld3(B->pos), a);
  v3 ac = v_sub(ld3(C->pos), a);
  v3 n = cross3(ab, ac);
  v3 ro = v_sub(r->o, a);
  v3 q = cross3(ro, r->d);
  float d = 1.0f / dot3(r->d, n);
  float u = d * neg_of(dot3(q, ac));
  float v = d * dot3(q, ab);
  if (u < 0.0f || v < 0.0f || u + v > 1.0f) return 0;
  float t = d * neg_of(dot3(n, ro));
  if (r->tmin <= t && t <= r->tmax) {
    its->t = t; its->u = u; its->v = v;
    return 1;   /* its.n = normalize(n) is overwritten by populate */
  }
  return 0;
}

/* shader/shapes/sphere.glsl:18-41 (3-argument overload) */
static inline int intersect_sphere(const ray_t* r, const hj_sphere* sp, its_t* its) {
  v3 l = v_sub(r->o, ld3(sp->center));
  float b = 2.0f * dot3(r->d, l);
  float c = dot3(l, l) - sp->radius * sp->radius;
  float d = b * b - 4.0f * c;
  if (d < 0.0f) return 0;
  d = sqrtf(d);
  float t0 = -0.5f * (b + d);
  if (r->tmin <= t0 && t0 <= r->tmax) { its->t = t0; return 1; }
  float t1 = -0.5f * (b - d);
  if (r->tmin <= t1 && t1 <= r->tmax) { its->t = t1; return 1; }
  return 0;
}

/* shader/shapes/quad.glsl:7-25 */
static inline int intersect_quad(const ray_t* r, const hj_quad* qd, its_t* its) {
  v3 e1 = ld3(qd->edge1), e2 = ld3(qd->edge2);
  v3 n = cross3(e1, e2);
  v3 ro = v_sub(r->o, ld3(qd->origin));
  v3 q = cross3(ro, r->d);
  float d = 1.0f / dot3(r->d, n);
  float u = d * neg_of(dot3(q, e2));
  float v = d * dot3(q, e1);
  if (u < 0.0f || u > 1.0f || v < 0.0f || v > 1.0f) return 0;
  float t = d * neg_of(dot3(n, ro));
  if (r->tmin <= t && t <= r->tmax) { its->t = t; its->u = u; its->v = v; return 1; }
  return 0;
}

/* shader/shapes/triangle.glsl:54-78 */
static inline void populate_triangle(const scene_t* S, uint32_t ix, its_t* its) {
  const hj_triangle* T = &S->sc->triangles[ix];
  const hj_vertex* A = &S->sc->vertices[T->v[0]];
  const hj_vertex* B = &S->sc->vertices[T->v[1]];
  const hj_vertex* C = &S->sc->vertices[T->v[2]];
  float l0 = (1.0f - its->u) - its->v, l1 = its->u, l2 = its->v;
  v3 ns = v_add(v_add(v_scale(ld3(A->normal), l0), v_scale(ld3(B->normal), l1)),
                v_scale(ld3(C->normal), l2));
  its->n = normalize3(ns);
  float uu = (A->u * l0 + B->u * l1) + C->u * l2;
  float vv = (A->v * l0 + B->v * l1) + C->v * l2;
  its->u = uu; its->v = vv;
  v3 bt = (fabsf(its->n.x) > fabsf(its->n.y)) ? V(0.f, 1.f, 0.f) : V(1.f, 0.f, 0.f);
  v3 t = normalize3(cross3(its->n, bt));
  bt = cross3(its->n, t);
  its->ft = t; its->fb = bt; its->fn = its->n;
}

/* shader/shapes/sphere.glsl:43-52 */
static inline void populate_sphere(const hj_sphere* sp, its_t* its) {
  v3 n = v_divs(v_sub(its->p, ld3(sp->center)), sp->radius);
  its->n = n;
  v3 t = normalize3(V(-n.z, 0.0f, n.x));
  v3 b = cross3(n, t);
  its->ft = t; its->fb = b; its->fn = n;
  float ux = 0.5f + hj_atan2(n.z, n.x) * (1.0f / TWO_PIF);
  float uy = 0.5f + hj_asin(f_min(f_max(n.y, -1.0f), 1.0f)) * INV_PIF;
  if (ux != ux) ux = 0.0f;
  its->u = ux; its->v = uy;
}

/* shader/shapes/quad.glsl:27-32 */
static inline void populate_quad(const hj_quad* qd, its_t* its) {
  v3 t = normalize3(ld3(qd->edge1));
  v3 b = normalize3(ld3(qd->edge2));
  v3 n = cross3(t, b);
  its->n = n; its->ft = t; its->fb = b; its->fn = n;
}

/* shader/scene.glsl:97-175 — both the USE_BVH and the linear-scan branch */
static int intersect_scene(const scene_t* S, ray_t ray, its_t* its, walk_ctr c) {
  its->id = -1;
  const uint32_t ns = S->ns, nq = S->nq, nt = S->nt;
  if (S->use_bvh) {
    const uint32_t nn = (uint32_t)S->sc->num_bvh_nodes;
    const hj_bvh_node* bvh = S->sc->bvh;
    if (g_dir_mode > 0) {
      const float dd[3] = {ray.d.x, ray.d.y, ray.d.z};
      bvh = g_dir_arrays + (size_t)hj_ray_direction_class(g_dir_mode, dd) * nn;
    }
    v3 inv = V(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
    v3 off = V(-(ray.o.x * inv.x), -(ray.o.y * inv.y), -(ray.o.z * inv.z));
    if (g_dir_mode < 0) {
      /* PROBE ONLY (tools/dir_order_probe.py): the bound a direction-dependent order can approach - every inner node's children
       * visited nearer-box-first for THIS ray (an explicit stack; the far child is re-tested against the tMax the near one left,
       * as the skip-link walk would).  Counts node records fetched like the walk below; finds the same hits but for epsilon ties. */
      uint32_t stack[256];
      int sp = 0;
      stack[sp++] = 0;
      while (sp) {
        const uint32_t cur = stack[--sp];
        const hj_bvh_node* nd = &bvh[cur];
        (*c.nodes)++;
        uint32_t shape = nd->shape_index;
        if (shape != HJ_BVH_INNER) {
          int hit;
          if (shape < ns) { hit = intersect_sphere(&ray, &S->sc->spheres[shape], its); (*c.sphere)++; }
          else if (shape < ns + nq) { hit = intersect_quad(&ray, &S->sc->quads[shape - ns], its); (*c.quad)++; }
          else { hit = intersect_triangle(S, &ray, shape - ns - nq, its); (*c.tri)++; }
          if (hit) { ray.tmax = its->t - M_EPSF; its->id = (int)shape; if (c.anyhit) return 1; }
          continue;
        }
        float tnx = fmaf(nd->aabb_min[0], inv.x, off.x), tpx = fmaf(nd->aabb_max[0], inv.x, off.x);
        float tny = fmaf(nd->aabb_min[1], inv.y, off.y), tpy = fmaf(nd->aabb_max[1], inv.y, off.y);
        float tnz = fmaf(nd->aabb_min[2], inv.z, off.z), tpz = fmaf(nd->aabb_max[2], inv.z, off.z);
        float t0 = f_max(f_max(f_min(tnx, tpx), f_min(tny, tpy)), f_min(tnz, tpz));
        float t1 = f_min(f_min(f_max(tnx, tpx), f_max(tny, tpy)), f_max(tnz, tpz));
        if (!(t0 < t1 + M_EPSF && t0 < ray.tmax && t1 > ray.tmin)) continue;
        const uint32_t l = cur + 1, r = bvh[l].exit_index;
        float e[2];
        for (int k = 0; k < 2; k++) {
          const hj_bvh_node* ch = &bvh[k ? r : l];
          float ax = fmaf(ch->aabb_min[0], inv.x, off.x), bx = fmaf(ch->aabb_max[0], inv.x, off.x);
          float ay = fmaf(ch->aabb_min[1], inv.y, off.y), by = fmaf(ch->aabb_max[1], inv.y, off.y);
          float az = fmaf(ch->aabb_min[2], inv.z, off.z), bz = fmaf(ch->aabb_max[2], inv.z, off.z);
          e[k] = f_max(f_max(f_min(ax, bx), f_min(ay, by)), f_min(az, bz));
        }
        if (sp + 2 > 256) continue;
        if (e[0] <= e[1]) { stack[sp++] = r; stack[sp++] = l; } else { stack[sp++] = l; stack[sp++] = r; }
      }
    } else
    for (uint32_t cur = 0; cur < nn;) {
      const hj_bvh_node* nd = &bvh[cur];
      (*c.nodes)++;
      if (c.hist) c.hist[cur]++;
      uint32_t shape = nd->shape_index, ex = nd->exit_index;
      if (shape != HJ_BVH_INNER) {
        int hit;
        if (shape < ns) { hit = intersect_sphere(&ray, &S->sc->spheres[shape], its); (*c.sphere)++; }
        else if (shape < ns + nq) { hit = intersect_quad(&ray, &S->sc->quads[shape - ns], its); (*c.quad)++; }
        else { hit = intersect_triangle(S, &ray, shape - ns - nq, its); (*c.tri)++; }
        if (hit) { ray.tmax = its->t - M_EPSF; its->id = (int)shape; if (c.anyhit) return 1; }
        cur = ex;
      } else {
        float tnx = fmaf(nd->aabb_min[0], inv.x, off.x), tpx = fmaf(nd->aabb_max[0], inv.x, off.x);
        float tny = fmaf(nd->aabb_min[1], inv.y, off.y), tpy = fmaf(nd->aabb_max[1], inv.y, off.y);
        float tnz = fmaf(nd->aabb_min[2], inv.z, off.z), tpz = fmaf(nd->aabb_max[2], inv.z, off.z);
        float t0 = f_max(f_max(f_min(tnx, tpx), f_min(tny, tpy)), f_min(tnz, tpz));
        float t1 = f_min(f_min(f_max(tnx, tpx), f_max(tny, tpy)), f_max(tnz, tpz));
        if (t0 < t1 + M_EPSF && t0 < ray.tmax && t1 > ray.tmin) cur = cur + 1;
        else cur = ex;
      }
    }
  } else {
    if (ns > 100 || nq > 100) return 0;   /* scene.glsl:135-138 "failsafe" */
    for (uint32_t i = 0; i < ns; i++) {
      (*c.sphere)++;
      if (intersect_sphere(&ray, &S->sc->spheres[i], its)) { ray.tmax = its->t - M_EPSF; its->id = (int)i; }
    }
    for (uint32_t i = 0; i < nq; i++) {
      (*c.quad)++;
      if (intersect_quad(&ray, &S->sc->quads[i], its)) { ray.tmax = its->t - M_EPSF; its->id = (int)(ns + i); }
    }
    for (uint32_t i = 0; i < nt; i++) {
      (*c.tri)++;
      if (intersect_triangle(S, &ray, i, its)) { ray.tmax = its->t - M_EPSF; its->id = (int)(ns + nq + i); }
    }
  }
  if (its->id == -1) return 0;
  its->raw_u = its->u; its->raw_v = its->v;   /* probe only: the hit's own (u,v) before populate* overwrites uv */
  its->p = V(fmaf(its->t, ray.d.x, ray.o.x), fmaf(its->t, ray.d.y, ray.o.y), fmaf(its->t, ray.d.z, ray.o.z));
  uint32_t id = (uint32_t)its->id;
  if (id < ns) populate_sphere(&S->sc->spheres[id], its);
  else if (id < ns + nq) populate_quad(&S->sc->quads[id - ns], its);
  else populate_triangle(S, id - ns - nq, its);
  return 1;
}

/* ---------------------------------------------------------- emitter sampling */

/* shader/shapes/triangle.glsl:81-102 */
static inline void sample_triangle(const scene_t* S, uint32_t ix, uint32_t* rng, srec_t* sr) {
  const hj_triangle* T = &S->sc->triangles[ix];
  const hj_vertex* A = &S->sc->vertices[T->v[0]];
  const hj_vertex* B = &S->sc->vertices[T->v[1]];
  const hj_vertex* C = &S->sc->vertices[T->v[2]];
  v3 a = ld3(A->pos), b = ld3(B->pos), cc = ld3(C->pos);
  v3 n = cross3(v_sub(b, a), v_sub(cc, a));
  float area = len3(n) * 0.5f;
  v3 l = rand_barycentric(rng);
  sr->n = normalize3(v_add(v_add(v_scale(ld3(A->normal), l.x), v_scale(ld3(B->normal), l.y)),
                           v_scale(ld3(C->normal), l.z)));
  sr->p = v_add(v_add(v_scale(a, l.x), v_scale(b, l.y)), v_scale(cc, l.z));
  sr->pdf = 1.0f / area;
}
/* shader/shapes/sphere.glsl:54-58 */
static inline void sample_sphere(const hj_sphere* sp, uint32_t* rng, srec_t* sr) {
  sr->n = rand_uniform_sphere(rng);
  sr->p = v_add(ld3(sp->center), v_scale(sr->n, sp->radius));
  sr->pdf = 1.0f / (((sp->radius * sp->radius) * 4.0f) * M_PIF);
}
/* shader/shapes/quad.glsl:34-45 */
static inline void sample_quad(const hj_quad* qd, uint32_t* rng, srec_t* sr) {
  v3 e1 = ld3(qd->edge1), e2 = ld3(qd->edge2);
  v3 n = cross3(e1, e2);
  float area = len3(n);
  sr->n = v_divs(n, area);
  float u = rng_float(rng), v = rng_float(rng);
  sr->p = v_add(v_add(ld3(qd->origin), v_scale(e1, u)), v_scale(e2, v));
  sr->pdf = 1.0f / area;
}

/* shader/scene.glsl:54-89.  Always consumes 3 random numbers.  With no
 * emitters the reference reads emitters[0] out of bounds (undefined); here
 * that case draws the 3 numbers and returns zero importance. */
static _Thread_local int g_last_emitter = -1;     /* sample_emitter's choice: the last column of the diagnostic ray log below */
static inline v3 sample_emitter(const scene_t* S, v3 ref, uint32_t* rng, ray_t* sh) {
  const hj_scene_desc* sc = S->sc;
  float xi = rng_float(rng);
  if (sc->num_emitters == 0) {
    rng_uint(rng); rng_uint(rng);
    sh->o = ref; sh->d = V(0, 0, 0); sh->tmin = 2.0f * M_EPSF; sh->tmax = 0.0f;
    return V(0, 0, 0);
  }
  uint32_t e = 0;
  for (uint32_t i = 0; i < (uint32_t)sc->num_emitters; i++) {
    xi -= sc->emitters[i].pdf;
    if (xi < 0.0f) { e = i; break; }
  }
  g_last_emitter = (int)e;
  uint32_t shape = sc->emitters[e].shape;
  srec_t sr;
  if (shape < S->ns) sample_sphere(&sc->spheres[shape], rng, &sr);
  else if (shape < S->ns + S->nq) sample_quad(&sc->quads[shape - S->ns], rng, &sr);
  else sample_triangle(S, shape - S->ns - S->nq, rng, &sr);
  uint32_t mat = sc->materials[shape];
  v3 power = ld3(sc->emissive[mat & HJ_MATERIAL_INDEX_MASK].power);
  v3 dir = v_sub(sr.p, ref);
  float dist = len3(dir);
  dir = v_divs(dir, dist);
  sh->o = ref; sh->d = dir; sh->tmin = 2.0f * M_EPSF; sh->tmax = dist - M_EPSF;
  float cosT = neg_of(dot3(dir, sr.n));
  if (cosT < 0.0f) return V(0, 0, 0);
  float pdf = (((sc->emitters[e].pdf * sr.pdf) * dist) * dist) / cosT;
  return v_divs(power, pdf);
}

/* ------------------------------------------------------------------- BSDFs */

/* shader/materials/diffusecb.glsl:6-13 */
static inline v3 checkerboard(const hj_diffuse_cb* m, float u, float v) {
  float fu = (0.5f * u) / m->scale_u, fv = (0.5f * v) / m->scale_v;
  fu = fu - floorf(fu); fv = fv - floorf(fv);
  int a = fu < 0.5f, b = fv < 0.5f;
  return (a != b) ? ld3(m->color_b) : ld3(m->color_a);
}

/* shader/material.glsl:18-30 (cosine folded in; non-diffuse -> 0) */
static inline v3 eval_bsdf(const scene_t* S, uint32_t mat, v3 wi, const its_t* its) {
  uint32_t tag = mat >> HJ_MATERIAL_TAG_SHIFT, idx = mat & HJ_MATERIAL_INDEX_MASK;
  v3 color;
  if (tag == HJ_MAT_DIFFUSE) color = ld3(S->sc->diffuse[idx].color);
  else if (tag == HJ_MAT_DIFFUSECBOARD) color = checkerboard(&S->sc->diffusecb[idx], its->u, its->v);
  else return V(0, 0, 0);
  float cs = dot3(its->n, wi);
  return v_scale(v_scale(color, cs), INV_PIF);
}

static inline v3 reflect3(v3 I, v3 N) {
  float s = 2.0f * dot3(N, I);
  return v_sub(I, v_scale(N, s));
}

/* shader/material.glsl:33-91.  Returns the BSDF weight; *alive = 0 for the
 * emissive case where the reference leaves `wo` unwritten and the weight is 0
 * (Appendix C-5: every later contribution is multiplied by zero). */
static inline v3 sample_bsdf(const scene_t* S, uint32_t mat, v3 wi, const its_t* its, uint32_t* rng,
                             v3* wo, v3* ext, int* alive) {
  uint32_t tag = mat >> HJ_MATERIAL_TAG_SHIFT, idx = mat & HJ_MATERIAL_INDEX_MASK;
  *alive = 1;
  switch (tag) {
    case HJ_MAT_DIFFUSE: {
      v3 l = rand_cos_hemisphere(rng);
      *wo = v_add(v_add(v_scale(its->ft, l.x), v_scale(its->fb, l.y)), v_scale(its->fn, l.z));
      return ld3(S->sc->diffuse[idx].color);
    }
    case HJ_MAT_DIFFUSECBOARD: {
      v3 l = rand_cos_hemisphere(rng);
      *wo = v_add(v_add(v_scale(its->ft, l.x), v_scale(its->fb, l.y)), v_scale(its->fn, l.z));
      return checkerboard(&S->sc->diffusecb[idx], its->u, its->v);
    }
    case HJ_MAT_MIRROR:
      *wo = reflect3(wi, its->n);
      return V(1, 1, 1);
    case HJ_MAT_DIELECTRIC: {
      const hj_dielectric* m = &S->sc->dielectric[idx];
      float eta = m->eta;
      float etaInv = 1.0f / eta;
      float cosI = neg_of(dot3(its->n, wi));
      v3 normal = its->n;
      int inside = cosI > 0.0f;                 /* sic: the reference's flag is inverted (C-3) */
      if (cosI < 0.0f) {
        eta = etaInv;
        etaInv = 1.0f / eta;
        normal = v_neg(normal);
        cosI = -cosI;
      }
      float k = 1.0f - (etaInv * etaInv) * (1.0f - cosI * cosI);
      if (k <= 0.0f) {
        *wo = reflect3(wi, normal);
      } else {
        float cosO = sqrtf(k);
        float rpar = (eta * cosI - cosO) / (eta * cosI + cosO);
        float rorth = (cosI - eta * cosO) / (cosI + eta * cosO);
        float fr = 0.5f * (rpar * rpar + rorth * rorth);
        if (rng_float(rng) < fr) {
          *wo = reflect3(wi, normal);
        } else {
          inside = !inside;
          v3 par = v_sub(wi, v_scale(normal, dot3(wi, normal)));
          *wo = v_sub(v_scale(par, etaInv), v_scale(normal, cosO));
        }
      }
      if (inside) *ext = ld3(m->extinction);
      return V(1, 1, 1);
    }
    default: /* HJ_MAT_EMISSIVE and unknown tags: weight 0 */
      *alive = 0;
      *wo = V(0, 0, 0);
      return V(0, 0, 0);
  }
}

/* ------------------------------------------------------------------- camera */

static inline void quat_mult(const float a[4], const float b[4], float r[4]) {
  /* shader/quaternion.glsl:1-6 */
  v3 av = V(a[0], a[1], a[2]), bv = V(b[0], b[1], b[2]);
  r[3] = a[3] * b[3] - dot3(av, bv);
  v3 c = cross3(av, bv);
  r[0] = (c.x + a[0] * b[3]) + b[0] * a[3];
  r[1] = (c.y + a[1] * b[3]) + b[1] * a[3];
  r[2] = (c.z + a[2] * b[3]) + b[2] * a[3];
}
static inline v3 quat_rotate(v3 v, const float q[4]) {
  /* shader/quaternion.glsl:15-19 */
  float vq[4] = {v.x, v.y, v.z, 0.0f}, tmp[4], qc[4] = {-q[0], -q[1], -q[2], q[3]}, out[4];
  quat_mult(q, vq, tmp);
  quat_mult(tmp, qc, out);
  return V(out[0], out[1], out[2]);
}
/* tan(radians(fov/2)) is a per-frame constant; evaluated in double and
 * rounded once (shader/render.glsl:28). */
static inline float tan_half_fov(float fov_deg) {
  return (float)tan((double)(0.5f * fov_deg) * (3.14159265358979323846 / 180.0));
}
/* shader/render.glsl:26-36 */
static inline ray_t camera_ray(const hj_camera* cam, float tanHalf, float px, float py, float W, float H) {
  float x = px - 0.5f * W, y = py - 0.5f * H;
  x = (x * tanHalf) / (0.5f * W);
  y = (y * tanHalf) / (0.5f * W);
  ray_t r;
  r.o = V(cam->position[0], cam->position[1], cam->position[2]);
  r.d = normalize3(quat_rotate(V(x, -y, -1.0f), cam->rotation));
  r.tmin = M_EPSF;
  r.tmax = INFINITY;   /* 1e100 is +inf in binary32 (Appendix C-13) */
  return r;
}

/* --------------------------------------------------------------- integrator */

typedef struct { float rgb[3]; float w; float n[3]; float depth; } sample_t;   /* layers 0,1 of render.glsl:172-173 */

/* Diagnostic (tests and tools only): hjo_set_ray_log(path) makes every ray of the single-threaded entry points (hjo_integrate_block)
 * append a record of eleven floats - o, d, tMin, tMax, kind (0 closest, 1 shadow), hit id or -1, index of the emitter a shadow ray
 * aims at (-1 for a closest-hit ray) - to `path`; NULL closes the log. */
static FILE* g_ray_log = NULL;
HJO_EXPORT void hjo_set_ray_log(const char* path) {
  if (g_ray_log) { fclose(g_ray_log); g_ray_log = NULL; }
  if (path) g_ray_log = fopen(path, "wb");
}
static void log_ray(const ray_t* r, int kind, int id) {
  if (!g_ray_log) return;
  const float rec[11] = {r->o.x, r->o.y, r->o.z, r->d.x, r->d.y, r->d.z, r->tmin, r->tmax, (float)kind, (float)id,
                         kind == 1 ? (float)g_last_emitter : -1.0f};
  fwrite(rec, sizeof rec, 1, g_ray_log);
}

/* shader/render.glsl:81-147 */
static void integrate_ray(const scene_t* S, ray_t ray, uint32_t* rng, uint32_t max_bounces, uint32_t rr_start,
                          sample_t* out) {
  hjo_counters* c = S->ctr;
  v3 ext = V(0, 0, 0), total = V(0, 0, 0), T = V(1, 1, 1);
  out->depth = 0.0f; out->n[0] = out->n[1] = out->n[2] = 0.0f;
  int was_discrete = 1;
  its_t its; memset(&its, 0, sizeof its);
  for (uint32_t bounce = 0; bounce < max_bounces; bounce++) {
    c->closest_calls++;
    {
      const int hit_ = intersect_scene(S, ray, &its, closest_ctr(c));
      log_ray(&ray, 0, hit_ ? its.id : -1);
      if (!hit_) break;
    }
    c->hits++;
    if (bounce == 0) { out->depth = its.t; out->n[0] = its.n.x; out->n[1] = its.n.y; out->n[2] = its.n.z; }
    uint32_t mat = S->sc->materials[its.id];
    uint32_t tag = mat >> HJ_MATERIAL_TAG_SHIFT, midx = mat & HJ_MATERIAL_INDEX_MASK;
    float dist = len3(v_sub(ray.o, its.p));
    T = v_mul(T, V(hj_exp(-ext.x * dist), hj_exp(-ext.y * dist), hj_exp(-ext.z * dist)));
    if (tag == HJ_MAT_EMISSIVE && was_discrete)
      total = v_add(total, v_mul(T, ld3(S->sc->emissive[midx].power)));
    if (tag == HJ_MAT_DIFFUSE || tag == HJ_MAT_DIFFUSECBOARD) {
      ray_t sh;
      c->nee_evals++;
      v3 imp = sample_emitter(S, its.p, rng, &sh);
      if (len3(imp) > M_EPSF && dot3(sh.d, its.n) > 0.0f) {
        its_t dummy; memset(&dummy, 0, sizeof dummy);
        c->shadow_calls++;
        int occluded = intersect_scene(S, sh, &dummy, shadow_ctr(c));   /* scene.glsl:92-96: full closest hit */
        log_ray(&sh, 1, occluded ? dummy.id : -1);
        c->shadow_hits += (uint64_t)occluded;
        if (!occluded) {
          v3 f = eval_bsdf(S, mat, sh.d, &its);
          total = v_add(total, v_mul(v_mul(T, f), imp));
        }
      }
    }
    v3 wo; int alive;
    v3 wgt = sample_bsdf(S, mat, ray.d, &its, rng, &wo, &ext, &alive);
    T = v_mul(T, wgt);
    if (!alive) break;
    ray.d = wo; ray.o = its.p; ray.tmin = 2.0f * M_EPSF; ray.tmax = INFINITY;
    was_discrete = (tag != HJ_MAT_DIFFUSE && tag != HJ_MAT_DIFFUSECBOARD);
    if (bounce >= rr_start) { /* `bounce > 3` (render.glsl:137) for rr_start = 4 */
      float q = f_min(0.99f, f_max(T.x, f_max(T.y, T.z)));
      if (rng_float(rng) > q) break;
      T = v_divs(T, q);
    }
  }
  out->rgb[0] = total.x; out->rgb[1] = total.y; out->rgb[2] = total.z; out->w = 1.0f;
}

/* shader/render.glsl:149-175: one block -> 128x128 sample records (row pitch = block width) */
static void integrate_block(const scene_t* S, const hj_image_block* b, const hj_render_opts* o, float tanHalf,
                            sample_t* out, uint32_t row0, uint32_t row1) {
  const float W = (float)b->original_dimension[0], H = (float)b->original_dimension[1];
  for (uint32_t ly = row0; ly < row1; ly++) {
    for (uint32_t lx = 0; lx < b->dimension[0]; lx++) {
      sample_t* s = &out[ly * b->dimension[0] + lx];
      if (lx >= b->original_dimension[0] || ly >= b->original_dimension[1]) { memset(s, 0, sizeof *s); continue; }
      uint32_t gx = lx + b->origin[0], gy = ly + b->origin[1];
      uint32_t rng = rng_seed(b->seed + lx + ly * b->dimension[0]);
      ray_t ray = camera_ray(&S->sc->camera, tanHalf, (float)gx + b->sample_offset[0],
                             (float)gy + b->sample_offset[1], W, H);
      S->ctr->paths++;
      integrate_ray(S, ray, &rng, o->max_bounces, o->rr_start, s);
    }
  }
}

/* ----------------------------------------------------------- reconstruction */

/* shader/reconstruction.glsl:22-66 for the output rows gy of the image with gy % row_mod == row_rem
 * (every pixel has ONE owner thread, so its sum keeps block order whatever the number of threads).
 * `smp` is the block's sample image (pitch = block width).  Centre loads outside the block read 0: for the
 * reference's own block lists (ragged blocks only at the right / bottom image edge, where the store is dropped)
 * that is exactly what reconstruction.glsl:31 sees; see DESIGN.md section 2 for custom interior blocks. */
static void reconstruct_block_rows(const hj_image_block* b, const hj_render_opts* o, const sample_t* smp,
                                   float* accum, uint32_t W, uint32_t H, uint32_t row_mod, uint32_t row_rem) {
  const int R = (int)o->recon_radius;
  const int Dx = (int)b->dimension[0], Dy = (int)b->dimension[1];
  const float gaussFac = -1.0f / ((2.0f * o->recon_stddev) * o->recon_stddev);
  const float curveOffset = hj_exp(gaussFac * (float)(R * R));
  for (int ly = -R; ly < Dy + R; ly++) {
    long gy = (long)b->origin[1] + ly;
    if (gy < 0 || gy >= (long)H || (uint32_t)gy % row_mod != row_rem) continue;
    for (int lx = -R; lx < Dx + R; lx++) {
      long gx = (long)b->origin[0] + lx;
      if (gx < 0 || gx >= (long)W) continue;
      float* px = &accum[((size_t)gy * W + (size_t)gx) * 4];
      float acc[4] = {px[0], px[1], px[2], px[3]};
      v3 nc = V(0, 0, 0);
      if (lx >= 0 && lx < Dx && ly >= 0 && ly < Dy) nc = ld3(smp[ly * Dx + lx].n);
      for (int dx = -R; dx <= R; dx++) {
        if (lx + dx < 0 || lx + dx >= Dx) continue;
        for (int dy = -R; dy <= R; dy++) {
          if (ly + dy < 0 || ly + dy >= Dy) continue;
          float sx = ((float)dx + b->sample_offset[0]) - 0.5f;
          float sy = ((float)dy + b->sample_offset[1]) - 0.5f;
          float w = hj_exp(gaussFac * (sx * sx + sy * sy)) - curveOffset;
          if (w < 0.0f) continue;
          const sample_t* sp = &smp[(ly + dy) * Dx + (lx + dx)];
          v3 no = v_sub(ld3(sp->n), nc);
          /* albedo layer is always 0 (render.glsl:84-85,174): its term is +0 */
          w *= hj_exp(-(dot3(no, no) * 2.0f));
          float v0 = w * sp->rgb[0], v1 = w * sp->rgb[1], v2 = w * sp->rgb[2], v3_ = w * sp->w;
          if (v0 != v0 || v1 != v1 || v2 != v2 || v3_ != v3_) continue;
          acc[0] += v0; acc[1] += v1; acc[2] += v2; acc[3] += v3_;
        }
      }
      px[0] = acc[0]; px[1] = acc[1]; px[2] = acc[2]; px[3] = acc[3];
    }
  }
}

/* ----------------------------------------------------------------- threading */

typedef struct job job_t;
typedef struct {
  job_t* job; int tid; hjo_counters ctr;
} worker_t;

struct job {
  const hj_scene_desc* sc; const hj_render_opts* opts; const hj_image_block* blocks; size_t nblocks;
  uint32_t W, H; float* accum; int nthreads; float tanHalf;
  sample_t** smp;               /* per block-in-batch sample images */
  size_t batch_begin, batch_n;
  volatile long next_item;      /* dynamic scheduler over (block,row-chunk) items */
  pthread_barrier_t bar;
};

#define ROW_CHUNK 8

static void* worker_main(void* arg) {
  worker_t* w = (worker_t*)arg; job_t* J = w->job;
  hjo_counters local; memset(&local, 0, sizeof local);   /* on this thread's stack: no false sharing between workers */
  scene_t S; S.sc = J->sc; S.ns = (uint32_t)J->sc->num_spheres; S.nq = (uint32_t)J->sc->num_quads;
  S.nt = (uint32_t)J->sc->num_triangles; S.use_bvh = (int)J->opts->use_bvh; S.ctr = &local;
  const size_t chunks_per_block = (HJ_BLOCK_SIZE + ROW_CHUNK - 1) / ROW_CHUNK;
  for (;;) {
    pthread_barrier_wait(&J->bar);            /* batch start */
    if (J->batch_n == 0) break;
    for (;;) {                                /* phase 1: integrate */
      long it = __sync_fetch_and_add(&J->next_item, 1);
      if ((size_t)it >= J->batch_n * chunks_per_block) break;
      size_t bi = (size_t)it / chunks_per_block, ch = (size_t)it % chunks_per_block;
      const hj_image_block* b = &J->blocks[J->batch_begin + bi];
      uint32_t r0 = (uint32_t)(ch * ROW_CHUNK), r1 = r0 + ROW_CHUNK;
      if (r0 >= b->dimension[1]) continue;
      if (r1 > b->dimension[1]) r1 = b->dimension[1];
      integrate_block(&S, b, J->opts, J->tanHalf, J->smp[bi], r0, r1);
    }
    pthread_barrier_wait(&J->bar);            /* samples complete */
    /* phase 2: accumulate, block order preserved per pixel.  Rows are dealt to the threads INTERLEAVED: a batch of
     * consecutive blocks covers a few block rows of a large frame, and contiguous row ranges left most threads
     * without work there (the 4096 x 4096 frame ran at 0.4 of the 1024 x 1024 frame's rate). */
    for (size_t bi = 0; bi < J->batch_n; bi++)
      reconstruct_block_rows(&J->blocks[J->batch_begin + bi], J->opts, J->smp[bi], J->accum, J->W, J->H,
                             (uint32_t)J->nthreads, (uint32_t)w->tid);
    pthread_barrier_wait(&J->bar);            /* batch end */
  }
  w->ctr = local;
  return NULL;
}

static void add_counters(hjo_counters* a, const hjo_counters* b) {
  uint64_t* x = (uint64_t*)a; const uint64_t* y = (const uint64_t*)b;
  for (size_t i = 0; i < sizeof(hjo_counters) / 8; i++) x[i] += y[i];
}

static int validate_blocks(const hj_image_block* blocks, size_t n, uint32_t W, uint32_t H) {
  for (size_t i = 0; i < n; i++) {
    const hj_image_block* b = &blocks[i];
    if (b->dimension[0] == 0 || b->dimension[1] == 0 || b->dimension[0] > HJ_BLOCK_SIZE ||
        b->dimension[1] > HJ_BLOCK_SIZE) return 0;
    if (b->original_dimension[0] != W || b->original_dimension[1] != H) return 0;
  }
  return 1;
}

/* Renders `blocks` in order into accum (W*H*4 floats, read-modify-write).
 * Returns 0 on success.  seconds_out (optional) = wall time of the render loop. */
HJO_EXPORT int hjo_render_blocks(const hj_scene_desc* sc, const hj_image_block* blocks, size_t nblocks,
                                 const hj_render_opts* opts, uint32_t W, uint32_t H, float* accum, int nthreads,
                                 hjo_counters* ctr_out, double* seconds_out) {
  if (!sc || !opts || !accum || (nblocks && !blocks)) return HJ_ERR_INVALID;
  if (opts->recon_radius != 2) return HJ_ERR_UNSUPPORTED;
  if (!validate_blocks(blocks, nblocks, W, H)) return HJ_ERR_INVALID;
  if (sc->num_materials != sc->num_spheres + sc->num_quads + sc->num_triangles) return HJ_ERR_INVALID;
  if (nthreads < 1) nthreads = 1;
  job_t J; memset(&J, 0, sizeof J);
  J.sc = sc; J.opts = opts; J.blocks = blocks; J.nblocks = nblocks; J.W = W; J.H = H; J.accum = accum;
  J.nthreads = nthreads; J.tanHalf = tan_half_fov(sc->camera.fov);
  size_t max_batch = (size_t)nthreads * 4; if (max_batch < 16) max_batch = 16;
  J.smp = (sample_t**)calloc(max_batch, sizeof(sample_t*));
  if (!J.smp) return HJ_ERR_NOMEM;
  for (size_t i = 0; i < max_batch; i++) {
    J.smp[i] = (sample_t*)malloc(sizeof(sample_t) * HJ_BLOCK_SIZE * HJ_BLOCK_SIZE);
    if (!J.smp[i]) return HJ_ERR_NOMEM;
  }
  pthread_barrier_init(&J.bar, NULL, (unsigned)nthreads + 1);
  worker_t* ws = (worker_t*)calloc((size_t)nthreads, sizeof(worker_t));
  pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
  for (int t = 0; t < nthreads; t++) { ws[t].job = &J; ws[t].tid = t; pthread_create(&th[t], NULL, worker_main, &ws[t]); }
  struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
  for (size_t begin = 0; begin < nblocks; begin += max_batch) {
    J.batch_begin = begin; J.batch_n = (nblocks - begin < max_batch) ? nblocks - begin : max_batch;
    J.next_item = 0;
    pthread_barrier_wait(&J.bar);   /* start */
    pthread_barrier_wait(&J.bar);   /* samples complete */
    pthread_barrier_wait(&J.bar);   /* end */
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  J.batch_n = 0;
  pthread_barrier_wait(&J.bar);     /* release workers to exit */
  hjo_counters total; memset(&total, 0, sizeof total);
  for (int t = 0; t < nthreads; t++) { pthread_join(th[t], NULL); add_counters(&total, &ws[t].ctr); }
  if (ctr_out) *ctr_out = total;
  if (seconds_out) *seconds_out = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  pthread_barrier_destroy(&J.bar);
  for (size_t i = 0; i < max_batch; i++) free(J.smp[i]);
  free(J.smp); free(ws); free(th);
  return HJ_OK;
}

/* --------------------------------------------- deterministic block generator */

static inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
/* Replaces `seed: rand::random()` (src/main.rs:675). */
HJO_EXPORT uint32_t hjo_block_seed(uint64_t master, uint32_t pass, uint32_t block_in_pass) {
  uint64_t h = splitmix64(splitmix64(master ^ 0x484A424Cull) + (((uint64_t)pass << 32) | block_in_pass));
  return (uint32_t)(h >> 32);
}
/* Replaces `sample_offset: rand::random()` (src/main.rs:643,670): two 24-bit uniforms in [0,1). */
HJO_EXPORT void hjo_pass_offset(uint64_t master, uint32_t k, float out[2]) {
  uint64_t h = splitmix64(splitmix64(master ^ 0x484A4F46ull) + k);
  out[0] = (float)(uint32_t)(h >> 40) * (1.0f / 16777216.0f);
  out[1] = (float)(uint32_t)((h >> 16) & 0xFFFFFFu) * (1.0f / 16777216.0f);
}
/* src/main.rs:648-682: pass-major raster order; ids run on across passes; the
 * LAST block of a pass already carries the NEXT pass's offset because the
 * generator refreshes sample_offset before building the returned block
 * (src/main.rs:664-680).  Writes up to `cap` blocks of passes
 * [pass_begin,pass_end); returns the number the range holds. */
HJO_EXPORT size_t hjo_make_blocks(uint32_t W, uint32_t H, uint32_t block, uint64_t master, uint32_t pass_begin,
                                  uint32_t pass_end, hj_image_block* out, size_t cap) {
  uint32_t nbx = (W + block - 1) / block, nby = (H + block - 1) / block;
  size_t n = 0;
  for (uint32_t p = pass_begin; p < pass_end; p++) {
    uint32_t j = 0;
    for (uint32_t by = 0; by < nby; by++)
      for (uint32_t bx = 0; bx < nbx; bx++, j++) {
        if (n < cap) {
          hj_image_block* b = &out[n];
          b->id = p * (nbx * nby) + j;
          b->seed = hjo_block_seed(master, p, j);
          b->origin[0] = bx * block; b->origin[1] = by * block;
          b->dimension[0] = (W - bx * block < block) ? W - bx * block : block;
          b->dimension[1] = (H - by * block < block) ? H - by * block : block;
          b->original_dimension[0] = W; b->original_dimension[1] = H;
          hjo_pass_offset(master, p + ((j == nbx * nby - 1) ? 1u : 0u), b->sample_offset);
        }
        n++;
      }
  }
  return n;
}

/* ---------------------------------------------------- function-level probes */
/* Small entry points so tests can pin individual functions (golden vectors). */

HJO_EXPORT uint32_t hjo_rng_seed(uint32_t seed) { return rng_seed(seed); }
HJO_EXPORT uint32_t hjo_rng_next(uint32_t* state) { return rng_uint(state); }
HJO_EXPORT float hjo_rng_float(uint32_t* state) { return rng_float(state); }
HJO_EXPORT float hjo_exp(float x) { return hj_exp(x); }
HJO_EXPORT void hjo_sincos2pi(float v, float* sc) { hj_sincos2pi(v, &sc[0], &sc[1]); }
HJO_EXPORT float hjo_atan2(float y, float x) { return hj_atan2(y, x); }
HJO_EXPORT float hjo_asin(float x) { return hj_asin(x); }
HJO_EXPORT void hjo_cos_hemisphere(uint32_t* state, float* out3) {
  v3 r = rand_cos_hemisphere(state); out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
HJO_EXPORT void hjo_barycentric(uint32_t* state, float* out3) {
  v3 r = rand_barycentric(state); out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
HJO_EXPORT void hjo_uniform_sphere(uint32_t* state, float* out3) {
  v3 r = rand_uniform_sphere(state); out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}

/* rays: n x 8 floats (o.xyz, d.xyz, tmin, tmax).  hits: n x 4 (id as int bits, t, u, v of the
 * raw hit BEFORE populate).  full (optional): n x 16 floats (p3, n3, uv2, t3(frame t), b3, pad2). */
static uint32_t* g_node_hist = NULL;   /* probe: per-node visit counts of hjo_intersect (single-threaded) */
HJO_EXPORT void hjo_set_node_histogram(uint32_t* hist) { g_node_hist = hist; }

HJO_EXPORT int hjo_intersect(const hj_scene_desc* sc, int use_bvh, const float* rays, size_t n, float* hits,
                             float* full) {
  hjo_counters c; memset(&c, 0, sizeof c);
  scene_t S; S.sc = sc; S.ns = (uint32_t)sc->num_spheres; S.nq = (uint32_t)sc->num_quads;
  S.nt = (uint32_t)sc->num_triangles; S.use_bvh = use_bvh; S.ctr = &c;
  for (size_t i = 0; i < n; i++) {
    const float* r = &rays[i * 8];
    ray_t ray; ray.o = V(r[0], r[1], r[2]); ray.d = V(r[3], r[4], r[5]); ray.tmin = r[6]; ray.tmax = r[7];
    its_t its; memset(&its, 0, sizeof its);
    walk_ctr wc = closest_ctr(&c); wc.hist = g_node_hist;
    int hit = intersect_scene(&S, ray, &its, wc);
    int32_t id = hit ? its.id : -1;
    memcpy(&hits[i * 4], &id, 4);
    hits[i * 4 + 1] = hit ? its.t : 0.0f;
    if (full) {
      float* f = &full[i * 16];
      memset(f, 0, 64);
      if (hit) {
        f[0] = its.p.x; f[1] = its.p.y; f[2] = its.p.z; f[3] = its.n.x; f[4] = its.n.y; f[5] = its.n.z;
        f[6] = its.u; f[7] = its.v; f[8] = its.ft.x; f[9] = its.ft.y; f[10] = its.ft.z;
        f[11] = its.fb.x; f[12] = its.fb.y; f[13] = its.fb.z;
      }
    }
    hits[i * 4 + 2] = hit ? its.raw_u : 0.0f;
    hits[i * 4 + 3] = hit ? its.raw_v : 0.0f;
  }
  return HJ_OK;
}

/* Camera rays for pixel centres + offset: out n x 6 (o, d). */
HJO_EXPORT void hjo_camera_rays(const hj_camera* cam, uint32_t W, uint32_t H, const float* pix_xy, size_t n,
                                float* out) {
  float th = tan_half_fov(cam->fov);
  for (size_t i = 0; i < n; i++) {
    ray_t r = camera_ray(cam, th, pix_xy[2 * i], pix_xy[2 * i + 1], (float)W, (float)H);
    float* o = &out[i * 6];
    o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; o[3] = r.d.x; o[4] = r.d.y; o[5] = r.d.z;
  }
}

/* Per-path samples of one block (no reconstruction): out = dim.x*dim.y x 8 floats
 * (rgb, w, normal, depth) = layers 0 and 1 of the intermediate image. */
HJO_EXPORT int hjo_integrate_block(const hj_scene_desc* sc, const hj_image_block* b, const hj_render_opts* opts,
                                   float* out, hjo_counters* ctr_out) {
  hjo_counters c; memset(&c, 0, sizeof c);
  scene_t S; S.sc = sc; S.ns = (uint32_t)sc->num_spheres; S.nq = (uint32_t)sc->num_quads;
  S.nt = (uint32_t)sc->num_triangles; S.use_bvh = (int)opts->use_bvh; S.ctr = &c;
  integrate_block(&S, b, opts, tan_half_fov(sc->camera.fov), (sample_t*)out, 0, b->dimension[1]);
  if (ctr_out) *ctr_out = c;
  return HJ_OK;
}

/* Reconstruction of one block from given samples into accum (whole image rows). */
HJO_EXPORT int hjo_reconstruct_block(const hj_image_block* b, const hj_render_opts* opts, const float* samples,
                                     float* accum, uint32_t W, uint32_t H) {
  reconstruct_block_rows(b, opts, (const sample_t*)samples, accum, W, H, 1u, 0u);
  return HJ_OK;
}

/* One tap weight of the reconstruction filter (Appendix B-7) before the bilateral factor. */
HJO_EXPORT float hjo_recon_gauss(int dx, int dy, float offx, float offy, float stddev, int radius) {
  float g = -1.0f / ((2.0f * stddev) * stddev);
  float c0 = hj_exp(g * (float)(radius * radius));
  float sx = ((float)dx + offx) - 0.5f, sy = ((float)dy + offy) - 0.5f;
  return hj_exp(g * (sx * sx + sy * sy)) - c0;
}

/* Fresnel reflectance + outgoing direction of the dielectric for a given draw (probe of
 * material.glsl:50-87).  out: wo3, fr, took_extinction. */
HJO_EXPORT void hjo_dielectric_probe(float eta, const float* n3, const float* wi3, uint32_t* rng, float* out5) {
  hj_dielectric m = {{0.25f, 0.5f, 0.75f}, eta};
  hj_scene_desc sc; memset(&sc, 0, sizeof sc); sc.dielectric = &m; sc.num_dielectric = 1;
  hjo_counters c; memset(&c, 0, sizeof c);
  scene_t S; S.sc = &sc; S.ns = S.nq = S.nt = 0; S.use_bvh = 1; S.ctr = &c;
  its_t its; memset(&its, 0, sizeof its); its.n = V(n3[0], n3[1], n3[2]);
  v3 wo, ext = V(0, 0, 0); int alive;
  sample_bsdf(&S, (HJ_MAT_DIELECTRIC << HJ_MATERIAL_TAG_SHIFT), V(wi3[0], wi3[1], wi3[2]), &its, rng, &wo, &ext, &alive);
  out5[0] = wo.x; out5[1] = wo.y; out5[2] = wo.z;
  out5[3] = 0.0f; out5[4] = (ext.x != 0.0f) ? 1.0f : 0.0f;
}

/* Probe of one shading step (render.glsl:94-135 without the visibility test): rays n x 8 (o, d, tMin, tMax) and one RNG
 * state per ray -> out n x 20: [0] objectID (-1 miss), [1..3] T*evalBSDF*importance of the NEE sample with T = 1 (0 when
 * the tests of render.glsl:121 reject it), [4..6] shadow direction, [7] shadow tMax, [8..10] wo, [11..13] sampleBSDF weight,
 * [14] alive, [15] RNG state bits after the step, [16..18] extinction after the step, [19] emitted radiance .r if emissive. */
HJO_EXPORT int hjo_shade_probe(const hj_scene_desc* sc, const float* rays, const uint32_t* rng_in, size_t n, float* out) {
  hjo_counters c; memset(&c, 0, sizeof c);
  scene_t S; S.sc = sc; S.ns = (uint32_t)sc->num_spheres; S.nq = (uint32_t)sc->num_quads;
  S.nt = (uint32_t)sc->num_triangles; S.use_bvh = 1; S.ctr = &c;
  for (size_t i = 0; i < n; i++) {
    const float* r = &rays[i * 8];
    float* o = &out[i * 20];
    memset(o, 0, 80);
    ray_t ray; ray.o = V(r[0], r[1], r[2]); ray.d = V(r[3], r[4], r[5]); ray.tmin = r[6]; ray.tmax = r[7];
    its_t its; memset(&its, 0, sizeof its);
    uint32_t rng = rng_in[i];
    int32_t id = -1;
    if (intersect_scene(&S, ray, &its, closest_ctr(&c))) {
      id = its.id;
      uint32_t mat = sc->materials[its.id];
      uint32_t tag = mat >> HJ_MATERIAL_TAG_SHIFT, midx = mat & HJ_MATERIAL_INDEX_MASK;
      if (tag == HJ_MAT_EMISSIVE) o[19] = sc->emissive[midx].power[0];
      if (tag == HJ_MAT_DIFFUSE || tag == HJ_MAT_DIFFUSECBOARD) {
        ray_t sh;
        v3 imp = sample_emitter(&S, its.p, &rng, &sh);
        o[4] = sh.d.x; o[5] = sh.d.y; o[6] = sh.d.z; o[7] = sh.tmax;
        if (len3(imp) > M_EPSF && dot3(sh.d, its.n) > 0.0f) {
          v3 f = v_mul(eval_bsdf(&S, mat, sh.d, &its), imp);
          o[1] = f.x; o[2] = f.y; o[3] = f.z;
        }
      }
      v3 wo = V(0, 0, 0), ext = V(0, 0, 0); int alive = 0;
      v3 w = sample_bsdf(&S, mat, ray.d, &its, &rng, &wo, &ext, &alive);
      o[8] = wo.x; o[9] = wo.y; o[10] = wo.z; o[11] = w.x; o[12] = w.y; o[13] = w.z; o[14] = (float)alive;
      o[16] = ext.x; o[17] = ext.y; o[18] = ext.z;
    }
    memcpy(&o[0], &id, 4);
    memcpy(&o[15], &rng, 4);
  }
  return HJ_OK;
}

HJO_EXPORT size_t hjo_sizeof_counters(void) { return sizeof(hjo_counters); }
