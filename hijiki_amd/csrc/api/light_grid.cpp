// Light-shaft visibility grid: which next-event shadow rays need no walk at all.
//
// The reference traces a shadow ray for every next-event sample (shader/render.glsl:117-126 -> scene.glsl:92-96, a full
// closest-hit walk; "TODO: optimize" upstream) and only its boolean is used.  On a Cornell-box-shaped scene most of those
// rays run from a wall to the light through empty space.  This grid proves that ahead of time, per (cell of a uniform grid
// over the scene, emitter): bit e of a cell is set only if EVERY shadow ray the renderer can generate from a hit point in
// that cell towards a sampled point of emitter e is unoccluded - whatever tree is walked, and including what float rounding
// can do to the shape tests - so the shade stage adds such a sample's contribution at once and no ray is queued.  A cleared
// bit costs nothing but the walk that would have run anyway.  Shaft culling in the sense of Haines & Wallace 1991.
//
// A bit is set when all of this holds (scale = extent of the scene incl. the camera; tol_p = 2e-6 scale bounds the distance
// of a computed hit point from its shape's plane, tol_s = 2e-7 scale is "coplanar", m = 1e-4 max(1, scale) pads every box):
//   1. every shape whose padded bounding box overlaps the padded cell is a triangle or quad whose float test is well conditioned
//      (cond <= 2, "tame shapes" below: a needle triangle reports hits of itself), and all of them lie in ONE plane P (every
//      vertex within tol_s): a hit point p in the cell lies on one of them, i.e. within 2 tol_p of P;
//   2. emitter e is a triangle or quad (plane Q) and all of it lies on one side of P at an angle: for every point y of its
//      padded box, |dist(y, P)| - tol_p >= 0.25 |y - x| for every x of the padded cell.  A ray leaving P that steeply is more
//      than tol_s away from P from t = 1e-4 on (tMin of a shadow ray is 2e-4, scene.glsl:85), so no shape coplanar with P
//      - the one p lies on, its neighbours in the wall - can be hit; the same with the roles exchanged at the emitter (sin >= 0.03
//      when the shapes in Q are exactly planar, 0.1 at most): shapes coplanar with Q, the emitter itself included, are met at
//      t >= dist - 2.5e-5 > tMax = dist - 1e-4;
//   3. no other shape's padded bounding box touches the convex hull of the padded cell and the padded box of the emitter
//      (every segment p -> y lies in that hull).  The hull of two boxes is the intersection, over the three axes, of the
//      extruded 2-D hulls of their projections (every facet normal of the hull is perpendicular to an axis), so a box is
//      outside it as soon as one projection separates them: by the union rectangle or by one of the four lines through
//      corresponding corners.  The test errs only towards "touches".
// Cells that hold several planes - corners, the facets of a mesh - are proven by bundle proofs (further down), for hit points the
// shade stage has checked against their shape: a second byte per cell.
// The leaves are enumerated through the uploaded skip-link array itself, with subtree bounds recomputed from the shapes
// (the array's own boxes are not trusted: tests upload trees with wrong boxes): every leaf whose bounds touch the shaft is
// looked at, in any well- or ill-formed array whose exits point forward.  Shapes the array does not hold cannot be hit.
#include "light_grid.hpp"
#include "hj_tuning.h"
#include "../kernels/hj_light_grid_const.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <thread>

namespace hjapi {

namespace {

struct Box { double lo[3], hi[3]; };
inline Box empty_box() { return {{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}}; }
inline void grow(Box& b, const double p[3]) { for (int k = 0; k < 3; k++) { b.lo[k] = std::min(b.lo[k], p[k]); b.hi[k] = std::max(b.hi[k], p[k]); } }
inline void join(Box& b, const Box& o) { for (int k = 0; k < 3; k++) { b.lo[k] = std::min(b.lo[k], o.lo[k]); b.hi[k] = std::max(b.hi[k], o.hi[k]); } }
inline Box pad(Box b, double m) { for (int k = 0; k < 3; k++) { b.lo[k] -= m; b.hi[k] += m; } return b; }
inline bool finite_box(const Box& b) { for (int k = 0; k < 3; k++) if (!(std::isfinite(b.lo[k]) && std::isfinite(b.hi[k]) && b.lo[k] <= b.hi[k])) return false; return true; }

// n.x - d = signed distance, |n| = 1; cond = 1 / sin of the angle between the two edges the reference's test takes its normal
// from (triangle.glsl / quad.glsl: cross(b - a, c - a), cross(e1, e2)) - what that float cross product's direction, and with it
// every t and (u, v) the test computes, loses to cancellation (a needle: cond >> 1)
struct Plane { double n[3], d, cond; bool ok; };

// Bounds as the scene's own floats (min / max of float coordinates are exact): half the memory of the double boxes in the two
// passes over a million shapes and two million nodes.
struct BoxF { float lo[3], hi[3]; };
inline BoxF empty_boxf() { return {{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}}; }
inline void joinf(BoxF& b, const BoxF& o) { for (int k = 0; k < 3; k++) { b.lo[k] = std::min(b.lo[k], o.lo[k]); b.hi[k] = std::max(b.hi[k], o.hi[k]); } }
inline Box widen(const BoxF& b) { Box r; for (int k = 0; k < 3; k++) { r.lo[k] = b.lo[k]; r.hi[k] = b.hi[k]; } return r; }

// fn(begin, end) over [0, n) in contiguous pieces on up to 16 threads (one piece per thread; small n: the caller's thread)
template <class Fn>
void parallel_pieces(size_t n, size_t min_per_thread, Fn fn) {
  unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  nt = (unsigned)std::max<size_t>(1, std::min<size_t>(nt, n / std::max<size_t>(1, min_per_thread)));
  if (nt <= 1) { fn((size_t)0, n, 0u); return; }
  std::vector<std::thread> pool;
  unsigned started = 1;
  try { for (unsigned t = 1; t < nt; t++) { pool.emplace_back(fn, n * t / nt, n * (t + 1) / nt, t); started++; } } catch (const std::exception&) {}
  fn((size_t)0, n / nt, 0u);
  for (unsigned t = started; t < nt; t++) fn(n * t / nt, n * (t + 1) / nt, t);     // (threads that could not be started)
  for (auto& th : pool) th.join();
}

struct Geometry {
  const hj_scene_desc* s;
  size_t ns, nq, nt;
  int vertices(size_t shape, double v[4][3]) const {          // 0 for a sphere
    if (shape < ns) return 0;
    if (shape < ns + nq) {
      const hj_quad& q = s->quads[shape - ns];
      for (int k = 0; k < 3; k++) {
        v[0][k] = q.origin[k]; v[1][k] = (double)q.origin[k] + q.edge1[k]; v[2][k] = (double)q.origin[k] + q.edge2[k];
        v[3][k] = (double)q.origin[k] + q.edge1[k] + q.edge2[k];
      }
      return 4;
    }
    const hj_triangle& t = s->triangles[shape - ns - nq];
    for (int c = 0; c < 3; c++) for (int k = 0; k < 3; k++) v[c][k] = s->vertices[t.v[c]].pos[k];
    return 3;
  }
  Box bounds(size_t shape) const {
    Box b = empty_box();
    if (shape < ns) {
      const hj_sphere& sp = s->spheres[shape];
      const double r = std::fabs((double)sp.radius);
      for (int k = 0; k < 3; k++) { b.lo[k] = sp.center[k] - r; b.hi[k] = sp.center[k] + r; }
      return b;
    }
    double v[4][3];
    const int n = vertices(shape, v);
    for (int c = 0; c < n; c++) grow(b, v[c]);
    return b;
  }
  BoxF boundsf(size_t shape) const {            // exact: a sphere's centre +- |radius| rounded OUTWARD, vertices as they are
    BoxF b = empty_boxf();
    if (shape < ns) {
      const hj_sphere& sp = s->spheres[shape];
      const double r = std::fabs((double)sp.radius);
      for (int k = 0; k < 3; k++) {
        b.lo[k] = std::nextafter((float)(sp.center[k] - r), -INFINITY);
        b.hi[k] = std::nextafter((float)(sp.center[k] + r), INFINITY);
      }
      return b;
    }
    double v[4][3];
    const int n = vertices(shape, v);
    for (int c = 0; c < n; c++) for (int k = 0; k < 3; k++) {
      b.lo[k] = std::min(b.lo[k], std::nextafter((float)v[c][k], -INFINITY));      // (quad corners are sums: outward by one ulp)
      b.hi[k] = std::max(b.hi[k], std::nextafter((float)v[c][k], INFINITY));
    }
    return b;
  }
  Plane plane(size_t shape) const {
    Plane p{};
    double v[4][3];
    if (vertices(shape, v) < 3) return p;
    const double a[3] = {v[1][0] - v[0][0], v[1][1] - v[0][1], v[1][2] - v[0][2]}, b[3] = {v[2][0] - v[0][0], v[2][1] - v[0][1], v[2][2] - v[0][2]};
    double n[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
    const double l = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    const double la = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]), lb = std::sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
    if (!(l > 1e-9 * la * lb) || !std::isfinite(l) || l == 0.0) return p;        // (needle triangles have no usable plane)
    for (int k = 0; k < 3; k++) p.n[k] = n[k] / l;
    p.d = p.n[0] * v[0][0] + p.n[1] * v[0][1] + p.n[2] * v[0][2];
    p.cond = la * lb / l;
    p.ok = true;
    return p;
  }
  int polygon(size_t shape, double v[8][3]) const {            // the corners in cyclic order (a quad: origin, +e1, +e1+e2, +e2); 0 for a sphere
    double q[4][3];
    const int n = vertices(shape, q);
    static const int quad_order[4] = {0, 1, 3, 2};
    for (int c = 0; c < n; c++) for (int k = 0; k < 3; k++) v[c][k] = q[n == 4 ? quad_order[c] : c][k];
    return n;
  }
  bool coplanar(size_t shape, const Plane& p, double tol) const {
    double v[4][3];
    const int n = vertices(shape, v);
    if (n == 0) return false;
    for (int c = 0; c < n; c++)
      if (!(std::fabs(p.n[0] * v[c][0] + p.n[1] * v[c][1] + p.n[2] * v[c][2] - p.d) <= tol)) return false;
    return true;
  }
};

// The shaft between two boxes, as its projections along the three axes.
struct Shaft {
  Box a, b, u;                                   // the two boxes and their union
  // per axis k (projection plane (i, j) = the other two axes): up to four supporting lines n.x <= c of the 2-D hull
  struct Line { double ni, nj, c; };
  Line lines[3][4];
  int nlines[3];
  Shaft(const Box& A, const Box& B) : a(A), b(B), u(A) {
    join(u, B);
    for (int k = 0; k < 3; k++) {
      const int i = (k + 1) % 3, j = (k + 2) % 3;
      nlines[k] = 0;
      for (int ci = 0; ci < 2; ci++) for (int cj = 0; cj < 2; cj++) {
        // the line through corner (ci, cj) of A's rectangle and the same corner of B's
        const double ax = ci ? A.hi[i] : A.lo[i], ay = cj ? A.hi[j] : A.lo[j], bx = ci ? B.hi[i] : B.lo[i], by = cj ? B.hi[j] : B.lo[j];
        double nx = -(by - ay), ny = bx - ax;                                  // a normal of the line
        if (nx == 0.0 && ny == 0.0) continue;
        // orient it away from both rectangles: every corner of A and B must satisfy n.x <= c (a supporting line), else skip
        for (int flip = 0; flip < 2; flip++) {
          const double sx = flip ? -nx : nx, sy = flip ? -ny : ny, c = sx * ax + sy * ay;
          double worst = -INFINITY;
          for (int di = 0; di < 2; di++) for (int dj = 0; dj < 2; dj++) {
            worst = std::max(worst, sx * (di ? A.hi[i] : A.lo[i]) + sy * (dj ? A.hi[j] : A.lo[j]) - c);
            worst = std::max(worst, sx * (di ? B.hi[i] : B.lo[i]) + sy * (dj ? B.hi[j] : B.lo[j]) - c);
          }
          const double slack = 1e-12 * (std::fabs(c) + std::fabs(sx) + std::fabs(sy));
          if (worst <= slack) { lines[k][nlines[k]++] = {sx, sy, c + slack}; break; }
        }
      }
    }
  }
  // true only if box q certainly does not touch the shaft
  bool outside(const Box& q) const {
    for (int k = 0; k < 3; k++) if (q.lo[k] > u.hi[k] || q.hi[k] < u.lo[k]) return true;
    for (int k = 0; k < 3; k++) {
      const int i = (k + 1) % 3, j = (k + 2) % 3;
      for (int l = 0; l < nlines[k]; l++) {
        const Line& L = lines[k][l];
        // the corner of q that lies deepest inside the half-plane n.x <= c
        const double best = L.ni * (L.ni > 0 ? q.lo[i] : q.hi[i]) + L.nj * (L.nj > 0 ? q.lo[j] : q.hi[j]);
        if (best > L.c) return true;
      }
    }
    return false;
  }
  // the same for a ball (a sphere shape with its padding): outside as soon as it lies beyond the union box along an axis or
  // beyond one supporting line of a projection by more than its radius (a disc outside a half-plane of the 2-D hull)
  bool outside_ball(const double c[3], double r) const {
    for (int k = 0; k < 3; k++) if (c[k] - r > u.hi[k] || c[k] + r < u.lo[k]) return true;
    for (int k = 0; k < 3; k++) {
      const int i = (k + 1) % 3, j = (k + 2) % 3;
      for (int l = 0; l < nlines[k]; l++) {
        const Line& L = lines[k][l];
        const double len = std::sqrt(L.ni * L.ni + L.nj * L.nj);
        if (L.ni * c[i] + L.nj * c[j] - L.c > r * len * (1.0 + 1e-12)) return true;
      }
    }
    return false;
  }
};

// ---- cells on MESHES (round 6; HJ_LIGHT_GRID_MESH) -------------------------------------------------------------------------------
// A cell whose shapes are flat (triangles, quads) but not coplanar.  A hit point p of such a cell lies on one of its shapes T;
// emitter e is a flat polygon E.  For one flat shape B that touches the shaft, `bundle_misses` proves that NO shadow ray from a
// point of T to a point of E can be a hit of B for the reference's test - with B's plane oriented so that E lies on its positive
// side (E on both sides: nothing is proven), f = signed distance to that plane, sin >= h_E / d_max the least sine at which any ray
// of the bundle meets the plane:
//   A.  every point of T has f >= tau_pos: the ray starts above B's plane and climbs: it never meets the plane (in float the
//       reference's t comes out negative, or beyond tMax when d.n rounds the wrong way: tau_pos / rounding > the shaft's length);
//   B.  sin >= 0.25 (as for planar cells) and every point of T has f >= -f0, f0 = half of tMin x sin: the ray meets the plane before
//       tMin / 2, far enough below tMin = 2e-4 (scene.glsl:85) for the rounding of t;
//   C.  otherwise the part of T below -f0 is cut out (a convex polygon), and every segment from one of its corners to a corner of E
//       is cut with B's plane: the convex hull of those points holds every point where a ray of the bundle can meet the plane
//       (the crossing point is a projective image of the segment's ends, and the hull of T- and E is spanned by those segments);
//       in the plane's own 2-D frame that hull has to stay clear of B by kappa - a separating axis among the edges of both
//       polygons - which is two orders of magnitude above what rounding does to the reference's (u, v).
// Everything errs towards "not proven".  T itself is one of the B's (f = 0: case B: the emitter must be seen steeply from T's
// own plane, which is condition 2 of the planar cells).
// Where the hit point is: o + t d with the float t of T's test.  Off T's PLANE it is tol_p x cond(T) at most, but ALONG the ray t is
// wrong by about (6 + 4 cond) ulp x |ro| / |cos(d, n)|, so a hit at a grazing angle leaves the point off T sideways by any amount,
// and the float (u, v) test lets rays pass that miss T's edge by as much.  The shade stage therefore uses these proofs only after it
// has CHECKED the hit point against T (kernels/hj_light_grid_const.h; the records come from here: shape_recs): then it is within
// sigma = 5e-6 of T, in any direction, and that is what "a point of T" means below.  B's own plane is uncertain by
// tol_p x cond(B) for the reference's arithmetic, its (u, v) by uv x cond(B) in the plane.
struct Poly { double v[8][3]; int n; };

inline int clip_below(const Poly& in, const double nrm[3], double d, double off, Poly& out) {   // the part with n.x - d <= off
  out.n = 0;
  for (int i = 0; i < in.n; i++) {
    const double* a = in.v[i];
    const double* b = in.v[(i + 1) % in.n];
    const double fa = nrm[0] * a[0] + nrm[1] * a[1] + nrm[2] * a[2] - d - off, fb = nrm[0] * b[0] + nrm[1] * b[1] + nrm[2] * b[2] - d - off;
    if (fa <= 0 && out.n < 8) { for (int k = 0; k < 3; k++) out.v[out.n][k] = a[k]; out.n++; }
    if ((fa < 0) != (fb < 0) && fa != fb && out.n < 8) {
      const double t = fa / (fa - fb);
      for (int k = 0; k < 3; k++) out.v[out.n][k] = a[k] + t * (b[k] - a[k]);
      out.n++;
    }
  }
  return out.n;
}

// 2-D: is the convex hull of pts (n points) further than `margin` from the convex polygon q (m corners, in order)?
inline bool hull_clear_of_polygon(const double (*pts)[2], int n, const double (*q)[2], int m, double margin) {
  if (n == 0) return true;
  // monotone chain
  int idx[64];
  for (int i = 0; i < n; i++) idx[i] = i;
  std::sort(idx, idx + n, [&](int a, int b) { return pts[a][0] < pts[b][0] || (pts[a][0] == pts[b][0] && pts[a][1] < pts[b][1]); });
  auto crs = [&](int o, int a, int b) { return (pts[a][0] - pts[o][0]) * (pts[b][1] - pts[o][1]) - (pts[a][1] - pts[o][1]) * (pts[b][0] - pts[o][0]); };
  int hull[130], h = 0;
  for (int i = 0; i < n; i++) { while (h >= 2 && crs(hull[h - 2], hull[h - 1], idx[i]) <= 0) h--; hull[h++] = idx[i]; }
  for (int i = n - 2, lower = h + 1; i >= 0; i--) { while (h >= lower && crs(hull[h - 2], hull[h - 1], idx[i]) <= 0) h--; hull[h++] = idx[i]; }
  if (h > 1) h--;                                                   // (the first point again)
  auto separates = [&](double ax, double ay) {
    const double l = std::sqrt(ax * ax + ay * ay);
    if (!(l > 0)) return false;
    ax /= l; ay /= l;
    double a0 = INFINITY, a1 = -INFINITY, b0 = INFINITY, b1 = -INFINITY;
    for (int i = 0; i < n; i++) { const double t = pts[i][0] * ax + pts[i][1] * ay; a0 = std::min(a0, t); a1 = std::max(a1, t); }
    for (int i = 0; i < m; i++) { const double t = q[i][0] * ax + q[i][1] * ay; b0 = std::min(b0, t); b1 = std::max(b1, t); }
    return a0 > b1 + margin || b0 > a1 + margin;
  };
  for (int i = 0; i < m; i++) { const double* a = q[i]; const double* b = q[(i + 1) % m]; if (separates(-(b[1] - a[1]), b[0] - a[0])) return true; }
  for (int i = 0; i < h; i++) {
    const double* a = pts[hull[i]];
    const double* b = pts[hull[(i + 1) % h]];
    if (separates(-(b[1] - a[1]), b[0] - a[0])) return true;
  }
  return false;
}

struct BundleTol {
  double tol_p, tol_s, tau_pos, sin_cell, t_min, sigma, uv_k;
  double at() const { return sigma + tol_s; }               // how far a (checked) hit point of T can be from T
};

// B's plane as the bundle of rays towards E sees it: oriented so that E lies on its positive side, the least sine at which a ray
// of the bundle meets it, and f0 (case B's threshold) - computed once per (B, E, cell), used for every T of the cell.
struct BundleSide { double nrm[3], d, f0, own, uv; bool usable, steep; };
inline BundleSide bundle_side(const Poly& E, const Plane& bp, double dmax, const BundleTol& tl) {   // (f0 still lacks the hit point's share: per T)
  BundleSide r{};
  double fe_min = INFINITY, fe_max = -INFINITY;
  for (int i = 0; i < E.n; i++) { const double f = bp.n[0] * E.v[i][0] + bp.n[1] * E.v[i][1] + bp.n[2] * E.v[i][2] - bp.d; fe_min = std::min(fe_min, f); fe_max = std::max(fe_max, f); }
  double sgn;
  const double own = tl.tol_p * bp.cond;                                                                  // what B's plane is uncertain by
  if (fe_min > tl.tau_pos + own) sgn = 1.0; else if (fe_max < -(tl.tau_pos + own)) sgn = -1.0; else return r;   // E on both sides of (or in) B's plane: nothing is proven
  r.usable = true;
  for (int k = 0; k < 3; k++) r.nrm[k] = sgn * bp.n[k];
  r.d = sgn * bp.d;
  const double sin_min = (sgn > 0 ? fe_min : -fe_max) / dmax;
  // a hit point with f >= -(tMin / 2) sin meets the plane before tMin / 2; the points of T whose hit points all satisfy that: f >= -f0
  r.f0 = 0.5 * tl.t_min * std::min(sin_min, 1.0) - own - tl.tol_s;
  r.steep = sin_min >= tl.sin_cell && r.f0 > 0;
  r.own = own;
  r.uv = tl.uv_k * bp.cond;
  return r;
}

// true: no ray from a hit point of T (within tl.at() of it) to a point of E can be a hit of B
inline bool bundle_misses(const Poly& T, const Poly& E, const Poly& B, const BundleSide& bs, const BundleTol& tl) {
  if (!bs.usable) return false;
  const double* nrm = bs.nrm;
  const double tin = tl.at();
  const double d = bs.d, f0 = bs.f0 - tin;
  double f_min = INFINITY;                                                                                // over T itself; a hit point: tol_in lower at most
  for (int i = 0; i < T.n; i++) f_min = std::min(f_min, nrm[0] * T.v[i][0] + nrm[1] * T.v[i][1] + nrm[2] * T.v[i][2] - d);
  if (f_min - tin - bs.own >= tl.tau_pos) return true;                                                    // A
  if (!bs.steep || !(f0 > 0)) return false;
  if (f_min >= -f0) return true;                                                                          // B
  Poly below;                                                                                             // C
  if (clip_below(T, nrm, d, -f0, below) == 0) return true;
  double pts[64][2], q[8][2];
  // a 2-D frame in B's plane
  double ax[3] = {std::fabs(nrm[0]) < 0.9 ? 1.0 : 0.0, std::fabs(nrm[0]) < 0.9 ? 0.0 : 1.0, 0.0};
  double u[3] = {nrm[1] * ax[2] - nrm[2] * ax[1], nrm[2] * ax[0] - nrm[0] * ax[2], nrm[0] * ax[1] - nrm[1] * ax[0]};
  const double ul = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
  for (int k = 0; k < 3; k++) u[k] /= ul;
  const double w[3] = {nrm[1] * u[2] - nrm[2] * u[1], nrm[2] * u[0] - nrm[0] * u[2], nrm[0] * u[1] - nrm[1] * u[0]};
  int np = 0;
  for (int i = 0; i < below.n; i++) {
    const double fa = nrm[0] * below.v[i][0] + nrm[1] * below.v[i][1] + nrm[2] * below.v[i][2] - d;
    for (int j = 0; j < E.n; j++) {
      const double fb = nrm[0] * E.v[j][0] + nrm[1] * E.v[j][1] + nrm[2] * E.v[j][2] - d;
      if (!(fa < 0 && fb > 0)) return false;
      const double t = fa / (fa - fb);
      double x[3];
      for (int k = 0; k < 3; k++) x[k] = below.v[i][k] + t * (E.v[j][k] - below.v[i][k]);
      if (np >= 64) return false;
      pts[np][0] = x[0] * u[0] + x[1] * u[1] + x[2] * u[2]; pts[np][1] = x[0] * w[0] + x[1] * w[1] + x[2] * w[2];
      np++;
    }
  }
  for (int i = 0; i < B.n; i++) { q[i][0] = B.v[i][0] * u[0] + B.v[i][1] * u[1] + B.v[i][2] * u[2]; q[i][1] = B.v[i][0] * w[0] + B.v[i][1] * w[1] + B.v[i][2] * w[2]; }
  return hull_clear_of_polygon(pts, np, q, B.n, 1.5 * (tin / tl.sin_cell + bs.uv));
}

}  // namespace

bool build_light_grid(const hj_scene_desc* s, uint32_t res, LightGrid& out) {
  out = LightGrid{};
  if (!s || res < 2 || res > 256 || s->num_emitters == 0 || s->num_bvh_nodes == 0) return false;
  const Geometry g{s, s->num_spheres, s->num_quads, s->num_triangles};
  const size_t shapes = g.ns + g.nq + g.nt, N = s->num_bvh_nodes;
  const bool timing = Tuning::from_env().light_grid_timing;       // wall time of the stages on stderr (no context here: hj_debug_light_grid has none)
  auto t_last = std::chrono::steady_clock::now();
  auto mark = [&](const char* what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "light grid: %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };

  // bounds of the scene (shapes) and the scale of the coordinates (incl. the camera: camera rays start there)
  std::vector<BoxF> sb(shapes);
  Box scene = empty_box();
  {
    std::vector<BoxF> part(16, empty_boxf());
    parallel_pieces(shapes, 65536, [&](size_t a, size_t b, unsigned t) {
      BoxF acc = empty_boxf();
      for (size_t i = a; i < b; i++) { sb[i] = g.boundsf(i); joinf(acc, sb[i]); }
      part[t] = acc;
    });
    for (const BoxF& b : part) { const Box w = widen(b); join(scene, w); }
  }
  if (!finite_box(scene)) return false;
  Box all = scene;
  { const double c[3] = {s->camera.position[0], s->camera.position[1], s->camera.position[2]}; if (std::isfinite(c[0] + c[1] + c[2])) grow(all, c); }
  double scale = 0, ext = 0;
  for (int k = 0; k < 3; k++) { scale = std::max(scale, std::max(all.hi[k] - all.lo[k], std::max(std::fabs(all.lo[k]), std::fabs(all.hi[k])))); ext = std::max(ext, scene.hi[k] - scene.lo[k]); }
  if (!(ext > 0)) return false;
  // A sphere is as much fatter for the reference as its discriminant is noisy (below: 1e-6 |l|^2 / r for a shadow ray from l away):
  // its box grows by that amount at the scene's diagonal, for the subtree bounds the shafts are culled against and for the cells
  // it counts as touching (a point-sized sphere fills the scene: no cell is proven then, which is what its test deserves).
  if (g.ns != 0) {
    double diag2 = 0;
    for (int k = 0; k < 3; k++) diag2 += (scene.hi[k] - scene.lo[k]) * (scene.hi[k] - scene.lo[k]);
    for (size_t i = 0; i < g.ns; i++) {
      const double r = std::fabs((double)s->spheres[i].radius), extra = 1e-6 * diag2 / std::max(r, 1e-30);
      for (int k = 0; k < 3; k++) {
        sb[i].lo[k] = std::nextafter((float)std::max((double)sb[i].lo[k] - extra, -3.0e38), -INFINITY);
        sb[i].hi[k] = std::nextafter((float)std::min((double)sb[i].hi[k] + extra, 3.0e38), INFINITY);
      }
    }
  }
  mark("shape bounds");
  const double tol_p = 2e-6 * scale, tol_s = 2e-7 * scale, m = 1e-4 * std::max(1.0, scale);
  constexpr double kSinCell = 0.25, kSinEmitter = 0.1, kTMin = 2e-4, kEps = 1e-4;
  // the two escape arguments of the header must hold with room to spare at this scale, or there is no grid
  if (!((tol_p + tol_s) / kSinCell < 0.5 * kTMin && 2 * tol_s / kSinEmitter < 0.5 * kEps)) return false;
  double scene_diag = 0, scene_maxabs = 0;                  // (what the emitter-side argument's rounding terms scale with)
  for (int k = 0; k < 3; k++) {
    scene_diag += (scene.hi[k] - scene.lo[k]) * (scene.hi[k] - scene.lo[k]);
    scene_maxabs = std::max(scene_maxabs, std::max(std::fabs(scene.lo[k]), std::fabs(scene.hi[k])));
  }
  scene_diag = std::sqrt(scene_diag);
  // Every flat shape's plane and conditioning, once.  `tame`: the reference's float test of this shape is accurate enough to reason
  // about - its t for a ray from anywhere in the scene (|ro| <= S) meeting the plane at sin >= 0.25 is off by 4 ulp x cond x S / 0.25,
  // and that has to stay below a quarter of tMin (the escape arguments leave half of tMin).  A needle triangle is not tame: a cell or
  // a shaft that holds one proves nothing.
  constexpr double kUlp = 5.97e-8;
  const double S = scene_diag + scene_maxabs;
  const double cond_limit = std::min(16.0, 5e-5 * kSinCell / (4.0 * kUlp * S));
  constexpr double kCondPlanar = 2.0;                      // shapes of a PLANAR cell (tol_p doubles for them)
  constexpr double kSigma = 5e-6;                          // a hit point the shade stage has checked is this close to its shape
  std::vector<Plane> planes(shapes);
  parallel_pieces(shapes, 65536, [&](size_t a, size_t b, unsigned) { for (size_t i = a; i < b; i++) planes[i] = g.plane(i); });
  auto tame = [&](size_t i) { return planes[i].ok && planes[i].cond <= cond_limit; };

  // subtree bounds from the shapes: sub[i] covers every leaf with an index in [i, exit(i)).  One reverse pass; on large
  // arrays the pass is cut at the top of the tree: the ranges [a, exit(a)) of a frontier of nodes are disjoint blocks of the
  // array, each done by its own thread, then the few nodes above the frontier.  A node whose exit leaves its block - no
  // tree - ends the attempt (no grid).
  std::vector<BoxF> sub(N);
  std::atomic<bool> malformed{false};
  auto reverse_pass = [&](size_t lo_i, size_t hi_i, const std::vector<uint8_t>* skip) {     // nodes [lo_i, hi_i), highest first
    for (size_t i = hi_i; i-- > lo_i;) {
      if (skip && (*skip)[i]) continue;
      const hj_bvh_node& nd = s->bvh[i];
      BoxF b = empty_boxf();
      if (nd.shape_index != HJ_BVH_INNER) { if (nd.shape_index >= shapes) { malformed = true; return; } b = sb[nd.shape_index]; }
      const size_t e = std::min<size_t>(nd.exit_index, N);
      if (e <= i || (!skip && e > hi_i)) { malformed = true; return; }
      int chain = 0;
      for (size_t j = i + 1; j < e; j = std::min<size_t>(s->bvh[j].exit_index, N)) {
        joinf(b, sub[j]);
        if (++chain > 8) { malformed = true; return; }      // not a tree this code wants to reason about
      }
      sub[i] = b;
    }
  };
  if (N < 200000) {
    reverse_pass(0, N, nullptr);
  } else {
    // frontier: open the node with the largest range until there are enough blocks
    std::vector<size_t> frontier{0};
    std::vector<uint8_t> done(N, 0);            // 1: inside a block (its thread computed it); 0: above the frontier
    while (frontier.size() < 64) {
      size_t best = 0, best_len = 0;
      for (size_t f = 0; f < frontier.size(); f++) {
        const size_t a = frontier[f], len = std::min<size_t>(s->bvh[a].exit_index, N) - a;
        if (s->bvh[a].shape_index == HJ_BVH_INNER && len > best_len) { best = f; best_len = len; }
      }
      if (best_len < 4096) break;
      const size_t a = frontier[best], e = std::min<size_t>(s->bvh[a].exit_index, N);
      frontier.erase(frontier.begin() + (long)best);
      int chain = 0;
      for (size_t j = a + 1; j < e; j = std::min<size_t>(s->bvh[j].exit_index, N)) {
        if (std::min<size_t>(s->bvh[j].exit_index, N) <= j || ++chain > 8) return false;
        frontier.push_back(j);
      }
    }
    std::atomic<size_t> next_block{0};
    parallel_pieces(frontier.size(), 1, [&](size_t, size_t, unsigned) {
      for (;;) {
        const size_t f = next_block.fetch_add(1);
        if (f >= frontier.size()) break;
        const size_t a = frontier[f], e = std::min<size_t>(s->bvh[a].exit_index, N);
        if (e <= a) { malformed = true; break; }
        reverse_pass(a, e, nullptr);
      }
    });
    if (malformed) return false;
    for (size_t a : frontier) for (size_t i = a, e = std::min<size_t>(s->bvh[a].exit_index, N); i < e; i++) done[i] = 1;
    // whatever is left (the opened nodes, and anything a strange array keeps outside every block), highest first
    reverse_pass(0, N, &done);
  }
  if (malformed) return false;

  mark("subtree bounds");
  // the grid over the padded scene bounds
  const Box gb = pad(scene, 4 * m);
  double cell[3];
  for (int k = 0; k < 3; k++) cell[k] = (gb.hi[k] - gb.lo[k]) / res;
  out.res = res;
  for (int k = 0; k < 3; k++) { out.lo[k] = (float)gb.lo[k]; out.inv[k] = (float)(1.0 / cell[k]); }
  // (the kernel computes the cell index in float from out.lo / out.inv: a cell's guarantee covers the padded cell, and the
  // padding - m on every side, plus the shapes' own - is orders of magnitude above what that rounding can move a point)
  const size_t ncell = (size_t)res * res * res;
  out.bits.assign(ncell, 0);

  // rasterise the shapes: per cell the first shape seen (in shape order), and whether all of them share its plane.  The cell
  // range of every shape first (parallel over shapes), then every thread owns a slab of z layers and takes, in shape order,
  // the shapes that reach into it: the same grid whatever the thread count.
  std::vector<uint32_t> first(ncell, 0xFFFFFFFFu);
  std::vector<uint8_t> bad(ncell, 0);
  // cells on meshes (HJ_LIGHT_GRID_MESH): a cell that is not planar can still be proven when everything in it is FLAT - `hard` marks
  // the cells that hold a sphere or a shape without a usable plane; `flat_ok` per shape.  Scenes of more than 200 000 shapes: not tried.
  const bool mesh_cells = Tuning::from_env().light_grid_mesh != 0 && shapes <= 200000;
  std::vector<uint8_t> hard(mesh_cells ? ncell : 0, 0), flat_ok(mesh_cells ? shapes : 0, 0);
  struct Flat { Plane pl; Poly poly; };
  std::vector<Flat> flat(mesh_cells ? shapes : 0);                 // plane and corners of every flat shape, once
  if (mesh_cells) parallel_pieces(shapes, 65536, [&](size_t a, size_t b, unsigned) {
    for (size_t i = a; i < b; i++) {
      flat[i].pl = planes[i];
      flat[i].poly.n = g.polygon(i, flat[i].poly.v);
      // (the shade stage evaluates n.(p - a) in float and adds what that can lose - 5 ulp of |p - a| - before it divides by |cos(d, n)|:
      // far from a on a wide shape only steep hits pass, whatever the shape)
      flat_ok[i] = tame(i) && flat[i].poly.n >= 3 ? 1 : 0;
    }
  });
  auto cell_range = [&](const Box& b, int lo[3], int hi[3]) {
    for (int k = 0; k < 3; k++) {
      lo[k] = std::max(0, (int)std::floor((b.lo[k] - gb.lo[k]) / cell[k]));
      hi[k] = std::min((int)res - 1, (int)std::floor((b.hi[k] - gb.lo[k]) / cell[k]));
    }
  };
  struct Range { uint8_t lo[3], hi[3]; };
  std::vector<Range> range(shapes);
  parallel_pieces(shapes, 65536, [&](size_t a, size_t b, unsigned) {
    for (size_t i = a; i < b; i++) {
      int lo[3], hi[3];
      cell_range(pad(widen(sb[i]), 2 * m), lo, hi);             // (shape and cell both padded by m)
      bool none = false;                                        // (a shape outside the grid: every shape lies inside today)
      for (int k = 0; k < 3; k++) { range[i].lo[k] = (uint8_t)lo[k]; range[i].hi[k] = (uint8_t)std::max(hi[k], 0); none = none || hi[k] < lo[k]; }
      if (none) { range[i].lo[2] = 255; range[i].hi[2] = 0; }   // the empty range, set AFTER the loop (k = 2 would overwrite it)
    }
  });
  parallel_pieces(res, 1, [&](size_t z0, size_t z1, unsigned) {
    for (size_t i = 0; i < shapes; i++) {
      const Range& r = range[i];
      if (r.lo[2] > r.hi[2] || r.hi[2] < z0 || r.lo[2] >= z1) continue;
      const bool planar_ok = planes[i].ok && planes[i].cond <= kCondPlanar && planes[i].cond <= cond_limit;
      for (size_t z = std::max<size_t>(r.lo[2], z0); z <= r.hi[2] && z < z1; z++) for (int y = r.lo[1]; y <= r.hi[1]; y++) for (int x = r.lo[0]; x <= r.hi[0]; x++) {
        const size_t c = (z * res + (size_t)y) * res + (size_t)x;
        if (mesh_cells && !flat_ok[i]) hard[c] = 1;
        if (bad[c]) continue;
        if (first[c] == 0xFFFFFFFFu) {
          first[c] = (uint32_t)i;
          if (!planar_ok) bad[c] = 1;
        } else if (!planar_ok || !g.coplanar(i, planes[first[c]], tol_s)) bad[c] = 1;
      }
    }
  });
  mark("planes + rasterisation");
  // emitters (bit e for e < 8)
  struct Em { size_t shape; Plane q; Box box; bool ok; double slack, sin_min; };
  std::vector<Em> ems;
  for (size_t e = 0; e < std::min<size_t>(s->num_emitters, 8); e++) {
    Em em{};
    em.shape = s->emitters[e].shape;
    em.ok = em.shape < shapes;
    if (em.ok) { em.q = planes[em.shape]; em.ok = tame(em.shape); em.box = pad(widen(sb[em.shape]), m); }
    if (em.ok) {
      // How steeply the emitter has to be seen: shapes "in Q" (the emitter, its neighbour triangle) are met at
      // t >= dist - slack / sin, and that has to stay behind tMax = dist - 1e-4 with room for the rounding of either side
      // (a quarter of eps for the geometry, the rest for the arithmetic: about 1e-5 at sin = 0.03).  slack = how far such shapes
      // really are from Q (measured; exactly 0 for an axis-aligned light) + how far a sampled point can be (4 ulp of the
      // emitter's largest coordinate).  0.1 as before when the measured figures give nothing better.
      double dev = 0.0, big = 0.0, cond_q = em.q.cond;       // (cond_q: the worst-conditioned shape in Q)
      for (size_t i = g.ns; i < shapes; i++) {
        if (!g.coplanar(i, em.q, tol_s)) continue;
        if (!tame(i)) { em.ok = false; break; }              // (a needle in the emitter's plane: its t is anybody's guess)
        cond_q = std::max(cond_q, planes[i].cond);
        double v[4][3];
        const int nv = g.vertices(i, v);
        for (int c = 0; c < nv; c++) dev = std::max(dev, std::fabs(em.q.n[0] * v[c][0] + em.q.n[1] * v[c][1] + em.q.n[2] * v[c][2] - em.q.d));
      }
      { double v[4][3]; const int nv = g.vertices(em.shape, v); for (int c = 0; c < nv; c++) for (int k = 0; k < 3; k++) big = std::max(big, std::fabs(v[c][k])); }
      em.slack = dev + 4.0 * 5.97e-8 * big;
      // ... and the ARITHMETIC's share grows with the scene: the reference's t = (1 / (d.n)) * (-(n.ro)) for a shape in Q loses, to the
      // cancellation in both dot products, about 6 ulp x |ro| / sin, and the rounding of ro = o - a another ulp of the largest
      // coordinate / sin (|ro| <= the scene's diagonal).  Half of eps for that, a quarter for the geometry: the view has to be
      // steeper in a larger scene, and where even 0.1 does not keep the sum inside eps this emitter proves nothing.
      // (of the 6 ulp, 4 are the normal's own cancellation: they grow with the conditioning of the shapes in Q)
      const double arith = 5.97e-8 * ((2.0 + 4.0 * cond_q) * scene_diag + 2.0 * std::max(big, scene_maxabs));
      em.sin_min = std::min(kSinEmitter, std::max(0.03, std::max(em.slack / (0.25 * kEps), arith / (0.5 * kEps))));
      if (!(em.ok && em.slack / em.sin_min <= 0.25 * kEps && arith / em.sin_min <= 0.5 * kEps)) em.ok = false;
    }
    ems.push_back(em);
  }

  std::vector<uint32_t> work;                              // cells that hold a planar surface
  for (size_t c = 0; c < ncell; c++) { if (first[c] != 0xFFFFFFFFu) out.cells_surface++; if (first[c] != 0xFFFFFFFFu && !bad[c]) work.push_back((uint32_t)c); }
  out.cells_planar = work.size();
  // mesh cells and their shapes (in shape order: the same lists whatever the thread count)
  const size_t planar_cells = work.size();
  std::vector<uint32_t> cell_first(mesh_cells ? ncell + 1 : 0, 0), cell_shapes;
  if (mesh_cells) {
    for (size_t c = 0; c < ncell; c++) if (first[c] != 0xFFFFFFFFu && bad[c] && !hard[c]) work.push_back((uint32_t)c);
    auto each_cell = [&](size_t i, auto fn) {
      const Range& r = range[i];
      if (r.lo[2] > r.hi[2]) return;
      for (size_t z = r.lo[2]; z <= r.hi[2]; z++) for (int y = r.lo[1]; y <= r.hi[1]; y++) for (int x = r.lo[0]; x <= r.hi[0]; x++) {
        const size_t c = (z * res + (size_t)y) * res + (size_t)x;
        if (first[c] != 0xFFFFFFFFu && bad[c] && !hard[c]) fn(c);
      }
    };
    for (size_t i = 0; i < shapes; i++) each_cell(i, [&](size_t c) { cell_first[c + 1]++; });
    for (size_t c = 0; c < ncell; c++) cell_first[c + 1] += cell_first[c];
    cell_shapes.resize(cell_first[ncell]);
    std::vector<uint32_t> fill(cell_first.begin(), cell_first.end() - 1);
    for (size_t i = 0; i < shapes; i++) each_cell(i, [&](size_t c) { cell_shapes[fill[c]++] = (uint32_t)i; });
  }
  // (the margin of case C is set per pair, in bundle_misses: 1.5 x (the hit point's slide / 0.25 + uv_k x cond(B)))
  const BundleTol btol{tol_p, tol_s, 1e-5 * std::max(1.0, scale), kSinCell, kTMin, kSigma, 16.0 * kUlp * S / kSinCell};
  if (mesh_cells) out.mesh_bits.assign(ncell, 0);

  std::atomic<size_t> next{0}, clear{0};
  auto run = [&] {
    for (;;) {
      const size_t w = next.fetch_add(64);
      if (w >= work.size()) break;
      for (size_t wi = w; wi < std::min(work.size(), w + 64); wi++) {
        const size_t c = work[wi];
        const int x = (int)(c % res), y = (int)((c / res) % res), z = (int)(c / ((size_t)res * res));
        Box cb;
        const int xyz[3] = {x, y, z};
        for (int k = 0; k < 3; k++) { cb.lo[k] = gb.lo[k] + xyz[k] * cell[k] - m; cb.hi[k] = gb.lo[k] + (xyz[k] + 1) * cell[k] + m; }
        const bool on_mesh = wi >= planar_cells;

        const Plane P = on_mesh ? Plane{} : planes[first[c]];
        const double tol_pc = kCondPlanar * tol_p;       // (a hit point on a shape of a planar cell)
        uint8_t bits = 0;
        for (size_t e = 0; e < ems.size() && on_mesh; e++) {
          // A cell on a mesh: the emitter side as for a planar cell; then every flat shape that touches the shaft - the cell's own
          // shapes among them - must be missed by the whole bundle of rays from EVERY shape of the cell to the emitter (bundle_misses)
          const Em& em = ems[e];
          if (!em.ok) continue;
          double dmax = 0;
          { double d2 = 0; for (int k = 0; k < 3; k++) { const double a = std::max(std::fabs(em.box.hi[k] - cb.lo[k]), std::fabs(cb.hi[k] - em.box.lo[k])); d2 += a * a; } dmax = std::sqrt(d2); }
          double lo_d = -em.q.d, hi_d = -em.q.d;
          for (int k = 0; k < 3; k++) { lo_d += em.q.n[k] * (em.q.n[k] > 0 ? cb.lo[k] : cb.hi[k]); hi_d += em.q.n[k] * (em.q.n[k] > 0 ? cb.hi[k] : cb.lo[k]); }
          const double near_q = lo_d > 0 ? lo_d : (hi_d < 0 ? -hi_d : 0.0);
          if (!(near_q - std::max(tol_s, em.slack) >= em.sin_min * dmax)) continue;
          const Poly& E = flat[em.shape].poly;
          if (E.n < 3) continue;
          bool blocked = false;
          // the cell's own shapes first (they touch the shaft by construction and fail most often: a facet that faces away from the
          // light, a light seen at a grazing angle): most unprovable cells end here, before the tree is walked
          for (uint32_t b = cell_first[c]; b < cell_first[c + 1] && !blocked; b++) {
            const uint32_t shp = cell_shapes[b];
            if (g.coplanar(shp, em.q, tol_s)) continue;
            const BundleSide bs = bundle_side(E, flat[shp].pl, dmax, btol);
            for (uint32_t k = cell_first[c]; k < cell_first[c + 1] && !blocked; k++)
              if (!bundle_misses(flat[cell_shapes[k]].poly, E, flat[shp].poly, bs, btol)) blocked = true;
          }
          if (blocked) continue;
          const Shaft sh(cb, em.box);
          for (size_t i = 0; i < N && !blocked;) {
            const hj_bvh_node& nd = s->bvh[i];
            if (sh.outside(pad(widen(sub[i]), m))) { i = std::min<size_t>(nd.exit_index, N); continue; }
            if (nd.shape_index != HJ_BVH_INNER) {
              const size_t shp = nd.shape_index;
              if (shp < g.ns) { blocked = true; break; }                         // (a sphere near a mesh cell's shaft: not tried)
              if (!g.coplanar(shp, em.q, tol_s)) {                               // (shapes in the emitter's plane: the emitter-side argument)
                if (!flat_ok[shp]) { blocked = true; break; }
                // (the cell's shapes as they are: cut to the cell they prove 5 % more cells and no more rays)
                const BundleSide bs = bundle_side(E, flat[shp].pl, dmax, btol);
                for (uint32_t k = cell_first[c]; k < cell_first[c + 1] && !blocked; k++)
                  if (!bundle_misses(flat[cell_shapes[k]].poly, E, flat[shp].poly, bs, btol)) blocked = true;
              }
            }
            i++;
          }
          if (!blocked) { bits |= (uint8_t)(1u << e); clear.fetch_add(1, std::memory_order_relaxed); }
        }
        for (size_t e = 0; e < ems.size() && !on_mesh; e++) {
          const Em& em = ems[e];
          if (!em.ok) continue;
          // 2. angles: the emitter's padded box on one side of P, steeply; the padded cell on one side of Q, not grazing
          double dmax = 0;                                 // the longest segment of the shaft
          { double d2 = 0; for (int k = 0; k < 3; k++) { const double a = std::max(std::fabs(em.box.hi[k] - cb.lo[k]), std::fabs(cb.hi[k] - em.box.lo[k])); d2 += a * a; } dmax = std::sqrt(d2); }
          auto side_dist = [&](const Plane& pl, const Box& b, double& lo_d, double& hi_d) {   // range of the signed distance over the box
            lo_d = -pl.d; hi_d = -pl.d;
            for (int k = 0; k < 3; k++) { lo_d += pl.n[k] * (pl.n[k] > 0 ? b.lo[k] : b.hi[k]); hi_d += pl.n[k] * (pl.n[k] > 0 ? b.hi[k] : b.lo[k]); }
          };
          double lo_d, hi_d;
          side_dist(P, em.box, lo_d, hi_d);
          const double near_p = lo_d > 0 ? lo_d : (hi_d < 0 ? -hi_d : 0.0);
          if (!(near_p - tol_pc >= kSinCell * dmax)) continue;
          // (the hit point is within tol_p of P, so its distance to Q is what the part of the cell near P has: the cell's box is a superset)
          side_dist(em.q, cb, lo_d, hi_d);
          const double near_q = lo_d > 0 ? lo_d : (hi_d < 0 ? -hi_d : 0.0);
          if (!(near_q - std::max(tol_s, em.slack) >= em.sin_min * dmax)) continue;
          // 3. the shaft holds nothing but shapes in P and shapes in Q
          const Shaft sh(cb, em.box);
          bool blocked = false;
          for (size_t i = 0; i < N && !blocked;) {
            const hj_bvh_node& nd = s->bvh[i];
            if (sh.outside(pad(widen(sub[i]), m))) { i = std::min<size_t>(nd.exit_index, N); continue; }
            if (nd.shape_index != HJ_BVH_INNER) {
              const size_t shp = nd.shape_index;
              if (shp < g.ns) {                              // a sphere: its ball, not its box, has to touch the shaft
                const hj_sphere& sp = s->spheres[shp];
                const double c[3] = {sp.center[0], sp.center[1], sp.center[2]};
                // (sphere.glsl:18-41 forms b^2 - 4 c of terms of size |l|^2 in float: it takes rays for hits that pass the sphere
                // at up to some 1e-7 |l|^2 / r - l = origin - centre, no longer than the shaft - so a small, distant sphere is as
                // much fatter for the reference as its discriminant is noisy; shadow directions are normalised afresh, |d| = 1)
                const double r = std::fabs((double)sp.radius);
                if (!sh.outside_ball(c, r + m + 1e-6 * dmax * dmax / std::max(r, 1e-30))) blocked = true;
              } else if (!((g.coplanar(shp, P, tol_s) && tame(shp)) || g.coplanar(shp, em.q, tol_s))) blocked = true;   // (in Q: all tame, or the emitter is not ok)
            }
            i++;
          }
          if (!blocked) { bits |= (uint8_t)(1u << e); clear.fetch_add(1, std::memory_order_relaxed); }
        }
        if (on_mesh) out.mesh_bits[c] = bits; else out.bits[c] = bits;
      }
    }
  };
  unsigned nthreads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  if (work.size() < 512) nthreads = 1;
  std::vector<std::thread> pool;
  try { for (unsigned t = 1; t < nthreads; t++) pool.emplace_back(run); } catch (const std::exception&) {}
  run();
  for (auto& t : pool) t.join();
  mark("shafts");
  if (timing) std::fprintf(stderr, "light grid: %zu planar cells, %zu cells on meshes\n", planar_cells, work.size() - planar_cells);
  out.pairs_clear = clear.load();
  if (out.pairs_clear == 0) { out = LightGrid{}; return false; }
  bool any_mesh = false;
  for (uint8_t b : out.mesh_bits) if (b) { any_mesh = true; break; }
  if (!any_mesh) out.mesh_bits.clear();
  else {
    // what the shade stage checks a hit point with (kernels/hj_light_grid_const.h): n, delta; a, kind.  delta: the float (u, v) of a
    // ray from anywhere (|ro| <= the diagonal of scene and camera) that meets the shape at sin >= kLightGridSinIn are off by
    // 7 ulp x cond x |ro| / (sin x edge) at most (the cross product ro x d, the dot products, the normal's own cancellation); twice
    // that (1 - u - v carries both errors) with a factor 1.5.  A shape that is not tame keeps an all-zero record.
    double dall = 0;
    for (int k = 0; k < 3; k++) dall += (all.hi[k] - all.lo[k]) * (all.hi[k] - all.lo[k]);
    dall = std::sqrt(dall);
    out.shape_recs.assign(8 * (shapes - g.ns), 0.0f);
    for (size_t i = g.ns; i < shapes; i++) {
      if (!flat_ok[i]) continue;
      double v[4][3];
      const int nv = g.vertices(i, v);
      double e1 = 0, e2 = 0;
      for (int k = 0; k < 3; k++) { e1 += (v[1][k] - v[0][k]) * (v[1][k] - v[0][k]); e2 += (v[2][k] - v[0][k]) * (v[2][k] - v[0][k]); }
      const double edge = std::sqrt(std::min(e1, e2));
      if (!(edge > 0)) continue;
      const double delta = 2.0 * 1.5 * 7.0 * kUlp * planes[i].cond * dall / ((double)hj::kLightGridSinIn * edge);
      float* r = &out.shape_recs[8 * (i - g.ns)];
      for (int k = 0; k < 3; k++) { r[k] = (float)planes[i].n[k]; r[4 + k] = (float)v[0][k]; }
      r[3] = std::nextafter((float)delta, INFINITY);
      r[7] = nv == 4 ? 1.0f : 0.0f;
    }
  }
  return true;
}

}  // namespace hjapi

// Test entries (no GPU needed): the grid hj_scene_upload would build for `s`.  bits: res^3 bytes (may be null: sizes only).
namespace {
int debug_grid(const hj_scene_desc* s, uint32_t res, hjapi::LightGrid& g) {
  // build_light_grid trusts what hj_scene_upload has validated; this entry takes a desc nobody has looked at: the same range checks
  // first (a desc that fails them has no grid)
  if (!s || !s->bvh || s->num_bvh_nodes == 0) return 0;
  const size_t shapes = s->num_spheres + s->num_quads + s->num_triangles;
  if ((s->num_spheres && !s->spheres) || (s->num_quads && !s->quads) || (s->num_triangles && !s->triangles) ||
      (s->num_vertices && !s->vertices) || (s->num_emitters && !s->emitters) || (shapes && !s->materials) || s->num_materials != shapes) return 0;
  for (size_t i = 0; i < s->num_triangles; i++)
    for (int c = 0; c < 3; c++) if (s->triangles[i].v[c] >= s->num_vertices) return 0;
  for (size_t i = 0; i < s->num_bvh_nodes; i++)
    if (s->bvh[i].shape_index != HJ_BVH_INNER && s->bvh[i].shape_index >= shapes) return 0;
  for (size_t e = 0; e < s->num_emitters; e++) if (s->emitters[e].shape >= shapes) return 0;
  return hjapi::build_light_grid(s, res, g) ? (int)g.res : 0;
}
void debug_grid_out(const hjapi::LightGrid& g, float lo[3], float inv[3], uint64_t stats[3]) {
  for (int k = 0; k < 3; k++) { if (lo) lo[k] = g.lo[k]; if (inv) inv[k] = g.inv[k]; }
  if (stats) { stats[0] = g.cells_surface; stats[1] = g.cells_planar; stats[2] = g.pairs_clear; }
}
}  // namespace

// bits = planar bits | mesh bits (the latter hold for hits that were not grazing only: hj_debug_light_grid_planes tells them apart)
extern "C" __attribute__((visibility("default"))) int hj_debug_light_grid(const hj_scene_desc* s, uint32_t res, uint8_t* bits, float lo[3],
                                                                           float inv[3], uint64_t stats[3]) {
  hjapi::LightGrid g;
  const int r = debug_grid(s, res, g);
  if (r == 0) return 0;
  if (bits) for (size_t i = 0; i < g.bits.size(); i++) bits[i] = (uint8_t)(g.bits[i] | (g.mesh_bits.empty() ? 0 : g.mesh_bits[i]));
  debug_grid_out(g, lo, inv, stats);
  return r;
}

// planar: res^3 bytes, proofs that hold for every hit point of the cell; mesh: res^3 bytes, bundle proofs (cells on meshes and in
// corners: for a hit point the shade stage has checked against its shape); recs: 8 floats per quad and triangle, what it checks with
// (n, delta; a, kind: kernels/hj_light_grid_const.h); limits: {kLightGridSinIn, kLightGridSlide}.  Any output may be null.
extern "C" __attribute__((visibility("default"))) int hj_debug_light_grid_planes(const hj_scene_desc* s, uint32_t res, uint8_t* planar, uint8_t* mesh,
                                                                                  float* recs, float limits[2], float lo[3], float inv[3],
                                                                                  uint64_t stats[3]) {
  hjapi::LightGrid g;
  const int r = debug_grid(s, res, g);
  if (r == 0) return 0;
  if (planar) std::memcpy(planar, g.bits.data(), g.bits.size());
  if (mesh) { if (g.mesh_bits.empty()) std::memset(mesh, 0, g.bits.size()); else std::memcpy(mesh, g.mesh_bits.data(), g.mesh_bits.size()); }
  if (recs) {
    const size_t n = 8 * (s->num_quads + s->num_triangles);
    if (g.shape_recs.empty()) std::memset(recs, 0, n * sizeof(float)); else std::memcpy(recs, g.shape_recs.data(), n * sizeof(float));
  }
  if (limits) { limits[0] = hj::kLightGridSinIn; limits[1] = hj::kLightGridSlide; }
  debug_grid_out(g, lo, inv, stats);
  return r;
}
