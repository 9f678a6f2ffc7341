// Synthetic Cornell-box-shaped bench scenes, built through the Scene API.
//
// The reference ships `scenes/cbox/cbox.obj` (GPL data, not copied).  These
// generators reproduce its *shape*: same box dimensions, light, camera,
// materials and triangle count (facts listed in SURVEY.md Appendix E, taken
// from scenes/cbox/cbox.{obj,mtl} and src/main.rs:417-425), with a
// procedurally tessellated smooth object standing in for the teapot.
#include <cmath>
#include <stdexcept>

#include "scene.hpp"
#include "../../../include/hijiki_host.h"

namespace hijiki {
namespace {

constexpr double kPi = 3.14159265358979323846;

int add_diffuse(Scene& s, float r, float g, float b) {
  Material m{};
  m.tag = HJ_MAT_DIFFUSE;
  m.diffuse = hj_diffuse{{r, g, b}, 0.f};
  s.materials.push_back(m);
  return (int)s.materials.size() - 1;
}
int add_emissive(Scene& s, float r, float g, float b) {
  Material m{};
  m.tag = HJ_MAT_EMISSIVE;
  m.emissive = hj_emissive{{r, g, b}, 0.f};
  s.materials.push_back(m);
  return (int)s.materials.size() - 1;
}

// One wall = one quad as 2 triangles, one flat vertex normal, uv (0,1) on
// every vertex as in the OBJ (Appendix E).
void add_wall(Scene& s, const float p[4][3], const float n[3], int material) {
  uint32_t base = (uint32_t)s.vertices.size();
  for (int i = 0; i < 4; i++) {
    hj_vertex v{};
    for (int k = 0; k < 3; k++) v.pos[k] = p[i][k], v.normal[k] = n[k];
    v.u = 0.f;
    v.v = 1.f;
    s.vertices.push_back(v);
  }
  Shape t{};
  t.kind = ShapeKind::Triangle;
  t.tri = hj_triangle{{base, base + 1, base + 2}};
  s.objects.emplace_back(t, material);
  t.tri = hj_triangle{{base, base + 2, base + 3}};
  s.objects.emplace_back(t, material);
}

// Closed lat-long surface with nu longitudes and nv latitude bands:
// 2*nu*(nv-1) triangles (two pole fans + nv-2 quad bands), smooth
// area-weighted vertex normals.  radius(theta,phi) shapes the blob.
template <class RadiusFn>
void add_blob(Scene& s, uint32_t nu, uint32_t nv, const double c[3], const double r[3], RadiusFn radius, int material) {
  const uint32_t base = (uint32_t)s.vertices.size();
  auto point = [&](double theta, double phi, float out[3]) {
    double k = radius(theta, phi);
    out[0] = (float)(c[0] + r[0] * k * std::sin(theta) * std::cos(phi));
    out[1] = (float)(c[1] + r[1] * k * std::cos(theta));
    out[2] = (float)(c[2] + r[2] * k * std::sin(theta) * std::sin(phi));
  };
  // vertex 0 = north pole, then (nv-1) rings of nu, last = south pole
  hj_vertex v{};
  point(0.0, 0.0, v.pos);
  s.vertices.push_back(v);
  for (uint32_t j = 1; j < nv; j++)
    for (uint32_t i = 0; i < nu; i++) {
      point(kPi * j / nv, 2.0 * kPi * i / nu, v.pos);
      s.vertices.push_back(v);
    }
  point(kPi, 0.0, v.pos);
  s.vertices.push_back(v);
  const uint32_t south = base + 1 + (nv - 1) * nu;
  auto ring = [&](uint32_t j, uint32_t i) { return base + 1 + (j - 1) * nu + (i % nu); };

  std::vector<hj_triangle> tris;
  tris.reserve((size_t)2 * nu * (nv - 1));
  for (uint32_t i = 0; i < nu; i++) tris.push_back({{base, ring(1, i + 1), ring(1, i)}});
  for (uint32_t j = 1; j + 1 < nv; j++)
    for (uint32_t i = 0; i < nu; i++) {
      tris.push_back({{ring(j, i), ring(j, i + 1), ring(j + 1, i + 1)}});
      tris.push_back({{ring(j, i), ring(j + 1, i + 1), ring(j + 1, i)}});
    }
  for (uint32_t i = 0; i < nu; i++) tris.push_back({{south, ring(nv - 1, i), ring(nv - 1, i + 1)}});

  // smooth normals: area-weighted face normals, accumulated in double
  std::vector<double> acc((size_t)(south - base + 1) * 3, 0.0);
  for (const auto& t : tris) {
    const float* a = s.vertices[t.v[0]].pos;
    const float* b = s.vertices[t.v[1]].pos;
    const float* cc = s.vertices[t.v[2]].pos;
    double e1[3], e2[3];
    for (int k = 0; k < 3; k++) e1[k] = (double)b[k] - a[k], e2[k] = (double)cc[k] - a[k];
    double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    for (int q = 0; q < 3; q++)
      for (int k = 0; k < 3; k++) acc[(size_t)(t.v[q] - base) * 3 + k] += n[k];
  }
  for (uint32_t vi = base; vi <= south; vi++) {
    double* n = &acc[(size_t)(vi - base) * 3];
    double len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    if (!(len > 0)) n[1] = 1.0, len = 1.0;
    for (int k = 0; k < 3; k++) s.vertices[vi].normal[k] = (float)(n[k] / len);
  }
  Shape sh{};
  sh.kind = ShapeKind::Triangle;
  for (const auto& t : tris) {
    sh.tri = t;
    s.objects.emplace_back(sh, material);
  }
}

// smallest (nu, nv) with 2*nu*(nv-1) == target and nu ~ 2*nv, else the closest below
void blob_resolution(uint32_t target, uint32_t& nu, uint32_t& nv) {
  if (target < 16) throw std::runtime_error("mesh_triangles must be >= 16");
  uint32_t best_nu = 4, best_nv = 3;
  uint64_t best_err = ~0ull;
  uint32_t guess = (uint32_t)std::sqrt((double)target / 4.0);  // nv ~ sqrt(T/4)
  uint32_t lo = guess > 40 ? guess - 40 : 3, hi = guess + 40;
  for (uint32_t v = lo; v <= hi; v++) {
    if (v < 3) continue;
    uint32_t u = target / (2 * (v - 1));
    if (u < 3) continue;
    uint64_t got = (uint64_t)2 * u * (v - 1);
    uint64_t err = (target - got) * 1000 + (uint64_t)std::llabs((long long)u - 2ll * v);
    if (err < best_err) best_err = err, best_nu = u, best_nv = v;
  }
  nu = best_nu;
  nv = best_nv;
}

}  // namespace

Scene make_synthetic(int kind, uint32_t mesh_triangles, uint32_t gen_seed) {
  Scene s;
  // camera of Scene::from_obj (src/main.rs:417-425)
  const float angle = -1.45f * (float)(kPi / 180.0);
  s.camera = hj_camera{};
  s.camera.position[0] = 0.f, s.camera.position[1] = 0.91f, s.camera.position[2] = 5.41f, s.camera.position[3] = 0.f;
  s.camera.rotation[0] = std::sin(0.5f * angle), s.camera.rotation[1] = 0.f, s.camera.rotation[2] = 0.f;
  s.camera.rotation[3] = std::cos(0.5f * angle);
  s.camera.fov = 27.7f;

  // material order of cbox.mtl: floor, light, porcelain, wall_blue, wall_gray, wall_red
  const int floor_m = add_diffuse(s, 0.455928f, 0.446495f, 0.427629f);
  const int light_m = add_emissive(s, 15.f, 15.f, 15.f);
  const int object_m = add_diffuse(s, 1.f, 0.979146f, 0.937447f);
  const int blue_m = add_diffuse(s, 0.161f, 0.133f, 0.427f);
  const int gray_m = add_diffuse(s, 0.725f, 0.71f, 0.68f);
  const int red_m = add_diffuse(s, 0.63f, 0.065f, 0.05f);

  // object first (as in the OBJ), then the walls
  uint32_t target = mesh_triangles ? mesh_triangles : 6320u;
  uint32_t nu, nv;
  blob_resolution(target, nu, nv);
  if (kind == HJH_SYNTH_CBOX_MESH) {
    // bumpy closed mesh ("dragon-like" multi-octave displacement), centred in the box
    const double c[3] = {0.0, 0.62, -0.1}, r[3] = {0.55, 0.55, 0.55};
    const double ph = 0.37 * (double)(gen_seed % 1000u);
    add_blob(s, nu, nv, c, r,
             [&](double th, double p) {
               double k = 1.0;
               k += 0.18 * std::sin(3.0 * p + ph) * std::sin(2.0 * th);
               k += 0.08 * std::sin(9.0 * p + 1.3 * ph) * std::sin(7.0 * th + 0.5);
               k += 0.035 * std::sin(27.0 * p + 2.1) * std::sin(23.0 * th + ph);
               k += 0.012 * std::sin(81.0 * p + 0.7 * ph) * std::sin(79.0 * th + 1.1);
               return k;
             },
             object_m);
  } else {
    // smooth blob inside the teapot's bounding box [-0.6,0,-0.4]..[0.687,0.63,0.4]
    const double c[3] = {0.0434, 0.315, 0.0}, r[3] = {0.55, 0.30, 0.34};
    const double ph = 0.37 * (double)(gen_seed % 1000u);
    add_blob(s, nu, nv, c, r,
             [&](double th, double p) { return 1.0 + 0.12 * std::sin(3.0 * p + ph) * std::sin(2.0 * th) * std::sin(th); },
             object_m);
  }

  const float x0 = -1.f, x1 = 1.f, y0 = 0.f, y1 = 1.59f, z0 = -1.04f, z1 = 0.99f;
  {
    const float p[4][3] = {{x1, y0, z1}, {x1, y0, z0}, {x1, y1, z0}, {x1, y1, z1}};
    const float n[3] = {-1, 0, 0};
    add_wall(s, p, n, blue_m);  // rightWall
  }
  {
    const float p[4][3] = {{x0, y0, z0}, {x0, y0, z1}, {x0, y1, z1}, {x0, y1, z0}};
    const float n[3] = {1, 0, 0};
    add_wall(s, p, n, red_m);  // leftWall
  }
  {
    const float p[4][3] = {{-0.24f, 1.58f, -0.22f}, {0.23f, 1.58f, -0.22f}, {0.23f, 1.58f, 0.16f}, {-0.24f, 1.58f, 0.16f}};
    const float n[3] = {0, -1, 0};
    add_wall(s, p, n, light_m);  // light
  }
  {
    const float p[4][3] = {{x0, y0, z0}, {x0, y1, z0}, {x1, y1, z0}, {x1, y0, z0}};
    const float n[3] = {0, 0, 1};
    add_wall(s, p, n, gray_m);  // backWall
  }
  {
    const float p[4][3] = {{x0, y0, z1}, {x0, y0, z0}, {x1, y0, z0}, {x1, y0, z1}};
    const float n[3] = {0, 1, 0};
    add_wall(s, p, n, floor_m);  // floor
  }
  {
    const float p[4][3] = {{x0, y1, z0}, {x0, y1, z1}, {x1, y1, z1}, {x1, y1, z0}};
    const float n[3] = {0, -1, 0};
    add_wall(s, p, n, gray_m);  // ceiling
  }

  if (kind == HJH_SYNTH_CBOX_SPHERES || kind == HJH_SYNTH_CBOX_CBOARD) {
    // sphere positions of --put-cbox-spheres (src/main.rs:1473-1482)
    Material mirror{};
    mirror.tag = HJ_MAT_MIRROR;
    s.materials.push_back(mirror);
    Material second{};
    if (kind == HJH_SYNTH_CBOX_SPHERES) {
      second.tag = HJ_MAT_DIELECTRIC;  // DielectricMaterial::clear(1.5), src/main.rs:129-133
      second.dielectric = hj_dielectric{{0.f, 0.f, 0.f}, 1.5f};
    } else {
      second.tag = HJ_MAT_DIFFUSECBOARD;  // the reference's live code, src/main.rs:1466-1471
      second.cboard = hj_diffuse_cb{{1.0f, 0.4f, 0.7f}, 0.1f, {0.4f, 0.7f, 1.0f}, 0.2f};
    }
    s.materials.push_back(second);
    Shape sp{};
    sp.kind = ShapeKind::Sphere;
    sp.sphere = hj_sphere{{-0.4214f, 0.3321f, -0.28f}, 0.3263f};
    s.objects.emplace_back(sp, (int)s.materials.size() - 2);
    sp.sphere = hj_sphere{{0.4458f, 0.3321f, 0.3767f}, 0.3263f};
    s.objects.emplace_back(sp, (int)s.materials.size() - 1);
  } else if (kind != HJH_SYNTH_CBOX && kind != HJH_SYNTH_CBOX_MESH) {
    throw std::runtime_error("unknown synthetic scene kind");
  }
  return s;
}

}  // namespace hijiki
