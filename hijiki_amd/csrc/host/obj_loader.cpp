// OBJ/MTL scene loader: `Scene::from_obj` of the reference (src/main.rs:414-530) on top of an own parser that
// follows the behaviour of the `tobj` 0.1.11 crate the reference calls (src/main.rs:415).  The crate's source is
// not part of the reference checkout; its semantics are restated from its documentation/behaviour:
//   * one model per `o`/`g` statement (and per `usemtl` change once faces exist), exported when the next starts;
//   * per model, vertices are re-indexed by unique (v, vt, vn) tuples in order of first use;
//   * polygons are triangulated as a fan (0, i, i+1); negative indices are relative to the current counts;
//   * `mtllib` is resolved next to the OBJ file; MTL statements the crate does not know (e.g. `Ke`) are kept as
//     strings in `unknown_param` — the reference reads the light's radiance from there (src/main.rs:434).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "../../../include/hijiki_host.h"
#include "scene.hpp"

namespace hijiki {
namespace {

struct ObjMesh {
  std::string name;
  std::vector<float> positions, normals, texcoords;
  std::vector<uint32_t> indices;
  int material_id = -1;   // tobj: Option<usize>
};
struct ObjMaterial {
  std::string name;
  float ambient[3] = {0, 0, 0}, diffuse[3] = {0, 0, 0}, specular[3] = {0, 0, 0};
  float shininess = 0.f, optical_density = 1.f, dissolve = 1.f;
  std::map<std::string, std::string> unknown_param;
};

std::vector<std::string> split_ws(const std::string& s) {
  std::vector<std::string> out;
  std::istringstream is(s);
  std::string tok;
  while (is >> tok) out.push_back(tok);
  return out;
}
std::string trim(const std::string& s) {
  size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
  return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}
float to_f(const std::string& s, const std::string& ctx) {
  char* end = nullptr;
  float v = std::strtof(s.c_str(), &end);
  if (end == s.c_str()) throw std::runtime_error("cannot parse number '" + s + "' in " + ctx);
  return v;
}
std::string dir_of(const std::string& path) {
  size_t p = path.find_last_of("/\\");
  return p == std::string::npos ? std::string() : path.substr(0, p + 1);
}

void load_mtl(const std::string& path, std::vector<ObjMaterial>& mats, std::map<std::string, int>& index) {
  std::ifstream in(path);
  if (!in) throw std::runtime_error("cannot open material library " + path);
  std::string line;
  ObjMaterial cur;
  bool have = false;
  auto flush = [&]() {
    if (have) {
      index[cur.name] = (int)mats.size();
      mats.push_back(cur);
    }
  };
  while (std::getline(in, line)) {
    line = trim(line);
    if (line.empty() || line[0] == '#') continue;
    std::vector<std::string> t = split_ws(line);
    const std::string& key = t[0];
    auto vec3 = [&](float out[3]) {
      if (t.size() < 4) throw std::runtime_error("MTL: '" + key + "' needs 3 numbers");
      for (int i = 0; i < 3; i++) out[i] = to_f(t[1 + i], path);
    };
    if (key == "newmtl") {
      flush();
      cur = ObjMaterial();
      cur.name = t.size() > 1 ? trim(line.substr(6)) : std::string();
      have = true;
    } else if (key == "Ka") vec3(cur.ambient);
    else if (key == "Kd") vec3(cur.diffuse);
    else if (key == "Ks") vec3(cur.specular);
    else if (key == "Ns" && t.size() > 1) cur.shininess = to_f(t[1], path);
    else if (key == "Ni" && t.size() > 1) cur.optical_density = to_f(t[1], path);
    else if (key == "d" && t.size() > 1) cur.dissolve = to_f(t[1], path);
    else if (key == "illum" || key == "map_Ka" || key == "map_Kd" || key == "map_Ks" || key == "map_Ns" || key == "map_d") {
      // known to the crate, irrelevant to the reference
    } else {
      cur.unknown_param[key] = trim(line.substr(key.size()));
    }
  }
  flush();
}

struct Corner { long v, vt, vn; };

void load_obj(const std::string& path, std::vector<ObjMesh>& models, std::vector<ObjMaterial>& mats) {
  std::ifstream in(path);
  if (!in) throw std::runtime_error("cannot open " + path);
  std::vector<float> pos, tex, nrm;
  std::vector<std::vector<Corner>> faces;
  std::map<std::string, int> mat_index;
  std::string name = "unnamed_object";
  int mat_id = -1;

  auto export_faces = [&]() {
    if (faces.empty()) return;
    ObjMesh m;
    m.name = name;
    m.material_id = mat_id;
    std::map<std::tuple<long, long, long>, uint32_t> remap;
    auto vertex = [&](const Corner& c) -> uint32_t {
      auto key = std::make_tuple(c.v, c.vt, c.vn);
      auto it = remap.find(key);
      if (it != remap.end()) return it->second;
      if (c.v < 0 || (size_t)c.v * 3 + 2 >= pos.size() + 0) throw std::runtime_error("OBJ: position index out of range");
      uint32_t idx = (uint32_t)(m.positions.size() / 3);
      for (int k = 0; k < 3; k++) m.positions.push_back(pos[(size_t)c.v * 3 + k]);
      if (c.vt >= 0) {
        if ((size_t)c.vt * 2 + 1 >= tex.size()) throw std::runtime_error("OBJ: texcoord index out of range");
        for (int k = 0; k < 2; k++) m.texcoords.push_back(tex[(size_t)c.vt * 2 + k]);
      }
      if (c.vn >= 0) {
        if ((size_t)c.vn * 3 + 2 >= nrm.size()) throw std::runtime_error("OBJ: normal index out of range");
        for (int k = 0; k < 3; k++) m.normals.push_back(nrm[(size_t)c.vn * 3 + k]);
      }
      remap.emplace(key, idx);
      return idx;
    };
    for (const auto& f : faces) {
      if (f.size() < 3) continue;   // points and lines carry no surface
      const uint32_t a = vertex(f[0]);
      uint32_t prev = vertex(f[1]);
      for (size_t i = 2; i < f.size(); i++) {   // fan triangulation
        const uint32_t c = vertex(f[i]);
        m.indices.push_back(a);
        m.indices.push_back(prev);
        m.indices.push_back(c);
        prev = c;
      }
    }
    models.push_back(std::move(m));
    faces.clear();
  };

  std::string line;
  size_t lineno = 0;
  while (std::getline(in, line)) {
    lineno++;
    line = trim(line);
    if (line.empty() || line[0] == '#') continue;
    std::vector<std::string> t = split_ws(line);
    const std::string& key = t[0];
    const std::string ctx = path + ":" + std::to_string(lineno);
    if (key == "v") {
      if (t.size() < 4) throw std::runtime_error(ctx + ": 'v' needs 3 numbers");
      for (int i = 0; i < 3; i++) pos.push_back(to_f(t[1 + i], ctx));
    } else if (key == "vt") {
      if (t.size() < 3) throw std::runtime_error(ctx + ": 'vt' needs 2 numbers");
      for (int i = 0; i < 2; i++) tex.push_back(to_f(t[1 + i], ctx));
    } else if (key == "vn") {
      if (t.size() < 4) throw std::runtime_error(ctx + ": 'vn' needs 3 numbers");
      for (int i = 0; i < 3; i++) nrm.push_back(to_f(t[1 + i], ctx));
    } else if (key == "f") {
      std::vector<Corner> f;
      for (size_t i = 1; i < t.size(); i++) {
        Corner c{-1, -1, -1};
        long* dst[3] = {&c.v, &c.vt, &c.vn};
        const long counts[3] = {(long)(pos.size() / 3), (long)(tex.size() / 2), (long)(nrm.size() / 3)};
        size_t start = 0;
        for (int part = 0; part < 3 && start <= t[i].size(); part++) {
          size_t slash = t[i].find('/', start);
          std::string tok = t[i].substr(start, slash == std::string::npos ? std::string::npos : slash - start);
          if (!tok.empty()) {
            long v = std::strtol(tok.c_str(), nullptr, 10);
            *dst[part] = v < 0 ? counts[part] + v : v - 1;   // negative = relative to the current count
          }
          if (slash == std::string::npos) break;
          start = slash + 1;
        }
        if (c.v < 0) throw std::runtime_error(ctx + ": bad face vertex '" + t[i] + "'");
        f.push_back(c);
      }
      faces.push_back(std::move(f));
    } else if (key == "o" || key == "g") {
      export_faces();
      name = t.size() > 1 ? trim(line.substr(1)) : std::string("unnamed_object");
    } else if (key == "mtllib") {
      for (size_t i = 1; i < t.size(); i++) load_mtl(dir_of(path) + t[i], mats, mat_index);
    } else if (key == "usemtl") {
      const std::string mname = t.size() > 1 ? trim(line.substr(6)) : std::string();
      auto it = mat_index.find(mname);
      const int new_id = it == mat_index.end() ? -1 : it->second;
      if (new_id != mat_id && !faces.empty()) export_faces();   // a material change splits the model
      mat_id = new_id;
    }
    // `s`, `l`, `p` and anything else: ignored
  }
  export_faces();
}

}  // namespace

// Scene::from_obj, src/main.rs:414-530
Scene scene_from_obj(const std::string& path) {
  std::vector<ObjMesh> models;
  std::vector<ObjMaterial> materials;
  load_obj(path, models, materials);

  Scene scene = make_synthetic(HJH_SYNTH_CBOX, 16, 0);   // only for the hard-coded camera (src/main.rs:417-425)
  scene.objects.clear();
  scene.vertices.clear();
  scene.materials.clear();

  for (const ObjMaterial& m : materials) {   // src/main.rs:432-458: material kind by NAME PREFIX
    Material out{};
    if (m.name.rfind("light", 0) == 0) {
      auto it = m.unknown_param.find("Ke");
      if (it == m.unknown_param.end()) throw std::runtime_error("material '" + m.name + "' has no Ke (reference: unwrap() panic, src/main.rs:434)");
      std::vector<std::string> parts = split_ws(it->second);
      if (parts.size() < 3) throw std::runtime_error("material '" + m.name + "': Ke needs 3 numbers");
      out.tag = HJ_MAT_EMISSIVE;
      out.emissive = hj_emissive{{to_f(parts[0], "Ke"), to_f(parts[1], "Ke"), to_f(parts[2], "Ke")}, 0.f};
    } else if (m.name.rfind("glass", 0) == 0) {
      out.tag = HJ_MAT_DIELECTRIC;
      out.dielectric = hj_dielectric{{0.f, 0.f, 0.f}, 1.5f};   // DielectricMaterial::clear(1.5)
    } else if (m.name.rfind("mirror", 0) == 0) {
      out.tag = HJ_MAT_MIRROR;
    } else {
      out.tag = HJ_MAT_DIFFUSE;
      out.diffuse = hj_diffuse{{m.diffuse[0], m.diffuse[1], m.diffuse[2]}, 0.f};
    }
    scene.materials.push_back(out);
  }

  for (const ObjMesh& mesh : models) {   // src/main.rs:460-527
    const uint32_t vertex_offset = (uint32_t)scene.vertices.size();
    const size_t nv = mesh.positions.size() / 3;
    for (size_t i = 0; i < nv; i++) {
      hj_vertex v{};
      for (int k = 0; k < 3; k++) v.pos[k] = mesh.positions[3 * i + k];
      if (2 * i + 1 < mesh.texcoords.size()) { v.u = mesh.texcoords[2 * i]; v.v = mesh.texcoords[2 * i + 1]; }   // else (0, 0)
      if (3 * i + 2 >= mesh.normals.size())
        throw std::runtime_error("model '" + mesh.name + "' has vertices without normals (reference: unwrap() panic, src/main.rs:467)");
      for (int k = 0; k < 3; k++) v.normal[k] = mesh.normals[3 * i + k];
      scene.vertices.push_back(v);
    }
    if (mesh.material_id < 0) continue;   // vertices are kept, faces dropped (src/main.rs:476-479)
    Shape t{};
    t.kind = ShapeKind::Triangle;
    for (size_t i = 0; i + 2 < mesh.indices.size(); i += 3) {
      t.tri = hj_triangle{{mesh.indices[i] + vertex_offset, mesh.indices[i + 1] + vertex_offset, mesh.indices[i + 2] + vertex_offset}};
      scene.objects.emplace_back(t, mesh.material_id);
    }
  }
  return scene;
}

// `--put-cbox-spheres`, src/main.rs:1463-1483: a mirror sphere and a checkerboard-diffuse sphere (the reference's
// comment says "glass sphere"; the live code gives it the checkerboard material).
void put_cbox_spheres(Scene& scene) {
  Material mirror{};
  mirror.tag = HJ_MAT_MIRROR;
  scene.materials.push_back(mirror);
  Material cb{};
  cb.tag = HJ_MAT_DIFFUSECBOARD;
  cb.cboard = hj_diffuse_cb{{1.0f, 0.4f, 0.7f}, 0.1f, {0.4f, 0.7f, 1.0f}, 0.2f};
  scene.materials.push_back(cb);
  Shape sp{};
  sp.kind = ShapeKind::Sphere;
  sp.sphere = hj_sphere{{-0.421400f, 0.332100f, -0.280000f}, 0.3263f};
  scene.objects.emplace_back(sp, (int)scene.materials.size() - 2);
  sp.sphere = hj_sphere{{0.445800f, 0.332100f, 0.376700f}, 0.3263f};
  scene.objects.emplace_back(sp, (int)scene.materials.size() - 1);
}

}  // namespace hijiki
