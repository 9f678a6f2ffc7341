// Light-shaft visibility grid (light_grid.cpp): built by hj_scene_upload on the host, read by the shade stage.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../../include/hijiki_hip.h"

namespace hjapi {

struct LightGrid {
  uint32_t res = 0;              // cells per axis; 0: no grid
  float lo[3] = {0, 0, 0};       // cell index along axis k = (int)((p[k] - lo[k]) * inv[k]), valid in [0, res)
  float inv[3] = {0, 0, 0};
  std::vector<uint8_t> bits;     // res^3 bytes (x fastest); bit e: every shadow ray from this cell to emitter e is unoccluded
  // cells on meshes and in corners (bundle proofs): bit e holds for a hit point that the shade stage has found to lie on its shape
  // (kernels/hj_light_grid_const.h); empty when no such cell is proven
  std::vector<uint8_t> mesh_bits;
  // with mesh_bits: 8 floats per quad and triangle (shape index - num_spheres) for that check: unit geometric normal, margin delta of
  // the (u, v) test; vertex a (the quad's origin), 0 for a triangle / 1 for a quad.  All 0: never on its shape.
  std::vector<float> shape_recs;
  size_t cells_surface = 0, cells_planar = 0, pairs_clear = 0;
};

// False (and an empty grid) when nothing can be proven for this scene.  Pure host code: no device call.
bool build_light_grid(const hj_scene_desc* s, uint32_t res, LightGrid& out);

}  // namespace hjapi
