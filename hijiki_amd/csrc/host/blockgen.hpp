// Deterministic ImageBlockGenerator (reference src/main.rs:619-682).
#pragma once
#include <cstdint>

#include "../../../include/hijiki_hip.h"

namespace hijiki {

uint32_t block_seed(uint64_t master, uint32_t pass, uint32_t block_in_pass);
void pass_offset(uint64_t master, uint32_t offset_index, float out[2]);

struct BlockGrid {
  uint32_t width, height, block_size, nbx, nby;
  BlockGrid(uint32_t w, uint32_t h, uint32_t block);
  uint32_t per_pass() const { return nbx * nby; }
  // Multi-GPU tile sharding: block j (column bx, row by) of pass p belongs to rank (bx + by + p) mod world — a
  // diagonal deal that moves one step per pass, so that over `world` passes every rank renders every block
  // position once.  Measured on cbox 1024^2 (8 x 8 blocks of 128^2), load-balance efficiency at world 8:
  //   j mod world          0.72  (one whole block column per rank; wall columns are cheaper than the centre)
  //   (bx + by) mod world  0.81  (8 fixed positions per rank still differ by 20 % in walk cost)
  //   (bx + by + p)        see DESIGN.md section 6
  // Callers that want all passes of one block on one rank (block interiors then sum in single-GPU order)
  // pass p = 0 for every pass (HJ_RENDER_STATIC_DEAL).
  uint32_t owner(uint32_t pass, uint32_t j, uint32_t world) const { return (j % nbx + j / nbx + pass) % world; }
  // block j (raster order) of pass `pass`
  hj_image_block make(uint64_t master, uint32_t pass, uint32_t j) const;
};

}  // namespace hijiki
