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
  // block j (raster order) of pass `pass`
  hj_image_block make(uint64_t master, uint32_t pass, uint32_t j) const;
};

}  // namespace hijiki
