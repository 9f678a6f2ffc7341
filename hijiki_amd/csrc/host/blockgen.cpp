// Deterministic stand-in for the OS-seeded `rand::random()` calls of the
// reference's ImageBlockGenerator (src/main.rs:643,670,675) and the block
// list it produces (src/main.rs:648-682).  Compiled into BOTH
// libhijiki_host.so and libhijiki_hip.so.
#include "blockgen.hpp"

namespace hijiki {

static inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

uint32_t block_seed(uint64_t master, uint32_t pass, uint32_t block_in_pass) {
  uint64_t h = splitmix64(splitmix64(master ^ 0x484A424Cull) + (((uint64_t)pass << 32) | block_in_pass));
  return (uint32_t)(h >> 32);
}

void pass_offset(uint64_t master, uint32_t k, float out[2]) {
  uint64_t h = splitmix64(splitmix64(master ^ 0x484A4F46ull) + k);
  out[0] = (float)(uint32_t)(h >> 40) * (1.0f / 16777216.0f);
  out[1] = (float)(uint32_t)((h >> 16) & 0xFFFFFFu) * (1.0f / 16777216.0f);
}

BlockGrid::BlockGrid(uint32_t w, uint32_t h, uint32_t block)
    : width(w), height(h), block_size(block), nbx((w + block - 1) / block), nby((h + block - 1) / block) {}

hj_image_block BlockGrid::make(uint64_t master, uint32_t pass, uint32_t j) const {
  hj_image_block b;
  const uint32_t bx = j % nbx, by = j / nbx;
  b.id = pass * per_pass() + j;  // ids run on across passes (src/main.rs:660-662)
  b.seed = block_seed(master, pass, j);
  b.origin[0] = bx * block_size;
  b.origin[1] = by * block_size;
  b.dimension[0] = (width - b.origin[0] < block_size) ? width - b.origin[0] : block_size;    // src/main.rs:658
  b.dimension[1] = (height - b.origin[1] < block_size) ? height - b.origin[1] : block_size;  // src/main.rs:659
  b.original_dimension[0] = width;
  b.original_dimension[1] = height;
  // The generator refreshes sample_offset when it wraps to the next pass and
  // only THEN builds the block it returns (src/main.rs:664-680): the last
  // block of pass p already carries the offset of pass p+1.
  pass_offset(master, pass + ((j == per_pass() - 1) ? 1u : 0u), b.sample_offset);
  return b;
}

}  // namespace hijiki
