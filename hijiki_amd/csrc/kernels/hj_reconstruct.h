// reference shader/reconstruction.glsl:22-66 as a per-pixel gather in block order.
#pragma once
#include "hj_device.h"

#pragma clang fp contract(off)

namespace hj {

// ------------------------------------------------------------ reconstruction

// One thread per output pixel; gathers, IN BLOCK ORDER, what every block of
// the batch splats onto it.  Per-pixel addition order == the reference's
// serial per-block dispatch order (reconstruction.glsl:22-66, main.rs:1316-1355).
// tile_off / tile_blk: for every 16x16 pixel tile the batch's blocks (ascending = list order) whose 2-pixel-extended
// rectangle touches the tile, built on the host while the path kernel runs (CSR layout).
// The 25 Gaussian tap weights of a block (uniform over the block because the sub-pixel offset is per block:
// reconstruction.glsl:27-28,43-44) are formed per (tile, block) in LDS by the first 25 threads.
__global__ __launch_bounds__(256) void k_reconstruct(BatchState st, float stddev,
                                                     const uint32_t* __restrict__ tile_off,
                                                     const uint32_t* __restrict__ tile_blk,
                                                     float4* __restrict__ accum, uint32_t W, uint32_t H) {
  const int tx0 = (int)(blockIdx.x * 16u), ty0 = (int)(blockIdx.y * 16u);
  const int x = tx0 + (int)(threadIdx.x & 15u), y = ty0 + (int)(threadIdx.x >> 4);
  const bool inimg = x < (int)W && y < (int)H;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  bool touched = false;     // pixels no block of this batch reaches are neither read nor written
  const uint32_t tile = blockIdx.y * gridDim.x + blockIdx.x;
  const uint32_t i0 = tile_off[tile], i1 = tile_off[tile + 1];
  // The samples a tile needs from one block (the tile + 2 pixels all round, 20 x 20) are staged in LDS once and
  // the 25 taps of its 256 pixels read them there instead of 50 global fetches per pixel.
  constexpr int TS = 20;
  __shared__ float4 s_rgb[TS * TS], s_nd[TS * TS];
  __shared__ float s_w[25];
  const int px = (int)(threadIdx.x & 15u), py = (int)(threadIdx.x >> 4);
  const float gq = -1.0f / ((2.0f * stddev) * stddev);
  const float c0 = hj_exp(gq * 4.0f);
  for (uint32_t idx = i0; idx < i1; idx++) {
    const uint32_t bi = tile_blk[idx];
    const hj_image_block b = st.blocks[bi];
    const int ox = (int)b.origin[0], oy = (int)b.origin[1], Dx = (int)b.dimension[0], Dy = (int)b.dimension[1];
    const uint32_t sbase = bi * kSlotsPerBlock;
    const int bx0 = tx0 - 2 - ox, by0 = ty0 - 2 - oy;        // block-local coordinates of LDS entry (0, 0)
    __syncthreads();                                          // previous block's taps are done with the LDS tile
    if (threadIdx.x < 25u) {
      const int dx = (int)(threadIdx.x / 5u) - 2, dy = (int)(threadIdx.x % 5u) - 2;
      const float sx = ((float)dx + b.sample_offset[0]) - 0.5f;
      const float sy = ((float)dy + b.sample_offset[1]) - 0.5f;
      s_w[threadIdx.x] = hj_exp(gq * (sx * sx + sy * sy)) - c0;
    }
    for (int e = (int)threadIdx.x; e < TS * TS; e += 256) {
      const int ex = bx0 + e % TS, ey = by0 + e / TS;
      if (ex >= 0 && ex < Dx && ey >= 0 && ey < Dy) {
        const uint32_t sp = sbase + (uint32_t)ey * HJ_BLOCK_SIZE + (uint32_t)ex;
        s_rgb[e] = st.smp_rgb[sp];
        s_nd[e] = st.smp_nd[sp];
      }
    }
    __syncthreads();
    const int lx = x - ox, ly = y - oy;
    if (!inimg || lx < -2 || lx >= Dx + 2 || ly < -2 || ly >= Dy + 2) continue;
    if (!touched) { acc = accum[(size_t)y * W + x]; touched = true; }   // reconstruction.glsl:26
    v3 nc = V(0, 0, 0);
    if (lx >= 0 && lx < Dx && ly >= 0 && ly < Dy) nc = xyz(s_nd[(py + 2) * TS + (px + 2)]);
    for (int dx = -2; dx <= 2; dx++) {
      if (lx + dx < 0 || lx + dx >= Dx) continue;
      for (int dy = -2; dy <= 2; dy++) {
        if (ly + dy < 0 || ly + dy >= Dy) continue;
        float w = s_w[(dx + 2) * 5 + (dy + 2)];
        if (w < 0.0f) continue;
        const int e = (py + 2 + dy) * TS + (px + 2 + dx);
        const float4 nd = s_nd[e];
        const v3 no = xyz(nd) - nc;
        const float dn = dot3(no, no) * 2.0f;
        if (dn != 0.0f) w *= hj_exp(-dn);     // equal normals (flat walls: most taps): hj_exp(-0) == 1 exactly, the product is w
        const float4 c = s_rgb[e];
        const float v0 = w * c.x, v1 = w * c.y, v2 = w * c.z, v3_ = w * c.w;
        if (v0 != v0 || v1 != v1 || v2 != v2 || v3_ != v3_) continue;
        acc.x += v0; acc.y += v1; acc.z += v2; acc.w += v3_;
      }
    }
  }
  if (touched) accum[(size_t)y * W + x] = acc;
}

}  // namespace hj
