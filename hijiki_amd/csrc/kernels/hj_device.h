// Device-side views: scene arrays as the kernels read them, and the SoA path
// state of one wavefront batch.  All pointers are device pointers.
#pragma once
#include "../../../include/hijiki_hip.h"
#include "hj_num.h"
#include "hj_light_grid_const.h"

namespace hj {

constexpr uint32_t kSlotsPerBlock = HJ_BLOCK_SIZE * HJ_BLOCK_SIZE;  // fixed 128x128 slot grid per ImageBlock
constexpr uint32_t kNumTags = 5;
// Pre-gathered emitter record, 7 x float4 (every value copied verbatim from the reference arrays):
//   r0 = (emitter.pdf, kind bits [0 sphere, 1 quad, 2 triangle], sphere radius, -)
//   r1..r3 = (sphere centre | quad origin, edge1, edge2 | triangle a, b, c).xyz, w = emissive power r, g, b
//   r4..r6 = triangle vertex normals
constexpr uint32_t kEmitRecF4 = 7;
#ifndef HJ_HOT_NODES
#define HJ_HOT_NODES 512   // 16 KB of LDS per workgroup (at the configurations' own frame sizes, interleaved: 384: -1.2 ... -1.8 % on the two
                           // box scenes, 640: -1 %; the 1 M-triangle scene does not care)
#endif
constexpr uint32_t kHotNodes = HJ_HOT_NODES;
constexpr uint32_t kRefillMin = 32;   // free lanes at which a wave of the persistent walk fetches new rays (hj_scene_upload: 24 on trees with
                                      // pair nodes; sweep on the first fused kernel: 16 -> 1.39, 32..48 -> 1.43 Gpaths/s, 64 -> 1.24)
constexpr uint32_t kInnerFlag = 0x80000000u;
constexpr uint32_t kPairFlag = 0x40000000u;    // with kInnerFlag: an inner node whose two children are triangle leaves
constexpr uint32_t kIndexMask = 0x3FFFFFFFu;
constexpr uint32_t kEndOfWalk = 0x3FFFFFFFu;   // the exit of the last nodes of a tree (the array holds two trees: no node count can serve)

// Scene data in HBM.  `nodes` are 32-byte records (two float4 per node) derived from the reference's
// skip-link array (same tree, same boxes, same visiting order) but RE-INDEXED: the kHotNodes nodes with the
// largest surface area come first (every workgroup keeps a copy of them in LDS: on cbox 64 nodes take 77 % of
// all node fetches, 256 take 84 %), the rest follow in the original pre-order.  Because "left child = next
// record" no longer holds, links are explicit:
//   n0 = (aabb_min.xyz, A)   A = shape index for a leaf, 0x80000000 | left-child index for an inner node,
//                            0xC0000000 | pair index for an inner node over two triangle leaves (those two leaves
//                            have no records of their own: hj_intersect.h leaf_test)
//   n1 = (aabb_max.xyz, B)   B = exit index (>= num_nodes ends the walk: kEndOfWalk)
// The walk starts at `root`.  Triangles are additionally pre-gathered per
// triangle so that a leaf test is ONE dependent fetch instead of the
// reference's index -> vertex chain (shader/shapes/triangle.glsl:16-18):
//   tri_isect[3i+0..2] = (a.xyz,-) (b-a .xyz,-) (c-a .xyz,-)        48 B
//   tri_shade[4i+0..3] = (na.xyz,ua) (nb.xyz,ub) (nc.xyz,uc) (va,vb,vc,-)  64 B
// b-a and c-a are the same single IEEE subtractions the shader performs.
struct DeviceScene {
  const float4* nodes;
  uint32_t num_nodes;
  uint32_t root;                // device index of the reference's node 0
  uint32_t root2;               // ... in the second copy of the tree, the reference's own (every node, no guards; pair nodes only): where
                                // rays that are not in general position start (kernels/hj_intersect.h general_position)
  uint32_t num_hot;             // nodes [0, num_hot) are the LDS-cached ones (<= kHotNodes)
  uint32_t inner_burst;         // max box steps per round of the persistent walk before leaf tests run
  uint32_t refill_min;          // free lanes that trigger a ray refill
  const float4* tri_isect;
  const float4* tri_pair;       // 6 x float4 per pair node: (a, b-a, c-a) of the left and of the right triangle, shape indices in [0].w, [3].w
  const float4* tri_shade;
  const float4* spheres;        // hj_sphere
  const float4* quads;          // hj_quad as 3 x float4
  const hj_triangle* triangles; // original indices (emitter sampling)
  const hj_vertex* vertices;    // original vertices (emitter sampling)
  const uint32_t* materials;
  const hj_emitter* emitters;
  const float4* emit_rec;       // kEmitRecF4 float4 per emitter, pre-gathered (see below)
  const float4* diffuse;
  const float4* diffusecb;      // 2 x float4 per record
  const float4* dielectric;
  const float4* emissive;
  uint32_t ns, nq, nt, num_emitters;
  uint32_t has_extinction;      // any dielectric with non-zero extinction
  uint32_t has_pairs;           // the node array holds pair nodes (tri_pair)
  uint32_t stream_state;        // large tree: path records and samples bypass the caches (non-temporal accesses)
  uint32_t group_tile;          // large tree: a 64-sample group is an 8 x 8 pixel tile of its block instead of 64 pixels of a row (hj_stages.h)
  hj_camera camera;
  float tan_half_fov;           // (float)tan(radians(fov/2)) evaluated in double on the host
  // Light-shaft visibility grid (api/light_grid.cpp): bit e of cell (x, y, z) says that EVERY next-event shadow ray from a hit
  // point in that cell to emitter e is unoccluded - the shade stage then adds the sample at once instead of queueing a ray.
  // Low byte: proofs that hold for every hit point of the cell (planar cells); high byte: cells on meshes and in corners, whose
  // proofs hold for a hit that was not grazing (|d.n| >= kLightGridSinIn |d| against lg_normals: hj_light_grid_const.h).
  // Behind the cells, 16-byte aligned, when any high byte is set: two float4 per quad and triangle (shape id - ns) for the check
  // that a hit point lies on its shape (hj_light_grid_const.h; one pointer for both: the kernels keep this struct in scalar registers).
  const uint16_t* light_grid;   // lg_res^3 cells, x fastest; null: no grid
  uint32_t lg_res;              // cells per axis | kLightGridHasRecords
  float lg_lo[3], lg_inv[3];    // cell index along axis k = (int)((p[k] - lg_lo[k]) * lg_inv[k])
};

// One wavefront batch = the samples of up to 4096 ImageBlocks.  Two index spaces:
//   SAMPLE   = block_in_batch * 16384 + ly * 128 + lx: the intermediate image of the batch (what reconstruction
//              reads); 64-sample groups are dealt round-robin over the workgroups, so each one samples the whole image;
//   POSITION = index into the workgroup's segment [g * pool, (g + 1) * pool) of the path arrays.  The paths in flight
//              are kept COMPACTED: the record of a path lives at its position in the current round's ray queue, shade
//              writes the record of a continuing path at its position in the NEXT round's queue (the arrays are
//              double-buffered by round parity), and the top-up appends NEW camera paths of the workgroup's sample
//              sequence behind them (path regeneration).  So the queue is implicit (entry i = record i), every stage
//              reads and writes the path arrays in queue order (coalesced), a path that ends simply is not written
//              again, and ~pool paths stay in flight per workgroup until its samples run out instead of decaying
//              bounce by bounce.  This is the "rays sorted/compacted by material and alive-mask" of the design brief.
// Workgroup g reads and appends only its own segments (appends: wave ballot + one LDS atomic): no stage touches a
// global atomic.  (The first version used chip-wide queue counters: ~88 M same-address atomics/s bounded every kernel.)
constexpr uint32_t kCameraFlag = 0x80000000u;   // in ray_o.w beside the sample index: camera ray (tMin = eps, render.glsl:33)
struct BatchState {
  // per sample
  float4* smp_rgb;      // layer 0 of the intermediate image: (radiance, 1)
  float4* smp_nd;       // layer 1: (first-hit normal, first-hit t)
  // per position, double-buffered by round parity: the paths in flight
  float4* ray_o[2];     // origin.xyz, sample index bits | kCameraFlag
  float4* ray_d[2];     // direction.xyz, RNG state bits
  float4* thr[2];       // throughput.rgb, flags bits (bit0 wasDiscrete, bits 1.. bounce index)
  float4* ext[2];       // current extinction (only touched if scene.has_extinction)
  // per position of the current round
  float4* hit;          // (t, objectID bits, u, v) of the raw hit
  uint8_t* hit_tag;     // material tag of the hit (0xFF: miss), written by the first pass of the hit compaction for its second
                        // pass: 1 byte per ray instead of the 16-byte record + the material word a second time
  uint32_t* q_hit;      // [kNumTags][num_wg][pool] positions of the hits, binned by material tag, in queue order
  // NEE shadow rays produced by shade, walked in the next round (self-contained: the path may be over by then)
  float4* sh_o;         // origin.xyz
  float4* sh_d;         // direction.xyz, tMax
  float4* sh_c;         // pending NEE contribution rgb, sample index bits
  uint32_t* cnt_ray[2];   // [num_wg]        (split-kernel path: counts between launches)
  uint32_t* cnt_hit;      // [num_wg][kNumTags]
  uint32_t* cnt_shadow;   // [num_wg]
  uint32_t* acc_closest;  // [num_wg] closest-hit rays traced by this workgroup over the batch (stats)
  uint32_t* acc_shadow;   // [num_wg] shadow rays
  uint32_t* acc_hits;     // [num_wg] closest-hit rays that hit something
  uint32_t* acc_unoccluded;   // [num_wg] shadow rays that reached their light
  uint32_t* acc_direct;       // [num_wg] of those: next-event samples the light-shaft grid answered (no ray was traced)
  const hj_image_block* blocks;  // the batch's ImageBlocks
  uint32_t num_blocks;
  uint32_t capacity;             // samples allocated
  uint32_t num_wg;               // grid size of every stage kernel
  uint32_t pool;                 // positions per workgroup, multiple of 64
  uint32_t xcd_deal;             // sample groups dealt per XCD (hj_stages.h: wg_group)
};

}  // namespace hj
