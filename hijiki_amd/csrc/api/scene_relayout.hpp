// hj_scene_upload's re-layout on the device (scene_relayout.hip); the host path in scene_upload.hip is the reference for it.
#pragma once
#include <cstddef>
#include <cstdint>

#include "hj_internal.h"

namespace hjapi {

struct RelayoutOut {
  const float4* nodes = nullptr;       // 2 x float4 per record, placed inside one 4 GiB window
  const float4* tri_isect = nullptr;
  const float4* tri_shade = nullptr;
  const float4* tri_pair = nullptr;
  uint32_t num_nodes = 0, root = 0, root2 = 0, num_hot = 0, num_pairs = 0;   // root2: the second copy of the tree (the reference's own)
  size_t kept = 0;                     // records without padding
};

// d_tris / d_verts: the scene's triangle and vertex arrays, already on the device.  node_order: HJ_NODE_ORDER (-1: by tree).
// HJ_ERR_UNSUPPORTED: the array is not a tree (or a node has more kept children than the kernels enumerate) - the caller takes the
// host path; buffers this call added to ctx->scene_bufs are the caller's to drop then.
int relayout_on_device(hj_context* ctx, const hj_scene_desc* s, const hj_triangle* d_tris, const hj_vertex* d_verts, bool pairs_on,
                       int node_order, float collapse_thr, bool timing, RelayoutOut& out, const hj_bvh_node* d_tree = nullptr);

}  // namespace hjapi
