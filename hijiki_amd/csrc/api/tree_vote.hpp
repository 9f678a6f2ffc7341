// The ray-voted child order on the device (kernels/hj_vote.h): host half, shared by hj_tune_bvh_device and hj_build_bvh_device.
#pragma once
#include "hj_internal.h"

namespace hjapi {

struct VoteShapes {               // the shape arrays of an hj_scene_desc, on the device
  const float4* spheres;          // hj_sphere
  const float4* quads;            // hj_quad as 3 x float4
  const hj_triangle* triangles;
  const hj_vertex* vertices;
};

struct VoteResult { uint32_t exchanged = 0, levels = 0; };

// d_nodes[N]: a flattened tree (pre-order skip-link records) over the shapes of `s`, on the device.  Samples `paths` camera paths
// of `s` (camera, materials, emitters: uploaded here), exchanges the children of the inner nodes where the sample finds the other
// order cheaper and writes the re-ordered array to d_out[N] (device, not d_nodes).  HJ_ERR_INVALID when the links are not a tree's.
int vote_on_device(hj_context* ctx, const hj_scene_desc* s, const VoteShapes& shapes, const hj_bvh_node* d_nodes, size_t N, size_t paths,
                   hj_bvh_node* d_out, bool timing, VoteResult* result);

// (position, record) pairs into a device array of records
int put_records(hj_context* ctx, const std::vector<std::pair<uint32_t, hj_bvh_node>>& records, hj_bvh_node* d_array);

}  // namespace hjapi
