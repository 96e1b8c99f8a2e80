// bvh_build.hpp — host BVH builder interface.
#pragma once
#include <vector>

#include "device_types.hpp"

namespace vpt {
// tris_in: world-space triangles in instance-major order (gid = index). Produces 128 B four-wide nodes
// (root = node 0) and the triangles permuted into leaf order.
void build_bvh(const std::vector<BvhTri>& tris_in, std::vector<BvhNode>& nodes_out, std::vector<BvhNodeWide>& wide_out, std::vector<BvhTri>& tris_out, int* depth_out);
}  // namespace vpt
