// bvh_build.hpp — host BVH builder interface.
#pragma once
#include <vector>

#include "device_types.hpp"

namespace vpt {
// tris_in: world-space triangles in instance-major order (gid = index). Produces 128 B four-wide nodes
// (root = node 0) and the triangles permuted into leaf order.
// nodes8_out (optional): the same binary tree collapsed eight-wide with octant-ordered slots, over the same tris_out.
void build_bvh(const std::vector<BvhTri>& tris_in, std::vector<BvhNode>& nodes_out, std::vector<BvhNodeWide>& wide_out, std::vector<BvhTri>& tris_out, int* depth_out,
               std::vector<BvhNode8>* nodes8_out = nullptr, bool spatial_splits = false, bool parallel = true);
// parallel: large subtrees are built by threads of their own; the result is the single-threaded builder's, bit for bit (tests/test_bvh_host.py).
// spatial_splits: SBVH — a node may also be cut by a plane, triangles crossing it are referenced from both children with the bounds of
// their clipped parts (tris_out then holds up to 1.5 x the triangles; hits are unchanged: ties in t go to the smaller global id).
}  // namespace vpt
