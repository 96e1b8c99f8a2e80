// bvh_build.hpp — host BVH builder interface.
#pragma once
#include <vector>

#include "device_types.hpp"

#ifndef VPT_BVH_PAIRWISE
#define VPT_BVH_PAIRWISE 0
#endif
namespace vpt {
// tris_in: world-space triangles in instance-major order (gid = index). Produces 128 B four-wide nodes
// (root = node 0) and the triangles permuted into leaf order.
// nodes8_out (optional): the same binary tree collapsed eight-wide with octant-ordered slots, over the same tris_out.
void build_bvh(const std::vector<BvhTri>& tris_in, std::vector<BvhNode>& nodes_out, std::vector<BvhNodeWide>& wide_out, std::vector<BvhTri>& tris_out, int* depth_out,
               std::vector<BvhNode8>* nodes8_out = nullptr, bool spatial_splits = false, bool parallel = true);
// The same with every knob of the builder study (tests/tools/vote_sim.cpp, profiles/r06_builder_study.md) and the trace lab's trees:
struct BvhBuildOptions {
    bool spatial_splits = false, parallel = true;
    int bins = 16;                                // SAH bins per axis (16: the product; at most 32)
    // How the binary tree is collapsed four-wide.  false: a node adopts grandchildren largest box first (up to three levels of the binary tree in one node);
    // true: always the two children's children ("pair-wise").  Same triangles, same leaves; only which inner boxes share a node changes.
    bool pairwise = VPT_BVH_PAIRWISE;
    std::vector<BvhNode8>* nodes8 = nullptr;      // the eight-wide tree of the BVH8 experiment
    // "split-order" four-wide tree (trace lab VPT_TRACE_VOTE4S): the binary tree collapsed so that slots 0,1 hold the left child's children
    // and slots 2,3 the right child's, with three 8-bit tables (one bit per ray-direction octant) in the mantissas of step_x / step_y that say
    // whether the ray meets the pair / the pairs in reverse: hit children are visited in that order instead of sorted by entry distance
    // (vote.hpp vote_node4s_step).  Same leaf-ordered triangles.
    std::vector<BvhNode>* nodes4s = nullptr;
    double* sah_cost = nullptr;                   // SAH cost of the binary tree: sum over inner nodes of kNodeCost x area + over leaves of count x area, / root area
};
void build_bvh_ex(const std::vector<BvhTri>& tris_in, std::vector<BvhNode>& nodes_out, std::vector<BvhNodeWide>& wide_out, std::vector<BvhTri>& tris_out, int* depth_out,
                  const BvhBuildOptions& opt);
// parallel: large subtrees are built by threads of their own; the result is the single-threaded builder's, bit for bit (tests/test_bvh_host.py).
// spatial_splits: SBVH — a node may also be cut by a plane, triangles crossing it are referenced from both children with the bounds of
// their clipped parts (tris_out then holds up to 1.5 x the triangles; hits are unchanged: ties in t go to the smaller global id).
}  // namespace vpt
