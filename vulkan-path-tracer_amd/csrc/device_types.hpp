// device_types.hpp — HBM data layout of the wavefront backend.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vpt.h"
#include "../../include/vpt_fp32.h"

namespace vpt {

using vptfp::V2;
using vptfp::V3;
using vptfp::V4;

// ---- BVH2, 64-byte nodes: both child boxes live in the parent so one 64 B fetch (4 x dwordx4)
// decides both children.  child >= 0: inner node index; child < 0: leaf, ~child = first<<3 | (count-1).
struct BvhNode {
    float lmin[3], lmax[3];
    float rmin[3], rmax[3];
    int32_t left, right;
    uint32_t pad0, pad1;
};
static_assert(sizeof(BvhNode) == 64, "node is 64 B");

// World-space triangle, 48 bytes (3 x dwordx4), stored in BVH leaf order.
struct BvhTri {
    float v0[3];
    float e1[3];
    float e2[3];
    uint32_t prim;  // PrimitiveIndex()
    uint32_t inst;  // InstanceIndex()
    uint32_t gid;   // instance-major global triangle id: the closest-hit tie-break key
};
static_assert(sizeof(BvhTri) == 48, "triangle is 48 B");

struct MeshDesc {  // slice of the pooled vertex / index buffers (uVertices[], uIndices[])
    uint32_t vertex_offset, index_offset, tri_count, pad;
};
struct InstanceDesc {  // uMaterialAndMeshIndices + ObjectToWorld3x4 + WorldToObject 3x3
    uint32_t mesh, material, tri_offset, pad;
    float xform[16];  // column-major mat4
    float inv3[9];    // row-major 3x3 of the inverse
    float pad2[3];
};
struct TexDesc {
    uint32_t offset;  // byte offset into the texel pool
    uint32_t w, h, c;
};
struct EmissiveDesc {  // == EmissiveMeshEntry (PathTracer.h:321-328), 80 B
    uint32_t mesh, material, tri_count, instance;
    float xform[16];
};
struct AliasEntry {  // == AliasMapEntry (Bindings.slang:1-5)
    uint32_t alias;
    float importance;
};

struct DeviceScene {
    const BvhNode* nodes;
    const BvhTri* tris;
    uint32_t node_count, tri_count;
    const vpt_vertex* vertices;
    const uint32_t* indices;
    const MeshDesc* meshes;
    const InstanceDesc* instances;
    const vpt_material* materials;
    const TexDesc* textures;
    const uint8_t* texels;
    const EmissiveDesc* emissive;
    uint32_t emissive_count, emissive_tris;
    const float* env;  // RGBA32F, alpha = pdf
    const AliasEntry* alias;
    uint32_t env_w, env_h;
    const float* lut_r;  // 64x64x32
    const float* lut_o;  // 128x128x32
    const float* lut_i;  // 128x128x32
};

struct RenderParams {
    float view_inv[16], proj_inv[16];
    uint32_t width, height;
    uint32_t shard_rank, shard_count, shard_rows, shard_pixels;
    uint32_t samples_per_frame, max_depth;
    float max_luminance, focus_distance, dof_strength;
    float sky_azimuth, sky_altitude, sky_intensity, emissive_pdf_bias;
    uint32_t flags, base_seed;
};

// ---- Wavefront path state, structure-of-arrays over `capacity` slots
// (slot = frame_in_flight * shard_pixels + shard_pixel).  All arrays are indexed by slot so a path
// never moves; queues hold slot ids.
struct PathState {
    uint32_t capacity;
    // payload (RTCommon.slang:5-35, surface subset) + raygen locals
    uint32_t* rng;
    float *ox, *oy, *oz;     // payload.Origin
    float *dx, *dy, *dz;     // payload.Direction
    float *tx, *ty, *tz;     // pathThroughput
    float *lx, *ly, *lz;     // pathLight
    float *bx, *by, *bz;     // payload.BxDF
    float* pdf;              // payload.PDF
    uint32_t* depth;         // payload.Depth
    uint32_t* medium_flag;   // payload.InMedium (bit 0) | sample index within the frame << 8
    float *mdensity, *maniso, *mcr, *mcg, *mcb;  // medium state
    // hit record written by extend (20 B)
    float *ht, *hu, *hv;
    uint32_t *hprim, *hinst;
    // pending contributions of the current bounce, joined in accumulate before the luminance clamp
    float *ex, *ey, *ez;     // emission / miss radiance
    float *skx, *sky, *skz;  // sky NEE contribution (if visible)
    float *lgx, *lgy, *lgz;  // light NEE contribution (if visible)
    uint32_t* vis;           // bit0 sky ray unoccluded, bit1 light ray reached the sampled triangle
    // per-slot sum over the samples of the frame (accumulatedLight)
    float *ax, *ay, *az;
};

// Compacted shadow ray, 32 B: one coalesced dwordx4 pair per lane.
struct ShadowRay {
    float ox, oy, oz;
    uint32_t slot_kind;  // slot | kind<<31 (0 sky, 1 light)
    float dx, dy, dz;
    uint32_t expect_gid;  // light: global id of the sampled triangle
};
static_assert(sizeof(ShadowRay) == 32, "shadow ray is 32 B");

struct Counters {
    uint32_t ray_count[2];   // active-path queue sizes (ping-pong)
    uint32_t shadow_count;
    uint32_t extend_head, shadow_head;  // persistent-kernel work cursors
    uint32_t pad[3];
    unsigned long long stat_nodes, stat_tris;          // extend kernel (count_traversal builds only)
    unsigned long long stat_shadow_nodes, stat_shadow_tris;  // shadow kernel
    unsigned long long stat_pad;
};

}  // namespace vpt
