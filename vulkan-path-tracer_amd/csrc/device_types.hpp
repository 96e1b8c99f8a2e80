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

// ---- BVH4, 64-byte nodes (two per cache line, 4 x dwordx4): the four child boxes are 8-bit offsets on a
// per-node grid, plane = origin + q * step per axis with step a power of two, rounded OUTWARD at build time so a decoded box
// always contains the (already padded) fp32 child box.  One fetch decides four children.  The steps are stored as the floats
// themselves (2^(e-127), e in [1, 254]; round 2 packed the three exponents into one word and left 8 bytes of the node unused:
// same values, six integer instructions more per visit).
//   child >= 0: inner node index; child < 0: leaf, ~child = first<<3 | (count-1).
// Unused slots hold the inverted box lo = 255, hi = 0 (empty interval on every axis) and a harmless leaf code.
struct BvhNode {
    float origin[3];
    float step_x;      // grid step along x
    uint32_t lo[3];    // per axis: byte k = lower plane of child k
    uint32_t hi[3];    // per axis: byte k = upper plane of child k
    float step_y, step_z;
    int32_t child[4];
    __host__ __device__ void set_step(int axis, uint32_t biased_exponent) {
        uint32_t bits = biased_exponent << 23; float f; __builtin_memcpy(&f, &bits, 4);
        (axis == 0 ? step_x : axis == 1 ? step_y : step_z) = f;
    }
};
static_assert(sizeof(BvhNode) == 64, "node is 64 B");
// The builder numbers the top of the tree breadth-first: nodes 0 .. kBvhTopNodes-1 (root, its children, ...) are the ones the ray-stream
// traversal kernels copy into LDS (4 KB per block next to the 14 KB of stacks: 8 blocks per CU still fit the 160 KB).
constexpr int kBvhTopNodes = 64;

// Eight-wide variant of the quantised node (96 B, 6 x dwordx4) for the BVH8 experiment (profiles/REJECTED.md): same grid
// quantisation, eight child boxes, children placed in slots by OCTANT — bit a of a slot index says on which side of the node's
// centre the child lies along axis a — so that a ray can visit the hit children in the fixed order slot XOR (sign bits of its
// direction) instead of sorting entry distances.  Unused slots hold the inverted box and a harmless leaf code.
struct BvhNode8 {
    float origin[3];
    uint32_t exps;        // ex | ey << 8 | ez << 16
    uint32_t lo[3][2];    // per axis: byte k of the 8 = lower plane of child k
    uint32_t hi[3][2];
    int32_t child[8];     // >= 0 inner node index; < 0 leaf, ~child = first << 3 | (count - 1)
};
static_assert(sizeof(BvhNode8) == 96, "BVH8 node is 96 B");

// The same tree with fp32 child boxes in SoA form (128 B), used only when the whole BVH is staged into LDS
// (scenes of a few dozen triangles): there the bytes are free and the plain slab test costs fewer VALU ops than
// decoding the grid.  Unused slots hold an unreachable point box at 1e30.
struct BvhNodeWide {
    float minx[4], miny[4], minz[4];
    float maxx[4], maxy[4], maxz[4];
    int32_t child[4];
    uint32_t pad[4];
};
static_assert(sizeof(BvhNodeWide) == 128, "wide node is 128 B");

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
// Derived tables, filled on the device by the same expressions the shade stage would otherwise evaluate
// per hit (kernels_path.hip k_precompute_*), so using them cannot change a bit of the result.
struct MatResolved {  // Material.Initialize (Material.slang:39-87), per material: every field whose texture is 1x1 (k_precompute_materials)
    float base[3], roughness;
    float emissive[3], metallic;
    float ax, ay, ior, inv_ior;
    float pm, pd, pg;
    uint32_t flags;   // kMat*: which fields are valid (the others are fetched per hit, shading.hpp material_issue / material_finish)
    float nmap[3], aspect;  // aspect = sqrt(1 - sqrt(Anisotropy) * 0.9), Material.slang:62
    float rot_sin, rot_cos, pad1, pad2;  // sincos(AnisotropyRotation in radians), Surface.slang:129-136
    // copies of textures[normal | base colour | roughness | metallic | emissive]: a hit's texel fetches are issued as soon as its
    // uv is known, all of them together, without waiting for a descriptor first
    TexDesc tex[5];
};
static_assert(sizeof(MatResolved) == 176, "MatResolved is 176 B");
enum : uint32_t {
    kMatAllValues = 1u,   // base / roughness / metallic / emissive textures are ALL 1x1: no value texture is sampled per hit
    kMatNormal = 2u,      // normal map 1x1 -> nmap valid
    kMatBase = 4u,        // base colour texture 1x1 -> base valid
    kMatRoughness = 8u,   // roughness texture 1x1 -> roughness, ax, ay valid
    kMatMetallic = 16u,   // metallic texture 1x1 -> metallic, pm, pd, pg valid
    kMatEmissive = 32u,   // emissive texture 1x1 -> emissive valid
};
// What SampleEmissiveTriangle (Sampler.slang:348-422) needs about the emissive mesh it has picked, gathered from EmissiveDesc,
// InstanceDesc, vpt_material and TexDesc into one record (k_precompute_lights): the sample costs two dependent fetches (this record,
// the light triangle) instead of six.
struct LightSampler {
    uint32_t tri_count, gid_base, tri_base, uniform;  // first global triangle id; first entry in emissive_tri; 1: the emissive texture is 1x1
    float emissive_color[3], pad0;
    float radiance[3], pad1;                          // uniform: EmissiveColor * the one texel
    TexDesc tex;                                      // otherwise: the emissive texture
};
static_assert(sizeof(LightSampler) == 64, "LightSampler is 64 B");
struct EmissiveTri {  // world-space light triangle as SampleEmissiveTriangle (Sampler.slang:375-404) derives it per sample
    float p0[3], area;
    float p1[3], u0;
    float p2[3], v0;
    float nrm[3], u1;
    float v1, u2, v2, pad;
};
static_assert(sizeof(EmissiveTri) == 80, "EmissiveTri is 80 B");

struct AliasEntry {  // == AliasMapEntry (Bindings.slang:1-5)
    uint32_t alias;
    float importance;
};

// A heterogeneous volume's density: dense raw values (x fastest) + the 32^3 table of per-block maxima of value/max.
struct DensityGrid {
    const float* values;
    const float* block_max;
    uint32_t dim[3];
    float max_density;
};

struct DeviceScene {
    const BvhNode* nodes;
    const BvhNodeWide* nodes_wide;  // non-null only for LDS-resident scenes
    const BvhNode8* nodes8;         // eight-wide tree over the SAME leaf-ordered triangles (trace lab only; nullptr unless built)
    const BvhNode* nodes4s;         // split-order four-wide tree over the same triangles (trace lab VPT_TRACE_VOTE4S only; nullptr unless built)
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
    uint32_t env_black;  // 1: every env texel (and so its pdf) is exactly 0 -> lookups return 0 without fetching
    uint32_t all_plain;  // always 0 from the host; the fused kernel's PLAIN instantiation sets it (a compile-time fact there): every material's textures
                         // are 1x1 (MatResolved.flags == 63, every LightSampler uniform), so no texel is ever fetched and the texture code drops out
    const float* lut_r;  // 64x64x32
    const float* lut_o;  // 128x128x32
    const float* lut_i;  // 128x128x32
    const MatResolved* mat_resolved;        // per material
    const float4* tri_ng;                   // per global triangle id: world-space geometric normal (Surface.slang:48-49)
    const float4* tri_shade;                // per global triangle id: 8 float4 = the three vertices (position | normal | uv, as in vpt_vertex) + tri_ng,
                                            // de-indexed into ONE 128-byte line, so a hit costs one line fetch instead of index triple + 3 vertices + normal
    const EmissiveTri* emissive_tri;        // per emissive triangle
    const uint32_t* emissive_tri_offset;    // per emissive mesh: first entry in emissive_tri
    const LightSampler* lights;             // per emissive mesh
    const uint32_t* tri_slot_of_gid;        // per global triangle id: position in the leaf-ordered triangle array
    const unsigned char* inst_class;        // per instance: shade class of its material (kShade*), the sort key of the shade queues
    uint32_t* stack_overflow;               // traversal stack entries beyond the LDS part, kStackOverflow per resident thread
    const vpt_volume* volumes;              // uVolumes (Volume.slang:9); volume_count == 0: none
    uint32_t volume_count, phase;           // PHASE_FUNCTION_* (PathTracer.h:76-81)
    uint32_t atm_on;                        // ENABLE_ATMOSPHERE
    uint32_t strict_hits;                   // VPT_FLAG_LOCAL_HITS: validate the winning hit's locality (traverse.hpp)
    uint32_t hetero;                        // some volume takes its density from a grid (its transmittance is tracked, not evaluated)
    const struct DensityGrid* grids;        // uNanoVDBBuffersDensity / uVolumeMaxDensities, densified (vpt_add_density_grid)
    vpt_atmosphere atm;
};

struct RenderParams {
    float view_inv[16], proj_inv[16];
    uint32_t width, height;
    uint32_t shard_rank, shard_count, shard_rows, shard_pixels;
    uint32_t samples_per_frame, max_depth;
    uint32_t split;                // ScreenSplitCount (RayGen.slang:16-25); 1 = every dispatch covers every pixel
    const uint32_t* launch_off;    // split > 1: prefix sums of the launch-grid sizes of the dispatches in the batch
    float max_luminance, focus_distance, dof_strength;
    float sky_azimuth, sky_altitude, sky_intensity, emissive_pdf_bias;
    uint32_t flags, base_seed;
    // Graph replays (vpt_render_async): when non-null, the batch's first dispatch index is read from here instead of the kernels'
    // dispatch_base / frame_base arguments, so that one captured batch serves every frame
    const uint32_t* dispatch_base_dev;
    // sincos_ of the four sky-rotation angles the shaders use, evaluated once per vpt_set_params with the shared fp32 contract
    // (vpt_api.hip sync_params): {sin, cos} of azimuth, altitude (ImportanceSampleEnvMap) and of -altitude, -azimuth (Miss)
    float sky_rot[8];
};

// ---- Wavefront path state: 16-byte records per slot (slot = frame_in_flight * shard_pixels +
// shard_pixel).  A path never moves; queues hold slot ids, so results cannot depend on queue order.
// Records are grouped by which stage touches them, so every access is one dwordx4 per lane: after
// compaction neighbouring lanes hold non-adjacent slots, and a 16 B record uses a fetched sector 4x
// better than 4 B structure-of-arrays gathers (and needs a quarter of the memory instructions).
struct PathState {
    uint32_t capacity;
    float4* A;       // payload.Origin.xyz | RNG state                        extend R, shade RW
    float4* B;       // payload.Direction.xyz | payload.Depth, bit31 = InMedium extend R, shade RW
    float4* T[2];    // pathThroughput.xyz | payload.PDF   ping-pong by bounce parity: shade(k) reads
                     // T[k&1] and writes T[(k+1)&1]; connect(k) still finds the pre-update throughput in T[k&1]
    float4* H;       // hit record t,u,v | PrimitiveIndex                      extend W, shade R
    uint32_t* hinst; // hit record InstanceIndex
    float4* CE;      // pending: emission / miss radiance .xyz | connect flags shade W, connect R
    float4* CS;      // pending: sky NEE contribution .xyz                     (only when a sky ray is queued)
    float4* CSO;     //          sky ray origin.xyz | dir.x
    float4* CSD;     //          sky ray dir.yz
    float4* CL;      // pending: light NEE contribution .xyz | global id of the sampled triangle
    float4* CLO;     //          light ray origin.xyz | dir.x
    float4* CLD;     //          light ray dir.yz
    float4* L;       // pathLight.xyz                                          connect RW   (round 1's stage kernels only: allocated with their records)
    float4* ACC;     // accumulatedLight of the frame (sum over samples_per_frame)  connect RW at path end
    float4* M;       // medium colour.rgb | density (only glass)               shade RW when refracting
    float* maniso;   // medium anisotropy
    uint32_t* sidx;  // sample index within the frame (samples_per_frame > 1 only)
    uint32_t* vdepth;  // payload.VolumeDepth (only touched while volumes are set)
    int32_t* cchan;    // payload.ColorChannel (only touched while the atmosphere is on)
};

// ---- Streams of the staged pipeline (kernels_stream.hip): records in the order the shade stage produced them, written by
// wave-private chunked appends (vote.hpp), so every read and write of them is coalesced.
struct StreamState {
    // path records in queue order, ping-pong by bounce parity: entry i of queue[p] is the path whose records are R*[p][i], so the
    // extend and shade stages read them as coalesced 16-byte streams (a surviving path's records MOVE to where its queue
    // entry goes; its frame sum, pathLight and medium stay addressed by slot)
    float4* RA[2];  // payload.Origin.xyz | RNG state
    float4* RB[2];  // payload.Direction.xyz | payload.Depth, bit31 = InMedium
    float4* RT[2];  // pathThroughput.xyz | payload.PDF
    float4* RL[2];  // pathLight.xyz: moves with the queue entry like the other records; the join stage adds to it where the path now is
    float4* SH;     // hit record t,u,v | PrimitiveIndex of queue entry i (extend W, shade R)
    uint32_t* SHI;  // hit record InstanceIndex
    // pending paths of this bounce (anything to join: emission, NEE candidates, end of sample)
    float4* PE;   // emission / miss radiance .xyz | connect flags
    float4* PS;   // sky NEE contribution .xyz | index of its ray in the sky-ray stream
    float4* PL;   // light NEE contribution .xyz | index of its ray in the light-ray stream
    float4* PT;   // pathThroughput before this bounce .xyz | the path's queue entry: in the next queue if kCF_Alive, else in this one (kHole: nobody wrote this entry)
    // shadow rays
    float4* SKO;  // sky rays: origin.xyz | dir.x
    float4* SKD;  //           dir.y, dir.z, 0xffffffff (0xfffffffe: hole), -
    float4* LTO;  // light rays: origin.xyz | dir.x
    float4* LTD;  //           dir.y, dir.z, global id of the sampled triangle (0xfffffffe: hole), -
    unsigned char* vis_sky;    // per sky ray: 1 = nothing hit
    unsigned char* vis_light;  // per light ray: 1 = the closest hit is the sampled triangle
    uint32_t cap;              // entries allocated per stream
};

// Streams of the media variant of the staged pipeline (kernels_media.hip), indexed by QUEUE ENTRY like the path records: allocated
// when a batch with volumes / atmosphere first runs on the streams pipeline (11 float4 per path).
struct MediaState {
    float4* MS;       // scatter decision: vol_index (int) | distance | atmosphere component + 1, bit 4 aborted, (colour channel + 1) << 8 | RNG state after the draws
    float4* MP[10];   // what shade_core<VOL> leaves for the tail stage (emission | rng, new origin | depth, new direction | pdf, BxDF | flags, the two
                      // NEE samples' ingredients, transmittance depths | VolumeDepth | ColorChannel, indices of the two shadow rays)
};

// Counters of the stream pipeline.  Every word several hundred waves hit with atomics sits in its OWN 256-byte block:
// atomics on one line serialise at ~11 ns each whichever word they address (MI355X_MICROARCH.md 'dequeue'), and the first
// version, with four stream lengths in one line, paid ~0.15 ms per launch for it.
struct alignas(256) HotWord {
    uint32_t v;
    uint32_t pad[63];
};
// Shade classes: the key the staged pipeline sorts its shade queue by (one queue and one shade launch per class, so a wave
// shades paths of ONE class).  The reference gets this grouping from hit-group / miss-shader dispatch
// (ClosestHit.slang:20, Miss.slang:8).  kShadePlain promises what its kernel instantiation relies on: every value texture
// and the normal map of the material are 1x1 (MatResolved.flags == 3), so no texel is ever fetched; the other hit classes
// run the general closest-hit code and differ only in what their waves have in common.
constexpr uint32_t kShadeMiss = 0, kShadePlain = 1, kShadeTextured = 2, kShadeGlass = 3, kShadeEmissive = 4, kShadeClasses = 5;
constexpr int kShadeAny = -1;  // template value: no class knowledge (fused kernels, round-1 shade stage)

struct StreamCounters {
    HotWord queue_len[2];   // ray queue lengths, holes included (ping-pong by bounce parity)
    HotWord alive[2];       // exact number of live paths in queue[p]: what the host and the resolve guard look at
    HotWord pend_len, sky_len, light_len;                // stream lengths, holes included
    HotWord extend_head, shade_head, sky_head, light_head;  // dynamic work cursors (work beyond each wave's static first 64 entries)
    HotWord finish_head;    // k_finish's cursor (0 between launches: k_finish_done resets it)
    // per shade class (filled by the extend stage's retire step, laid out by k_prepare_classes)
    HotWord class_len[kShadeClasses];    // class queue lengths (dense: the classify step writes no holes)
    HotWord class_head[kShadeClasses];   // shade work cursors
    uint32_t class_active[kShadeClasses];   // waves of the shade grid that take part in the class's launch
    uint32_t class_exact[kShadeClasses];    // 1: the launch appends exactly (short queue), 0: chunked with static first chunks
    uint32_t class_base[kShadeClasses];     // first entry of the launch's static chunks in every stream it appends to
    uint32_t classify_done;                 // blocks of the classify launch that have finished (the last one lays the streams out)
    // Path regeneration by refill (vpt_config.resident_frames, kernels_stream.hip k_refill_plan): a batch of F frames keeps at most
    // `cap` paths resident; behind every shade stage the free room of the next ray queue is filled with the batch's next unstarted
    // samples — a CONTIGUOUS run of sample ids, i.e. coherent camera rays appended as one block — until the samples are used up.
    uint32_t refill_next, refill_total;     // next unstarted sample id of the batch | samples of the batch (== : nothing left to start)
    uint32_t refill_entry, refill_first, refill_count;   // the plan of the current refill: first queue entry, first sample id, how many
};

// Stream appends (vote.hpp WaveAppender): chunk size of the wave-private chunked appends, and the queue length below which a launch
// appends exactly instead (no holes).  The host sizes the streams' slack for unwritten chunk tails by these (vpt_api.hip alloc_path_buffers).
constexpr uint32_t kAppendChunk = 256;
constexpr uint32_t kAppendExactBelow = 1u << 21;
// The fused per-bounce kernel's threshold.  Measured at 2^18 (round 4, profiles/r04_latency_probe_exact18.json: the 2M-path launches of a
// 1-frame batch at 1080p then append in chunks): bounce 0 234 -> 206 us, but the seven later bounces 605 -> 733 us (holes in queues
// that are short anyway, tiles of mixed holes without the regrouping ring) — slower overall, so it stays where the streams' is.
constexpr uint32_t kFusedExactBelow = kAppendExactBelow;

// connect flags (CE.w)
constexpr uint32_t kCF_Sky = 1u, kCF_Light = 2u, kCF_Finalize = 4u, kCF_Clamp = 8u;
constexpr uint32_t kCF_Alive = 16u;   // streams pipeline: the path lives on, PT.w is its entry in the NEXT queue (else: in this one)

struct Counters {
    uint32_t ray_count[2];    // active-path queue sizes (ping-pong by bounce parity)
    uint32_t connect_front;   // connect queue: entries with shadow rays grow from the front,
    uint32_t connect_back;    //                emission/finalize-only entries from the back
    uint32_t extend_head, connect_head;  // persistent-kernel work cursors
    uint32_t shadow_rays;     // shadow rays queued by shade this bounce
    uint32_t pad;
    // fused pipeline: three rotating queue sizes — bounce k reads rc3[k % 3], appends to rc3[(k + 1) % 3] and zeroes
    // rc3[(k + 2) % 3] (idle during bounce k), so no separate reset kernel sits between two bounces
    uint32_t rc3[4];          // dynamically reserved part of the three rotating queues' lengths
    uint32_t rc3_static[4];   // static part: one chunk per wave of the producing launch (0 when it appended exactly)
    uint32_t alive3[4];       // exact number of live paths in each of the three queues
    unsigned long long stat_closest, stat_shadow, stat_connect;  // folded per bounce by k_prepare / k_fold
    unsigned long long stat_nodes, stat_tris;                // extend kernel (count_traversal builds only)
    unsigned long long stat_shadow_nodes, stat_shadow_tris;  // connect kernel
    unsigned long long stat_primary_hits, stat_primary_alive, stat_primary_rays;  // bounce 0: hits, survivors, shadow rays
    unsigned long long stat_finish_paths, stat_finish_closest, stat_finish_shadow;  // k_finish (streams pipeline): paths taken over, closest-hit rays, shadow rays (both also in stat_closest / stat_shadow)
};

}  // namespace vpt
