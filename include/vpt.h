/* vpt.h — C-ABI of the MI355X wavefront render backend.
 *
 * Drop-in boundary for the hot path of Zydak/Vulkan-Path-Tracer.
 * The reference has no FFI; the seam is the C++ class surface its Editor calls:
 *   PathTracer     (PathTracer/PathTracer.h:83-183)
 *   PostProcessor  (PathTracer/PostProcessor.h:8-33)
 * Each entry point below names the reference member(s) it replaces.  The C++
 * facade in vulkan-path-tracer_amd/host/ (class PathTracer / PostProcessor /
 * FlyCamera with the reference's method names) is a thin shim over these calls;
 * INTEGRATION.md shows the binding a maintainer would add upstream.
 *
 * Conventions: plain pointers and sizes only; return 0 or a negative VPT_ERR_*;
 * never aborts; device memory is owned by the context, host buffers by the
 * caller; one host thread per context; matrices are float[16] column-major
 * exactly as glm stores a mat4; all struct layouts are the reference's scalar
 * layouts (Bindings.slang) so existing host data can be passed through.
 */
#ifndef VPT_H
#define VPT_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VPT_OK 0
#define VPT_ERR_INVALID_ARGUMENT (-1)
#define VPT_ERR_NO_DEVICE (-2)     /* HIP device/runtime missing: the backend never falls back to a CPU path */
#define VPT_ERR_OUT_OF_MEMORY (-3)
#define VPT_ERR_NO_SCENE (-4)
#define VPT_ERR_DEVICE (-5)        /* a HIP call failed; see vpt_last_error() */
#define VPT_ERR_UNSUPPORTED (-6)
#define VPT_ERR_LIMIT (-7)         /* MAX_ENTITIES / MAX_INSTANCES / MAX_EMISSIVE_MESHES, PathTracer.h:192-195 */

/* Limits, PathTracer.h:192-195 (asserts PathTracer.cpp:182-184). */
#define VPT_MAX_ENTITIES 10000u
#define VPT_MAX_INSTANCES 100000u
#define VPT_MAX_EMISSIVE_MESHES 10000u

/* Feature flags == the Slang #define set assembled at PathTracer.cpp:621-654. */
#define VPT_FLAG_SKY_MIS (1u << 0)             /* ENABLE_SKY_MIS            SetSkyMIS            */
#define VPT_FLAG_MESH_MIS (1u << 1)            /* ENABLE_MESH_MIS           SetMeshMIS           */
#define VPT_FLAG_SHOW_ENV_DIRECTLY (1u << 2)   /* SHOW_ENV_MAP_DIRECTLY     SetEnvMapShownDirectly */
#define VPT_FLAG_GEOMETRY_NORMALS (1u << 3)    /* USE_ONLY_GEOMETRY_NORMALS SetUseOnlyGeometryNormals */
#define VPT_FLAG_ENERGY_COMPENSATION (1u << 4) /* USE_ENERGY_COMPENSATION   SetUseEnergyCompensation */
#define VPT_FLAG_FURNACE (1u << 5)             /* FURNACE_TEST_MODE         SetFurnaceTestMode   */
#define VPT_FLAG_RAY_QUERIES (1u << 6)         /* USE_RAY_QUERIES: shadow / distance queries as ray queries (RTCommon.slang:52-63, 88-101: direction as it is,
                                                * TMin 1e-4 / 1e-5, TMax 1e6, closest committed hit).  CLEAR = the TraceRay forms (RTCommon.slang:64-84, 103-117,
                                                * MissShadow.slang:4-9): normalised direction, TMin 1e-5, TMax 1000, accept-first-hit.  Upstream, the light-identity
                                                * compare of an emissive-mesh NEE sample then reads payload.TriangleIdx / InstanceIdx, which nothing on that path
                                                * writes (closest-hit shader skipped, MissShadow leaves them alone): an UNDEFINED word.  Pinned here (oracle and
                                                * kernels alike) as "never equal": in this mode an emissive-mesh NEE sample is drawn and never visible, so meshes
                                                * light the scene through BSDF-sampled hits only; sky visibility and GetDistanceToGeometry are well defined and exact */
/* Tonemap.slang:170 samples the bloom image with a sampler whose filter is VulkanHelper's
 * default (PostProcessor.cpp:67, unpinned): set = LINEAR (default), clear = NEAREST. */
#define VPT_FLAG_TONEMAP_LINEAR_BLOOM_TAP (1u << 7)
/* Strict, structure-independent hit rule (no reference counterpart; off by default): a ray/triangle candidate only counts
 * where the ray is inside the triangle's own bounding box (vpt_fp32.h hit_is_local).  Without it, a ray grazing a small
 * triangle at < 1e-3 rad can be given a hit up to ~1e-2 scene units off the triangle by fp32 rounding, which a box
 * hierarchy sees or not depending on its shape — about one ray in 1e9, 1e-11 relative L2 on a 531 M-sample image.
 * With it the image equals brute-force intersection bit for bit at any size.  Cost in throughput, measured at 1080p on the default
 * pipelines (profiles/r04_strict_rate.json): Cornell box (fused per-bounce kernels) 11.5 %, atrium (streams) 6.8 %, glass bust (streams, depth 32) 0.5 %. */
#define VPT_FLAG_LOCAL_HITS (1u << 8)
#define VPT_FLAGS_DEFAULT                                                                                   \
    (VPT_FLAG_SKY_MIS | VPT_FLAG_MESH_MIS | VPT_FLAG_SHOW_ENV_DIRECTLY | VPT_FLAG_ENERGY_COMPENSATION |    \
     VPT_FLAG_RAY_QUERIES | VPT_FLAG_TONEMAP_LINEAR_BLOOM_TAP) /* PathTracer.h:211-221 */

/* == PathTracer::Material (PathTracer.h:12-34) == CPUMaterial (Bindings.slang:55-77); 112 bytes. */
typedef struct vpt_material {
    float base_color[3];
    float emissive_color[3];
    float specular_color[3];
    float medium_color[3];
    float medium_emissive_color[3];
    float metallic;
    float roughness;
    float ior;
    float transmission;
    float anisotropy;
    float anisotropy_rotation;
    float medium_density;
    float medium_anisotropy;
    uint32_t base_color_texture;
    uint32_t normal_texture;
    uint32_t roughness_texture;
    uint32_t metallic_texture;
    uint32_t emissive_texture;
} vpt_material;

/* == Vertex (Bindings.slang:7-12) == VulkanHelper::LoadedMeshVertex (PathTracer.cpp:190-194); 32 bytes. */
typedef struct vpt_vertex {
    float position[3];
    float normal[3];
    float texcoord[2];
} vpt_vertex;

/* == VulkanHelper Mesh as used at PathTracer.cpp:204-225. */
typedef struct vpt_mesh {
    const vpt_vertex* vertices;
    uint32_t vertex_count;
    const uint32_t* indices;
    uint32_t index_count;
} vpt_mesh;

/* == VulkanHelper::MeshInstance as used at PathTracer.cpp:449-481. */
typedef struct vpt_instance {
    uint32_t mesh_index;
    uint32_t material_index;
    float transform[16];
} vpt_instance;

/* == TextureAsset, PathTracer.cpp:815-836: channels 4 = RGBA8 UNORM, 1 = R8 UNORM. */
typedef struct vpt_texture {
    uint32_t width;
    uint32_t height;
    uint32_t channels;
    const uint8_t* data;
} vpt_texture;

/* Scene as PathTracer::SetScene assembles it (PathTracer.cpp:158-676).  Texture indices in the
 * materials index `textures`.  `env_rgba` is the RGBA32F environment image (alpha is ignored;
 * importance, alias table and the per-texel pdf in alpha are derived by the backend exactly as
 * LoadEnvironmentMap does, PathTracer.cpp:1137-1332).  The three energy-compensation tables are
 * the contents of Assets/LookupTables/ (.bin files; 64x64x32, 128x128x32, 128x128x32 floats). */
typedef struct vpt_scene_desc {
    const vpt_mesh* meshes;
    uint32_t mesh_count;
    const vpt_material* materials;
    uint32_t material_count;
    const vpt_instance* instances;
    uint32_t instance_count;
    const vpt_texture* textures;
    uint32_t texture_count;
    const float* env_rgba;
    uint32_t env_width;
    uint32_t env_height;
    const float* lut_reflection;
    const float* lut_refraction_outside;
    const float* lut_refraction_inside;
} vpt_scene_desc;

/* Scalar state of PathTracer (PathTracer.h:197-233) that reaches the shaders through
 * PathTracerUniform (PathTracer.h:271-302) plus the feature-flag set, plus the one thing the
 * reference does not expose: the seed.  Reference: Seed = PCGHash(wall-clock ms)
 * (PathTracer.cpp:127-140); here Seed(dispatch k) = PCGHash(base_seed + k). */
typedef struct vpt_params {
    uint32_t samples_per_frame; /* SetSamplesPerFrame, default 1 */
    uint32_t max_samples;       /* SetMaxSamplesAccumulated, default 5000 */
    uint32_t max_depth;         /* SetMaxDepth, default 200 */
    float max_luminance;        /* SetMaxLuminance, default 500 */
    float focus_distance;       /* SetFocusDistance, default 1 */
    float dof_strength;         /* SetDepthOfFieldStrength, default 0 */
    float sky_azimuth;          /* SetSkyAzimuth, degrees */
    float sky_altitude;         /* SetSkyAltitude, degrees */
    float sky_intensity;        /* SetSkyIntensity, default 1 */
    uint32_t screen_chunk_count;/* SetSplitScreenCount, default 1 */
    float emissive_pdf_bias;    /* SetEmissiveMeshSamplingPDFBias, default 0 */
    uint32_t flags;             /* VPT_FLAG_* */
    uint32_t base_seed;
} vpt_params;

/* PostProcessor::TonemappingData + BloomData (PostProcessor.h:8-21). */
typedef struct vpt_post_params {
    float exposure;        /* 1.0 */
    float gamma;           /* 2.2 */
    float bloom_threshold; /* 2.0 */
    float bloom_strength;  /* 1.0 */
    uint32_t mip_count;    /* 10 */
    float falloff_range;   /* 5.0 */
    uint32_t schedule;     /* VPT_POST_FUSED (default) | VPT_POST_REFERENCE_PASSES; the output is identical, byte for byte */
} vpt_post_params;
#define VPT_POST_FUSED 0u            /* threshold inside the first down-sample; middle mips down in one launch, up in one; the smallest down and up in one; last up-sample + tonemap in one */
#define VPT_POST_REFERENCE_PASSES 1u /* one kernel per pass as PostProcessor.cpp:193-246 records them (the A/B baseline) */

typedef struct vpt_config {
    int device;          /* HIP device ordinal */
    uint32_t width;      /* output image, RGBA32F (PathTracer.cpp:507-512, 698) */
    uint32_t height;
    uint32_t shard_rank; /* this context renders rows y with y % shard_count == shard_rank */
    uint32_t shard_count;/* 1 = whole image */
    uint32_t frames_in_flight; /* largest batch (frames per wavefront batch) the context will hold; 0 = ~448M paths (226 frames at 1080p), never more than 60 % of
                                * the free device memory, at most 8192 frames.  This is a CAP, not an allocation: vpt_create allocates the path records of ONE
                                * frame (~0.8 GB at 1080p) and the buffers grow to the largest batch a vpt_render / vpt_render_async call actually asks for
                                * (min(dispatches, cap) frames, 380 B per path; vpt_stats.frames_allocated) — an interactive host that renders a frame per
                                * call never holds more than that one frame.  Contexts whose batches run as whole-path launches (VPT_PIPELINE_WHOLE, or AUTO
                                * where it applies) keep their paths in registers and allocate 36 B per sample + the records of one frame
                                * (vpt_stats.resident_frames == 1).
                                * DEFAULT SCHEDULE (this field 0 AND resident_frames 0): a context that regenerates paths (see resident_frames) or runs whole-path
                                * launches takes batches of 4 x F frames, F being the cap above (904 frames at 1080p), with F / 2 frames of paths resident
                                * (113; none for whole-path launches): 126 GB on a 1080p scene whose BVH lives in memory (profiles/r05_frames_sweep.json).
                                * To CAP THE MEMORY of a drop-in host set this field: e.g. frames_in_flight = 64, resident_frames = 16 holds
                                * 64 x pixels x 36 B + 16 x pixels x 290 B = 14.4 GB at 1080p (INTEGRATION.md "Device memory"). */
    uint32_t profile;    /* 1 = bracket every kernel launch with hipEvents (vpt_get_stats kernel times) */
    uint32_t count_traversal; /* 1 = count BVH node/triangle visits (slower; for the roofline's algorithmic bytes) */
    uint32_t pipeline;   /* VPT_PIPELINE_* */
    uint32_t build_flags; /* VPT_BUILD_*: how vpt_set_scene builds the BVH of this context (reported back in vpt_stats.build_flags) */
    /* Path regeneration: frames of PATHS a batch keeps in flight at a time.  A batch of F frames has F x pixels samples; with K < F resident
     * frames the camera-ray launch starts the first K x pixels samples, and behind every shade stage the room that ended paths left in the
     * next ray queue is REFILLED with the batch's next unstarted samples — a contiguous run of sample ids, i.e. a block of coherent camera
     * rays appended behind the survivors (seeds depend on pixel and frame only, a sample's result lands in its own slot of the frame sums
     * and the running mean is applied in frame order when the batch has finished) — so every launch works on ~K x pixels paths until the
     * samples run out: ~290 B per resident path + 36 B per sample instead of 380 B per sample.  Applies to the streams pipeline (scenes whose
     * BVH lives in memory; VPT_PIPELINE_AUTO / _STAGED / _STAGED_SORTED).  Whole-path launches (scenes that ride in LDS) hold no path records
     * at all, whatever this says; the fused per-bounce kernels, media batches, split-screen dispatch and VPT_PIPELINE_STAGED_R1 keep every
     * sample resident.  0 (default): with frames_in_flight == 0 too, batches of 4 x F frames keep F / 2 frames of paths resident (the default schedule
     * described at frames_in_flight); with an explicit frames_in_flight, every sample resident.  A value >= the batch size: every sample resident.
     * Measured trade: profiles/r05_frames_sweep.json
     * (DESIGN.md §3 "Regeneration by refill").  Images do not depend on it, bit for bit.  Reference loop being unrolled: RayGen.slang:28-33,116-159. */
    uint32_t resident_frames;
} vpt_config;

/* Spatial splits in the BVH builder (bvh_build.hpp): identical images, pays on scenes of uneven triangle sizes only (profiles/REJECTED.md).
 * Per context, never read from the environment: two contexts of one process cannot silently build different trees. */
#define VPT_BUILD_SBVH 1u
/* Keep the general instantiation of the whole-path / fused per-bounce kernels even when the scene qualifies for the class-specialised one
 * (every texture 1x1 and a black environment: k_whole<PLAIN>, k_bounce<PLAIN>, kernels_path.hip).  Images are identical; this is the A/B switch of that choice. */
#define VPT_BUILD_GENERAL_KERNELS 2u
/* Streams pipeline: never hand the rest of a batch to the one-launch finisher (kernels_path.hip k_finish), i.e. run every bounce of every batch
 * through the stream stages.  By default a small batch (<= 6M samples: a frame or two per call) goes there after three bounces and a large one once
 * the host sees fewer than 262,144 paths alive: seven dependent launches per bounce on a short queue cost more than the finisher's one launch of
 * persistent waves.  Images are identical; this is the A/B switch of that choice. */
#define VPT_BUILD_STREAMS_ONLY 4u

/* AUTO = WHOLE where it applies (BVH in LDS, no media, one sample per pixel and frame), FUSED for the other scenes whose BVH fits in LDS next to
 * the traversal stacks, else STAGED.  Results are identical. */
#define VPT_PIPELINE_AUTO 0u
#define VPT_PIPELINE_FUSED 1u   /* one kernel per bounce */
#define VPT_PIPELINE_STAGED 2u  /* extend -> shade -> connect with compacted queues; traversal on the vote-scheduled persistent kernels */
#define VPT_PIPELINE_STAGED_SORTED 4u /* STAGED with the shade queue sorted by material class (miss | plain | textured | glass | emissive),
                                       * one shade launch per class, the miss and plain ones specialised.  Bit-identical; measured
                                       * 10-14 % SLOWER than STAGED on the BASELINE scenes (profiles/REJECTED.md), so AUTO never picks it */
#define VPT_PIPELINE_WHOLE 5u  /* ONE launch per batch: persistent waves run every path from its camera ray to its end, a lane whose path has ended takes
                               * the batch's next sample (kernels_path.hip k_whole; the reference's own shape: one RayGen thread = one whole path).
                               * For scenes whose BVH rides in LDS, no media, samples_per_frame == 1, every sample resident — VPT_ERR_UNSUPPORTED
                               * otherwise.  Bit-identical.  AUTO takes it wherever it applies: measured faster than FUSED at every batch size
                               * (Cornell box 1080p: 8.3 vs 7.4 Gsamples/s at 226 frames per batch, 3.8 vs 2.4 at one; profiles/r04_whole_ab.json) */
#define VPT_PIPELINE_STAGED_R1 3u /* LABORATORY build only (include/vpt_lab.h, libvpt_hip_lab.so): the same stages with round 1's kernels (slot-addressed records, 64 rays per
                                   * wave at a time), kept as the measured baseline.  vpt_create of the product library answers VPT_ERR_UNSUPPORTED */

#define VPT_KERNEL_COUNT 10
enum vpt_kernel_id {
    VPT_K_PRIMARY = 0,  /* whole-path pipeline: the batch's one launch (k_whole); fused pipeline: bounce 0 (camera ray + extend + shade + connect);
                         * staged pipeline: raygen */
    VPT_K_EXTEND = 1,
    VPT_K_SHADE = 2,
    VPT_K_CONNECT = 3,  /* shadow rays + light accumulation + end-of-sample in one kernel (VPT_PIPELINE_STAGED_R1, laboratory build) */
    VPT_K_BOUNCE = 4,   /* bounce >= 1 fused (LDS-resident scenes): extend + shade + connect in one kernel */
    VPT_K_RESOLVE = 5,
    VPT_K_BLOOM = 6,
    VPT_K_TONEMAP = 7,
    VPT_K_SHADOW = 8,   /* staged pipeline: shadow-ray streams (sky rays, light rays) traced as any-hit searches */
    VPT_K_JOIN = 9      /* staged pipeline: NEE contributions joined with the emission, pathLight, end of sample */
};

typedef struct vpt_stats {
    uint64_t samples;          /* camera paths finished (GetSamplesAccumulated * pixels) */
    uint64_t frames;           /* m_FrameCount */
    uint64_t dispatches;       /* m_DispatchCount */
    uint64_t closest_rays;     /* rays traced by the extend kernel */
    uint64_t shadow_rays;      /* rays traced by the connect kernel */
    uint64_t connect_paths;    /* path-bounces that went through the connect kernel (staged pipelines); whole-path launches: hits of bounces >= 1, whose
                                * pathLight waits in the frame-sum slot while the hit is parked (16 B written + 16 B read each) */
    uint64_t primary_hits;     /* camera rays that hit geometry */
    uint64_t primary_survivors;   /* paths that continue after bounce 0 */
    uint64_t primary_shadow_rays; /* shadow rays traced inside the primary kernel */
    uint64_t nodes_visited;    /* extend kernel, only when count_traversal */
    uint64_t tris_tested;      /* extend kernel, only when count_traversal */
    uint64_t shadow_nodes_visited; /* connect kernel, only when count_traversal */
    uint64_t shadow_tris_tested;   /* connect kernel, only when count_traversal */
    uint64_t kernel_launches[VPT_KERNEL_COUNT];
    double kernel_ms[VPT_KERNEL_COUNT]; /* only when profile */
    uint64_t total_vertex_count; /* GetTotalVertexCount */
    uint64_t total_index_count;  /* GetTotalIndexCount */
    uint32_t bvh_nodes;
    uint32_t bvh_triangles;
    uint32_t bvh_node_bytes;
    uint32_t bvh_tri_bytes;
    uint32_t emissive_mesh_count;
    uint32_t emissive_triangle_count;
    uint32_t frames_in_flight; /* the largest batch this context renders at once, in frames: NOT an echo of vpt_config.frames_in_flight — on the default
                                * schedule it is 4 x that cap (see vpt_config.frames_in_flight) */
    uint32_t shard_pixels;
    uint32_t bvh8_nodes;       /* laboratory build: eight-wide nodes of the BVH8 experiment (0 in the product library) */
    uint32_t build_flags;      /* VPT_BUILD_* the scene's BVH was built with */
    uint32_t frames_allocated; /* frames of samples the buffers currently hold (grows with the largest batch requested, <= frames_in_flight) */
    uint32_t resident_frames;  /* frames of paths the queues / stream records hold (vpt_config.resident_frames; < frames_allocated when paths are regenerated) */
    uint32_t reserved0;
    uint32_t graph_launches;   /* batches replayed from a captured hipGraph by vpt_render_async since vpt_reset_stats */
    /* Words of the traversal stacks' global SPILL regions written since vpt_set_scene (a lane's stack is 14 LDS entries, deeper entries
     * spill to a per-thread region; vpt_api.hip keeps one region per concurrently running traversal grid): [0] the context's main
     * stream, [1] the second stream the shadow kernels of bounce k run on beside the extend of bounce k + 1.  Counted from the regions
     * themselves (they are preset to a pattern no stack entry can be), so the figure belongs to the product kernels, not to counting variants. */
    uint64_t stack_spills[2];
    double set_scene_ms;       /* wall time of the last vpt_set_scene (validation, BVH build, uploads, derived tables) */
    double bvh_build_ms;       /* of which: the host-side BVH build (bvh_build.cpp; the reference builds BLAS / TLAS on the device, PathTracer.cpp:484-505) */
    /* Streams pipeline, the one-launch finisher (kernels_path.hip k_finish, timed under VPT_K_BOUNCE): paths it took over from the streams, and the
     * closest-hit / shadow rays it traced — both also counted in closest_rays / shadow_rays above. */
    uint64_t finish_paths;
    uint64_t finish_closest_rays;
    uint64_t finish_shadow_rays;
} vpt_stats;

typedef struct vpt_ctx vpt_ctx;

/* PathTracer::New (PathTracer.cpp:21-120) + CreateOutputImageView (692-710). NULL on failure;
 * *err (if non-NULL) receives the VPT_ERR_* code.  Fails with VPT_ERR_NO_DEVICE when no HIP
 * device is usable: there is no CPU fallback behind this API. */
vpt_ctx* vpt_create(const vpt_config* cfg, int* err);
void vpt_destroy(vpt_ctx* ctx);
const char* vpt_last_error(const vpt_ctx* ctx);

/* PathTracer::SetScene (PathTracer.cpp:158-676): uploads geometry/materials/textures, derives the
 * emissive-mesh list (449-469) and the env importance/alias tables (1137-1332), builds the BVH
 * (replaces BLASBuilder/TLAS, 488-505; builder options: vpt_config.build_flags), resets accumulation. Arrays are borrowed
 * for the call only. */
int vpt_set_scene(vpt_ctx* ctx, const vpt_scene_desc* scene);
/* PathTracer::SetMaterial (PathTracer.cpp:712-810): patches one material, rebuilds the emissive list
 * if emission changed, resets accumulation. */
int vpt_set_material(vpt_ctx* ctx, uint32_t index, const vpt_material* material);
int vpt_get_material(const vpt_ctx* ctx, uint32_t index, vpt_material* out);
/* ---- participating media: homogeneous box volumes ---------------------------------
 * PathTracer::Volume / VolumeGPU (PathTracer.h:36-74, 341-400) as the shaders read it (Volume.slang:19-52).
 * corner_min / corner_max are the WORLD-space box, i.e. Position + Corner * Scale already applied
 * (PathTracer.h:395-396).  Heterogeneous volumes take their density from a DENSE grid (vpt_add_density_grid below:
 * the OpenVDB / NanoVDB tree of the reference, densified), including emission from temperature / blackbody.
 * The integrator side is RayGen.slang:162-380 (free-flight sampling per box,
 * nearest scatter vs. distance to geometry, NEE towards sky and emissive meshes through every box's Beer-Lambert
 * transmittance, phase-function scattering) and ClosestHit.slang:332-333,364 (volumes shadow surface NEE). */
typedef struct vpt_volume {
    float corner_min[3];
    float corner_max[3];
    float color[3];              /* single-scattering albedo */
    float emissive_color[3];
    float density;               /* extinction per unit length */
    float anisotropy;            /* g of Henyey-Greenstein / Draine */
    float alpha;                 /* Draine alpha */
    float droplet_size;          /* HG+Draine fit parameter d (micrometres) */
    int32_t density_data_index;  /* -1: homogeneous; >= 0: a grid added with vpt_add_density_grid */
    int32_t approximated_scattering;          /* ApproximatedScatteringForClouds: g^(1+depth), density * falloff^depth */
    float approximated_scattering_falloff;
    float grid_sharpness;                     /* GridSharpness (heterogeneous only) */
    /* Emission from temperature (Volume.slang:233-258).  Upstream the host writes the normalised temperature INTO the
     * density grid wherever it is positive (PathTracer.cpp:1444-1454) and the shader reads that same grid
     * (Volume.slang:238): has_temperature_data = 1 means "the grid attached to this volume is that merged grid". */
    int32_t has_temperature_data;
    int32_t use_blackbody;                    /* 1: Blackbody(kelvin), 0: temperature_color */
    float temperature_color[3];
    float temperature_gamma, temperature_scale, emissive_color_gamma;
    int32_t kelvin_min, kelvin_max;
} vpt_volume;
#define VPT_MAX_VOLUMES 32       /* the reference sorts into fixed float[100] / int[100] arrays (RayGen.slang:165-166) */
#define VPT_PHASE_HENYEY_GREENSTEIN 0        /* PathTracer.h:76-81 PhaseFunction */
#define VPT_PHASE_DRAINE 1
#define VPT_PHASE_HENYEY_GREENSTEIN_PLUS_DRAINE 2
/* AddVolume / RemoveVolume / SetVolume (PathTracer.h:157-159): the whole list is replaced; count 0 removes all
 * volumes.  Resets accumulation. */
int vpt_set_volumes(vpt_ctx* ctx, const vpt_volume* volumes, uint32_t count);
/* AddDensityDataToVolume (PathTracer.cpp:1347-1516) with the .vdb file already decoded by the caller: a dense grid of
 * raw densities over the active-voxel bounding box (x fastest, then y, then z, in the file's index order).  The library
 * does what the reference does after reading the file: MaxDensityInTheGrid, the 32x32x32 table of per-block maxima of
 * density / max used for empty-space skipping (y flipped "for Vulkan", :1425-1442), upload.  Lookups mirror
 * SampleNanoVDBBuffer (Volume.slang:69-117): position normalised in the box, y flipped, floor to a voxel, +-1 voxel
 * of random jitter per axis (three PCG draws), clamp to the grid.  Returns the grid's index (the value to put in
 * vpt_volume.density_data_index) or a negative VPT_ERR_*.  At most VPT_MAX_DENSITY_GRIDS grids. */
#define VPT_MAX_DENSITY_GRIDS 16
int vpt_add_density_grid(vpt_ctx* ctx, uint32_t dim_x, uint32_t dim_y, uint32_t dim_z, const float* density);
int vpt_clear_density_grids(vpt_ctx* ctx);   /* RemoveDensityDataFromVolume for all; volumes must not reference grids afterwards */
/* SetPhaseFunction (PathTracer.h:106); default VPT_PHASE_HENYEY_GREENSTEIN (PathTracer.h:219).  Resets accumulation. */
int vpt_set_phase_function(vpt_ctx* ctx, uint32_t phase_function);
/* ---- atmosphere ----------------------------------------------------------------------
 * SetEnableAtmosphere + the planet / density setters (PathTracer.h:168-179; defaults :221-232; UBO fields
 * :276-288).  With an atmosphere the sky is no environment map: rays that leave the scene return black
 * (Miss.slang:11-14) and all sky light is in-scattered sunlight — delta-tracked scatter events on Rayleigh / Mie /
 * ozone profiles (Atmosphere.slang:131-201, RayGen.slang:212-262,382-470), ONE colour channel per path once it has
 * scattered (ColorChannel, RayGen.slang:120-129), sun-disk NEE (Sampler.slang:431-462; its direction comes from
 * sky_azimuth / sky_altitude) through ratio-tracked transmittance (Atmosphere.slang:33-107).  Units are metres.
 * Runs on the fused pipeline. */
typedef struct vpt_atmosphere {
    float planet_position[3];
    float planet_radius;
    float atmosphere_height;
    float rayleigh_density_falloff, mie_density_falloff, ozone_density_falloff, ozone_peak;
    float rayleigh_multiplier[3];   /* RayleighScatteringCoefficientMultiplier */
    float mie_multiplier[3];        /* MieScatteringCoefficientMultiplier */
    float ozone_multiplier[3];      /* OzoneAbsorptionCoefficientMultiplier */
    float sun_color[3];
} vpt_atmosphere;
void vpt_default_atmosphere(vpt_atmosphere* out);                       /* PathTracer.h:222-232 */
int vpt_set_atmosphere(vpt_ctx* ctx, const vpt_atmosphere* atmosphere); /* NULL: SetEnableAtmosphere(false). Resets accumulation. */
/* SetCameraViewInverse / SetCameraProjectionInverse. */
int vpt_set_camera(vpt_ctx* ctx, const float view_inverse[16], const float projection_inverse[16]);
/* All scalar setters + the #define toggles (PathTracer.cpp:1010-1015, 1623-1716). Resets accumulation — except when
 * max_samples is the only field that changed (SetMaxSamplesAccumulated keeps the image, PathTracer.cpp:1003-1006). */
int vpt_set_params(vpt_ctx* ctx, const vpt_params* params);
void vpt_default_params(vpt_params* params);
void vpt_default_post_params(vpt_post_params* params);
/* PathTracer::ResizeImage. */
int vpt_resize(vpt_ctx* ctx, uint32_t width, uint32_t height);
/* PathTracer::ResetPathTracing (PathTracer.h:183). */
int vpt_reset(vpt_ctx* ctx);

/* PathTracer::PathTrace x dispatches (PathTracer.cpp:122-156); blocking. *done (if non-NULL) is
 * PathTrace's return value: 1 once samples accumulated >= max_samples (then nothing is launched). */
int vpt_render(vpt_ctx* ctx, uint32_t dispatches, int* done);

/* ---- the reference's asynchronous per-frame shape --------------------------------------------------------------------
 * PathTracer::PathTrace(cmd) RECORDS the dispatch and returns (PathTracer.cpp:122-156); Editor::Draw calls it once and then
 * PostProcessor::PostProcess(cmd) every frame (Editor.cpp:116,129); both outputs stay on the device (PathTracer.h:94-95
 * GetOutputImage, PostProcessor GetOutputImageView) and the host waits on a fence of an EARLIER frame.  Same here:
 *   vpt_render_async      enqueues `dispatches` dispatches on the context's stream and returns; bookkeeping (frame count, *done) advances
 *                         at enqueue time as PathTrace's does.  *ticket (optional) identifies the work enqueued so far.
 *   vpt_postprocess_device the post chain behind it on the same stream, no synchronisation, RGBA8 left in the context's output image
 *                         (vpt_output_device) and, if rgba8_device != NULL, copied there device-to-device (an interop / swapchain image).
 *   vpt_wait(ticket)      blocks until that ticket's work has finished (0: everything enqueued so far).
 * When every path of a batch provably ends within max_depth * samples_per_frame bounces (no material scatters inside a medium, no
 * volumes / atmosphere) and that number is <= VPT_ASYNC_MAX_BOUNCES — or when the batch is ONE whole-path launch (VPT_PIPELINE_WHOLE, which
 * AUTO takes for LDS-resident scenes: every path runs to its end inside it, whatever max_depth is) — a batch is a fixed schedule: nothing in
 * it waits for the host, any number of frames can be in flight, and such a batch — the whole-path / fused pipelines', and the streams pipeline's ~7 launches
 * per bounce on one stream — is captured once as a hipGraph and replayed while nothing changes (vpt_stats.graph_launches).  Otherwise the enqueued part is the first VPT_ASYNC_MAX_BOUNCES bounces and the NEXT call on the context
 * (or vpt_wait) finishes the batch first, exactly as vpt_render would have.  Images are bit-identical to vpt_render's either way.
 * Every other entry point that reads or changes device state drains outstanding work first. */
#define VPT_ASYNC_MAX_BOUNCES 32u
int vpt_render_async(vpt_ctx* ctx, uint32_t dispatches, int* done, uint64_t* ticket);
int vpt_postprocess_device(vpt_ctx* ctx, const vpt_post_params* params, void* rgba8_device, uint64_t* ticket);
int vpt_wait(vpt_ctx* ctx, uint64_t ticket);
/* GetOutputImageView(): device pointer to the RGBA8 UNORM image of the last vpt_postprocess / vpt_postprocess_device (width*height*4 bytes,
 * owned by the context, valid until vpt_resize / vpt_destroy); NULL before the first post-process. */
const void* vpt_output_device(vpt_ctx* ctx);
/* The same image read back (waits for outstanding work): width*height*4 bytes to host memory.  VPT_ERR_INVALID_ARGUMENT before the first post-process. */
int vpt_get_output(vpt_ctx* ctx, uint8_t* rgba8_host);

/* GetOutputImage(): the RGBA32F accumulation image (alpha 1). Whole image (shard_count==1, or after
 * vpt_assemble_shards) to a caller-owned host / device buffer of width*height*4 floats. */
int vpt_get_radiance(vpt_ctx* ctx, float* rgba_host);
int vpt_get_radiance_device(vpt_ctx* ctx, void* rgba_device);
/* Checkpoint/resume hook: overwrite the accumulation image and frame counter. */
int vpt_set_radiance(vpt_ctx* ctx, const float* rgba_host, uint32_t frame_count);

/* Multi-GPU: this context's rows (y % shard_count == shard_rank), packed in increasing y, as
 * rows*width*4 floats in device memory (the buffer handed to the RCCL gather), and the inverse on
 * the gathering rank: `gathered` holds shard_count shards back to back, each padded to
 * vpt_shard_floats(ctx) floats. */
size_t vpt_shard_floats(const vpt_ctx* ctx);
int vpt_get_shard_device(vpt_ctx* ctx, void* shard_device);
int vpt_assemble_shards(vpt_ctx* ctx, const void* gathered_device, uint32_t shard_count);

/* The one collective of the path.  The reference's only image partition is the interleaved
 * split-screen chunking of RayGen.slang:16-25 / PathTracer.cpp:141-153; here shard r owns rows y % N == r and
 * the finished shards meet ONCE, on `root`, over xGMI.  Nothing else ever crosses devices.
 *   process per GPU (torchrun / mpirun launch): vpt_comm_unique_id() on one rank, the 128 bytes handed to the
 *     others out of band (any control plane), vpt_comm_init() = ncclCommInitRank on the context's device,
 *     vpt_comm_gather_shards() = ncclGather of the padded shards (vpt_shard_floats each) on the context's stream
 *     followed, on root, by the row re-interleave: root's vpt_get_radiance / vpt_postprocess then see the whole
 *     image.  rank must equal the context's shard_rank and world its shard_count.
 *   one process driving N devices (the C++ host, `vpt_render --gpus N`): vpt_multi_gather_shards() over the N
 *     contexts, shard k at index k: peer copies (hipMemcpyPeerAsync, direct xGMI links) into root + re-interleave.
 * Both return VPT_ERR_DEVICE with the RCCL / HIP message in vpt_last_error() on failure. */
#define VPT_COMM_ID_BYTES 128
int vpt_comm_unique_id(void* id_out);
int vpt_comm_init(vpt_ctx* ctx, const void* id, int rank, int world);
int vpt_comm_gather_shards(vpt_ctx* ctx, int root);
int vpt_comm_destroy(vpt_ctx* ctx);
/* What the communicator actually is: the RCCL the process has MAPPED (a host that loaded another librccl first — PyTorch ships its
 * own — gets that one behind this library's calls, whatever it was linked against), the one this library was compiled against,
 * and what RCCL itself reports for the communicator.  vpt_comm_init refuses (VPT_ERR_DEVICE) a mapped RCCL whose major version
 * differs from the compiled one.  Valid after vpt_comm_init; without a communicator nranks = 0 and only the versions / path are filled. */
typedef struct vpt_comm_info {
    int32_t rccl_version_runtime;   /* ncclGetVersion() of the mapped library, e.g. 22606 */
    int32_t rccl_version_compiled;  /* NCCL_VERSION_CODE of the headers this library was built with */
    int32_t nranks;                 /* ncclCommCount */
    int32_t rank;                   /* ncclCommUserRank */
    int32_t device;                 /* ncclCommCuDevice */
    int32_t reserved;
    char library_path[232];         /* file the ncclGather symbol in use comes from (dladdr) */
} vpt_comm_info;
int vpt_comm_get_info(vpt_ctx* ctx, vpt_comm_info* out);
/* "<PCI bus id>" of the context's device (e.g. 0000:05:00.0): lets the host's control plane refuse two ranks on one device before
 * ncclCommInitRank is entered (RCCL rejects that configuration itself, but only after its bootstrap has run). out: >= 32 bytes. */
int vpt_device_identity(vpt_ctx* ctx, char* out, uint32_t out_bytes);
int vpt_multi_gather_shards(vpt_ctx* const* ctxs, uint32_t count, uint32_t root);

/* PostProcessor::SetTonemappingData/SetBloomData + PostProcess (PostProcessor.cpp:193-246) on the
 * accumulation image: threshold -> (mip-1)x down -> (mip-1)x up -> tonemap. rgba8_host receives
 * GetOutputImageView() (RGBA8 UNORM, width*height*4 bytes); bloom0_host (optional) receives bloom
 * mip 0 after the up-sample chain (RGBA32F) for testing. */
int vpt_postprocess(vpt_ctx* ctx, const vpt_post_params* params, uint8_t* rgba8_host, float* bloom0_host);

int vpt_get_stats(vpt_ctx* ctx, vpt_stats* out);
int vpt_reset_stats(vpt_ctx* ctx);

/* Test hook on the extend kernel alone: closest hit of n rays (origin xyz, tmin, dir xyz, tmax —
 * 32 B each, host memory) -> n hits {t, u, v, primitive, instance} (20 B each). */
typedef struct vpt_ray { float origin[3]; float tmin; float direction[3]; float tmax; } vpt_ray;
typedef struct vpt_hit { float t; float u; float v; uint32_t primitive; uint32_t instance; } vpt_hit;
int vpt_trace_rays(vpt_ctx* ctx, const vpt_ray* rays_host, uint32_t n, vpt_hit* hits_host);

/* ---- energy-compensation lookup tables -------------------------------------------
 * Replaces LookupTableCalculator::CalculateTable(tableSize, sampleCount) (LookupTableCalculator.cpp:44-157)
 * with its shaders LookupReflect.slang / LookupRefract.slang (+ ABOVE_SURFACE / BELOW_SURFACE): sampleCount/20
 * passes of 20 samples per cell, pass i seeded with PCG(i*2 + sampleCount + PCG(time_ms)), summed in pass
 * order and divided by the pass count.  The reference puts wall-clock milliseconds into time_ms (one reading
 * per pass); here it is one caller-chosen value, so a table is reproducible.  Application.cpp:41,54,67 use
 * sizes 64x64x32 (reflect) and 128x128x32 (refract) with 10'000'000 samples.
 * out_host receives size_x*size_y*size_z floats, x fastest.  Needs no context (runs on `device`).
 * Known disagreement with the shipped tables (tests/test_oracle_lut_fp64.py): in the grazing near-mirror corner of
 * the two refraction tables (rows y <= 4 with x <= 31, and layer z = 0) this generator — like a float64 evaluation of the same
 * algorithm — gives 0.896 where Assets/LookupTables/RefractionLookup*.bin hold 0.819.  Everywhere else the generated tables match the
 * shipped ones within Monte-Carlo error.  Rendering is unaffected: the integrator consumes whichever tables vpt_set_scene is given. */
#define VPT_LUT_REFLECT 0
#define VPT_LUT_REFRACT_ABOVE 1
#define VPT_LUT_REFRACT_BELOW 2
int vpt_lut_calculate(int device, uint32_t kind, uint32_t size_x, uint32_t size_y, uint32_t size_z, uint32_t sample_count,
                      uint32_t time_ms, float* out_host);

#ifdef __cplusplus
}
#endif
#endif /* VPT_H */
