// kernels.hpp — host-callable launch wrappers of the HIP stages (definitions in kernels_*.hip).
#pragma once
#include "device_types.hpp"
#include "../../include/vpt_lab.h"   // VPT_TRACE_* / VPT_LAB_* constants (the functions exist in the laboratory build only)
#ifndef VPT_LAB
#define VPT_LAB 0
#endif

namespace vpt {

void launch_bounce(hipStream_t s, uint32_t blocks, bool lds_scene, bool count, bool first, const DeviceScene& sc, const RenderParams& P,
                   const PathState& ps, const StreamState& ss, const uint32_t* queue, uint32_t* queue_next, Counters* ctr, uint32_t parity, uint32_t n_slots,
                   uint32_t dispatch_base, uint32_t k3, bool plain = false);
int bounce_blocks_per_cu(bool lds_scene, const DeviceScene& sc, bool plain = false);
void launch_whole(hipStream_t s, uint32_t blocks, bool count, const DeviceScene& sc, const RenderParams& P, const PathState& ps, Counters* ctr, uint32_t n_slots,
                  uint32_t dispatch_base, bool plain, uint32_t static_rounds, uint32_t chunk_tiles);
int whole_blocks_per_cu(const DeviceScene& sc, bool plain);
// the rest of a streams batch in one launch (kernels_path.hip k_finish): every path of queue[parity] run to its end
void launch_finish(hipStream_t s, uint32_t blocks, bool count, const DeviceScene& sc, const RenderParams& P, const PathState& ps, const StreamState& ss, const uint32_t* queue,
                   StreamCounters* sctr, Counters* ctr, uint32_t parity);
int finish_blocks_per_cu(const DeviceScene& sc);
void launch_raygen(hipStream_t s, const RenderParams& P, const PathState& ps, uint32_t* queue, Counters* ctr, uint32_t n_slots, uint32_t dispatch_base);
void launch_prepare(hipStream_t s, Counters* ctr, uint32_t parity);
void launch_fold(hipStream_t s, Counters* ctr);
void launch_extend(hipStream_t s, uint32_t blocks, bool lds_scene, bool count, const DeviceScene& sc, const PathState& ps,
                   const uint32_t* queue, Counters* ctr, uint32_t parity);
void launch_shade(hipStream_t s, uint32_t blocks, const DeviceScene& sc, const RenderParams& P, const PathState& ps,
                  const uint32_t* queue, uint32_t* queue_next, uint32_t* cqueue, Counters* ctr, uint32_t parity);
void launch_connect(hipStream_t s, uint32_t blocks, bool lds_scene, bool count, const DeviceScene& sc, const RenderParams& P,
                    const PathState& ps, const uint32_t* cqueue, Counters* ctr, uint32_t parity);
void launch_resolve(hipStream_t s, const RenderParams& P, const PathState& ps, float* image, uint32_t frames, uint32_t frame_base, const uint32_t* guard);
void launch_trace_rays(hipStream_t s, uint32_t blocks, const DeviceScene& sc, const vpt_ray* rays, uint32_t n, vpt_hit* hits);
void launch_scatter_rows(hipStream_t s, const float* gathered, float* full, uint32_t w, uint32_t h, uint32_t shard_count, uint32_t stride_px);
void launch_precompute_materials(hipStream_t s, const DeviceScene& sc, uint32_t flags, MatResolved* out, uint32_t n);
void launch_precompute_tri_ng(hipStream_t s, const DeviceScene& sc, float4* out);
void launch_precompute_tri_shade(hipStream_t s, const DeviceScene& sc, float4* out);
void launch_precompute_emissive(hipStream_t s, const DeviceScene& sc, EmissiveTri* out, uint32_t total);
size_t traverse_lds_bytes(const DeviceScene& sc, bool lds_scene);
size_t stack_overflow_bytes(uint32_t blocks);  // per-thread spill region of the traversal stacks for a grid of `blocks`
int traverse_blocks_per_cu(bool lds_scene, const DeviceScene& sc);
int shade_blocks_per_cu();

// ray-stream traversal kernels (kernels_trace.hip)
struct TraceArgs {
    const float4* ro;       // origin.xyz | -
    const float4* rd;       // direction.xyz | -
    const uint32_t* order;  // optional: entry i of the stream is ray order[i]
    const uint32_t* valid;  // optional (with order == nullptr): entry i is a hole when valid[i] == 0xffffffff
    float4* hit;            // closest: t (< 0 miss), u, v | primitive; any-hit: x = 1 occluded / -1 clear
    uint32_t* hinst;        // closest: instance
    uint32_t n;
    const uint32_t* n_dev;  // optional: the stream length lives in device memory (a queue size word); overrides n
    uint32_t* head;         // work cursor for entries beyond the waves' static first 64 (zeroed before the launch)
    float tmin, tmax;
    uint32_t normalize_dir; // RayGen.slang:70 normalises the payload direction before tracing
    uint32_t store_gid;     // 1: the hit record's fourth word is the GLOBAL triangle id (what the shade stage wants), 0: PrimitiveIndex (lab, vpt_hit)
    uint32_t param;         // variant parameter (vote: idle lanes that trigger a fetch step; 0 = default)
    unsigned char* cls;     // closest-hit only, optional: per queue entry, the shade class of what the ray hit (kShade*; 0xff for a hole)
    uint32_t cull;          // closest-hit only: 1 = drop stale stack entries at the pop (vote.hpp pop_or_done_cull); hits are unchanged
    uint32_t one_tri;       // trace lab: 1 = ONE triangle per triangle step (round 3's step) instead of the product's up to two (vote.hpp vote_tri2_step_*); hits are unchanged
    uint32_t packed;        // trace lab: 1 = the node step's plane arithmetic in packed fp32 instructions (traverse.hpp node_entries_pk); hits are unchanged
};
void launch_trace(hipStream_t s, uint32_t blocks, uint32_t variant, bool any, bool count, const DeviceScene& sc, const TraceArgs& a, Counters* ctr);
int trace_blocks_per_cu(uint32_t variant, bool any);

// staged pipeline on compact streams (kernels_stream.hip)
void launch_stream_begin(hipStream_t s, StreamCounters* sc, uint32_t n_first, uint32_t n_total);
// path regeneration by refill: the room the ended paths left in the next ray queue is filled with the batch's next unstarted samples
void launch_refill(hipStream_t s, uint32_t blocks, const RenderParams& P, const PathState& ps, const StreamState& ss, uint32_t* queue_next, StreamCounters* sc, uint32_t parity_next,
                   uint32_t cap, uint32_t dispatch_base);
void launch_raygen_stream(hipStream_t s, const RenderParams& P, const PathState& ps, const StreamState& ss, uint32_t* queue, uint32_t n_slots, uint32_t dispatch_base, bool media);
void launch_prepare_stream(hipStream_t s, StreamCounters* sc, uint32_t parity);
void launch_classify(hipStream_t s, const uint32_t* queue, const unsigned char* cls, uint32_t* const* class_queue, StreamCounters* sc, uint32_t parity, uint32_t max_entries, uint32_t shade_waves);
void launch_layout_single(hipStream_t s, StreamCounters* sc, uint32_t parity, uint32_t shade_waves);
void launch_shade_stream(hipStream_t s, uint32_t blocks, uint32_t cls, bool sorted, const DeviceScene& sc, const RenderParams& P, const PathState& ps, const StreamState& ss,
                         const uint32_t* queue, const uint32_t* order, uint32_t* queue_next, Counters* ctr, StreamCounters* sctr, uint32_t parity);
void launch_classify_instances(hipStream_t s, const DeviceScene& sc, unsigned char* out, uint32_t n);
void launch_trace_shadow(hipStream_t s, uint32_t blocks, bool light, bool count, const DeviceScene& sc, const StreamState& ss, Counters* ctr,
                         StreamCounters* sctr, uint32_t param, uint32_t ray_queries = 1u);   // ray_queries: USE_RAY_QUERIES (RTCommon.slang:52 / :64)
void launch_join(hipStream_t s, uint32_t blocks, const RenderParams& P, const PathState& ps, const StreamState& ss, const StreamCounters* sctr, const uint32_t* queue,
                 const uint32_t* queue_next, uint32_t parity);
int shade_stream_blocks_per_cu();
int join_blocks_per_cu();
// participating media on the streams (kernels_media.hip)
void launch_media_scatter(hipStream_t s, uint32_t blocks, const DeviceScene& sc, const PathState& ps, const StreamState& ss, const MediaState& ms, const uint32_t* queue,
                          const StreamCounters* sctr, uint32_t parity);
void launch_layout_media(hipStream_t s, StreamCounters* sctr, uint32_t parity, uint32_t shade_waves, uint32_t tail_waves);
void launch_shade_media(hipStream_t s, uint32_t blocks, const DeviceScene& sc, const RenderParams& P, const PathState& ps, const StreamState& ss, const MediaState& ms,
                        const uint32_t* queue, Counters* ctr, StreamCounters* sctr, uint32_t parity);
void launch_media_tail(hipStream_t s, uint32_t blocks, const DeviceScene& sc, const RenderParams& P, const PathState& ps, const StreamState& ss, const MediaState& ms,
                       const uint32_t* queue, uint32_t* queue_next, Counters* ctr, StreamCounters* sctr, uint32_t parity);
int shade_media_blocks_per_cu();
int media_tail_blocks_per_cu();
int trace_shadow_blocks_per_cu();

// lookup-table generator (kernels_lut.hip)
void launch_lut(hipStream_t s, int kind, float* table, uint32_t sx, uint32_t sy, uint32_t sz, uint32_t sample_count, uint32_t time_hash,
                uint32_t first_dispatch, uint32_t n_dispatches);

// post-process chain (kernels_post.hip)
void launch_bloom_threshold(hipStream_t s, const float* in, float* out, uint32_t w, uint32_t h, float threshold, float falloff);
void launch_bloom_down(hipStream_t s, const float* in, uint32_t iw, uint32_t ih, float* out, uint32_t ow, uint32_t oh, float strength);
void launch_bloom_up(hipStream_t s, const float* in, uint32_t iw, uint32_t ih, float* out, uint32_t ow, uint32_t oh, float strength);
// levels mips[0 .. n - 1] receive their up-samples in one launch (only mips[0] is written); mips[n] is read as it stands; n <= kBloomChainMax
constexpr uint32_t kBloomChainMax = 4;
// outs[0 .. n - 1] = successive down-samples of `in`, all written, one launch; n = 2 or 3 (kBloomDownChainMax)
constexpr uint32_t kBloomDownChainMax = 3;
constexpr uint32_t kBloomDownChainTexels = 160000;   // 480 x 270 and below: smaller than what fills the chip for the length of a launch
void launch_bloom_down_chain(hipStream_t s, const float* in, uint32_t iw, uint32_t ih, float* const* outs, const uint32_t* w, const uint32_t* h, uint32_t n, float strength);
void launch_bloom_up_chain(hipStream_t s, float* const* mips, const uint32_t* w, const uint32_t* h, uint32_t n, float strength);
// fused schedule: threshold inside the first down-sample, the small mips down and up in one launch, last up-sample + tonemap in one
void launch_bloom_down_first(hipStream_t s, const float* hdr, uint32_t iw, uint32_t ih, float* out, uint32_t ow, uint32_t oh, float strength, float threshold, float falloff);
// staged (base mip <= 8448 texels, bloom_tail_is_staged): levels w[0..] go down and up in LDS and level 0, finished, is written to top_out — the
// up-sample into the base is the caller's (the up chain's); otherwise (huge base) the base itself is updated and top_out is not used
bool bloom_tail_is_staged(uint32_t bw, uint32_t bh);
void launch_bloom_tail(hipStream_t s, float* base, float* top_out, uint32_t bw, uint32_t bh, const uint32_t* w, const uint32_t* h, uint32_t levels, float strength);
void launch_post_final(hipStream_t s, const float* hdr, const float* mip1, uint32_t mw, uint32_t mh, float* bloom0_out, uint8_t* out, uint32_t w, uint32_t h,
                       float threshold, float falloff, float strength, float exposure, float gamma, bool linear_tap);
constexpr uint32_t kBloomTailMaxTexels = 2048, kBloomTailMaxLevels = 8;   // largest mip the one-launch tail keeps in LDS; levels it can hold
void launch_tonemap(hipStream_t s, const float* hdr, const float* bloom0, uint8_t* out, uint32_t w, uint32_t h, float exposure,
                    float gamma, bool linear_tap);

}  // namespace vpt
