// vpt_api.hip — implementation of the C-ABI in include/vpt.h: context, scene preparation (host
// arithmetic of PathTracer.cpp restated), the wavefront render loop, sharding, post-process schedule.
// There is no CPU fallback in this file: without a HIP device vpt_create() fails.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h>

#include "bvh_build.hpp"
#include "kernels.hpp"

using namespace vpt;

constexpr uint64_t kBytesPerPath = 380;          // core slot records 68 + stream records ~290 (slack included) + queues 8 + image share
constexpr uint32_t kMaxFramesInFlight = 8192;   // frames of one batch.  A 1/8 shard of 1080p holds ~448M paths at 1728 frames and its default batch is four times that
                                                // (6912 frames = 1.79 G samples, what a whole 1080p image renders per batch at 904 frames): with round 5's bound of 2048 a rank of an 8-GPU
                                                // job rendered 3.5 x shorter batches than a 1-GPU job, i.e. bench.py's "weak" scaling did not keep the per-rank work fixed
constexpr uint64_t kResidentPaths = 448ull << 20;   // samples of a batch by default (see check_render_size)
// The rest of a streams batch goes to k_finish (kernels_path.hip) — one launch instead of seven per bounce —
//  * after kFinishAfterBounces (3) bounces when the batch is small from the start (a frame or two per call: its launches never fill the chip for long), and
//  * as soon as the host sees fewer than kFinishBelowPaths paths alive in a large one (the last of depth-32 paths: 20 bounce-sets on nearly empty queues).
#ifndef VPT_FINISH_AFTER   // (-D overrides: the A/B builds of tests/tools/ab_variants.sh)
#define VPT_FINISH_AFTER 3
#endif
#ifndef VPT_FINISH_BELOW
#define VPT_FINISH_BELOW (1u << 18)
#endif
constexpr uint32_t kFinishSmallBatchPaths = 6u << 20, kFinishAfterBounces = VPT_FINISH_AFTER, kFinishBelowPaths = VPT_FINISH_BELOW;

// One wavefront batch in progress: what render_batch's stages hand to each other (and what an asynchronous batch leaves behind for
// the call that finishes it).
struct BatchState {
    uint32_t frames = 0, dispatch_base = 0, n_slots = 0;
    uint32_t n_first = 0;   // slots the camera-ray launch starts (n_slots, or the resident part of it when paths are regenerated)
    bool fused = false, stream = false, media_stream = false, sorted = false, overlap = false, count = false;
    bool whole = false;     // the whole batch is ONE launch of k_whole (kernels_path.hip): no bounces follow
    bool regen = false;     // only n_first of the batch's n_slots samples start with the camera-ray launch; refills start the rest (kernels_stream.hip k_refill_plan)
    uint32_t finish_at = 0; // streams: once this many bounces have run, ONE launch of k_finish runs what is left of the batch to its end (0: never)
    bool finished = false;  // ... and it has been enqueued: no bounce follows
    uint32_t parity = 0, k3 = 0;
    bool join_pending = false;
    uint64_t iter = 0, iter_cap = 0, min_bounces = 0;
};
// Counter words a finished batch copies to pinned host memory (asynchronously, behind its resolve).
struct HostCounters {
    Counters ctr;
    uint32_t alive[2], queue_len[2];
    uint32_t refill_next;   // regenerating batches: samples started so far (StreamCounters::refill_next)
};
constexpr int kTickets = 16;

struct vpt_ctx {
    vpt_config cfg{};
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;             // staged pipeline: the shadow-ray kernels and the join of bounce k run here, beside the extend of bounce k + 1
    hipEvent_t ev_shade = nullptr, ev_join = nullptr;
    std::string err;
    int cu_count = 256;

    // host copies (SetMaterial / emissive list maintenance / stats)
    bool has_scene = false;
    bool buffers_ok = false;     // render buffers allocated for the current size (false after a failed vpt_resize)
    uint32_t texture_count = 0;  // of the current scene: vpt_set_material validates texture indices against it
    ncclComm_t comm = nullptr;   // vpt_comm_init: one rank per process
    int comm_rank = -1, comm_world = 0;
    float* gather_buf = nullptr; // root: shard_count padded shards back to back
    // trace lab (vpt_lab_*): a resident ray set
    float4 *lab_ro = nullptr, *lab_rd = nullptr, *lab_hit = nullptr;
    uint32_t *lab_hinst = nullptr, *lab_order = nullptr;
    uint32_t lab_n = 0; float lab_tmin = 0.0f, lab_tmax = 0.0f;
    std::vector<vpt_material> materials;
    std::vector<MeshDesc> meshes;
    std::vector<InstanceDesc> instances;
    std::vector<EmissiveDesc> emissive;
    uint32_t emissive_tris = 0;
    uint64_t total_vertices = 0, total_indices = 0;
    uint32_t bvh_depth = 0;

    DeviceScene dsc{};
    std::vector<void*> scene_allocs;
    vpt_material* d_materials = nullptr;
    EmissiveDesc* d_emissive = nullptr;
    MatResolved* d_mat_resolved = nullptr;
    EmissiveTri* d_emissive_tri = nullptr;
    uint32_t* d_emissive_tri_offset = nullptr;
    float4* d_tri_ng = nullptr;
    float4* d_tri_shade = nullptr;
    unsigned char* d_inst_class = nullptr;   // shade class per instance (kernels_path.hip k_classify_instances)
    std::vector<BvhTri> bvh_input;           // the triangles the BVH was built from (trace lab: the eight-wide tree is built from them on first use)
    bool lds_scene = false;
    bool scene_plain = false;    // every material's five textures are 1x1 and the environment is black: the fused kernel's PLAIN instantiation serves it
    std::vector<unsigned char> tex_1x1;   // per texture of the scene
    bool sbvh = false;
    uint32_t* stack_overflow2 = nullptr;   // spill region of the traversal kernels launched on stream2
    int trav_blocks = 1024;

    vpt_params params{};
    RenderParams P{};
    uint32_t frames_in_flight = 1;   // largest batch the context will render at once (the cap; vpt_config.frames_in_flight)
    uint32_t frames_cap = 0;     // upper bound of frames_in_flight after an out-of-memory failure of a size the library chose itself
    uint32_t long_factor = 4;    // batch_cap(): contexts that keep only part (or none) of a batch's paths resident take batches this many times frames_in_flight
    uint32_t frames_alloc = 0;   // frames of SAMPLES the slot-addressed buffers hold now: they grow to the largest batch actually requested (ensure_path_buffers)
    uint32_t resident_alloc = 0; // frames of PATHS the queues and stream records hold (<= frames_alloc; less when paths are regenerated)
    bool ps_has_sidx = false, ps_has_media = false;   // the per-sample words only some batches touch are allocated only for them: sample index (samples_per_frame > 1), VolumeDepth / ColorChannel (media)
    int whole_blocks = 0;        // persistent grid of the whole-path kernel (kernels_path.hip k_whole), 0: the scene does not ride in LDS
    uint32_t lab_whole_sched = 4u;   // how k_whole's waves get their tiles (VPT_LAB_WHOLE_SCHED): tiles per atomic | static-rounds mode << 4
    uint32_t lab_whole_frames = 0xffffffffu;   // VPT_PIPELINE_AUTO runs batches of at most this many frames as ONE whole-path launch (VPT_LAB_WHOLE_FRAMES); default: every batch
                                               // (Cornell box 1080p, Msamples/s whole vs per-bounce at 1 / 4 / 16 / 64 / 226 frames per batch: 3821 / 6413 / 7737 / 8183 / 8315 vs
                                               // 2401 / 4753 / 6493 / 7244 / 7359; general instantiation 8840 vs 7713: profiles/r04_whole_ab.json)
    bool depth_bounded = true;   // every path ends within max_depth * samples_per_frame bounces (no material scatters inside a medium): see vpt_render_async
    // asynchronous batches (vpt_render_async / vpt_postprocess_device / vpt_wait)
    hipEvent_t tick_ev[kTickets] = {};
    uint64_t tick_issued = 0;
    HostCounters* h_ctr = nullptr;   // pinned
    bool async_dirty = false;        // work has been enqueued without a host synchronisation behind it
    bool out_active = false;         // an enqueued batch whose paths may outlive the bounces enqueued so far: the next call finishes it
    BatchState out_batch;
    uint64_t out_ticket = 0;
    uint32_t* d_dispatch_base = nullptr;   // graph replays read the batch's first dispatch index from here (RenderParams::dispatch_base_dev)
    hipGraphExec_t graph = nullptr;
    uint64_t graph_gen = 0, state_gen = 1;   // state_gen: bumped by everything a captured batch bakes in (scene tables' addresses, params, camera, buffers)
    uint32_t graph_frames = 0, graph_bounces = 0, graph_streak = 0;
    uint64_t graph_streak_gen = 0;
    uint64_t graph_kernel_launches[VPT_KERNEL_COUNT] = {};
    bool graph_broken = false;       // a capture failed once on this context: stay on plain launches
    bool capturing = false;          // a batch is being captured into a hipGraph: one stream only (no shadow / join overlap on stream2)
    // Pipelined 1-frame batches (vpt_render_async): a frame of the fused fixed schedule is a chain of ~9 dependent launches, each bounded
    // below by the latency of one bounce (~60-90 us on nearly empty queues), so one frame at a time leaves most of the chip idle
    // (profiles/r04_latency_probe.json: 0.95 ms of kernels per 1080p frame against 0.33 ms per frame in 16-frame batches).  Consecutive
    // frames are independent until their resolve, so they go round-robin over kLanes lanes — the context itself and kLanes - 1 lane
    // contexts with a stream, counters, 1-frame path buffers and a spill region of their own, sharing the scene tables and the
    // accumulation image — and only the resolves are ordered (frame k's waits for frame k - 1's: the running mean is applied in frame order).
    vpt_ctx* lanes[2] = {nullptr, nullptr};
    vpt_ctx* owner = nullptr;        // set in a lane: the context whose scene and image it borrows
    void* lane_spill = nullptr;      // a lane's own traversal spill region
    hipEvent_t ev_resolved = nullptr;    // recorded behind this lane's latest resolve
    vpt_ctx* order_lane = nullptr;   // owner only: the lane the latest resolve was enqueued on (nullptr: nothing pipelined since the last drain)
    hipEvent_t ev_post = nullptr;    // owner only: recorded behind the latest vpt_postprocess_device — the next frame's resolve must not touch the image before
    bool post_pending = false;
    uint32_t lane_rr = 0;
    // vpt_lab_set; defaults = what tests/tools/latency_probe.py measured best (profiles/r04_latency_probe.json): a frame goes to the first
    // lane whose previous frame has been resolved (so a host with two frames in flight alternates between two lanes, one with three
    // uses all three), every lane launches the full persistent grid, and the bounces >= 2 of a 1-frame batch — queues of a quarter of
    // the frame's paths and less — a third of it, which leaves room for the other lanes' blocks
    uint32_t lab_lanes = 3, lab_lane_grid = 1, lab_tail_grid = 3;
    int tail_blocks = 0;             // > 0: grid of the bounces >= 2 of the batch being enqueued (pipelined 1-frame batches)
    BatchState graph_batch;          // the captured batch as it stands before its resolve
    BatchState last_fixed;           // the latest fixed-schedule batch enqueued on this lane (drain checks that nothing outlived it)
    bool last_fixed_valid = false;
    uint32_t stack_overflow_words = 0;   // words per spill region
    unsigned long long* d_spill_count = nullptr;
    bool spill_dirty = true;         // traversal kernels have run since the spill regions were last counted (vpt_get_stats counts lazily)
    uint64_t spill_cached[2] = {0, 0};
    double set_scene_ms = 0.0, bvh_build_ms = 0.0;

    void* ps_block = nullptr;    // slot-addressed records every pipeline uses (L, ACC, M + the dword arrays)
    void* ps_legacy = nullptr;   // round 1's stage kernels only (A, B, T, H, C*, hinst): allocated on their first use
    PathState ps{};
    uint32_t* queue[2] = {nullptr, nullptr};
    uint32_t* cqueue = nullptr;  // connect queue (two-ended)
    void* ss_block = nullptr;    // stream records of the staged pipeline (kernels_stream.hip)
    uint32_t* class_queue[kShadeClasses] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // the shade queue sorted by class
    unsigned char* cls_q = nullptr;          // shade class per ray-queue entry, written by the extend stage
    StreamState ss{};
    void* media_block = nullptr; // streams of the media variant of the staged pipeline (kernels_media.hip): allocated on first use
    MediaState ms{};
    int shade_media_blocks = 768, media_tail_blocks = 768;
    uint32_t media_frames = 0;   // frames a media batch on the streams can hold (ensure_media_buffers)
    uint32_t class_present = 0x1fu;   // shade classes some instance of the scene belongs to (bit kShadeMiss always set): the others get no launch
    uint32_t stream_slack = 0;   // entries a stream may hold beyond its true count: unwritten chunk tails (vote.hpp WaveAppender)
    int shade_stream_blocks = 768, shadow_blocks = 2048, finish_blocks = 768;
    std::vector<vpt_volume> volumes;       // homogeneous box volumes (vpt_set_volumes)
    vpt_volume* d_volumes = nullptr;
    std::vector<DensityGrid> grids;        // device pointers inside (vpt_add_density_grid)
    DensityGrid* d_grids = nullptr;
    uint32_t phase = VPT_PHASE_HENYEY_GREENSTEIN;
    uint32_t* d_launch_off = nullptr;  // split-screen: launch-grid prefix sums of the dispatches of a batch
    int shade_blocks = 1024, primary_blocks = 768, max_blocks = 1536, join_blocks = 2048;
    int primary_blocks_general = 768, primary_blocks_plain = 768;   // grids of the fused kernel's two instantiations (primary_blocks = the one scene_plain picks)
    int vote_blocks = 2048;   // persistent grid of the vote-scheduled traversal kernels
    uint32_t vote_param = 256u + 16u;  // weighted vote, fetch step at 16 idle lanes (profiles/r02_trace_lab_*.json)
    Counters* ctr = nullptr;
    StreamCounters* sctr = nullptr;   // stream pipeline: lengths, exact live counts and work cursors, one cache line each
    float* image = nullptr;       // this shard's rows, RGBA32F
    float* full_image = nullptr;  // whole image when shard_count > 1 (after vpt_assemble_shards)
    bool full_valid = false;

    uint64_t dispatch_count = 0;
    uint32_t frame_count = 0, samples_accum = 0;
    vpt_stats stats{};

    // post
    std::vector<float*> mips;
    std::vector<std::pair<uint32_t, uint32_t>> mip_sizes;
    uint8_t* post_out = nullptr;
    uint32_t post_w = 0, post_h = 0;

    // profiling events
    std::vector<hipEvent_t> ev_pool;
    struct Pending { int kernel; hipEvent_t a, b; };
    std::vector<Pending> pending;
    size_t ev_next = 0;
};

namespace {

#define HIPCHK(ctx, call)                                                                                      \
    do {                                                                                                       \
        hipError_t e_ = (call);                                                                                \
        if (e_ != hipSuccess) {                                                                                \
            char buf_[512];                                                                                    \
            snprintf(buf_, sizeof(buf_), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            (ctx)->err = buf_;                                                                                 \
            return (e_ == hipErrorOutOfMemory) ? VPT_ERR_OUT_OF_MEMORY : VPT_ERR_DEVICE;                       \
        }                                                                                                      \
    } while (0)

int fail(vpt_ctx* c, int code, const char* msg) { c->err = msg; return code; }

template <class T>
int upload(vpt_ctx* c, const std::vector<T>& v, const T** out, size_t min_elems = 1) {
    size_t n = std::max(v.size(), min_elems);
    void* d = nullptr;
    HIPCHK(c, hipMalloc(&d, n * sizeof(T)));
    c->scene_allocs.push_back(d);
    HIPCHK(c, hipMemset(d, 0, n * sizeof(T)));
    if (!v.empty()) HIPCHK(c, hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T*)d;
    return VPT_OK;
}

void free_scene(vpt_ctx* c) {
    for (void* p : c->scene_allocs) (void)hipFree(p);
    c->scene_allocs.clear();
    c->has_scene = false;
    c->d_materials = nullptr; c->d_emissive = nullptr;
}
void free_lab(vpt_ctx* c) {
    for (void* p : {(void*)c->lab_ro, (void*)c->lab_rd, (void*)c->lab_hit, (void*)c->lab_hinst, (void*)c->lab_order}) if (p) (void)hipFree(p);
    c->lab_ro = c->lab_rd = c->lab_hit = nullptr; c->lab_hinst = c->lab_order = nullptr; c->lab_n = 0;
}
// Everything sized by (frames held) x (shard pixels): path records, queues, streams.
void free_path_buffers(vpt_ctx* c) {
    if (c->ps_block) (void)hipFree(c->ps_block);
    c->ps_block = nullptr;
    if (c->ps_legacy) (void)hipFree(c->ps_legacy);
    c->ps_legacy = nullptr;
    for (int i = 0; i < 2; i++) { if (c->queue[i]) (void)hipFree(c->queue[i]); c->queue[i] = nullptr; }
    if (c->cqueue) (void)hipFree(c->cqueue);
    c->cqueue = nullptr;
    if (c->ss_block) (void)hipFree(c->ss_block);
    c->ss_block = nullptr;
    if (c->media_block) (void)hipFree(c->media_block);
    c->media_block = nullptr; c->ms = MediaState{}; c->media_frames = 0;
    for (uint32_t k = 0; k < kShadeClasses; k++) { if (c->class_queue[k]) (void)hipFree(c->class_queue[k]); c->class_queue[k] = nullptr; }
    if (c->cls_q) (void)hipFree(c->cls_q);
    c->cls_q = nullptr;
    c->ps = PathState{}; c->ss = StreamState{};
    c->frames_alloc = 0; c->resident_alloc = 0;
    c->state_gen++;   // a captured batch holds these addresses
}
void free_render_buffers(vpt_ctx* c) {
    free_path_buffers(c);
    if (c->image) (void)hipFree(c->image);
    c->image = nullptr;
    if (c->full_image) (void)hipFree(c->full_image);
    c->full_image = nullptr;
    if (c->gather_buf) (void)hipFree(c->gather_buf);
    c->gather_buf = nullptr;
    for (float* m : c->mips) (void)hipFree(m);
    c->mips.clear(); c->mip_sizes.clear();
    if (c->post_out) (void)hipFree(c->post_out);
    c->post_out = nullptr; c->post_w = c->post_h = 0;
}

uint32_t shard_rows_of(uint32_t height, uint32_t rank, uint32_t count) { return rank < height ? (height - rank + count - 1) / count : 0; }

// Size checks of a (width, height) before anything is freed or changed; *frames_out = the largest batch this context will render.
int check_render_size(vpt_ctx* c, uint32_t width, uint32_t height, uint32_t* frames_out) {
    const uint64_t rows = shard_rows_of(height, c->cfg.shard_rank, c->cfg.shard_count);
    const uint64_t px = rows * width;
    if (px == 0) return fail(c, VPT_ERR_INVALID_ARGUMENT, "empty shard");
    if (px >= (1ull << 31) || (uint64_t)width * height >= (1ull << 31)) return fail(c, VPT_ERR_INVALID_ARGUMENT, "image too large");
    uint64_t F = c->cfg.frames_in_flight;
    // ~448M resident paths whatever the shard size (226 frames at 1080p; kBytesPerPath x 448M = ~170 GB of the 288 GB, and never more
    // than 60 % of the memory that is free right now): the last bounces of a batch are short launches that cannot fill 256 CUs, and
    // a larger batch makes them longer for the same fixed cost.  Msamples/s from 32M to 128M paths: Cornell +8 %, atrium +16 %, glass
    // bust (depth 32) +78 %; 128M to 256M: +0 / +2 / +18 %; 256M to 512M: +0 / +2.4 / +9.1 % (profiles/r03_frames_sweep.json).
    // This is the CAP of a batch; the buffers hold what has actually been asked for (ensure_path_buffers).
    if (F == 0) {
        uint64_t paths = kResidentPaths;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) paths = std::min<uint64_t>(paths, (uint64_t)(free_b * 0.6) / kBytesPerPath);
        F = paths / px;
    }
    F = std::max<uint64_t>(1, std::min<uint64_t>(F, kMaxFramesInFlight));
    if (px * F >= (1ull << 31)) return fail(c, VPT_ERR_INVALID_ARGUMENT, "too many paths in flight");
    *frames_out = (uint32_t)F;
    return VPT_OK;
}

// Buffers for batches of up to `frames` frames of this shard of which `resident` frames of paths are in flight at a time (the caller
// has drained the streams): slot-addressed records (frame sum, medium, per-sample words: 36 B per SAMPLE of the batch) and the queues
// and stream records (~290 B per RESIDENT path).
int alloc_path_buffers(vpt_ctx* c, uint32_t frames, uint32_t resident) {
    free_path_buffers(c);
    const RenderParams& P = c->P;
    resident = std::min(resident, frames);
    if ((uint64_t)P.shard_pixels * frames >= (1ull << 31)) return fail(c, VPT_ERR_INVALID_ARGUMENT, "too many paths in flight");
    const uint32_t samples = P.shard_pixels * frames;
    uint32_t cap = P.shard_pixels * resident;
    // slot-addressed records every pipeline uses: 2 float4 records + up to 4 dword arrays per slot (device_types.hpp PathState): the medium
    // anisotropy always; the sample index only for samples_per_frame > 1, VolumeDepth / ColorChannel only with media — 36 B per sample of
    // a plain batch (the kernels touch those words under exactly these conditions; path_words_ok() replaces the buffers when a batch needs more).
    // The records of round 1's stage kernels come with ensure_legacy_buffers()
    const bool want_sidx = P.samples_per_frame > 1u, want_media = !c->volumes.empty() || c->dsc.atm_on;
    const size_t kRecords = 2, kWords = 1u + (want_sidx ? 1u : 0u) + (want_media ? 2u : 0u);
    size_t stride = ((size_t)samples + 63) & ~(size_t)63;
    HIPCHK(c, hipMalloc(&c->ps_block, stride * (16 * kRecords + 4 * kWords)));
    float4* rb = (float4*)c->ps_block;
    PathState& s = c->ps;
    s = PathState{};
    s.capacity = cap;
    s.ACC = rb; s.M = rb + stride;
    uint32_t* wb = (uint32_t*)(rb + stride * kRecords);
    s.maniso = (float*)wb; wb += stride;
    if (want_sidx) { s.sidx = wb; wb += stride; }
    if (want_media) { s.vdepth = wb; s.cchan = (int32_t*)(wb + stride); }
    c->ps_has_sidx = want_sidx; c->ps_has_media = want_media;
    // streams written by chunked appends hold up to one unwritten chunk tail per wave that appended to them: at most
    // 256 entries per 64 items processed, and never more than one per resident wave of the largest persistent grid.  A launch
    // appends in chunks only when its queue holds >= kFusedExactBelow (fused kernel) / kAppendExactBelow (streams) entries, holes
    // included; below that every append is exact and no stream ever holds a hole, so buffers that cannot reach that length need no slack.
    // (the appending kernels' persistent grids: blocks per CU from the occupancy query, the fused kernel's at most 3 by its LDS; checked
    // against the real grids by check_stream_slack once the scene is known.  Round 3 reserved for 8192 blocks: 2 GB of a large batch's streams)
    const int fused_per_cu = (std::max(c->primary_blocks_general, c->primary_blocks_plain) + std::max(c->cu_count, 1) - 1) / std::max(c->cu_count, 1);   // (the scene's, once one is set)
    const int per_cu = std::max(std::max(4, fused_per_cu), std::max(shade_stream_blocks_per_cu(), std::max(shade_media_blocks_per_cu(), media_tail_blocks_per_cu())));
    const uint64_t max_tails = (uint64_t)c->cu_count * (uint64_t)per_cu * 4u * kAppendChunk;
    c->stream_slack = cap < kFusedExactBelow ? 256u : (uint32_t)std::min<uint64_t>((uint64_t)cap * 4 + 256, max_tails);
    const size_t scap = (size_t)cap + c->stream_slack;
    for (int i = 0; i < 2; i++) HIPCHK(c, hipMalloc((void**)&c->queue[i], scap * 4));
    {
        const size_t sst = (scap + 63) & ~(size_t)63;
        HIPCHK(c, hipMalloc(&c->ss_block, sst * (16 * 17 + 4 + 2)));
        float4* q = (float4*)c->ss_block;
        StreamState& t = c->ss;
        t.PE = q; t.PS = q + sst; t.PL = q + 2 * sst; t.PT = q + 3 * sst; t.SKO = q + 4 * sst; t.SKD = q + 5 * sst; t.LTO = q + 6 * sst; t.LTD = q + 7 * sst;
        t.RA[0] = q + 8 * sst; t.RA[1] = q + 9 * sst; t.RB[0] = q + 10 * sst; t.RB[1] = q + 11 * sst; t.RT[0] = q + 12 * sst; t.RT[1] = q + 13 * sst;
        t.RL[0] = q + 14 * sst; t.RL[1] = q + 15 * sst;
        t.SH = q + 16 * sst; t.SHI = (uint32_t*)(q + 17 * sst);
        t.vis_sky = (unsigned char*)(t.SHI + sst); t.vis_light = t.vis_sky + sst;
        t.cap = (uint32_t)scap;
    }
    c->frames_alloc = frames; c->resident_alloc = resident;
    return VPT_OK;
}

// Every wave of a launch that appends to a stream may leave one unwritten chunk tail in it (vote.hpp WaveAppender): the streams are
// allocated with room for stream_slack such entries.  Refuse — not after a kernel has written past a stream — if a device with more
// CUs / other occupancy than the allocation assumed ever needs more.  Called wherever the grids (vpt_set_scene) or the buffers change.
int check_stream_slack(vpt_ctx* c) {
    if (!c->has_scene || c->frames_alloc == 0 || c->ps.capacity < kFusedExactBelow) return VPT_OK;   // short streams are appended to exactly: no tails
    const uint64_t appending_waves = 4ull * (uint64_t)std::max(std::max(c->shade_stream_blocks, c->primary_blocks), std::max(c->shade_media_blocks, c->media_tail_blocks));
    if (appending_waves * kAppendChunk > (uint64_t)c->stream_slack && (uint64_t)c->ps.capacity * 4 + 256 > (uint64_t)c->stream_slack)
        return fail(c, VPT_ERR_DEVICE, "internal: the stream slack allocated for chunk tails is smaller than one chunk per appending wave of this device");
    return VPT_OK;
}

// (Re)allocates everything that depends on the image size: the accumulation image(s) and the path buffers of ONE frame.  On failure
// the context keeps NO render buffers and says so (buffers_ok == false): vpt_render / vpt_get_* / vpt_postprocess then return an error
// instead of touching freed memory.
int alloc_render_buffers(vpt_ctx* c) {
    c->buffers_ok = false;
    c->frames_cap = 0;
    uint32_t F = 1;
    int rc = check_render_size(c, c->cfg.width, c->cfg.height, &F);
    if (rc != VPT_OK) return rc;
    free_render_buffers(c);
    RenderParams& P = c->P;
    P.width = c->cfg.width; P.height = c->cfg.height;
    P.shard_rank = c->cfg.shard_rank; P.shard_count = c->cfg.shard_count;
    P.shard_rows = shard_rows_of(P.height, P.shard_rank, P.shard_count);
    P.shard_pixels = P.shard_rows * P.width;
    c->frames_in_flight = F;
    auto images = [&]() -> int {
        // padded to the largest shard's row count (vpt_shard_floats): the buffer is handed to ncclGather as it is
        const size_t image_bytes = (size_t)shard_rows_of(P.height, 0, P.shard_count) * P.width * 16;
        HIPCHK(c, hipMalloc((void**)&c->image, image_bytes));
        HIPCHK(c, hipMemset(c->image, 0, image_bytes));
        if (P.shard_count > 1) {
            HIPCHK(c, hipMalloc((void**)&c->full_image, (size_t)P.width * P.height * 16));
            HIPCHK(c, hipMemset(c->full_image, 0, (size_t)P.width * P.height * 16));
        }
        return VPT_OK;
    };
    rc = images();
    if (rc == VPT_OK) rc = alloc_path_buffers(c, 1, 1);
    if (rc != VPT_OK) {
        std::string keep = c->err;
        free_render_buffers(c);
        (void)hipGetLastError();
        c->err = keep;
        return rc;
    }
    c->full_valid = false;
    c->buffers_ok = true;
    return check_stream_slack(c);
}

// One launch per batch (kernels_path.hip k_whole): the scene rides in LDS, no media, one sample per pixel and frame.
// VPT_PIPELINE_WHOLE asks for it; AUTO takes it wherever it applies (lab_whole_frames bounds the batch size, for the A/B).
bool whole_possible(const vpt_ctx* c) {
    const bool vol = !c->volumes.empty() || c->dsc.atm_on;
    return c->has_scene && c->lds_scene && c->whole_blocks > 0 && !vol && c->P.samples_per_frame == 1u;
}
// ... as far as scene, parameters and configuration go (the buffers are the callers' business)
bool whole_policy(const vpt_ctx* c, uint32_t frames) {
    if (!whole_possible(c)) return false;
    return c->cfg.pipeline == VPT_PIPELINE_WHOLE || (c->cfg.pipeline == VPT_PIPELINE_AUTO && frames <= c->lab_whole_frames);
}
// Does a batch of `frames` frames need only its per-sample buffers (36 B per sample: frame sum, medium state), not the ~290 B of records per
// resident path?  A whole-path launch keeps its paths in registers (vpt_config.resident_frames means nothing to it: no path of it is resident in memory).
bool whole_without_records(const vpt_ctx* c, uint32_t frames) { return whole_policy(c, frames); }
bool whole_applies(const vpt_ctx* c, uint32_t frames) { return whole_policy(c, frames) && frames <= c->frames_alloc; }

// Frames of paths a batch of `frames` frames keeps resident.  Paths are regenerated — the next ray queue refilled with fresh camera rays
// behind every shade stage, kernels_stream.hip k_refill_plan — on the STREAMS pipeline (scenes whose BVH lives in memory, no media, whole-frame
// dispatches).  Everything else keeps every sample of a batch resident: the fused per-bounce kernels (scenes that ride in LDS run whole-path
// launches, which hold no records at all; what is left for k_bounce — media, several samples per frame — is not worth a second mechanism),
// round 1's stage kernels (records by slot), media batches (per-entry media streams), split-screen dispatches (launch indices map to
// pixels per dispatch).
bool regen_allowed(const vpt_ctx* c) {
    if (!c->has_scene) return false;
    const bool vol = !c->volumes.empty() || c->dsc.atm_on;
    if (vol || c->P.split != 1u) return false;
    if (c->cfg.pipeline == VPT_PIPELINE_STAGED || c->cfg.pipeline == VPT_PIPELINE_STAGED_SORTED) return true;   // the stream kernels, whatever the scene's size
    return c->cfg.pipeline == VPT_PIPELINE_AUTO && !c->lds_scene;
}
// Default schedule of a context that regenerates (profiles/r05_frames_sweep.json, 1080p): batches of 4 x frames_in_flight frames with HALF of
// frames_in_flight frames of paths resident — 904 / 113 frames: 126 GB instead of 142 GB for 226 / 226, glass bust (depth 32) 3272 against 3088
// Msamples/s (the tail of a batch, ~max_depth bounce-sets on emptying queues, is paid once per 1.9 G samples instead of once per 469 M), atrium
// 1477 against 1485.  vpt_config.resident_frames != 0: as asked (a value >= the batch keeps every sample resident).
uint32_t resident_frames_for(const vpt_ctx* c, uint32_t frames) {
    if (whole_without_records(c, frames)) return 1u;   // (one frame of records stays: what the context would need for a per-bounce batch of one frame)
    if (!regen_allowed(c)) return frames;
    uint64_t k = c->cfg.resident_frames;
    if (k == 0) k = c->cfg.frames_in_flight != 0 ? frames : std::max(1u, c->frames_in_flight / 2u);   // (an explicit batch size without an explicit residency: all resident, as before)
    return (uint32_t)std::min<uint64_t>(frames, k);
}
// The largest batch this context renders at once.  frames_in_flight is what fits with EVERY sample resident (380 B per sample; or what the caller
// asked for).  A context whose batches keep half of that many frames of paths resident (regeneration by refill: 36 B per sample + 290 B per
// resident path) — or none at all (whole-path launches: 36 B per sample) — takes batches long_factor (4) times as long, in less memory.
uint32_t batch_cap(const vpt_ctx* c) {
    const uint32_t F = c->frames_in_flight;
    if (c->cfg.frames_in_flight != 0 || c->cfg.resident_frames != 0 || !c->has_scene) return F;
    const uint64_t by_slots = ((1ull << 31) - 1ull) / std::max<uint64_t>(1, c->P.shard_pixels);
    const uint32_t cap = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>((uint64_t)F * c->long_factor, kMaxFramesInFlight), by_slots));
    return (regen_allowed(c) || whole_policy(c, cap)) ? cap : F;
}

// Grows the buffers so that a batch of `want` frames (<= frames_in_flight) fits; the caller has drained the streams.  A size the
// library chose itself (vpt_config.frames_in_flight == 0) is halved and tried again when the device runs out of memory after all
// (another process, fragmentation) — frames_in_flight then drops to what was obtained and the caller renders in smaller batches;
// an explicit size fails as it is and the context keeps the buffers it had.
// Do the per-sample word arrays allocated cover what the next batch touches (samples_per_frame / media may have changed since)?
bool path_words_ok(const vpt_ctx* c) {
    const bool want_sidx = c->P.samples_per_frame > 1u, want_media = !c->volumes.empty() || c->dsc.atm_on;
    return (!want_sidx || c->ps_has_sidx) && (!want_media || c->ps_has_media);
}
int ensure_path_buffers(vpt_ctx* c, uint32_t want) {
    if (want <= c->frames_alloc && resident_frames_for(c, want) <= c->resident_alloc && path_words_ok(c)) return VPT_OK;
    const uint32_t old = std::max(c->frames_alloc, 1u), old_res = std::max(c->resident_alloc, 1u);
    uint32_t tryf = std::max(want, old);
    while (true) {
        // (a long batch — tryf beyond frames_in_flight — is sized for resident_frames_for(tryf) resident frames: keeping a larger old residency there would exceed the budget F was chosen for)
        const uint32_t keep_res = tryf > c->frames_in_flight ? resident_frames_for(c, tryf) : std::max(resident_frames_for(c, tryf), std::min(old_res, tryf));
        int rc = alloc_path_buffers(c, tryf, keep_res);
        if (rc == VPT_OK) break;
        std::string keep = c->err;
        free_path_buffers(c);
        (void)hipGetLastError();
        const bool oom = rc == VPT_ERR_OUT_OF_MEMORY || rc == VPT_ERR_DEVICE;
        if (!oom || c->cfg.frames_in_flight != 0 || tryf <= old) {
            if (alloc_path_buffers(c, old, old_res) != VPT_OK) { free_path_buffers(c); (void)hipGetLastError(); c->buffers_ok = false; }
            c->err = keep;
            return rc;
        }
        if (tryf > c->frames_in_flight && c->long_factor > 1u) c->long_factor /= 2u;   // a long batch did not fit: shorter long batches from now on
        tryf = std::max(tryf / 2, old);
        c->frames_cap = tryf;
        c->frames_in_flight = std::min(c->frames_in_flight, tryf);
    }
    return check_stream_slack(c);
}

// Round 1's stage kernels (VPT_PIPELINE_STAGED_R1, VPT_FLAG_LOCAL_HITS, an LDS-sized scene forced into the staged pipeline)
// keep a path's records by slot: 13 more float4 records (pathLight among them), the hit instance and the two-ended connect queue, 216 bytes per path,
// allocated when such a batch is first rendered and kept until the next resize.
// The class queues of VPT_PIPELINE_STAGED_SORTED (21 bytes per path), likewise on first use.
// Media on the streams pipeline: 11 more float4 streams per queue entry (176 bytes per path), allocated when such a batch is first
// rendered and kept until the next resize.  They are sized to what is free then (at most 85 % of it): a media batch holds
// media_frames frames, which may be fewer than frames_in_flight (vpt_render then renders in more, smaller batches; the image is
// the same for any batch size).
int ensure_media_buffers(vpt_ctx* c) {
    if (c->media_block) return VPT_OK;
    const uint64_t px = c->P.shard_pixels;
    uint64_t frames = c->resident_alloc;   // what the queues and streams hold now (free_path_buffers drops this block with them)
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        const uint64_t fit = (uint64_t)(free_b * 0.85) / (16ull * 11ull);
        const uint64_t fit_frames = fit > c->stream_slack ? (fit - c->stream_slack) / px : 0;
        frames = std::min<uint64_t>(frames, fit_frames);
    }
    if (frames == 0) return fail(c, VPT_ERR_DEVICE, "out of device memory for the media streams (176 bytes per resident path)");
    const size_t sst = (((size_t)frames * px + c->stream_slack) + 63) & ~(size_t)63;
    if (hipMalloc(&c->media_block, sst * 16 * 11) != hipSuccess) {
        (void)hipGetLastError();
        c->media_block = nullptr;
        return fail(c, VPT_ERR_DEVICE, "out of device memory for the media streams (176 bytes per resident path): lower vpt_config.frames_in_flight");
    }
    c->media_frames = (uint32_t)frames;
    float4* q = (float4*)c->media_block;
    c->ms.MS = q;
    for (int k = 0; k < 10; k++) c->ms.MP[k] = q + (size_t)(k + 1) * sst;
    return VPT_OK;
}

int ensure_sorted_buffers(vpt_ctx* c) {
    if (c->cls_q) return VPT_OK;
    const size_t scap = c->ss.cap;
    for (uint32_t k = 0; k < kShadeClasses; k++) if (!c->class_queue[k]) HIPCHK(c, hipMalloc((void**)&c->class_queue[k], scap * 4));
    HIPCHK(c, hipMalloc((void**)&c->cls_q, scap));
    return VPT_OK;
}

int ensure_legacy_buffers(vpt_ctx* c) {
    const size_t cap = c->ps.capacity, stride = (cap + 63) & ~(size_t)63;
    if (!c->ps_legacy) {
        HIPCHK(c, hipMalloc(&c->ps_legacy, stride * (16 * 13 + 4)));
        float4* q = (float4*)c->ps_legacy;
        PathState& s = c->ps;
        s.L = q + stride * 12;
        s.A = q; s.B = q + stride; s.T[0] = q + stride * 2; s.T[1] = q + stride * 3; s.H = q + stride * 4;
        s.CE = q + stride * 5; s.CS = q + stride * 6; s.CSO = q + stride * 7; s.CSD = q + stride * 8; s.CL = q + stride * 9; s.CLO = q + stride * 10; s.CLD = q + stride * 11;
        s.hinst = (uint32_t*)(q + stride * 13);
    }
    if (!c->cqueue) HIPCHK(c, hipMalloc((void**)&c->cqueue, cap * 4));   // (a failed call leaves what it got; the next one completes it)
    return VPT_OK;
}

void sync_params(vpt_ctx* c) {
    RenderParams& P = c->P;
    const vpt_params& p = c->params;
    P.samples_per_frame = p.samples_per_frame; P.max_depth = p.max_depth;
    P.max_luminance = p.max_luminance; P.focus_distance = p.focus_distance; P.dof_strength = p.dof_strength;
    P.sky_azimuth = p.sky_azimuth; P.sky_altitude = p.sky_altitude; P.sky_intensity = p.sky_intensity;
    P.emissive_pdf_bias = p.emissive_pdf_bias; P.flags = p.flags; P.base_seed = p.base_seed;
    P.split = p.screen_chunk_count; P.launch_off = c->d_launch_off;
    // the angles exactly as the shaders form them (Sampler.slang:333-334, Miss.slang:28-29); same header, same bits as on the device
    const float VPT_PI = 3.1415926535897F;   // shading.hpp's M_PI
    vptfp::sincos_(P.sky_azimuth / 180.0f * VPT_PI, &P.sky_rot[0], &P.sky_rot[1]);
    vptfp::sincos_(P.sky_altitude / 180.0f * VPT_PI, &P.sky_rot[2], &P.sky_rot[3]);
    vptfp::sincos_(-(P.sky_altitude / 180.0f * VPT_PI), &P.sky_rot[4], &P.sky_rot[5]);
    vptfp::sincos_(-(P.sky_azimuth / 180.0f * VPT_PI), &P.sky_rot[6], &P.sky_rot[7]);
}

void reset_accum(vpt_ctx* c) { c->frame_count = 0; c->dispatch_count = 0; c->samples_accum = 0; }  // PathTracer.h:183

// Emissive-mesh list, PathTracer.cpp:449-469 (and SetMaterial's rebuild, 712-794: same resulting order
// only for additions at the end; we rebuild from instance order, which is what SetScene produces).
void build_emissive(vpt_ctx* c) {
    c->emissive.clear(); c->emissive_tris = 0;
    for (uint32_t i = 0; i < c->instances.size(); i++) {
        const vpt_material& m = c->materials[c->instances[i].material];
        if (m.emissive_color[0] != 0.0f || m.emissive_color[1] != 0.0f || m.emissive_color[2] != 0.0f) {
            EmissiveDesc e;
            e.mesh = c->instances[i].mesh; e.material = c->instances[i].material;
            e.tri_count = c->meshes[e.mesh].tri_count; e.instance = i;
            memcpy(e.xform, c->instances[i].xform, 64);
            c->emissive.push_back(e); c->emissive_tris += e.tri_count;
        }
    }
}
int upload_emissive(vpt_ctx* c) {
    if (c->emissive.size() > VPT_MAX_EMISSIVE_MESHES) return fail(c, VPT_ERR_LIMIT, "too many emissive meshes");
    if (!c->emissive.empty())
        HIPCHK(c, hipMemcpy(c->d_emissive, c->emissive.data(), c->emissive.size() * sizeof(EmissiveDesc), hipMemcpyHostToDevice));
    c->dsc.emissive_count = (uint32_t)c->emissive.size();
    c->dsc.emissive_tris = c->emissive_tris;
    // per-light-triangle table (world-space corners, normal, area)
    std::vector<uint32_t> off(std::max<size_t>(1, c->emissive.size()), 0u);
    uint32_t total = 0;
    for (size_t k = 0; k < c->emissive.size(); k++) { off[k] = total; total += c->emissive[k].tri_count; }
    HIPCHK(c, hipMemcpy(c->d_emissive_tri_offset, off.data(), off.size() * 4, hipMemcpyHostToDevice));
    launch_precompute_emissive(c->stream, c->dsc, c->d_emissive_tri, total);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VPT_OK;
}

// LoadEnvironmentMap, PathTracer.cpp:1161-1296: per-texel importance = solid angle * max(rgb), alias
// table (Vose-style pairing with the reference's pre-increment partition quirk), pdf into alpha.
void build_env_tables(const float* rgba, uint32_t w, uint32_t h, std::vector<float>& env, std::vector<AliasEntry>& alias) {
    const uint64_t size = (uint64_t)w * h;
    env.assign(rgba, rgba + size * 4);
    std::vector<float> importance(size);
    float cos_prev = 1.0f;
    const float step_phi = 2.0f * 3.14159265358979323846f / (float)w;
    const float step_theta = 3.14159265358979323846f / (float)h;
    for (uint32_t y = 0; y < h; y++) {
        float cos_next = vptfp::cos_((float)(y + 1) * step_theta);
        float area = (cos_prev - cos_next) * step_phi;
        cos_prev = cos_next;
        const float* row = &env[(size_t)y * w * 4];
        for (uint32_t x = 0; x < w; x++) importance[(size_t)y * w + x] = area * std::max(row[x * 4], std::max(row[x * 4 + 1], row[x * 4 + 2]));
    }
    float sum = 0.0f;
    for (uint64_t i = 0; i < size; i++) sum = sum + importance[i];  // std::accumulate in fp32, in order
    const float average = sum / (float)size;
    alias.resize(size);
    for (uint64_t i = 0; i < size; i++) { alias[i].importance = (average == 0.0f) ? 0.0f : importance[i] / average; alias[i].alias = (uint32_t)i; }
    std::vector<uint32_t> table(size + 1, 0u);
    uint32_t lo = 0, hi = (uint32_t)size;
    for (uint32_t i = 0; i < size; i++) {
        if (alias[i].importance < 1.0f) table[++lo] = i;  // upstream pre-increments: slot 0 stays 0
        else table[--hi] = i;
    }
    for (lo = 0; lo < hi && hi < size; lo++) {
        const uint32_t l = table[lo], g = table[hi];
        alias[l].alias = g;
        alias[g].importance -= 1.0f - alias[l].importance;
        if (alias[g].importance < 1.0f) hi++;
    }
    for (uint64_t i = 0; i < size; i++) {
        float m = std::max(env[i * 4], std::max(env[i * 4 + 1], env[i * 4 + 2]));
        env[i * 4 + 3] = (sum == 0.0f) ? 0.0f : m / sum;
    }
}

// Does every path end within max_depth * samples_per_frame bounces?  A bounce either raises payload.Depth or ends the path — except a
// scattering event INSIDE a medium (ClosestHit.slang:80-116: depth unchanged), which needs a transmissive material whose medium has a
// density and an anisotropy other than 1 (shade_core.hpp); media (volumes / atmosphere) raise the depth per event but their batches
// run stages with host-visible fallbacks, so they count as unbounded too.
void update_depth_bounded(vpt_ctx* c) {
    bool bounded = true;
    for (const vpt_material& m : c->materials)
        if (m.transmission > 0.0f && m.medium_density != 0.0f && m.medium_anisotropy != 1.0f) bounded = false;
    c->depth_bounded = bounded;
    // the scene class the fused kernel is specialised for (kernels_path.hip k_bounce<PLAIN>): what k_precompute_materials turns into
    // MatResolved.flags == 63 for every material, and k_precompute_lights into uniform light samplers
    bool plain = c->dsc.env_black != 0u && !(c->cfg.build_flags & VPT_BUILD_GENERAL_KERNELS);
    auto one = [&](uint32_t t) { return t < c->tex_1x1.size() && c->tex_1x1[t] != 0; };
    for (const vpt_material& m : c->materials)
        if (!(one(m.base_color_texture) && one(m.normal_texture) && one(m.roughness_texture) && one(m.metallic_texture) && one(m.emissive_texture))) plain = false;
    c->scene_plain = plain;
    c->primary_blocks = (plain && c->lds_scene) ? c->primary_blocks_plain : c->primary_blocks_general;
}

constexpr int kSpillPatternByte = 0x7f;
constexpr uint32_t kSpillPattern = 0x7f7f7f7fu;
__global__ __launch_bounds__(256) void k_count_spilled(const uint32_t* p, uint32_t n, unsigned long long* out) {
    unsigned long long cnt = 0ull;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) cnt += p[i] != kSpillPattern ? 1ull : 0ull;
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
    if ((threadIdx.x & 63u) == 0u && cnt) atomicAdd(out, cnt);
}

// Which shade classes occur in the scene: the staged pipeline launches the shade stage once per class that does.
int update_class_present(vpt_ctx* c) {
    std::vector<unsigned char> cls(c->instances.size());
    if (!cls.empty()) HIPCHK(c, hipMemcpy(cls.data(), c->d_inst_class, cls.size(), hipMemcpyDeviceToHost));
    c->class_present = 1u << kShadeMiss;
    for (unsigned char k : cls) if (k < kShadeClasses) c->class_present |= 1u << k;
    return VPT_OK;
}

void begin_timing(vpt_ctx* c, int kernel, hipEvent_t* a, hipEvent_t* b) {
    *a = *b = nullptr;
    c->stats.kernel_launches[kernel]++;
    if (!c->cfg.profile) return;
    if (c->ev_next + 2 > c->ev_pool.size()) {
        for (int i = 0; i < 64; i++) { hipEvent_t e; (void)hipEventCreate(&e); c->ev_pool.push_back(e); }
    }
    *a = c->ev_pool[c->ev_next++]; *b = c->ev_pool[c->ev_next++];
    (void)hipEventRecord(*a, c->stream);
    c->pending.push_back({kernel, *a, *b});
}
void end_timing(vpt_ctx* c, hipEvent_t b) { if (b) (void)hipEventRecord(b, c->stream); }
void collect_timing(vpt_ctx* c) {  // call after a stream sync
    for (auto& p : c->pending) { float ms = 0.0f; if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) c->stats.kernel_ms[p.kernel] += ms; }
    c->pending.clear(); c->ev_next = 0;
}
#define TIMED(ctx, kid, launch_expr)            \
    do {                                        \
        hipEvent_t ea_, eb_;                    \
        begin_timing(ctx, kid, &ea_, &eb_);     \
        launch_expr;                            \
        end_timing(ctx, eb_);                   \
    } while (0)

// ---- One batch of `frames` consecutive dispatches starting at dispatch index `dispatch_base`, in stages: batch_begin (camera rays /
// bounce 0), batch_bounces (k more bounces), batch_resolve (the guarded resolve + the counters on their way to pinned host memory),
// batch_check (host synchronisation: how many paths are still alive).  The bounce loop runs without host round-trips: every stage
// reads its queue size from device memory, so the host only looks at the counters every few bounces (render_batch) or not at all
// until somebody waits (vpt_render_async).
bool media_on_streams(const vpt_ctx* c) {
    const bool vol = !c->volumes.empty() || c->dsc.atm_on;
    return vol && !c->lds_scene && (c->cfg.pipeline == VPT_PIPELINE_AUTO || c->cfg.pipeline == VPT_PIPELINE_STAGED);
}
int batch_begin(vpt_ctx* c, uint32_t frames, uint32_t dispatch_base, BatchState& b) {
    hipStream_t s = c->stream;
    b = BatchState{};
    b.frames = frames; b.dispatch_base = dispatch_base;
    (c->owner ? c->owner : c)->spill_dirty = true;
    if (frames == 0 || frames > c->frames_alloc) return fail(c, VPT_ERR_DEVICE, "internal: batch larger than the path buffers");
    // path regeneration: only `resident` frames of the batch's samples are in flight; the room ended paths leave in the next ray queue is
    // refilled with the batch's next unstarted samples behind every shade stage (kernels_stream.hip k_refill_plan), until they are used up
    const bool whole = whole_applies(c, frames);   // (keeps its paths in registers: no records, nothing to regenerate)
    const uint32_t resident = whole ? frames : std::min(frames, c->resident_alloc);
    const bool regen = resident < frames;
    if (regen && !regen_allowed(c)) return fail(c, VPT_ERR_DEVICE, "internal: this batch needs all of its samples resident");
    uint32_t n_slots = frames * c->P.shard_pixels;  // samples of the batch
    const uint32_t S = c->P.split;
    if (S > 1) {  // RayTrace(ceil(W/S), ceil(H/S)) per dispatch, in-bounds part only (PathTracer.cpp:145-150, RayGen.slang:24)
        std::vector<uint32_t> off(frames + 1, 0u);
        for (uint32_t k = 0; k < frames; k++) {
            uint32_t ch = (dispatch_base + k) % (S * S), cx = ch % S, cy = ch / S;
            uint32_t lw = cx < c->P.width ? (c->P.width - cx + S - 1) / S : 0, lh = cy < c->P.height ? (c->P.height - cy + S - 1) / S : 0;
            off[k + 1] = off[k] + lw * lh;
        }
        n_slots = off[frames];
        HIPCHK(c, hipMemcpyAsync(c->d_launch_off, off.data(), off.size() * 4, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipStreamSynchronize(s));  // `off` is a stack-lifetime staging buffer
    }
    b.n_slots = n_slots;
    b.regen = regen;
    b.n_first = regen ? resident * c->P.shard_pixels : n_slots;   // launch-grid size of the camera-ray kernel = the most paths ever resident
    b.count = c->cfg.count_traversal != 0;
    // fused (one kernel per bounce, bounce 0 included) when the BVH rides in LDS; staged otherwise
    const bool vol = !c->volumes.empty() || c->dsc.atm_on;
    // AUTO: fused for LDS-resident scenes, the stream pipeline otherwise (atrium 1150 vs 575, glass bust 2410 vs 1290, Cornell box with
    // the 960-triangle glass sphere 2880 vs 2600 Msamples/s: no scene measured prefers the fused kernel once its BVH lives in memory)
    // Media (volumes / atmosphere): on the streams when the BVH lives in memory (kernels_media.hip: the traversal then runs on the
    // vote-scheduled kernels), in the fused per-bounce kernel when it rides in LDS or when the fused pipeline is asked for.
    b.media_stream = media_on_streams(c);
    b.fused = (vol && !b.media_stream) || c->cfg.pipeline == VPT_PIPELINE_FUSED || (c->cfg.pipeline == VPT_PIPELINE_AUTO && c->lds_scene) || c->cfg.pipeline == VPT_PIPELINE_WHOLE;
    b.whole = whole;
    if (c->cfg.pipeline == VPT_PIPELINE_WHOLE && !b.whole)
        return fail(c, VPT_ERR_UNSUPPORTED, "VPT_PIPELINE_WHOLE needs a scene whose BVH rides in LDS, no media, samples_per_frame == 1 and every sample of a batch resident");
    b.stream = !b.fused && c->cfg.pipeline != VPT_PIPELINE_STAGED_R1;   // (a scene that rides in LDS and is forced into the staged pipeline runs the stream kernels on its tree in memory; round 1's stage kernels: laboratory build, VPT_PIPELINE_STAGED_R1)
    if (vol && c->cfg.pipeline == VPT_PIPELINE_STAGED && c->lds_scene)
        return fail(c, VPT_ERR_UNSUPPORTED, "media with VPT_PIPELINE_STAGED need a scene whose BVH lives in memory (this one rides in LDS: use VPT_PIPELINE_AUTO or _FUSED)");
    if (b.media_stream) {
        int rl = ensure_media_buffers(c); if (rl != VPT_OK) return rl;
        if (frames > c->media_frames) return fail(c, VPT_ERR_DEVICE, "internal: media batch larger than the media streams");
    }
#if VPT_LAB
    if (!b.fused && !b.stream) { int rl = ensure_legacy_buffers(c); if (rl != VPT_OK) return rl; }
#else
    if (!b.fused && !b.stream) return fail(c, VPT_ERR_UNSUPPORTED, "VPT_PIPELINE_STAGED_R1 needs the laboratory build");
#endif
    if (b.stream && c->cfg.pipeline == VPT_PIPELINE_STAGED_SORTED) { int rl = ensure_sorted_buffers(c); if (rl != VPT_OK) return rl; }
    b.min_bounces = (uint64_t)c->P.max_depth * c->P.samples_per_frame;
    b.iter_cap = (b.min_bounces * 4ull + 1024ull) * ((frames + resident - 1) / resident);
    b.sorted = c->cfg.pipeline == VPT_PIPELINE_STAGED_SORTED;
    // Two streams: the shadow-ray kernels and the join of bounce k run beside the extend of bounce k + 1 (which needs only the ray
    // queue shade k wrote), so the tail of one persistent traversal kernel is filled by the next one's first blocks.  Off while
    // kernels are timed or visits counted (one kernel at a time then) and in the sorted pipeline.
    b.overlap = b.stream && !b.sorted && !c->cfg.profile && !b.count && !b.media_stream && !c->capturing && !c->owner;   // (a lane's batch stays on its one stream: the lanes overlap each other)
    const bool finisher = !(c->cfg.build_flags & VPT_BUILD_STREAMS_ONLY);
    if (finisher && b.stream && !b.media_stream && !b.regen && n_slots <= kFinishSmallBatchPaths) b.finish_at = kFinishAfterBounces;
    if (n_slots == 0) return VPT_OK;
    HIPCHK(c, hipMemsetAsync(c->ctr, 0, offsetof(Counters, stat_closest), s));  // queue words only, stat_* keep running
    if (b.whole) {  // the batch's paths from camera ray to their end in one launch; no queue is written, alive3[] stays 0 for the resolve's guard
        const uint32_t grid = std::max(1u, std::min<uint32_t>((uint32_t)std::min(c->whole_blocks, c->primary_blocks), (n_slots + 255u) / 256u));   // (blocks of 256 lanes)
        // tiles of 64 samples: `rounds` per wave; mode 0: the first round static, mode 1: all but the last, mode 2: half of them; the rest through the counter
        const uint32_t n_waves = grid * 4u, rounds = ((n_slots + 63u) / 64u) / n_waves, mode = c->lab_whole_sched >> 4;
        const uint32_t static_rounds = (mode == 0u || mode == 3u) ? std::min(rounds, 1u) : mode == 1u ? (rounds >= 2u ? rounds - 1u : 0u) : rounds / 2u;
        // (mode 3: mode 0 with guided chunks — at most the given tiles per atomic, fewer towards the end of the batch)
        TIMED(c, VPT_K_PRIMARY, launch_whole(s, grid, b.count, c->dsc, c->P, c->ps, c->ctr, n_slots, dispatch_base, c->scene_plain, static_rounds, std::max(1u, c->lab_whole_sched & 15u) | (mode == 3u ? 0x100u : 0u)));
        b.parity = 1; b.k3 = 1; b.iter = 1;
    } else if (b.fused) {  // bounce 0 of every slot needs no input records; survivors land in queue[1]
        TIMED(c, VPT_K_PRIMARY, launch_bounce(s, (uint32_t)c->primary_blocks, c->lds_scene, b.count, true, c->dsc, c->P, c->ps, c->ss, nullptr, c->queue[1], c->ctr, 0u, b.n_first, dispatch_base, 0u, c->scene_plain));
        b.parity = 1; b.k3 = 1; b.iter = 1;
    } else if (b.stream) {
        TIMED(c, VPT_K_PRIMARY, launch_raygen_stream(s, c->P, c->ps, c->ss, c->queue[0], b.n_first, dispatch_base, b.media_stream));
        launch_stream_begin(s, c->sctr, b.n_first, n_slots);
        b.parity = 0;
    } else {
#if VPT_LAB
        TIMED(c, VPT_K_PRIMARY, launch_raygen(s, c->P, c->ps, c->queue[0], c->ctr, n_slots, dispatch_base));
#endif
        b.parity = 0;
    }
    return VPT_OK;
}

int batch_bounces(vpt_ctx* c, BatchState& b, uint32_t bounces) {
    hipStream_t s = c->stream;
    const bool count = b.count, sorted = b.sorted, overlap = b.overlap;
    uint32_t& parity = b.parity;
    const uint32_t n_slots = b.n_first;   // (upper bound of a queue's live entries)
    if (n_slots == 0 || b.whole) return VPT_OK;   // (a whole-path batch has no bounces left to launch)
    for (uint32_t j = 0; j < bounces; j++) {
        b.iter++;
        if (b.fused) {  // no reset kernel in between: the bounce kernels rotate three queue-size words
            const int grid = (c->tail_blocks > 0 && b.iter >= 3) ? c->tail_blocks : c->primary_blocks;   // (b.iter counts bounce 0)
            TIMED(c, VPT_K_BOUNCE, launch_bounce(s, (uint32_t)grid, c->lds_scene, count, false, c->dsc, c->P, c->ps, c->ss, c->queue[parity], c->queue[parity ^ 1u], c->ctr, parity, 0u, 0u, b.k3, c->scene_plain));
            parity ^= 1u; b.k3 = (b.k3 + 1u) % 3u;
            continue;
        }
        // a memory-resident BVH runs the staged pipeline on the vote-scheduled traversal kernels and compact streams
        // (kernels_trace.hip, kernels_stream.hip); round 1's stage kernels serve LDS-resident scenes forced into the staged
        // pipeline and VPT_PIPELINE_STAGED_R1
        if (b.media_stream) {   // distance -> scatter -> extend -> shade -> sky rays, light rays -> tail (kernels_media.hip), one stream
            launch_prepare_stream(s, c->sctr, parity);
            TraceArgs a{};
            a.ro = c->ss.RA[parity]; a.rd = c->ss.RB[parity]; a.order = nullptr; a.valid = c->queue[parity]; a.hit = c->ss.SH; a.hinst = c->ss.SHI; a.cls = nullptr;
            a.n = 0; a.n_dev = &c->sctr->queue_len[parity].v; a.store_gid = 1u; a.param = c->vote_param;
            // GetDistanceToGeometry (RTCommon.slang:86-101): the payload direction as it is, TMin 1e-5, TMax 1e6
            a.head = &c->sctr->shade_head.v; a.tmin = 0.00001f; a.tmax = 1000000.0f; a.normalize_dir = 0u;
            if (!(c->P.flags & VPT_FLAG_RAY_QUERIES)) { a.tmax = 1000.0f; a.normalize_dir = 1u; }   // RTCommon.slang:103-117
            TIMED(c, VPT_K_EXTEND, launch_trace(s, (uint32_t)c->vote_blocks, VPT_TRACE_VOTE, false, count, c->dsc, a, c->ctr));
            TIMED(c, VPT_K_SHADE, launch_media_scatter(s, (uint32_t)c->shade_blocks, c->dsc, c->ps, c->ss, c->ms, c->queue[parity], c->sctr, parity));
            a.head = &c->sctr->extend_head.v; a.tmin = 0.01f; a.tmax = 100000.0f; a.normalize_dir = 1u;
            TIMED(c, VPT_K_EXTEND, launch_trace(s, (uint32_t)c->vote_blocks, VPT_TRACE_VOTE, false, count, c->dsc, a, c->ctr));
            launch_layout_media(s, c->sctr, parity, (uint32_t)c->shade_media_blocks * 4u, (uint32_t)c->media_tail_blocks * 4u);
            TIMED(c, VPT_K_SHADE, launch_shade_media(s, (uint32_t)c->shade_media_blocks, c->dsc, c->P, c->ps, c->ss, c->ms, c->queue[parity], c->ctr, c->sctr, parity));
            TIMED(c, VPT_K_SHADOW, launch_trace_shadow(s, (uint32_t)c->shadow_blocks, false, count, c->dsc, c->ss, c->ctr, c->sctr, c->vote_param, (c->P.flags & VPT_FLAG_RAY_QUERIES) ? 1u : 0u));
            TIMED(c, VPT_K_SHADOW, launch_trace_shadow(s, (uint32_t)c->shadow_blocks, true, count, c->dsc, c->ss, c->ctr, c->sctr, c->vote_param, (c->P.flags & VPT_FLAG_RAY_QUERIES) ? 1u : 0u));
            TIMED(c, VPT_K_JOIN, launch_media_tail(s, (uint32_t)c->media_tail_blocks, c->dsc, c->P, c->ps, c->ss, c->ms, c->queue[parity], c->queue[parity ^ 1u], c->ctr, c->sctr, parity));
            parity ^= 1u;
            continue;
        }
        if (b.stream && b.finished) continue;   // k_finish has been enqueued: nothing is alive behind it
        if (b.stream && b.finish_at != 0u && b.iter > b.finish_at) {   // (b.iter counts this bounce): the rest of the batch in one launch
            if (b.join_pending) { HIPCHK(c, hipStreamWaitEvent(s, c->ev_join, 0)); b.join_pending = false; }   // pathLight of the queue's entries is final behind the previous join
            TIMED(c, VPT_K_BOUNCE, launch_finish(s, (uint32_t)c->finish_blocks, count, c->dsc, c->P, c->ps, c->ss, c->queue[parity], c->sctr, c->ctr, parity));
            b.finished = true;
            continue;
        }
        if (b.stream) {   // stream pipeline: extend -> classify -> shade per class (streams out) -> sky rays, light rays -> join
            launch_prepare_stream(s, c->sctr, parity);
            TraceArgs a{};
            a.ro = c->ss.RA[parity]; a.rd = c->ss.RB[parity]; a.order = nullptr; a.valid = c->queue[parity]; a.hit = c->ss.SH; a.hinst = c->ss.SHI; a.cls = c->cls_q;
            a.n = 0; a.n_dev = &c->sctr->queue_len[parity].v; a.head = &c->sctr->extend_head.v;
            a.tmin = 0.01f; a.tmax = 100000.0f; a.normalize_dir = 1u; a.store_gid = 1u; a.param = c->vote_param;
            if (!sorted) a.cls = nullptr;
            TIMED(c, VPT_K_EXTEND, launch_trace(s, (uint32_t)c->vote_blocks, VPT_TRACE_VOTE, false, count, c->dsc, a, c->ctr));
            // the shade stage of this bounce overwrites the pending records and shadow-ray streams the join of the previous
            // bounce reads (overlapped mode: that join runs on the second stream, beside the extend launched above)
            if (overlap && b.join_pending) { HIPCHK(c, hipStreamWaitEvent(s, c->ev_join, 0)); b.join_pending = false; }
            if (sorted) {   // the shade queue sorted by material class: one dense queue and one launch per class present in the scene
                TIMED(c, VPT_K_SHADE, launch_classify(s, c->queue[parity], c->cls_q, c->class_queue, c->sctr, parity, n_slots + c->stream_slack, (uint32_t)c->shade_stream_blocks * 4u));
                for (uint32_t k = 0; k < kShadeClasses; k++)
                    if (c->class_present & (1u << k))
                        TIMED(c, VPT_K_SHADE, launch_shade_stream(s, (uint32_t)c->shade_stream_blocks, k, true, c->dsc, c->P, c->ps, c->ss, c->queue[parity], c->class_queue[k], c->queue[parity ^ 1u], c->ctr, c->sctr, parity));
            } else {
                launch_layout_single(s, c->sctr, parity, (uint32_t)c->shade_stream_blocks * 4u);
                TIMED(c, VPT_K_SHADE, launch_shade_stream(s, (uint32_t)c->shade_stream_blocks, 0u, false, c->dsc, c->P, c->ps, c->ss, c->queue[parity], nullptr, c->queue[parity ^ 1u], c->ctr, c->sctr, parity));
            }
            hipStream_t sb = s;
            DeviceScene dsc_shadow = c->dsc;
            if (overlap) {   // shadow rays and join of this bounce on the second stream: the next bounce's extend does not depend on them
                HIPCHK(c, hipEventRecord(c->ev_shade, s));
                HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_shade, 0));
                sb = c->stream2;
                dsc_shadow.stack_overflow = c->stack_overflow2;   // its own stack spill region: it runs beside the next extend
            }
            // regeneration: fresh camera rays into the room the ended paths left in the next queue (entries behind the ones the join of this
            // bounce addresses, so it may run beside the shadow kernels and the join)
            if (b.regen) TIMED(c, VPT_K_PRIMARY, launch_refill(s, 2048u, c->P, c->ps, c->ss, c->queue[parity ^ 1u], c->sctr, parity ^ 1u, b.n_first, b.dispatch_base));
            TIMED(c, VPT_K_SHADOW, launch_trace_shadow(sb, (uint32_t)c->shadow_blocks, false, count, dsc_shadow, c->ss, c->ctr, c->sctr, c->vote_param, (c->P.flags & VPT_FLAG_RAY_QUERIES) ? 1u : 0u));
            TIMED(c, VPT_K_SHADOW, launch_trace_shadow(sb, (uint32_t)c->shadow_blocks, true, count, dsc_shadow, c->ss, c->ctr, c->sctr, c->vote_param, (c->P.flags & VPT_FLAG_RAY_QUERIES) ? 1u : 0u));
            TIMED(c, VPT_K_JOIN, launch_join(sb, (uint32_t)c->join_blocks, c->P, c->ps, c->ss, c->sctr, c->queue[parity], c->queue[parity ^ 1u], parity));
            if (overlap) { HIPCHK(c, hipEventRecord(c->ev_join, c->stream2)); b.join_pending = true; }
            parity ^= 1u;
            continue;
        }
#if VPT_LAB
        launch_prepare(s, c->ctr, parity);
        TIMED(c, VPT_K_EXTEND, launch_extend(s, (uint32_t)c->trav_blocks, c->lds_scene, count, c->dsc, c->ps, c->queue[parity], c->ctr, parity));
        TIMED(c, VPT_K_SHADE, launch_shade(s, (uint32_t)c->shade_blocks, c->dsc, c->P, c->ps, c->queue[parity], c->queue[parity ^ 1u], c->cqueue, c->ctr, parity));
        TIMED(c, VPT_K_CONNECT, launch_connect(s, (uint32_t)c->trav_blocks, c->lds_scene, count, c->dsc, c->P, c->ps, c->cqueue, c->ctr, parity));
#endif
        parity ^= 1u;
    }
    return VPT_OK;
}

// The resolve rides right behind the bounces that are expected to be the last ones; it does nothing if a path is still alive
// (in-medium walks do not consume depth), in which case more bounces and another resolve follow.  Behind it the counters travel to
// pinned host memory, for whoever synchronises next.
int batch_resolve(vpt_ctx* c, BatchState& b) {
    hipStream_t s = c->stream;
    if (b.n_slots == 0) return VPT_OK;
#if VPT_LAB
    if (!b.fused && !b.stream) launch_fold(s, c->ctr);
#endif
    if (b.overlap && b.join_pending) { HIPCHK(c, hipStreamWaitEvent(s, c->ev_join, 0)); b.join_pending = false; }   // the resolve reads the frame sums the join writes
    const uint32_t* guard = b.fused ? &c->ctr->alive3[b.k3] : b.stream ? &c->sctr->alive[b.parity].v : &c->ctr->ray_count[b.parity];
    TIMED(c, VPT_K_RESOLVE, launch_resolve(s, c->P, c->ps, c->image, b.frames, b.dispatch_base, guard));
    HIPCHK(c, hipMemcpyAsync(&c->h_ctr->ctr, c->ctr, sizeof(Counters), hipMemcpyDeviceToHost, s));
    if (b.stream) {  // the exact number of live paths, and the queue length (holes included), which must fit the queue allocation
        HIPCHK(c, hipMemcpyAsync(&c->h_ctr->alive[0], &c->sctr->alive[b.parity].v, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(&c->h_ctr->queue_len[0], &c->sctr->queue_len[b.parity].v, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(&c->h_ctr->refill_next, &c->sctr->refill_next, 4, hipMemcpyDeviceToHost, s));
    }
    return VPT_OK;
}

// The device-side ray statistics are running totals per lane (Counters::stat_*), copied to pinned memory behind every resolve.
void update_ray_stats(vpt_ctx* c) {
    vpt_ctx* root = c->owner ? c->owner : c;
    unsigned long long v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto add = [&](const vpt_ctx* x) {
        const Counters& h = x->h_ctr->ctr;
        v[0] += h.stat_closest; v[1] += h.stat_shadow; v[2] += h.stat_connect; v[3] += h.stat_primary_hits; v[4] += h.stat_primary_alive; v[5] += h.stat_primary_rays;
        v[6] += h.stat_finish_paths; v[7] += h.stat_finish_closest; v[8] += h.stat_finish_shadow;
    };
    add(root);
    for (vpt_ctx* L : root->lanes) if (L) add(L);
    root->stats.closest_rays = v[0]; root->stats.shadow_rays = v[1]; root->stats.connect_paths = v[2];
    root->stats.primary_hits = v[3]; root->stats.primary_survivors = v[4]; root->stats.primary_shadow_rays = v[5];
    root->stats.finish_paths = v[6]; root->stats.finish_closest_rays = v[7]; root->stats.finish_shadow_rays = v[8];
}

// Host synchronisation: statistics, overflow checks, *alive = paths of the batch still in flight.
int batch_check(vpt_ctx* c, BatchState& b, uint32_t* alive) {
    *alive = 0;
    if (b.n_slots == 0) return VPT_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    collect_timing(c);
    const Counters& h = c->h_ctr->ctr;
    update_ray_stats(c);
    const bool counted = c->cfg.count_traversal != 0;   // (the finisher and the fused kernel on a tree in memory always count: reported only when asked for, so the figures are never partial)
    c->stats.nodes_visited = counted ? h.stat_nodes : 0;
    c->stats.tris_tested = counted ? h.stat_tris : 0;
    c->stats.shadow_nodes_visited = counted ? h.stat_shadow_nodes : 0;
    c->stats.shadow_tris_tested = counted ? h.stat_shadow_tris : 0;
    uint32_t n = b.fused ? h.alive3[b.k3] : h.ray_count[b.parity];
    if (b.stream) {
        n = c->h_ctr->alive[0];
        const uint64_t len = c->h_ctr->queue_len[0];
        const uint64_t room = b.media_stream ? (uint64_t)c->media_frames * c->P.shard_pixels + c->stream_slack : (uint64_t)c->ps.capacity + c->stream_slack;
        if (len > room || len > (uint64_t)c->ps.capacity + c->stream_slack) { (void)hipStreamSynchronize(c->stream2); return fail(c, VPT_ERR_DEVICE, "internal: stream overflow"); }
    }
    if (n > b.n_first) { (void)hipStreamSynchronize(c->stream2); return fail(c, VPT_ERR_DEVICE, "internal: queue overflow"); }
    if (n != 0 && b.iter > b.iter_cap) { (void)hipStreamSynchronize(c->stream2); return fail(c, VPT_ERR_DEVICE, "internal: bounce loop did not terminate"); }
    *alive = n;
    return VPT_OK;
}

// Runs a begun batch to its end: resolve + check, and while paths are alive four more bounces at a time.
int batch_finish(vpt_ctx* c, BatchState& b, bool resolve_enqueued) {
    while (true) {
        if (!resolve_enqueued) { int rc = batch_resolve(c, b); if (rc) return rc; }
        resolve_enqueued = false;
        uint32_t n = 0;
        int rc = batch_check(c, b, &n);
        if (rc) return rc;
        if (n == 0) break;
        // few paths left — and, in a regenerating batch, no sample left to start: the next launch finishes them
        if (!(c->cfg.build_flags & VPT_BUILD_STREAMS_ONLY) && b.stream && !b.media_stream && (!b.regen || c->h_ctr->refill_next >= b.n_slots) && !b.finished && b.finish_at == 0u && n < kFinishBelowPaths) b.finish_at = (uint32_t)b.iter;
        rc = batch_bounces(c, b, b.regen ? 8u : 4u);   // (a regenerating batch runs many more launches than max_depth: fewer host round trips)
        if (rc) return rc;
    }
    HIPCHK(c, hipGetLastError());
    return VPT_OK;
}

int render_batch(vpt_ctx* c, uint32_t frames, uint32_t dispatch_base) {
    BatchState b;
    int rc = batch_begin(c, frames, dispatch_base, b);
    if (rc) return rc;
    if (b.n_slots == 0) return VPT_OK;
    // the host looks at the queue after max_depth bounces (when a surface-only batch is done) or after eight, whichever comes first
    const uint32_t first = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(b.min_bounces - (b.fused ? 1 : 0), 1), 8);
    rc = batch_bounces(c, b, first);
    if (rc) return rc;
    rc = batch_finish(c, b, false);
    if (rc) return rc;
    c->stats.samples += (uint64_t)b.n_slots * c->P.samples_per_frame;
    return VPT_OK;
}

// ---- asynchronous batches -------------------------------------------------------------------------------------------------
uint64_t issue_ticket(vpt_ctx* c, hipStream_t on) {
    c->tick_issued++;
    (void)hipEventRecord(c->tick_ev[c->tick_issued % kTickets], on);
    c->async_dirty = true;
    return c->tick_issued;
}
// An enqueued batch whose paths may outlive the bounces enqueued with it: finish it exactly as render_batch would have.
int finish_outstanding(vpt_ctx* c) {
    if (!c->out_active) return VPT_OK;
    c->out_active = false;
    return batch_finish(c, c->out_batch, true);
}
// Everything enqueued so far — on every lane — has finished when this returns (and an unfinished batch has been finished).
int drain(vpt_ctx* c) {
    int rc = finish_outstanding(c);
    if (rc) return rc;
    if (!c->async_dirty) return VPT_OK;
    c->async_dirty = false;
    for (vpt_ctx* L : c->lanes)
        if (L) {
            HIPCHK(c, hipStreamSynchronize(L->stream));
            if (L->last_fixed_valid && (L->last_fixed.stream ? L->h_ctr->alive[0] : L->h_ctr->ctr.alive3[L->last_fixed.k3]) != 0u) return fail(c, VPT_ERR_DEVICE, "internal: a path outlived a fixed-schedule batch");
            L->last_fixed_valid = false;
            for (int k = 0; k < VPT_KERNEL_COUNT; k++) { c->stats.kernel_launches[k] += L->stats.kernel_launches[k]; L->stats.kernel_launches[k] = 0; }
            c->stats.graph_launches += L->stats.graph_launches; L->stats.graph_launches = 0;
        }
    c->order_lane = nullptr; c->post_pending = false;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream2));
    if (c->last_fixed_valid) {   // a fixed-schedule batch on the main lane: nothing may have outlived it either (its guarded resolve would have been a no-op)
        c->last_fixed_valid = false;
        const BatchState& f = c->last_fixed;
        if (f.stream) {
            if (c->h_ctr->alive[0] != 0u) return fail(c, VPT_ERR_DEVICE, "internal: a path outlived a fixed-schedule batch");
            if ((uint64_t)c->h_ctr->queue_len[0] > (uint64_t)c->ps.capacity + c->stream_slack) return fail(c, VPT_ERR_DEVICE, "internal: stream overflow");
        } else if (f.fused && c->h_ctr->ctr.alive3[f.k3] != 0u) return fail(c, VPT_ERR_DEVICE, "internal: a path outlived a fixed-schedule batch");
    }
    collect_timing(c);
    update_ray_stats(c);   // fixed-schedule batches copy their counters to pinned memory too
    HIPCHK(c, hipGetLastError());
    return VPT_OK;
}
void destroy_graph(vpt_ctx* c) {
    if (c->graph) (void)hipGraphExecDestroy(c->graph);
    c->graph = nullptr; c->graph_gen = 0;
}
// A whole batch as a fixed schedule: bounce 0 (or the camera rays) and `bounces_total` bounces in all; the guarded resolve is the caller's.
int enqueue_fixed(vpt_ctx* c, uint32_t frames, uint32_t dispatch_base, uint32_t bounces_total, BatchState& b) {
    int rc = batch_begin(c, frames, dispatch_base, b);
    if (rc) return rc;
    if (b.n_slots == 0) return VPT_OK;
    return batch_bounces(c, b, b.fused ? bounces_total - 1u : bounces_total);
}
// The same through a captured hipGraph: the fused pipeline's batch (memset, bounce 0, bounces) with the first dispatch index read from
// device memory, captured once per (state, frames, bounces) and replayed.  b: the batch as it stands before its resolve.
int enqueue_graph(vpt_ctx* c, uint32_t frames, uint32_t dispatch_base, uint32_t bounces_total, bool* used, BatchState& b) {
    *used = false;
    if (c->graph_broken) return VPT_OK;
    if (!c->graph || c->graph_gen != c->state_gen || c->graph_frames != frames || c->graph_bounces != bounces_total) {
        destroy_graph(c);
        uint64_t before[VPT_KERNEL_COUNT];
        memcpy(before, c->stats.kernel_launches, sizeof(before));
        c->P.dispatch_base_dev = c->d_dispatch_base;
        hipGraph_t g = nullptr;
        bool ok = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
        c->capturing = true;
        int rc = ok ? enqueue_fixed(c, frames, 0u, bounces_total, c->graph_batch) : VPT_ERR_DEVICE;
        c->capturing = false;
        if (ok && hipStreamEndCapture(c->stream, &g) != hipSuccess) { ok = false; g = nullptr; }
        c->P.dispatch_base_dev = nullptr;
        for (int k = 0; k < VPT_KERNEL_COUNT; k++) { c->graph_kernel_launches[k] = c->stats.kernel_launches[k] - before[k]; c->stats.kernel_launches[k] = before[k]; }
        if (ok && rc == VPT_OK && g && hipGraphInstantiate(&c->graph, g, nullptr, nullptr, 0) != hipSuccess) { ok = false; c->graph = nullptr; }
        if (g) (void)hipGraphDestroy(g);
        if (!ok || rc != VPT_OK || !c->graph) {   // capture is an optimisation: without it the batch goes out as plain launches
            (void)hipGetLastError();
            destroy_graph(c);
            c->graph_broken = true;
            c->err.clear();
            return VPT_OK;
        }
        c->graph_gen = c->state_gen; c->graph_frames = frames; c->graph_bounces = bounces_total;
    }
    HIPCHK(c, hipMemsetD32Async((hipDeviceptr_t)c->d_dispatch_base, (int)dispatch_base, 1, c->stream));
    HIPCHK(c, hipGraphLaunch(c->graph, c->stream));
    for (int k = 0; k < VPT_KERNEL_COUNT; k++) c->stats.kernel_launches[k] += c->graph_kernel_launches[k];
    c->stats.graph_launches++;
    b = c->graph_batch;
    b.dispatch_base = dispatch_base;
    *used = true;
    return VPT_OK;
}

// ---- lanes (see vpt_ctx::lanes)
int init_ctx_resources(vpt_ctx* c) {
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_shade, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_resolved, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_post, hipEventDisableTiming) != hipSuccess ||
        hipMalloc((void**)&c->ctr, sizeof(Counters)) != hipSuccess) return VPT_ERR_DEVICE;
    (void)hipMemset(c->ctr, 0, sizeof(Counters));
    if (hipMalloc((void**)&c->sctr, sizeof(StreamCounters)) != hipSuccess) return VPT_ERR_DEVICE;
    (void)hipMemset(c->sctr, 0, sizeof(StreamCounters));
    if (hipMalloc((void**)&c->d_launch_off, (kMaxFramesInFlight + 1) * 4) != hipSuccess) return VPT_ERR_DEVICE;
    for (int k = 0; k < kTickets; k++)
        if (hipEventCreateWithFlags(&c->tick_ev[k], hipEventDisableTiming) != hipSuccess) { c->tick_ev[k] = nullptr; return VPT_ERR_DEVICE; }
    if (hipHostMalloc((void**)&c->h_ctr, sizeof(HostCounters), hipHostMallocDefault) != hipSuccess) { c->h_ctr = nullptr; return VPT_ERR_DEVICE; }
    memset(c->h_ctr, 0, sizeof(HostCounters));
    if (hipMalloc((void**)&c->d_dispatch_base, 256) != hipSuccess || hipMalloc((void**)&c->d_spill_count, 256) != hipSuccess) return VPT_ERR_DEVICE;
    return VPT_OK;
}
void destroy_lane(vpt_ctx* L);
// Lane k of the owner: created on first use, holds one frame of path buffers for the owner's image size.
vpt_ctx* get_lane(vpt_ctx* c, int k) {
    if (c->lanes[k]) return c->lanes[k];
    vpt_ctx* L = new vpt_ctx();
    L->owner = c; L->cfg = c->cfg; L->cfg.frames_in_flight = 1; L->cfg.resident_frames = 1; L->cfg.profile = 0; L->cfg.count_traversal = 0;
    L->cu_count = c->cu_count;
    L->P = c->P;
    bool ok = init_ctx_resources(L) == VPT_OK && alloc_path_buffers(L, 1, 1) == VPT_OK;
    if (ok) {
        const size_t bytes = stack_overflow_bytes((uint32_t)std::max(std::max(std::max(c->primary_blocks_general, c->primary_blocks_plain), c->whole_blocks), c->max_blocks));   // (the stream kernels' grids included)
        ok = hipMalloc(&L->lane_spill, bytes) == hipSuccess && hipMemset(L->lane_spill, 0x7f, bytes) == hipSuccess;   // (kSpillPatternByte: vpt_get_stats counts what was spilled)
        L->stack_overflow_words = (uint32_t)(bytes / 4);
    }
    if (!ok) { (void)hipGetLastError(); destroy_lane(L); return nullptr; }
    L->frames_in_flight = 1; L->buffers_ok = true;
    c->lanes[k] = L;
    return L;
}
// What a lane borrows from its owner, refreshed before every use: scene tables, parameters, camera, the accumulation image.
int sync_lane(vpt_ctx* c, vpt_ctx* L) {
    L->P = c->P; L->P.dispatch_base_dev = nullptr;
    L->dsc = c->dsc; L->dsc.stack_overflow = (uint32_t*)L->lane_spill;
    L->params = c->params;
    L->lds_scene = c->lds_scene; L->scene_plain = c->scene_plain; L->depth_bounded = c->depth_bounded; L->has_scene = true;
    L->primary_blocks = c->primary_blocks; L->whole_blocks = c->whole_blocks; L->lab_whole_frames = c->lab_whole_frames; L->lab_whole_sched = c->lab_whole_sched;
    L->vote_blocks = c->vote_blocks; L->shadow_blocks = c->shadow_blocks; L->shade_stream_blocks = c->shade_stream_blocks; L->join_blocks = c->join_blocks; L->finish_blocks = c->finish_blocks;
    L->max_blocks = c->max_blocks; L->vote_param = c->vote_param; L->class_present = c->class_present; L->stack_overflow2 = (uint32_t*)L->lane_spill;   // (a lane's batches run on one stream: no second region in use)
    L->image = c->image;
    // (vpt_set_params drained every lane before samples_per_frame changed: nothing of this lane is in flight when its per-sample words are replaced)
    // New buffers first — free_path_buffers() bumps the lane's own generation — and the owner's generation assigned BEHIND that: a lane left one
    // generation ahead of its owner would, after one more vpt_set_camera / vpt_set_params on the owner, find its stale captured batch "current" again.
    if (!path_words_ok(L)) {
        destroy_graph(L);   // the captured batch holds the old buffers' addresses
        if (alloc_path_buffers(L, 1, 1) != VPT_OK) { L->buffers_ok = false; c->err = L->err.empty() ? "lane: out of memory" : L->err; return VPT_ERR_OUT_OF_MEMORY; }
    }
    L->state_gen = c->state_gen;   // the owner's generation invalidates the lane's captured batch too
    return VPT_OK;
}
void destroy_lane(vpt_ctx* L) {
    if (!L) return;
    if (L->stream) (void)hipStreamSynchronize(L->stream);
    destroy_graph(L);
    L->image = nullptr; L->full_image = nullptr;   // borrowed
    free_render_buffers(L);
    if (L->lane_spill) (void)hipFree(L->lane_spill);
    for (int k = 0; k < kTickets; k++) if (L->tick_ev[k]) (void)hipEventDestroy(L->tick_ev[k]);
    if (L->h_ctr) (void)hipHostFree(L->h_ctr);
    if (L->d_dispatch_base) (void)hipFree(L->d_dispatch_base);
    if (L->d_spill_count) (void)hipFree(L->d_spill_count);
    if (L->ctr) (void)hipFree(L->ctr);
    if (L->sctr) (void)hipFree(L->sctr);
    if (L->d_launch_off) (void)hipFree(L->d_launch_off);
    if (L->ev_shade) (void)hipEventDestroy(L->ev_shade);
    if (L->ev_join) (void)hipEventDestroy(L->ev_join);
    if (L->ev_resolved) (void)hipEventDestroy(L->ev_resolved);
    if (L->ev_post) (void)hipEventDestroy(L->ev_post);
    if (L->stream2) (void)hipStreamDestroy(L->stream2);
    if (L->stream) (void)hipStreamDestroy(L->stream);
    delete L;
}
void destroy_lanes(vpt_ctx* c) {
    for (vpt_ctx*& L : c->lanes) { destroy_lane(L); L = nullptr; }
    c->order_lane = nullptr;
}

int ensure_post_buffers(vpt_ctx* c) {
    const uint32_t w = c->P.width, h = c->P.height;
    if (c->post_w == w && c->post_h == h && !c->mips.empty()) return VPT_OK;
    for (float* m : c->mips) (void)hipFree(m);
    c->mips.clear(); c->mip_sizes.clear();
    if (c->post_out) (void)hipFree(c->post_out);
    c->post_out = nullptr;
    uint32_t cw = w, ch = h;
    for (int i = 0; i < 10; i++) {  // PostProcessor.cpp:136-157 (MAX_BLOOM_LEVELS = 10)
        float* m = nullptr;
        HIPCHK(c, hipMalloc((void**)&m, (size_t)cw * ch * 16));
        c->mips.push_back(m); c->mip_sizes.push_back({cw, ch});
        if (cw % 2 != 0) cw -= 1;
        if (ch % 2 != 0) ch -= 1;
        cw /= 2; ch /= 2;
        if (cw < 2 || ch < 2) break;
    }
    HIPCHK(c, hipMalloc((void**)&c->post_out, (size_t)w * h * 4));
    c->post_w = w; c->post_h = h;
    return VPT_OK;
}

const float* whole_image(vpt_ctx* c) { return c->P.shard_count > 1 ? c->full_image : c->image; }

}  // namespace

extern "C" {

void vpt_default_params(vpt_params* p) {  // PathTracer.h:197-233
    p->samples_per_frame = 1; p->max_samples = 5000; p->max_depth = 200; p->max_luminance = 500.0f;
    p->focus_distance = 1.0f; p->dof_strength = 0.0f; p->sky_azimuth = 0.0f; p->sky_altitude = 0.0f; p->sky_intensity = 1.0f;
    p->screen_chunk_count = 1; p->emissive_pdf_bias = 0.0f; p->flags = VPT_FLAGS_DEFAULT; p->base_seed = 1;
}
void vpt_default_post_params(vpt_post_params* p) {  // PostProcessor.h:8-21
    p->schedule = VPT_POST_FUSED;
    p->exposure = 1.0f; p->gamma = 2.2f; p->bloom_threshold = 2.0f; p->bloom_strength = 1.0f; p->mip_count = 10; p->falloff_range = 5.0f;
}

vpt_ctx* vpt_create(const vpt_config* cfg, int* err) {
    auto set = [&](int e) { if (err) *err = e; };
    if (!cfg || cfg->width == 0 || cfg->height == 0 || cfg->shard_count == 0 || cfg->shard_rank >= cfg->shard_count || cfg->pipeline > VPT_PIPELINE_WHOLE) { set(VPT_ERR_INVALID_ARGUMENT); return nullptr; }
#if !VPT_LAB
    if (cfg->pipeline == VPT_PIPELINE_STAGED_R1) { set(VPT_ERR_UNSUPPORTED); return nullptr; }   // round 1's stage kernels live in the laboratory build (libvpt_hip_lab.so)
#endif
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) { set(VPT_ERR_NO_DEVICE); return nullptr; }
    if (hipSetDevice(cfg->device) != hipSuccess) { set(VPT_ERR_NO_DEVICE); return nullptr; }
    vpt_ctx* c = new vpt_ctx();
    c->cfg = *cfg;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess) c->cu_count = prop.multiProcessorCount;
    if (init_ctx_resources(c) != VPT_OK) { (void)hipGetLastError(); set(VPT_ERR_DEVICE); vpt_destroy(c); return nullptr; }
    vpt_default_params(&c->params);
    const float id[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    memcpy(c->P.view_inv, id, 64); memcpy(c->P.proj_inv, id, 64);
    sync_params(c);
    int rc = alloc_render_buffers(c);
    if (rc != VPT_OK) { set(rc); vpt_destroy(c); return nullptr; }
    set(VPT_OK);
    return c;
}

void vpt_destroy(vpt_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    destroy_lanes(c);
    destroy_graph(c);
    if (c->comm) { (void)ncclCommDestroy(c->comm); c->comm = nullptr; }
    if (c->ev_resolved) (void)hipEventDestroy(c->ev_resolved);
    if (c->ev_post) (void)hipEventDestroy(c->ev_post);
    for (int k = 0; k < kTickets; k++) if (c->tick_ev[k]) (void)hipEventDestroy(c->tick_ev[k]);
    if (c->h_ctr) (void)hipHostFree(c->h_ctr);
    if (c->d_dispatch_base) (void)hipFree(c->d_dispatch_base);
    if (c->d_spill_count) (void)hipFree(c->d_spill_count);
    free_lab(c);
    free_scene(c);
    free_render_buffers(c);
    if (c->ctr) (void)hipFree(c->ctr);
    if (c->sctr) (void)hipFree(c->sctr);
    if (c->d_launch_off) (void)hipFree(c->d_launch_off);
    if (c->d_volumes) (void)hipFree(c->d_volumes);
    for (DensityGrid& g : c->grids) { (void)hipFree((void*)g.values); (void)hipFree((void*)g.block_max); }
    if (c->d_grids) (void)hipFree(c->d_grids);
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->ev_shade) (void)hipEventDestroy(c->ev_shade);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->stream2) { (void)hipStreamSynchronize(c->stream2); (void)hipStreamDestroy(c->stream2); }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* vpt_last_error(const vpt_ctx* c) { return c ? c->err.c_str() : "null context"; }

int vpt_set_scene(vpt_ctx* c, const vpt_scene_desc* sd) {
    if (!c || !sd) return VPT_ERR_INVALID_ARGUMENT;
    if (sd->mesh_count == 0 || !sd->meshes) return fail(c, VPT_ERR_INVALID_ARGUMENT, "No meshes found in scene");  // PathTracer.cpp:180
    if (sd->mesh_count >= VPT_MAX_ENTITIES || sd->material_count >= VPT_MAX_ENTITIES) return fail(c, VPT_ERR_LIMIT, "too many meshes/materials");
    if (sd->instance_count >= VPT_MAX_INSTANCES) return fail(c, VPT_ERR_LIMIT, "too many mesh instances");
    if (!sd->materials || sd->material_count == 0 || !sd->instances || !sd->textures || sd->texture_count == 0 || !sd->env_rgba ||
        sd->env_width == 0 || sd->env_height == 0 || !sd->lut_reflection || !sd->lut_refraction_outside || !sd->lut_refraction_inside)
        return fail(c, VPT_ERR_INVALID_ARGUMENT, "incomplete scene description");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    { int rd = drain(c); if (rd) return rd; }
    const auto t_scene0 = std::chrono::steady_clock::now();
    // ---- validate the whole description first: a rejected scene leaves the current one untouched
    {
        uint64_t nv = 0, ni = 0, texel_bytes = 0;
        for (uint32_t m = 0; m < sd->mesh_count; m++) {
            const vpt_mesh& me = sd->meshes[m];
            if (!me.vertices || !me.indices || me.index_count % 3 != 0) return fail(c, VPT_ERR_INVALID_ARGUMENT, "bad mesh");
            for (uint32_t k = 0; k < me.index_count; k++) if (me.indices[k] >= me.vertex_count) return fail(c, VPT_ERR_INVALID_ARGUMENT, "mesh index out of range");
            nv += me.vertex_count; ni += me.index_count;
        }
        if (nv > 0xffffffffull || ni > 0xffffffffull) return fail(c, VPT_ERR_LIMIT, "more than 2^32 pooled vertices / indices");
        for (uint32_t i = 0; i < sd->material_count; i++) {
            const vpt_material& m = sd->materials[i];
            if (m.base_color_texture >= sd->texture_count || m.normal_texture >= sd->texture_count || m.roughness_texture >= sd->texture_count ||
                m.metallic_texture >= sd->texture_count || m.emissive_texture >= sd->texture_count)
                return fail(c, VPT_ERR_INVALID_ARGUMENT, "material texture index out of range");
        }
        for (uint32_t i = 0; i < sd->instance_count; i++) {
            if (sd->instances[i].mesh_index >= sd->mesh_count) return fail(c, VPT_ERR_INVALID_ARGUMENT, "instance mesh index out of range");
            if (sd->instances[i].material_index >= sd->material_count) return fail(c, VPT_ERR_INVALID_ARGUMENT, "Mesh instance has invalid material index");  // PathTracer.cpp:454
        }
        for (uint32_t t = 0; t < sd->texture_count; t++) {
            const vpt_texture& tx = sd->textures[t];
            if (!tx.data || tx.width == 0 || tx.height == 0 || (tx.channels != 1 && tx.channels != 4)) return fail(c, VPT_ERR_INVALID_ARGUMENT, "bad texture");
            texel_bytes += (uint64_t)tx.width * tx.height * tx.channels + 3;
        }
        if (texel_bytes > 0xffffffffull) return fail(c, VPT_ERR_LIMIT, "texel pool over 4 GiB (TexDesc offsets are 32-bit)");
    }
    free_scene(c);
    reset_accum(c);
    destroy_lanes(c);   // (their spill regions are sized by this scene's grids)
    c->state_gen++;
    // ---- geometry pools
    std::vector<vpt_vertex> verts; std::vector<uint32_t> idx;
    c->meshes.clear(); c->total_vertices = 0; c->total_indices = 0;
    for (uint32_t m = 0; m < sd->mesh_count; m++) {
        const vpt_mesh& me = sd->meshes[m];
        MeshDesc d; d.vertex_offset = (uint32_t)verts.size(); d.index_offset = (uint32_t)idx.size(); d.tri_count = me.index_count / 3; d.pad = 0;
        verts.insert(verts.end(), me.vertices, me.vertices + me.vertex_count);
        idx.insert(idx.end(), me.indices, me.indices + me.index_count);
        c->meshes.push_back(d);
        c->total_vertices += me.vertex_count; c->total_indices += me.index_count;
    }
    c->materials.assign(sd->materials, sd->materials + sd->material_count);
    c->texture_count = sd->texture_count;
    // ---- instances, flattened world-space triangles (instance-major global ids)
    c->instances.clear();
    std::vector<BvhTri> tris;
    uint32_t total_tris = 0;
    for (uint32_t i = 0; i < sd->instance_count; i++) {
        const vpt_instance& in = sd->instances[i];
        InstanceDesc d; memset(&d, 0, sizeof(d));
        d.mesh = in.mesh_index; d.material = in.material_index; d.tri_offset = total_tris;
        memcpy(d.xform, in.transform, 64);
        vptfp::inverse3x3_from_mat4(in.transform, d.inv3);
        c->instances.push_back(d);
        const MeshDesc& me = c->meshes[d.mesh];
        for (uint32_t t = 0; t < me.tri_count; t++) {
            const uint32_t* ii = &idx[me.index_offset + t * 3];
            vptfp::V3 p[3];
            for (int k = 0; k < 3; k++) {
                const vpt_vertex& v = verts[me.vertex_offset + ii[k]];
                p[k] = vptfp::mat_point(d.xform, vptfp::v3(v.position[0], v.position[1], v.position[2]));
            }
            vptfp::V3 e1 = p[1] - p[0], e2 = p[2] - p[0];
            BvhTri bt;
            bt.v0[0] = p[0].x; bt.v0[1] = p[0].y; bt.v0[2] = p[0].z;
            bt.e1[0] = e1.x; bt.e1[1] = e1.y; bt.e1[2] = e1.z;
            bt.e2[0] = e2.x; bt.e2[1] = e2.y; bt.e2[2] = e2.z;
            bt.prim = t; bt.inst = i; bt.gid = total_tris++;  // instance-major id over ALL triangles (tie-break key)
            if (!vptfp::triangle_degenerate(e1, e2)) tris.push_back(bt);  // slivers are not intersectable (vpt_fp32.h)
        }
    }
    std::vector<BvhNode> nodes; std::vector<BvhNodeWide> wide; std::vector<BvhTri> leaf_tris; int depth = 0;
    c->sbvh = (c->cfg.build_flags & VPT_BUILD_SBVH) != 0u;   // spatial splits in the builder: a per-context option
    const auto t_bvh0 = std::chrono::steady_clock::now();
    build_bvh(tris, nodes, wide, leaf_tris, &depth, nullptr, c->sbvh);
    c->bvh_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_bvh0).count();
    c->bvh_input = tris; c->dsc.nodes8 = nullptr; c->dsc.nodes4s = nullptr;
    c->bvh_depth = (uint32_t)depth;
    // ---- textures
    c->tex_1x1.clear();
    std::vector<TexDesc> tds; std::vector<uint8_t> texels;
    for (uint32_t t = 0; t < sd->texture_count; t++) {
        const vpt_texture& tx = sd->textures[t];
        while (texels.size() % 4) texels.push_back(0);
        TexDesc d; d.offset = (uint32_t)texels.size(); d.w = tx.width; d.h = tx.height; d.c = tx.channels;
        c->tex_1x1.push_back(tx.width == 1 && tx.height == 1 ? 1 : 0);
        texels.insert(texels.end(), tx.data, tx.data + (size_t)tx.width * tx.height * tx.channels);
        tds.push_back(d);
    }
    // TexDesc.offset and the shade stage's texel addresses are 32-bit byte offsets into the pool
    if (texels.size() > 0xffffffffull) return fail(c, VPT_ERR_LIMIT, "the scene's textures exceed the 4 GiB texel pool");
    // ---- environment + tables
    std::vector<float> env; std::vector<AliasEntry> alias;
    build_env_tables(sd->env_rgba, sd->env_width, sd->env_height, env, alias);
    std::vector<float> lr(sd->lut_reflection, sd->lut_reflection + 64 * 64 * 32);
    std::vector<float> lo(sd->lut_refraction_outside, sd->lut_refraction_outside + 128 * 128 * 32);
    std::vector<float> li(sd->lut_refraction_inside, sd->lut_refraction_inside + 128 * 128 * 32);
    build_emissive(c);
    // ---- upload
    DeviceScene& D = c->dsc;
    int rc;
    if ((rc = upload(c, nodes, &D.nodes))) return rc;
    // small scenes ride in LDS next to the traversal stacks, in the fp32 node form: up to 3 KB, so that three blocks of the fused
    // kernel (14 KB of stacks + 36 KB of regrouping ring + the scene each) still fit the 160 KB of a CU
    c->lds_scene = (nodes.size() * sizeof(BvhNodeWide) + leaf_tris.size() * sizeof(BvhTri)) <= 3072;
    D.nodes_wide = nullptr;
    if (c->lds_scene && (rc = upload(c, wide, &D.nodes_wide))) return rc;
    if ((rc = upload(c, leaf_tris, &D.tris))) return rc;
    D.node_count = (uint32_t)nodes.size(); D.tri_count = (uint32_t)leaf_tris.size();
    {
        std::vector<uint32_t> slot_of(total_tris, 0xffffffffu);  // 0xffffffff: a sliver, in no leaf
        for (size_t i = 0; i < leaf_tris.size(); i++) slot_of[leaf_tris[i].gid] = (uint32_t)i;
        if ((rc = upload(c, slot_of, &D.tri_slot_of_gid))) return rc;
    }
    if ((rc = upload(c, verts, &D.vertices))) return rc;
    if ((rc = upload(c, idx, &D.indices))) return rc;
    if ((rc = upload(c, c->meshes, &D.meshes))) return rc;
    if ((rc = upload(c, c->instances, &D.instances))) return rc;
    const vpt_material* dm = nullptr;
    if ((rc = upload(c, c->materials, &dm))) return rc;
    D.materials = dm; c->d_materials = const_cast<vpt_material*>(dm);
    if ((rc = upload(c, tds, &D.textures))) return rc;
    if ((rc = upload(c, texels, &D.texels, 4))) return rc;
    {
        void* d = nullptr;
        HIPCHK(c, hipMalloc(&d, sizeof(EmissiveDesc) * std::max<size_t>(1, c->instances.size())));
        c->scene_allocs.push_back(d);
        c->d_emissive = (EmissiveDesc*)d; D.emissive = c->d_emissive;
    }
    {
        void *d1 = nullptr, *d2 = nullptr, *d3 = nullptr, *d4 = nullptr;
        HIPCHK(c, hipMalloc(&d1, sizeof(MatResolved) * c->materials.size())); c->scene_allocs.push_back(d1);
        HIPCHK(c, hipMalloc(&d2, sizeof(EmissiveTri) * std::max<size_t>(1, total_tris))); c->scene_allocs.push_back(d2);
        HIPCHK(c, hipMalloc(&d3, 4 * std::max<size_t>(1, c->instances.size()))); c->scene_allocs.push_back(d3);
        HIPCHK(c, hipMalloc(&d4, 16 * std::max<size_t>(1, total_tris))); c->scene_allocs.push_back(d4);
        void* d6 = nullptr;
        HIPCHK(c, hipMalloc(&d6, 128 * std::max<size_t>(1, total_tris))); c->scene_allocs.push_back(d6);
        c->d_tri_shade = (float4*)d6; D.tri_shade = c->d_tri_shade;
        void* d7 = nullptr;   // one LightSampler per emissive mesh (at most one per instance)
        HIPCHK(c, hipMalloc(&d7, sizeof(LightSampler) * std::max<size_t>(1, c->instances.size()))); c->scene_allocs.push_back(d7);
        D.lights = (const LightSampler*)d7;
        void* d5 = nullptr;
        HIPCHK(c, hipMalloc(&d5, std::max<size_t>(1, c->instances.size()))); c->scene_allocs.push_back(d5);
        c->d_inst_class = (unsigned char*)d5; D.inst_class = c->d_inst_class;
        c->d_mat_resolved = (MatResolved*)d1; c->d_emissive_tri = (EmissiveTri*)d2; c->d_emissive_tri_offset = (uint32_t*)d3; c->d_tri_ng = (float4*)d4;
        D.mat_resolved = c->d_mat_resolved; D.emissive_tri = c->d_emissive_tri; D.emissive_tri_offset = c->d_emissive_tri_offset; D.tri_ng = c->d_tri_ng;
    }
    if ((rc = upload_emissive(c))) return rc;
    if ((rc = upload(c, env, &D.env))) return rc;
    if ((rc = upload(c, alias, &D.alias))) return rc;
    D.env_w = sd->env_width; D.env_h = sd->env_height;
    D.env_black = 1u;
    for (size_t i = 0; i < env.size(); i++) if (env[i] != 0.0f) { D.env_black = 0u; break; }
    if ((rc = upload(c, lr, &D.lut_r))) return rc;
    if ((rc = upload(c, lo, &D.lut_o))) return rc;
    if ((rc = upload(c, li, &D.lut_i))) return rc;
#if VPT_LAB
    c->trav_blocks = traverse_blocks_per_cu(c->lds_scene, D) * c->cu_count;
    c->shade_blocks = shade_blocks_per_cu() * c->cu_count;
#else
    c->trav_blocks = 0;
    c->shade_blocks = 4 * c->cu_count;   // (the media scatter stage's grid-stride launch)
#endif
    c->join_blocks = join_blocks_per_cu() * c->cu_count;
    c->primary_blocks_general = bounce_blocks_per_cu(c->lds_scene, D, false) * c->cu_count;
    c->primary_blocks_plain = bounce_blocks_per_cu(c->lds_scene, D, true) * c->cu_count;
    c->primary_blocks = std::max(c->primary_blocks_general, c->primary_blocks_plain);   // (sizes the spill regions below; update_depth_bounded picks the grid)
    c->whole_blocks = c->lds_scene ? std::max(whole_blocks_per_cu(D, false), whole_blocks_per_cu(D, true)) * c->cu_count : 0;
    c->shade_stream_blocks = shade_stream_blocks_per_cu() * c->cu_count;
    c->finish_blocks = finish_blocks_per_cu(D) * c->cu_count;
    c->shade_media_blocks = shade_media_blocks_per_cu() * c->cu_count;
    c->media_tail_blocks = media_tail_blocks_per_cu() * c->cu_count;
    c->shadow_blocks = trace_shadow_blocks_per_cu() * c->cu_count;
    c->vote_blocks = std::min(trace_blocks_per_cu(VPT_TRACE_VOTE, false), trace_blocks_per_cu(VPT_TRACE_VOTE, true)) * c->cu_count;
    {   // per-thread overflow region of the traversal stacks, for the largest persistent grid launched
        c->max_blocks = std::max(std::max(std::max(std::max(c->trav_blocks, c->shade_blocks), std::max(c->primary_blocks, c->whole_blocks)), c->vote_blocks), std::max(std::max(c->shade_stream_blocks, c->finish_blocks), c->shadow_blocks));
        void* d = nullptr;
        // two regions: the shadow kernels of bounce k run on the second stream beside the extend kernel of bounce k + 1, and a
        // spill slot is addressed by (block, thread) alone, so concurrent grids must not share one region (round 2 did)
        const size_t region = stack_overflow_bytes((uint32_t)c->max_blocks);
        HIPCHK(c, hipMalloc(&d, 2 * region));
        c->scene_allocs.push_back(d);
        D.stack_overflow = (uint32_t*)d;
        c->stack_overflow2 = (uint32_t*)((char*)d + region);
        // preset to a word no stack entry can be (a node index of 2.1e9; leaf codes are negative): vpt_get_stats counts what was spilled
        HIPCHK(c, hipMemset(d, kSpillPatternByte, 2 * region));
        c->stack_overflow_words = (uint32_t)(region / 4);
        c->spill_dirty = true;
    }
    launch_precompute_tri_ng(c->stream, D, c->d_tri_ng);
    launch_precompute_tri_shade(c->stream, D, c->d_tri_shade);
    launch_precompute_materials(c->stream, D, c->params.flags, c->d_mat_resolved, (uint32_t)c->materials.size());
    launch_classify_instances(c->stream, D, c->d_inst_class, (uint32_t)c->instances.size());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if ((rc = update_class_present(c))) return rc;
    HIPCHK(c, hipGetLastError());
    c->has_scene = true;
    update_depth_bounded(c);
    // (after a failed vpt_resize there is no image to clear: the scene is installed all the same, rendering needs a successful resize first)
    if (c->buffers_ok) HIPCHK(c, hipMemset(c->image, 0, (size_t)c->P.shard_pixels * 16));
    c->set_scene_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_scene0).count();
    return check_stream_slack(c);
}

int vpt_set_material(vpt_ctx* c, uint32_t index, const vpt_material* m) {
    if (!c || !m) return VPT_ERR_INVALID_ARGUMENT;
    if (!c->has_scene) return fail(c, VPT_ERR_NO_SCENE, "no scene");
    if (index >= c->materials.size()) return fail(c, VPT_ERR_INVALID_ARGUMENT, "material index out of range");
    if (m->base_color_texture >= c->texture_count || m->normal_texture >= c->texture_count || m->roughness_texture >= c->texture_count ||
        m->metallic_texture >= c->texture_count || m->emissive_texture >= c->texture_count)
        return fail(c, VPT_ERR_INVALID_ARGUMENT, "material texture index out of range");   // the shade stage indexes textures[] unchecked
    HIPCHK(c, hipSetDevice(c->cfg.device));
    { int rd = drain(c); if (rd) return rd; }   // batches in flight read the tables patched below
    c->state_gen++;
    const vpt_material& old = c->materials[index];
    bool emissive_changed = old.emissive_color[0] != m->emissive_color[0] || old.emissive_color[1] != m->emissive_color[1] || old.emissive_color[2] != m->emissive_color[2];
    c->materials[index] = *m;
    HIPCHK(c, hipMemcpy(c->d_materials + index, m, sizeof(vpt_material), hipMemcpyHostToDevice));
    if (emissive_changed) { build_emissive(c); int rc = upload_emissive(c); if (rc) return rc; }
    launch_precompute_materials(c->stream, c->dsc, c->params.flags, c->d_mat_resolved, (uint32_t)c->materials.size());
    launch_classify_instances(c->stream, c->dsc, c->d_inst_class, (uint32_t)c->instances.size());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    { int rc2 = update_class_present(c); if (rc2) return rc2; }
    update_depth_bounded(c);
    reset_accum(c);
    return VPT_OK;
}
int vpt_get_material(const vpt_ctx* c, uint32_t index, vpt_material* out) {
    if (!c || !out || index >= c->materials.size()) return VPT_ERR_INVALID_ARGUMENT;
    *out = c->materials[index];
    return VPT_OK;
}

int vpt_set_camera(vpt_ctx* c, const float* vi, const float* pi) {
    if (!c || !vi || !pi) return VPT_ERR_INVALID_ARGUMENT;
    memcpy(c->P.view_inv, vi, 64); memcpy(c->P.proj_inv, pi, 64);   // host state only: batches already enqueued carry their own copy
    c->state_gen++;
    reset_accum(c);
    return VPT_OK;
}

int vpt_set_params(vpt_ctx* c, const vpt_params* p) {
    if (!c || !p) return VPT_ERR_INVALID_ARGUMENT;
    if (p->samples_per_frame == 0 || p->samples_per_frame > 0xffffffu) return fail(c, VPT_ERR_INVALID_ARGUMENT, "samples_per_frame must be >= 1");
    // MAX_DEPTH (Defines.slang:16) marks a finished path; a larger MaxDepth would make the reference loop forever on a miss
    if (p->max_depth == 0 || p->max_depth > 1000000u) return fail(c, VPT_ERR_INVALID_ARGUMENT, "max_depth must be in [1, 1000000]");
    if (p->screen_chunk_count == 0 || p->screen_chunk_count > 64) return fail(c, VPT_ERR_INVALID_ARGUMENT, "screen_chunk_count must be in [1, 64]");
    if (p->screen_chunk_count != 1 && c->P.shard_count != 1) return fail(c, VPT_ERR_UNSUPPORTED, "split-screen dispatch needs the whole image in one context (shard_count == 1): its first dispatch copies pixels across rows");
    {   // SetMaxSamplesAccumulated alone keeps the accumulated image (PathTracer.cpp:1003-1006 does not reset)
        vpt_params same = *p; same.max_samples = c->params.max_samples;
        if (p->max_samples != c->params.max_samples && memcmp(&same, &c->params, sizeof(vpt_params)) == 0) { c->params.max_samples = p->max_samples; return VPT_OK; }
    }
    { int rd = drain(c); if (rd) return rd; }
    c->state_gen++;
    const bool flags_changed = c->params.flags != p->flags;
    c->dsc.strict_hits = (p->flags & VPT_FLAG_LOCAL_HITS) ? 1u : 0u;
    c->params = *p;
    sync_params(c);
    reset_accum(c);
    if (flags_changed && c->has_scene) {  // FURNACE_TEST_MODE is baked into the resolved-material table
        HIPCHK(c, hipSetDevice(c->cfg.device));
        launch_precompute_materials(c->stream, c->dsc, c->params.flags, c->d_mat_resolved, (uint32_t)c->materials.size());
        launch_classify_instances(c->stream, c->dsc, c->d_inst_class, (uint32_t)c->instances.size());
        HIPCHK(c, hipStreamSynchronize(c->stream));
        int rc2 = update_class_present(c); if (rc2) return rc2;
    }
    return VPT_OK;
}

int vpt_set_volumes(vpt_ctx* c, const vpt_volume* v, uint32_t count) {
    if (!c || (count && !v)) return VPT_ERR_INVALID_ARGUMENT;
    if (count > VPT_MAX_VOLUMES) return fail(c, VPT_ERR_LIMIT, "more than VPT_MAX_VOLUMES volumes");
    if (count && (c->cfg.pipeline > VPT_PIPELINE_STAGED || (c->cfg.pipeline == VPT_PIPELINE_STAGED && c->has_scene && c->lds_scene)))
        return fail(c, VPT_ERR_UNSUPPORTED, "volumes run on the fused pipeline or, for a scene whose BVH lives in memory, on the streams (VPT_PIPELINE_AUTO, _FUSED, _STAGED)");
    for (uint32_t i = 0; i < count; i++) {
        if (v[i].density_data_index < -1 || v[i].density_data_index >= (int)c->grids.size())
            return fail(c, VPT_ERR_INVALID_ARGUMENT, "density_data_index must be -1 or an index returned by vpt_add_density_grid");
        if (v[i].has_temperature_data && v[i].density_data_index < 0) return fail(c, VPT_ERR_INVALID_ARGUMENT, "has_temperature_data needs a density grid");
        if (!(v[i].density > 0.0f)) return fail(c, VPT_ERR_INVALID_ARGUMENT, "volume density must be > 0");  // -log(u)/0 (Sampler.slang:427)
    }
    HIPCHK(c, hipSetDevice(c->cfg.device));
    { int rd = drain(c); if (rd) return rd; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->state_gen++;
    if (c->d_volumes) { (void)hipFree(c->d_volumes); c->d_volumes = nullptr; }
    c->volumes.assign(v, v + count);
    if (count) {
        HIPCHK(c, hipMalloc((void**)&c->d_volumes, (size_t)count * sizeof(vpt_volume)));
        HIPCHK(c, hipMemcpy(c->d_volumes, v, (size_t)count * sizeof(vpt_volume), hipMemcpyHostToDevice));
    }
    c->dsc.volumes = c->d_volumes; c->dsc.volume_count = count; c->dsc.phase = c->phase;
    c->dsc.hetero = 0u;
    for (uint32_t i = 0; i < count; i++) if (v[i].density_data_index >= 0) c->dsc.hetero = 1u;
    reset_accum(c);
    return VPT_OK;
}
// AddDensityDataToVolume, PathTracer.cpp:1390-1442, on a dense grid
int vpt_add_density_grid(vpt_ctx* c, uint32_t dx, uint32_t dy, uint32_t dz, const float* d) {
    if (!c || !d || dx == 0 || dy == 0 || dz == 0 || (uint64_t)dx * dy * dz > (1ull << 31)) return VPT_ERR_INVALID_ARGUMENT;
    if (c->grids.size() >= VPT_MAX_DENSITY_GRIDS) return fail(c, VPT_ERR_LIMIT, "more than VPT_MAX_DENSITY_GRIDS density grids");
    const size_t n = (size_t)dx * dy * dz;
    float mx = 0.0f;
    for (size_t i = 0; i < n; i++) mx = std::max(mx, d[i]);
    if (!(mx > 0.0f)) return fail(c, VPT_ERR_INVALID_ARGUMENT, "density grid has no positive value");
    std::vector<float> block_max(32768, 0.0f);
    for (uint32_t z = 0; z < dz; z++)
        for (uint32_t y = 0; y < dy; y++)
            for (uint32_t x = 0; x < dx; x++) {
                const float raw = d[(size_t)x + (size_t)(dy - 1 - y) * dx + (size_t)z * dx * dy];  // "Y has to be flipped for vulkan" (:1435)
                const float dens = vptfp::clamp_(raw / mx, 0.0f, 1.0f);
                const uint32_t bi = ((x * 32u) / dx) + ((y * 32u) / dy) * 32u + ((z * 32u) / dz) * 1024u;
                if (block_max[bi] < dens) block_max[bi] = dens;
            }
    HIPCHK(c, hipSetDevice(c->cfg.device));
    { int rd = drain(c); if (rd) return rd; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->state_gen++;
    DensityGrid g{};
    float *dv = nullptr, *db = nullptr;
    HIPCHK(c, hipMalloc((void**)&dv, n * 4));
    if (hipMalloc((void**)&db, 32768 * 4) != hipSuccess) { (void)hipFree(dv); return fail(c, VPT_ERR_OUT_OF_MEMORY, "hipMalloc block maxima"); }
    HIPCHK(c, hipMemcpy(dv, d, n * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(db, block_max.data(), 32768 * 4, hipMemcpyHostToDevice));
    g.values = dv; g.block_max = db; g.dim[0] = dx; g.dim[1] = dy; g.dim[2] = dz; g.max_density = mx;
    c->grids.push_back(g);
    if (c->d_grids) (void)hipFree(c->d_grids);
    HIPCHK(c, hipMalloc((void**)&c->d_grids, c->grids.size() * sizeof(DensityGrid)));
    HIPCHK(c, hipMemcpy(c->d_grids, c->grids.data(), c->grids.size() * sizeof(DensityGrid), hipMemcpyHostToDevice));
    c->dsc.grids = c->d_grids;
    return (int)c->grids.size() - 1;
}
int vpt_clear_density_grids(vpt_ctx* c) {
    if (!c) return VPT_ERR_INVALID_ARGUMENT;
    for (const vpt_volume& v : c->volumes) if (v.density_data_index >= 0) return fail(c, VPT_ERR_INVALID_ARGUMENT, "a volume still references a density grid");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    { int rd = drain(c); if (rd) return rd; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->state_gen++;
    for (DensityGrid& g : c->grids) { (void)hipFree((void*)g.values); (void)hipFree((void*)g.block_max); }
    c->grids.clear();
    if (c->d_grids) { (void)hipFree(c->d_grids); c->d_grids = nullptr; }
    c->dsc.grids = nullptr;
    return VPT_OK;
}
void vpt_default_atmosphere(vpt_atmosphere* a) {  // PathTracer.h:222-232
    if (!a) return;
    a->planet_position[0] = 0.0f; a->planet_position[1] = 6360e3f + 1000.0f; a->planet_position[2] = 0.0f;
    a->planet_radius = 6360e3f; a->atmosphere_height = 100e3f;
    a->rayleigh_density_falloff = 8000.0f; a->mie_density_falloff = 1200.0f; a->ozone_density_falloff = 5000.0f; a->ozone_peak = 22000.0f;
    for (int k = 0; k < 3; k++) { a->rayleigh_multiplier[k] = 1.0f; a->mie_multiplier[k] = 1.0f; a->ozone_multiplier[k] = 1.0f; }
    a->sun_color[0] = 1.0f; a->sun_color[1] = 0.956f; a->sun_color[2] = 0.88f;
}
int vpt_set_atmosphere(vpt_ctx* c, const vpt_atmosphere* a) {
    if (!c) return VPT_ERR_INVALID_ARGUMENT;
    if (a && (c->cfg.pipeline > VPT_PIPELINE_STAGED || (c->cfg.pipeline == VPT_PIPELINE_STAGED && c->has_scene && c->lds_scene)))
        return fail(c, VPT_ERR_UNSUPPORTED, "the atmosphere runs on the fused pipeline or, for a scene whose BVH lives in memory, on the streams (VPT_PIPELINE_AUTO, _FUSED, _STAGED)");
    if (a && (!(a->planet_radius > 0.0f) || !(a->atmosphere_height > 0.0f) || !(a->rayleigh_density_falloff > 0.0f) || !(a->mie_density_falloff > 0.0f) ||
              !(a->ozone_density_falloff > 0.0f)))
        return fail(c, VPT_ERR_INVALID_ARGUMENT, "planet radius, atmosphere height and the density falloffs must be > 0");
    { int rd = drain(c); if (rd) return rd; }
    c->state_gen++;
    c->dsc.atm_on = a ? 1u : 0u;
    if (a) c->dsc.atm = *a;
    reset_accum(c);
    return VPT_OK;
}
int vpt_set_phase_function(vpt_ctx* c, uint32_t phase) {
    if (!c) return VPT_ERR_INVALID_ARGUMENT;
    if (phase > VPT_PHASE_HENYEY_GREENSTEIN_PLUS_DRAINE) return fail(c, VPT_ERR_INVALID_ARGUMENT, "unknown phase function");
    { int rd = drain(c); if (rd) return rd; }
    c->state_gen++;
    c->phase = phase; c->dsc.phase = phase;
    reset_accum(c);
    return VPT_OK;
}

int vpt_resize(vpt_ctx* c, uint32_t w, uint32_t h) {
    if (!c || w == 0 || h == 0) return VPT_ERR_INVALID_ARGUMENT;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    uint32_t F = 1;
    int rc = check_render_size(c, w, h, &F);   // nothing is freed or changed for a size this context cannot hold
    if (rc != VPT_OK) return rc;
    if ((rc = drain(c))) return rc;
    destroy_lanes(c);   // (they hold path buffers of the old size)
    c->state_gen++;
    c->cfg.width = w; c->cfg.height = h;
    reset_accum(c);
    return alloc_render_buffers(c);
}

int vpt_reset(vpt_ctx* c) { if (!c) return VPT_ERR_INVALID_ARGUMENT; reset_accum(c); return VPT_OK; }

// The next batch of a render call: how many dispatches it takes (PathTrace's accounting), with the path buffers grown to hold them.
// *nf == 0: max_samples reached (PathTrace returns true and launches nothing).
int next_batch(vpt_ctx* c, uint32_t left, uint32_t* nf) {
    *nf = 0;
    if (c->samples_accum >= c->params.max_samples) return VPT_OK;   // PathTracer.cpp:124-125
    // dispatches until PathTrace would return true: samples = floor(dispatches / S^2) * spp (PathTracer.cpp:151-153)
    const uint64_t S2 = (uint64_t)c->params.screen_chunk_count * c->params.screen_chunk_count;
    const uint64_t frames_needed = ((uint64_t)c->params.max_samples + c->params.samples_per_frame - 1) / c->params.samples_per_frame;
    const uint64_t disp_left = frames_needed * S2 - c->dispatch_count;
    uint32_t n = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(left, batch_cap(c)), disp_left);
    if (n > c->frames_alloc || resident_frames_for(c, n) > c->resident_alloc || !path_words_ok(c)) {   // the buffers grow to the largest batch asked for (and to the words it touches); nothing may be in flight while they are replaced
        int rc = drain(c);
        if (rc) return rc;
        if ((rc = ensure_path_buffers(c, n))) return rc;
        n = std::min(n, c->frames_alloc);   // (a size the library chose itself may have been halved)
    }
    if (!regen_allowed(c) && !whole_without_records(c, n)) n = std::min(n, c->resident_alloc);
    if (media_on_streams(c)) {   // media on the streams: the batch is what the media streams hold
        int rm = ensure_media_buffers(c);
        if (rm) return rm;
        n = std::min(n, c->media_frames);
    }
    *nf = n;
    return VPT_OK;
}
void advance_counts(vpt_ctx* c, uint32_t nf) {
    const uint64_t S2 = (uint64_t)c->params.screen_chunk_count * c->params.screen_chunk_count;
    c->dispatch_count += nf;
    c->frame_count = (uint32_t)(c->dispatch_count / S2);
    c->samples_accum = c->frame_count * c->params.samples_per_frame;
    c->full_valid = false;
}

int vpt_render(vpt_ctx* c, uint32_t dispatches, int* done) {
    if (!c) return VPT_ERR_INVALID_ARGUMENT;
    if (!c->has_scene) return fail(c, VPT_ERR_NO_SCENE, "vpt_render before vpt_set_scene");
    if (!c->buffers_ok) return fail(c, VPT_ERR_DEVICE, "no render buffers: the last vpt_resize failed");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    { int rd = drain(c); if (rd) return rd; }
    if (done) *done = 0;
    uint32_t left = dispatches;
    while (left > 0) {
        uint32_t nf = 0;
        int rc = next_batch(c, left, &nf);
        if (rc) return rc;
        if (nf == 0) { if (done) *done = 1; break; }
        rc = render_batch(c, nf, (uint32_t)c->dispatch_count);  // returns with the stream drained
        if (rc) return rc;
        advance_counts(c, nf);
        left -= nf;
    }
    return VPT_OK;
}

// PathTrace(cmd) as the reference has it: recorded, not waited for (include/vpt.h).
int vpt_render_async(vpt_ctx* c, uint32_t dispatches, int* done, uint64_t* ticket) {
    if (!c) return VPT_ERR_INVALID_ARGUMENT;
    if (!c->has_scene) return fail(c, VPT_ERR_NO_SCENE, "vpt_render_async before vpt_set_scene");
    if (!c->buffers_ok) return fail(c, VPT_ERR_DEVICE, "no render buffers: the last vpt_resize failed");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (done) *done = 0;
    uint32_t left = dispatches;
    while (left > 0) {
        int rc = finish_outstanding(c);   // the path buffers are single: an unfinished batch goes first
        if (rc) return rc;
        uint32_t nf = 0;
        if ((rc = next_batch(c, left, &nf))) return rc;
        if (nf == 0) { if (done) *done = 1; break; }
        const uint64_t bounds = (uint64_t)c->P.max_depth * c->P.samples_per_frame;
        const bool vol = !c->volumes.empty() || c->dsc.atm_on;
        // a fixed schedule: every path has ended after `bounds` bounces, whatever the random numbers say
        // ... or the batch is ONE launch that runs every path to its end (k_whole), whatever max_depth is
        const bool whole = whole_applies(c, nf);
        const bool fused_auto = (whole || c->cfg.pipeline == VPT_PIPELINE_FUSED || (c->cfg.pipeline == VPT_PIPELINE_AUTO && c->lds_scene)) && !vol;
        const bool streams_pipe = !vol && !fused_auto && (c->cfg.pipeline == VPT_PIPELINE_AUTO || c->cfg.pipeline == VPT_PIPELINE_STAGED || c->cfg.pipeline == VPT_PIPELINE_STAGED_SORTED);
        // a small batch of the streams pipeline ends in ONE launch that runs every path to its end (k_finish): a fixed schedule whatever max_depth is
        const bool stream_finish = streams_pipe && !(c->cfg.build_flags & VPT_BUILD_STREAMS_ONLY) && nf <= c->resident_alloc && (uint64_t)nf * c->P.shard_pixels <= kFinishSmallBatchPaths;
        const bool fixed = whole || stream_finish || (c->depth_bounded && !vol && bounds <= VPT_ASYNC_MAX_BOUNCES && nf <= c->resident_alloc);   // (a regenerating batch has no fixed length)
        const uint32_t enq = stream_finish ? kFinishAfterBounces + 1u : (uint32_t)std::min<uint64_t>(bounds, VPT_ASYNC_MAX_BOUNCES);
        const uint32_t base = (uint32_t)c->dispatch_count;
        const bool plain_launches = c->cfg.profile || c->cfg.count_traversal || c->P.split != 1u;
        // the streams pipeline's fixed batch (a frame per call on a scene whose BVH lives in memory: seven launches per bounce, every one of them short)
        // goes over the lanes and is replayed from a captured graph too — each lane's batch on ONE stream
        const bool stream_fixed = fixed && streams_pipe && c->cfg.pipeline != VPT_PIPELINE_STAGED_SORTED;
        if (c->graph_streak_gen == c->state_gen) c->graph_streak++; else { c->graph_streak = 0; c->graph_streak_gen = c->state_gen; }
        // the fused pipeline's fixed 1-frame batch goes to the next lane (vpt_ctx::lanes); asked for again with nothing changed since the
        // last two calls it is replayed from the lane's captured graph
        vpt_ctx* X = c;
        if (fixed && (fused_auto || stream_fixed) && !plain_launches && nf == 1u) {
            const uint32_t max_lanes = std::max(1u, std::min(c->lab_lanes, 3u));
            vpt_ctx* idle = nullptr;
            uint32_t have = 1;
            if (hipEventQuery(c->ev_resolved) == hipSuccess) idle = c;
            for (uint32_t k = 0; k + 1 < max_lanes; k++) {
                vpt_ctx* L = c->lanes[k];
                if (!L) { if (!idle) { L = get_lane(c, (int)k); if (L) idle = L; } break; }   // every existing lane is busy: one more
                have++;
                if (!idle && hipEventQuery(L->ev_resolved) == hipSuccess) idle = L;
            }
            (void)hipGetLastError();   // (hipErrorNotReady is not an error)
            if (!idle) {   // all lanes busy: round robin
                const uint32_t k = c->lane_rr++ % have;
                idle = k == 0u ? c : c->lanes[k - 1];
            }
            X = idle;
            if (X != c) { const int rc_lane = sync_lane(c, X); if (rc_lane != VPT_OK) return rc_lane; }
        }
        // a batch on the main lane behind pipelined frames: their resolves come first (frame order), and the records it overwrites are the main lane's own
        if (X == c && c->order_lane && c->order_lane != c) HIPCHK(c, hipStreamWaitEvent(c->stream, c->order_lane->ev_resolved, 0));
        BatchState b;
        bool graphed = false;
        const bool pipelined = fixed && fused_auto && !plain_launches && nf == 1u && c->graph_streak >= 2u;
        // Frames in steady accumulation share the chip: each lane's kernels take a third of the persistent grid (one block per CU of the
        // three the fused kernel's LDS allows), so that the three lanes' chains are co-resident and the tail of one frame — launches that are
        // bounded by one bounce's latency, not by throughput — runs beside the first bounces of the next two.  (A full-size grid fills every
        // CU's LDS and keeps the other lanes' blocks out until it retires.)
        const int full_grid = X->primary_blocks;
        auto part = [&](uint32_t div) { return std::max(c->cu_count, (c->primary_blocks / (int)std::max(1u, div) / std::max(c->cu_count, 1)) * c->cu_count); };
        if (pipelined) {
            X->primary_blocks = part(std::max(1u, c->lab_lane_grid));
            X->tail_blocks = c->lab_tail_grid > 1u ? std::min(X->primary_blocks, part(c->lab_tail_grid)) : 0;
        }
        if (fixed && (fused_auto || stream_fixed) && !plain_launches && c->graph_streak >= 2u) {
            rc = enqueue_graph(X, nf, base, enq, &graphed, b);
            if (rc) { X->primary_blocks = full_grid; X->tail_blocks = 0; if (X != c) c->err = X->err; return rc; }
        }
        if (!graphed) {
            rc = enqueue_fixed(X, nf, base, enq, b);
            if (rc) { X->primary_blocks = full_grid; X->tail_blocks = 0; if (X != c) c->err = X->err; return rc; }
        }
        X->primary_blocks = full_grid; X->tail_blocks = 0;
        if (fixed && b.n_slots) {   // frames resolve in order: this one's resolve waits for the previous frame's, whichever lane that ran on
            if (c->order_lane && c->order_lane != X) HIPCHK(c, hipStreamWaitEvent(X->stream, c->order_lane->ev_resolved, 0));
            if (c->post_pending && X != c) HIPCHK(c, hipStreamWaitEvent(X->stream, c->ev_post, 0));   // ... and for the post-process that is still reading the image
        }
        if ((rc = batch_resolve(X, b))) { if (X != c) c->err = X->err; return rc; }
        if (fixed && b.n_slots) {
            HIPCHK(c, hipEventRecord(X->ev_resolved, X->stream));
            c->order_lane = X;
            X->last_fixed = b; X->last_fixed_valid = true;
        }
        c->stats.samples += (uint64_t)b.n_slots * c->P.samples_per_frame;
        advance_counts(c, nf);
        left -= nf;
        const uint64_t t = issue_ticket(c, X->stream);
        if (!fixed && b.n_slots) { c->out_active = true; c->out_batch = b; c->out_ticket = t; }
    }
    if (ticket) *ticket = c->tick_issued;
    return VPT_OK;
}

int vpt_wait(vpt_ctx* c, uint64_t ticket) {
    if (!c) return VPT_ERR_INVALID_ARGUMENT;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (ticket == 0 || ticket >= c->tick_issued) return drain(c);
    if (c->out_active && c->out_ticket <= ticket) { int rc = finish_outstanding(c); if (rc) return rc; }
    // the events are reused round-robin and recorded in stream order: the latest record of ticket's event belongs to a ticket >= it
    HIPCHK(c, hipEventSynchronize(c->tick_ev[ticket % kTickets]));
    return VPT_OK;
}

int vpt_get_radiance_device(vpt_ctx* c, void* dst) {
    if (!c || !dst) return VPT_ERR_INVALID_ARGUMENT;
    if (!c->buffers_ok) return fail(c, VPT_ERR_DEVICE, "no render buffers: the last vpt_resize failed");
    if (c->P.shard_count > 1 && !c->full_valid) return fail(c, VPT_ERR_INVALID_ARGUMENT, "sharded context: call vpt_assemble_shards first");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    { int rd = drain(c); if (rd) return rd; }
    HIPCHK(c, hipMemcpyAsync(dst, whole_image(c), (size_t)c->P.width * c->P.height * 16, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VPT_OK;
}
int vpt_get_radiance(vpt_ctx* c, float* dst) {
    if (!c || !dst) return VPT_ERR_INVALID_ARGUMENT;
    if (!c->buffers_ok) return fail(c, VPT_ERR_DEVICE, "no render buffers: the last vpt_resize failed");
    if (c->P.shard_count > 1 && !c->full_valid) return fail(c, VPT_ERR_INVALID_ARGUMENT, "sharded context: call vpt_assemble_shards first");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    { int rd = drain(c); if (rd) return rd; }
    HIPCHK(c, hipMemcpy(dst, whole_image(c), (size_t)c->P.width * c->P.height * 16, hipMemcpyDeviceToHost));
    return VPT_OK;
}
int vpt_set_radiance(vpt_ctx* c, const float* src, uint32_t frame_count) {
    if (!c || !src) return VPT_ERR_INVALID_ARGUMENT;
    if (!c->buffers_ok) return fail(c, VPT_ERR_DEVICE, "no render buffers: the last vpt_resize failed");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    { int rd = drain(c); if (rd) return rd; }
    const uint32_t W = c->P.width;
    if (c->P.shard_count == 1) {
        HIPCHK(c, hipMemcpy(c->image, src, (size_t)W * c->P.height * 16, hipMemcpyHostToDevice));
    } else {
        HIPCHK(c, hipMemcpy(c->full_image, src, (size_t)W * c->P.height * 16, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy2D(c->image, (size_t)W * 16, src + (size_t)c->P.shard_rank * W * 4, (size_t)W * 16 * c->P.shard_count, (size_t)W * 16,
                              c->P.shard_rows, hipMemcpyHostToDevice));
        c->full_valid = true;
    }
    c->frame_count = frame_count;
    c->dispatch_count = (uint64_t)frame_count * c->params.screen_chunk_count * c->params.screen_chunk_count;
    c->samples_accum = frame_count * c->params.samples_per_frame;
    return VPT_OK;
}

size_t vpt_shard_floats(const vpt_ctx* c) {
    if (!c) return 0;
    uint32_t max_rows = shard_rows_of(c->P.height, 0, c->P.shard_count);
    return (size_t)max_rows * c->P.width * 4;
}
int vpt_get_shard_device(vpt_ctx* c, void* dst) {
    if (!c || !dst) return VPT_ERR_INVALID_ARGUMENT;
    if (!c->buffers_ok) return fail(c, VPT_ERR_DEVICE, "no render buffers: the last vpt_resize failed");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    { int rd = drain(c); if (rd) return rd; }
    size_t bytes = (size_t)c->P.shard_pixels * 16, padded = vpt_shard_floats(c) * 4;
    HIPCHK(c, hipMemcpyAsync(dst, c->image, bytes, hipMemcpyDeviceToDevice, c->stream));
    if (padded > bytes) HIPCHK(c, hipMemsetAsync((char*)dst + bytes, 0, padded - bytes, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VPT_OK;
}
int vpt_assemble_shards(vpt_ctx* c, const void* gathered, uint32_t shard_count) {
    if (!c || !gathered) return VPT_ERR_INVALID_ARGUMENT;
    if (shard_count != c->P.shard_count) return fail(c, VPT_ERR_INVALID_ARGUMENT, "shard_count mismatch");
    if (!c->buffers_ok) return fail(c, VPT_ERR_DEVICE, "no render buffers: the last vpt_resize failed");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    { int rd = drain(c); if (rd) return rd; }
    float* dst = c->P.shard_count > 1 ? c->full_image : c->image;
    launch_scatter_rows(c->stream, (const float*)gathered, dst, c->P.width, c->P.height, shard_count, (uint32_t)(vpt_shard_floats(c) / 4));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    c->full_valid = true;
    return VPT_OK;
}

// PostProcessor::PostProcess, PostProcessor.cpp:193-246: the launches, on the context's stream, nothing waited for.
namespace {
int enqueue_post(vpt_ctx* c, const vpt_post_params* pp, bool bloom0) {
    int rc = ensure_post_buffers(c);
    if (rc) return rc;
    hipStream_t s = c->stream;
    const float* hdr = whole_image(c);
    const uint32_t W = c->P.width, H = c->P.height;
    uint32_t mip_count = std::max(1u, std::min(pp->mip_count, (uint32_t)c->mips.size()));
    const bool linear_tap = (c->params.flags & VPT_FLAG_TONEMAP_LINEAR_BLOOM_TAP) != 0;
    auto MW = [&](uint32_t i) { return c->mip_sizes[i].first; };
    auto MH = [&](uint32_t i) { return c->mip_sizes[i].second; };
    if (pp->schedule == VPT_POST_REFERENCE_PASSES) {   // PostProcessor.cpp:193-246 pass by pass: threshold, down x (n-1), up x (n-1), tonemap
        TIMED(c, VPT_K_BLOOM, launch_bloom_threshold(s, hdr, c->mips[0], W, H, pp->bloom_threshold, pp->falloff_range));
        for (uint32_t i = 1; i < mip_count; i++)
            TIMED(c, VPT_K_BLOOM, launch_bloom_down(s, c->mips[i - 1], MW(i - 1), MH(i - 1), c->mips[i], MW(i), MH(i), pp->bloom_strength));
        for (uint32_t i = mip_count - 1; i > 0; i--)
            TIMED(c, VPT_K_BLOOM, launch_bloom_up(s, c->mips[i], MW(i), MH(i), c->mips[i - 1], MW(i - 1), MH(i - 1), pp->bloom_strength));
        TIMED(c, VPT_K_TONEMAP, launch_tonemap(s, hdr, c->mips[0], c->post_out, W, H, pp->exposure, pp->gamma, linear_tap));
    } else {
        // Fused schedule, same values (kernels_post.hip): mip 0 is never materialised unless the caller asks for it.
        //   T = first mip the one-launch tail keeps in LDS (<= kBloomTailMaxTexels texels, and >= 2: its base mip must exist in memory)
        uint32_t T = mip_count;
        for (uint32_t i = 2; i < mip_count; i++) if ((uint64_t)MW(i) * MH(i) <= kBloomTailMaxTexels) { T = i; break; }
        if (mip_count - T > kBloomTailMaxLevels) T = mip_count;   // cannot happen with <= 10 mips; the per-pass kernels cover it
        if (mip_count >= 2) TIMED(c, VPT_K_BLOOM, launch_bloom_down_first(s, hdr, W, H, c->mips[1], MW(1), MH(1), pp->bloom_strength, pp->bloom_threshold, pp->falloff_range));
        {   // down-samples between mip 1 and the tail's base: one launch each while the levels are large, the last (up to three, at most
            // kBloomDownChainTexels texels in the first of them) in one launch
            const uint32_t last = std::min(T, mip_count) - 1;   // last level produced here
            uint32_t i = 2;
            while (i <= last) {
                const uint32_t left = last - i + 1;
                if (left >= 2 && left <= kBloomDownChainMax && (uint64_t)MW(i) * MH(i) <= kBloomDownChainTexels) {
                    float* lv[kBloomDownChainMax]; uint32_t lw[kBloomDownChainMax], lh[kBloomDownChainMax];
                    for (uint32_t k = 0; k < left; k++) { lv[k] = c->mips[i + k]; lw[k] = MW(i + k); lh[k] = MH(i + k); }
                    TIMED(c, VPT_K_BLOOM, launch_bloom_down_chain(s, c->mips[i - 1], MW(i - 1), MH(i - 1), lv, lw, lh, left, pp->bloom_strength));
                    i += left;
                } else {
                    TIMED(c, VPT_K_BLOOM, launch_bloom_down(s, c->mips[i - 1], MW(i - 1), MH(i - 1), c->mips[i], MW(i), MH(i), pp->bloom_strength));
                    i++;
                }
            }
        }
        uint32_t up_from = std::min(T, mip_count) - 1;   // the highest level that is final once the tail has run
        if (T < mip_count) {
            uint32_t tw[kBloomTailMaxLevels], th[kBloomTailMaxLevels];
            for (uint32_t i = T; i < mip_count; i++) { tw[i - T] = MW(i); th[i - T] = MH(i); }
            const bool staged = bloom_tail_is_staged(MW(T - 1), MH(T - 1));
            TIMED(c, VPT_K_BLOOM, launch_bloom_tail(s, c->mips[T - 1], c->mips[T], MW(T - 1), MH(T - 1), tw, th, mip_count - T, pp->bloom_strength));
            if (staged) up_from = T;   // the staged tail leaves mip T finished in memory and mip T - 1 as the down-samples left it
        }
        // up-samples of the levels between the tail and mip 1: up to kBloomChainMax of them per launch, only the lowest level written
        // (the levels between are read by nothing else)
        for (uint32_t top = up_from; top > 1;) {
            const uint32_t n = std::min(top - 1u, kBloomChainMax), base = top - n;
            if (n == 1) {
                TIMED(c, VPT_K_BLOOM, launch_bloom_up(s, c->mips[top], MW(top), MH(top), c->mips[base], MW(base), MH(base), pp->bloom_strength));
            } else {
                float* lv[kBloomChainMax + 1]; uint32_t lw[kBloomChainMax + 1], lh[kBloomChainMax + 1];
                for (uint32_t k = 0; k <= n; k++) { lv[k] = c->mips[base + k]; lw[k] = MW(base + k); lh[k] = MH(base + k); }
                TIMED(c, VPT_K_BLOOM, launch_bloom_up_chain(s, lv, lw, lh, n, pp->bloom_strength));
            }
            top = base;
        }
        TIMED(c, VPT_K_TONEMAP, launch_post_final(s, hdr, mip_count >= 2 ? c->mips[1] : nullptr, mip_count >= 2 ? MW(1) : 0u, mip_count >= 2 ? MH(1) : 0u,
                                                  bloom0 ? c->mips[0] : nullptr, c->post_out, W, H, pp->bloom_threshold, pp->falloff_range, pp->bloom_strength,
                                                  pp->exposure, pp->gamma, linear_tap));
    }
    return VPT_OK;
}
int post_preconditions(vpt_ctx* c) {
    if (!c->buffers_ok) return fail(c, VPT_ERR_DEVICE, "no render buffers: the last vpt_resize failed");
    if (c->P.shard_count > 1 && !c->full_valid) return fail(c, VPT_ERR_INVALID_ARGUMENT, "sharded context: call vpt_assemble_shards first");
    return VPT_OK;
}
}  // namespace

int vpt_postprocess(vpt_ctx* c, const vpt_post_params* pp, uint8_t* out8, float* bloom0) {
    if (!c || !pp || !out8) return VPT_ERR_INVALID_ARGUMENT;
    int rc = post_preconditions(c);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if ((rc = drain(c))) return rc;
    if ((rc = enqueue_post(c, pp, bloom0 != nullptr))) return rc;
    const uint32_t W = c->P.width, H = c->P.height;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    collect_timing(c);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpy(out8, c->post_out, (size_t)W * H * 4, hipMemcpyDeviceToHost));
    if (bloom0) HIPCHK(c, hipMemcpy(bloom0, c->mips[0], (size_t)W * H * 16, hipMemcpyDeviceToHost));
    return VPT_OK;
}

// PostProcess(cmd) as the reference has it: recorded behind the render on the same stream, the RGBA8 image stays on the device.
int vpt_postprocess_device(vpt_ctx* c, const vpt_post_params* pp, void* rgba8_device, uint64_t* ticket) {
    if (!c || !pp) return VPT_ERR_INVALID_ARGUMENT;
    int rc = post_preconditions(c);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if ((rc = finish_outstanding(c))) return rc;   // the image must be complete: an unfinished batch is finished first
    if (c->order_lane && c->order_lane != c) HIPCHK(c, hipStreamWaitEvent(c->stream, c->order_lane->ev_resolved, 0));   // the latest frame was resolved on another lane
    if ((rc = enqueue_post(c, pp, false))) return rc;
    if (rgba8_device) HIPCHK(c, hipMemcpyAsync(rgba8_device, c->post_out, (size_t)c->P.width * c->P.height * 4, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipEventRecord(c->ev_post, c->stream));
    c->post_pending = true;
    const uint64_t t = issue_ticket(c, c->stream);
    if (ticket) *ticket = t;
    return VPT_OK;
}
const void* vpt_output_device(vpt_ctx* c) { return c ? c->post_out : nullptr; }
int vpt_get_output(vpt_ctx* c, uint8_t* out8) {
    if (!c || !out8) return VPT_ERR_INVALID_ARGUMENT;
    if (!c->post_out) return fail(c, VPT_ERR_INVALID_ARGUMENT, "vpt_get_output before the first post-process");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    { int rd = drain(c); if (rd) return rd; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out8, c->post_out, (size_t)c->post_w * c->post_h * 4, hipMemcpyDeviceToHost));
    return VPT_OK;
}

int vpt_get_stats(vpt_ctx* c, vpt_stats* out) {
    if (!c || !out) return VPT_ERR_INVALID_ARGUMENT;
    if (c->async_dirty || c->out_active) {
        HIPCHK(c, hipSetDevice(c->cfg.device));
        int rd = drain(c); if (rd) return rd;
    }
    vpt_stats s = c->stats;
    s.frames = c->frame_count; s.dispatches = c->dispatch_count;
    s.total_vertex_count = c->total_vertices; s.total_index_count = c->total_indices;
    s.bvh_nodes = c->dsc.node_count; s.bvh_triangles = c->dsc.tri_count;
    s.bvh_node_bytes = c->lds_scene ? sizeof(BvhNodeWide) : sizeof(BvhNode); s.bvh_tri_bytes = sizeof(BvhTri);
    s.emissive_mesh_count = (uint32_t)c->emissive.size(); s.emissive_triangle_count = c->emissive_tris;
    s.frames_in_flight = batch_cap(c); s.shard_pixels = c->P.shard_pixels;   // (the largest batch the context renders at once with its current scene and parameters)
    s.build_flags = (c->sbvh ? VPT_BUILD_SBVH : 0u) | (c->cfg.build_flags & (VPT_BUILD_GENERAL_KERNELS | VPT_BUILD_STREAMS_ONLY));
    s.frames_allocated = c->frames_alloc; s.resident_frames = c->resident_alloc;
    s.set_scene_ms = c->set_scene_ms; s.bvh_build_ms = c->bvh_build_ms;
    // what the traversal kernels have written into their spill regions: counted when something has run since the last count (the scan reads
    // ~0.4 GB: a host that asks for the statistics after every frame would otherwise pay 0.1-0.2 ms per call for a number that does not change)
    if (c->has_scene && c->stack_overflow_words && c->dsc.stack_overflow && c->spill_dirty) {
        HIPCHK(c, hipSetDevice(c->cfg.device));
        unsigned long long h[2] = {0ull, 0ull};
        HIPCHK(c, hipMemsetAsync(c->d_spill_count, 0, 16, c->stream));
        hipLaunchKernelGGL(k_count_spilled, dim3(1024), dim3(256), 0, c->stream, c->dsc.stack_overflow, c->stack_overflow_words, c->d_spill_count);
        hipLaunchKernelGGL(k_count_spilled, dim3(1024), dim3(256), 0, c->stream, c->stack_overflow2, c->stack_overflow_words, c->d_spill_count + 1);
        for (vpt_ctx* L : c->lanes)   // the lanes' own regions (pipelined asynchronous frames) count towards the first figure
            if (L && L->lane_spill && L->stack_overflow_words)
                hipLaunchKernelGGL(k_count_spilled, dim3(1024), dim3(256), 0, c->stream, (const uint32_t*)L->lane_spill, L->stack_overflow_words, c->d_spill_count);
        HIPCHK(c, hipMemcpyAsync(h, c->d_spill_count, 16, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->spill_cached[0] = h[0]; c->spill_cached[1] = h[1];
        c->spill_dirty = false;
    }
    s.stack_spills[0] = c->has_scene ? c->spill_cached[0] : 0; s.stack_spills[1] = c->has_scene ? c->spill_cached[1] : 0;
    *out = s;
    return VPT_OK;
}
int vpt_reset_stats(vpt_ctx* c) {
    if (!c) return VPT_ERR_INVALID_ARGUMENT;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    { int rd = drain(c); if (rd) return rd; }
    c->stats = vpt_stats{};
    HIPCHK(c, hipMemset(c->ctr, 0, sizeof(Counters)));
    memset(c->h_ctr, 0, sizeof(HostCounters));
    for (vpt_ctx* L : c->lanes) if (L) { HIPCHK(c, hipMemset(L->ctr, 0, sizeof(Counters))); memset(L->h_ctr, 0, sizeof(HostCounters)); L->stats = vpt_stats{}; }
    return VPT_OK;
}

int vpt_trace_rays(vpt_ctx* c, const vpt_ray* rays, uint32_t n, vpt_hit* hits) {
    if (!c || (n && (!rays || !hits))) return VPT_ERR_INVALID_ARGUMENT;
    if (!c->has_scene) return fail(c, VPT_ERR_NO_SCENE, "no scene");
    if (n == 0) return VPT_OK;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    { int rd = drain(c); if (rd) return rd; }
    vpt_ray* dr = nullptr; vpt_hit* dh = nullptr;
    HIPCHK(c, hipMalloc((void**)&dr, (size_t)n * sizeof(vpt_ray)));
    if (hipMalloc((void**)&dh, (size_t)n * sizeof(vpt_hit)) != hipSuccess) { (void)hipFree(dr); return fail(c, VPT_ERR_OUT_OF_MEMORY, "hipMalloc hits"); }
    int rc = VPT_OK;
    if (hipMemcpy(dr, rays, (size_t)n * sizeof(vpt_ray), hipMemcpyHostToDevice) != hipSuccess) rc = VPT_ERR_DEVICE;
    if (!rc) {
        c->spill_dirty = true;   // a traversal kernel runs: vpt_get_stats recounts the spill regions
        launch_trace_rays(c->stream, (uint32_t)c->max_blocks, c->dsc, dr, n, dh);
        if (hipStreamSynchronize(c->stream) != hipSuccess || hipGetLastError() != hipSuccess) rc = VPT_ERR_DEVICE;
    }
    if (!rc && hipMemcpy(hits, dh, (size_t)n * sizeof(vpt_hit), hipMemcpyDeviceToHost) != hipSuccess) rc = VPT_ERR_DEVICE;
    (void)hipFree(dr); (void)hipFree(dh);
    if (rc) c->err = "vpt_trace_rays: device error";
    return rc;
}


// ---- the one collective of the path (include/vpt.h) ----
namespace {
int ensure_gather_buf(vpt_ctx* c) {
    if (c->gather_buf) return VPT_OK;
    HIPCHK(c, hipMalloc((void**)&c->gather_buf, vpt_shard_floats(c) * 4 * (size_t)c->P.shard_count));
    return VPT_OK;
}
int nccl_fail(vpt_ctx* c, const char* what, ncclResult_t r) {
    c->err = std::string(what) + " failed: " + ncclGetErrorString(r);
    return VPT_ERR_DEVICE;
}
// root: gather_buf -> full image (rows re-interleaved); shard_count == 1: the image already is the whole image
int assemble_from_gather_buf(vpt_ctx* c) {
    if (c->P.shard_count == 1) return VPT_OK;
    launch_scatter_rows(c->stream, c->gather_buf, c->full_image, c->P.width, c->P.height, c->P.shard_count, (uint32_t)(vpt_shard_floats(c) / 4));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    c->full_valid = true;
    return VPT_OK;
}
}  // namespace

int vpt_comm_unique_id(void* id_out) {
    if (!id_out) return VPT_ERR_INVALID_ARGUMENT;
    static_assert(sizeof(ncclUniqueId) == VPT_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return VPT_ERR_DEVICE;
    memcpy(id_out, &id, sizeof(id));
    return VPT_OK;
}
int vpt_comm_init(vpt_ctx* c, const void* id, int rank, int world) {
    if (!c || !id) return VPT_ERR_INVALID_ARGUMENT;
    if (world < 1 || rank < 0 || rank >= world || (uint32_t)rank != c->P.shard_rank || (uint32_t)world != c->P.shard_count)
        return fail(c, VPT_ERR_INVALID_ARGUMENT, "vpt_comm_init: rank / world must equal the context's shard_rank / shard_count");
    if (c->comm) return fail(c, VPT_ERR_INVALID_ARGUMENT, "vpt_comm_init: the context already has a communicator");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    {   // the RCCL behind these calls is whichever librccl the process mapped first, not necessarily the one linked against
        int v = 0;
        if (ncclGetVersion(&v) != ncclSuccess) return fail(c, VPT_ERR_DEVICE, "ncclGetVersion failed");
        if (v / 10000 != NCCL_VERSION_CODE / 10000) {
            char msg[160]; snprintf(msg, sizeof(msg), "vpt_comm_init: the mapped RCCL is version %d, this library was built against %d (different major version)", v, (int)NCCL_VERSION_CODE);
            return fail(c, VPT_ERR_DEVICE, msg);
        }
    }
    ncclUniqueId uid; memcpy(&uid, id, sizeof(uid));
    ncclResult_t r = ncclCommInitRank(&c->comm, world, uid, rank);
    if (r != ncclSuccess) { c->comm = nullptr; return nccl_fail(c, "ncclCommInitRank", r); }
    c->comm_rank = rank; c->comm_world = world;
    return VPT_OK;
}
int vpt_comm_gather_shards(vpt_ctx* c, int root) {
    if (!c) return VPT_ERR_INVALID_ARGUMENT;
    if (!c->comm) return fail(c, VPT_ERR_INVALID_ARGUMENT, "vpt_comm_gather_shards before vpt_comm_init");
    if (root < 0 || root >= c->comm_world) return fail(c, VPT_ERR_INVALID_ARGUMENT, "root out of range");
    if (!c->buffers_ok) return fail(c, VPT_ERR_DEVICE, "no render buffers: the last vpt_resize failed");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    { int rd = drain(c); if (rd) return rd; }
    const bool is_root = c->comm_rank == root;
    if (is_root) { int rc = ensure_gather_buf(c); if (rc) return rc; }
    // every rank contributes its rows padded to the largest shard (the image buffer is allocated at that size);
    // the launch is ordered behind the renders already on the context's stream
    ncclResult_t r = ncclGather(c->image, is_root ? c->gather_buf : nullptr, vpt_shard_floats(c), ncclFloat32, root, c->comm, c->stream);
    if (r != ncclSuccess) return nccl_fail(c, "ncclGather", r);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return is_root ? assemble_from_gather_buf(c) : VPT_OK;
}
int vpt_comm_get_info(vpt_ctx* c, vpt_comm_info* out) {
    if (!c || !out) return VPT_ERR_INVALID_ARGUMENT;
    memset(out, 0, sizeof(*out));
    int v = 0;
    if (ncclGetVersion(&v) == ncclSuccess) out->rccl_version_runtime = v;
    out->rccl_version_compiled = (int32_t)NCCL_VERSION_CODE;
    Dl_info di{};
    if (dladdr((const void*)&ncclGather, &di) && di.dli_fname) snprintf(out->library_path, sizeof(out->library_path), "%s", di.dli_fname);
    out->rank = -1; out->device = -1;
    if (c->comm) {
        int n = 0, r = -1, d = -1;
        if (ncclCommCount(c->comm, &n) != ncclSuccess || ncclCommUserRank(c->comm, &r) != ncclSuccess || ncclCommCuDevice(c->comm, &d) != ncclSuccess)
            return fail(c, VPT_ERR_DEVICE, "ncclCommCount / ncclCommUserRank / ncclCommCuDevice failed");
        out->nranks = n; out->rank = r; out->device = d;
    }
    return VPT_OK;
}
int vpt_device_identity(vpt_ctx* c, char* out, uint32_t out_bytes) {
    if (!c || !out || out_bytes < 32) return VPT_ERR_INVALID_ARGUMENT;
    HIPCHK(c, hipDeviceGetPCIBusId(out, (int)out_bytes, c->cfg.device));
    return VPT_OK;
}
int vpt_comm_destroy(vpt_ctx* c) {
    if (!c) return VPT_ERR_INVALID_ARGUMENT;
    if (c->comm) {
        (void)hipSetDevice(c->cfg.device);
        ncclResult_t r = ncclCommDestroy(c->comm);
        c->comm = nullptr; c->comm_rank = -1; c->comm_world = 0;
        if (r != ncclSuccess) return nccl_fail(c, "ncclCommDestroy", r);
    }
    return VPT_OK;
}
int vpt_multi_gather_shards(vpt_ctx* const* ctxs, uint32_t count, uint32_t root) {
    if (!ctxs || count == 0 || root >= count || !ctxs[root]) return VPT_ERR_INVALID_ARGUMENT;
    vpt_ctx* R = ctxs[root];
    if (R->P.shard_count != count) return fail(R, VPT_ERR_INVALID_ARGUMENT, "vpt_multi_gather_shards: count must equal shard_count");
    for (uint32_t k = 0; k < count; k++) {
        vpt_ctx* c = ctxs[k];
        if (!c || c->P.shard_rank != k || c->P.shard_count != count || c->P.width != R->P.width || c->P.height != R->P.height || !c->buffers_ok)
            return fail(R, VPT_ERR_INVALID_ARGUMENT, "vpt_multi_gather_shards: context k must be shard k of the same image");
    }
    for (uint32_t k = 0; k < count; k++) { HIPCHK(R, hipSetDevice(ctxs[k]->cfg.device)); int rd = drain(ctxs[k]); if (rd) return rd; }
    HIPCHK(R, hipSetDevice(R->cfg.device));
    int rc = ensure_gather_buf(R);
    if (rc) return rc;
    const size_t stride = vpt_shard_floats(R) * 4;
    for (uint32_t k = 0; k < count; k++) {   // direct peer copies: xGMI is point to point, every shard takes its own link into root
        vpt_ctx* c = ctxs[k];
        HIPCHK(R, hipSetDevice(c->cfg.device));
        if (c->cfg.device != R->cfg.device) {
            hipError_t e = hipDeviceEnablePeerAccess(R->cfg.device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); }  // the copy below then stages through the host
            else (void)hipGetLastError();
        }
        HIPCHK(R, hipMemcpyPeerAsync((char*)R->gather_buf + (size_t)k * stride, R->cfg.device, c->image, c->cfg.device, stride, c->stream));
    }
    for (uint32_t k = 0; k < count; k++) {
        HIPCHK(R, hipSetDevice(ctxs[k]->cfg.device));
        HIPCHK(R, hipStreamSynchronize(ctxs[k]->stream));
    }
    HIPCHK(R, hipSetDevice(R->cfg.device));
    return assemble_from_gather_buf(R);
}

#if VPT_LAB   // ---- the laboratory's entry points (include/vpt_lab.h): absent from the product library
int vpt_lab_set(vpt_ctx* c, uint32_t key, uint32_t value) {
    if (!c || (value > 3u && key != VPT_LAB_WHOLE_FRAMES && key != VPT_LAB_WHOLE_SCHED)) return VPT_ERR_INVALID_ARGUMENT;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    { int rd = drain(c); if (rd) return rd; }
    if (key == VPT_LAB_LANES && value >= 1u) c->lab_lanes = value;
    else if (key == VPT_LAB_LANE_GRID && value >= 1u) c->lab_lane_grid = value;
    else if (key == VPT_LAB_TAIL_GRID && value >= 1u) c->lab_tail_grid = value;
    else if (key == VPT_LAB_WHOLE_SCHED && (value & 15u) >= 1u && (value >> 4) <= 3u) c->lab_whole_sched = value;
    else if (key == VPT_LAB_WHOLE_FRAMES) c->lab_whole_frames = value == 0xffffu ? 0xffffffffu : value;   // (0xffff: no bound, the default)
    else return VPT_ERR_INVALID_ARGUMENT;
    c->state_gen++;   // captured batches hold the old grids
    return VPT_OK;
}
int vpt_lab_set_rays(vpt_ctx* c, const vpt_ray* rays, uint32_t n) {
    if (!c || !rays || n == 0) return VPT_ERR_INVALID_ARGUMENT;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    free_lab(c);
    std::vector<float4> ro(n), rd(n);
    for (uint32_t i = 0; i < n; i++) {
        ro[i] = make_float4(rays[i].origin[0], rays[i].origin[1], rays[i].origin[2], 0.0f);
        rd[i] = make_float4(rays[i].direction[0], rays[i].direction[1], rays[i].direction[2], 0.0f);
    }
    HIPCHK(c, hipMalloc((void**)&c->lab_ro, (size_t)n * 16)); HIPCHK(c, hipMalloc((void**)&c->lab_rd, (size_t)n * 16));
    HIPCHK(c, hipMalloc((void**)&c->lab_hit, (size_t)n * 16)); HIPCHK(c, hipMalloc((void**)&c->lab_hinst, (size_t)n * 4));
    HIPCHK(c, hipMalloc((void**)&c->lab_order, (size_t)n * 4));
    HIPCHK(c, hipMemcpy(c->lab_ro, ro.data(), (size_t)n * 16, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->lab_rd, rd.data(), (size_t)n * 16, hipMemcpyHostToDevice));
    c->lab_n = n; c->lab_tmin = rays[0].tmin; c->lab_tmax = rays[0].tmax;
    return VPT_OK;
}
int vpt_lab_trace(vpt_ctx* c, uint32_t variant, uint32_t any_hit, const uint32_t* order, uint32_t param, uint32_t reps, vpt_hit* hits, float* best_ms,
                  uint64_t* visits) {
    if (!c || variant > VPT_TRACE_VOTE4S || reps == 0) return VPT_ERR_INVALID_ARGUMENT;
    if ((variant == VPT_TRACE_POOL || variant == VPT_TRACE_PAIR) && any_hit) return fail(c, VPT_ERR_UNSUPPORTED, "VPT_TRACE_POOL / _PAIR are closest-hit variants");
    if (!c->has_scene) return fail(c, VPT_ERR_NO_SCENE, "no scene");
    if (c->lds_scene) return fail(c, VPT_ERR_UNSUPPORTED, "the trace lab runs on scenes whose BVH lives in memory");
    if (c->lab_n == 0) return fail(c, VPT_ERR_INVALID_ARGUMENT, "vpt_lab_trace before vpt_lab_set_rays");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    { int rd = drain(c); if (rd) return rd; }
    if (variant == VPT_TRACE_VOTE4S && !c->dsc.nodes4s) {   // split-order experiment: the same binary tree collapsed pair-wise with order tables, over the same leaf-ordered triangles
        std::vector<BvhNode> n4, n4s; std::vector<BvhNodeWide> w4; std::vector<BvhTri> lt; int d = 0;
        BvhBuildOptions opt; opt.spatial_splits = c->sbvh; opt.nodes4s = &n4s;
        build_bvh_ex(c->bvh_input, n4, w4, lt, &d, opt);
        int rc4 = upload(c, n4s, &c->dsc.nodes4s);
        if (rc4) return rc4;
    }
    if (variant == VPT_TRACE_VOTE8 && !c->dsc.nodes8) {   // BVH8 experiment: the same binary tree collapsed eight-wide, over the same leaf-ordered triangles
        std::vector<BvhNode> n4; std::vector<BvhNodeWide> w4; std::vector<BvhTri> lt; std::vector<BvhNode8> n8; int d = 0;
        build_bvh(c->bvh_input, n4, w4, lt, &d, &n8, c->sbvh);
        if (n8.empty()) return fail(c, VPT_ERR_UNSUPPORTED, "no eight-wide tree for an empty scene");
        int rc8 = upload(c, n8, &c->dsc.nodes8);
        if (rc8) return rc8;
        c->stats.bvh8_nodes = (uint32_t)n8.size();
    }
    const uint32_t n = c->lab_n;
    c->spill_dirty = true;   // traversal kernels run: vpt_get_stats recounts the spill regions
    if (order) HIPCHK(c, hipMemcpy(c->lab_order, order, (size_t)n * 4, hipMemcpyHostToDevice));
    TraceArgs a{};
    a.ro = c->lab_ro; a.rd = c->lab_rd; a.order = order ? c->lab_order : nullptr; a.hit = c->lab_hit; a.hinst = c->lab_hinst;
    a.n = n; a.head = &c->ctr->extend_head; a.tmin = c->lab_tmin; a.tmax = c->lab_tmax; a.normalize_dir = 0u; a.param = (variant == VPT_TRACE_POOL || variant == VPT_TRACE_PAIR) ? param : param & 0xfff1ffffu;
    a.cull = variant == VPT_TRACE_VOTE ? (param >> 17) & 1u : 0u;     // lab: bit 17 = stale-entry culling (closest-hit, VPT_TRACE_VOTE)
    a.one_tri = variant == VPT_TRACE_VOTE ? (param >> 19) & 1u : 0u;     // lab: bit 19 = one triangle per triangle step, as before round 4 (VPT_TRACE_VOTE, product vote parameters)
    a.packed = variant == VPT_TRACE_VOTE ? (param >> 18) & 1u : 0u;   // lab: bit 18 = packed plane arithmetic in the node step (VPT_TRACE_VOTE, product vote parameters)
    // (the pool variant's spill region is indexed by slot: 512 slots per block against 256 threads)
    const uint32_t blocks = (uint32_t)std::min(trace_blocks_per_cu(variant, any_hit != 0) * c->cu_count, (variant == VPT_TRACE_POOL || variant == VPT_TRACE_PAIR) ? c->max_blocks / 2 : c->max_blocks);
    hipEvent_t e0, e1;
    HIPCHK(c, hipEventCreate(&e0)); HIPCHK(c, hipEventCreate(&e1));
    float best = 1e30f;
    for (uint32_t r = 0; r < reps + (visits ? 1u : 0u); r++) {
        const bool count = visits && r == reps;
        HIPCHK(c, hipMemsetAsync(c->ctr, 0, sizeof(Counters), c->stream));
        HIPCHK(c, hipEventRecord(e0, c->stream));
        launch_trace(c->stream, blocks, variant, any_hit != 0, count, c->dsc, a, c->ctr);
        HIPCHK(c, hipEventRecord(e1, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipGetLastError());
        float ms = 0.0f;
        HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
        if (!count) best = std::min(best, ms);
        else {
            Counters h{};
            HIPCHK(c, hipMemcpy(&h, c->ctr, sizeof(Counters), hipMemcpyDeviceToHost));
            visits[0] = h.stat_nodes; visits[1] = h.stat_tris;
        }
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    HIPCHK(c, hipMemset(c->ctr, 0, sizeof(Counters)));
    if (best_ms) *best_ms = best;
    if (hits) {
        std::vector<float4> h4(n); std::vector<uint32_t> hi(n, 0xffffffffu);
        HIPCHK(c, hipMemcpy(h4.data(), c->lab_hit, (size_t)n * 16, hipMemcpyDeviceToHost));
        if (!any_hit) HIPCHK(c, hipMemcpy(hi.data(), c->lab_hinst, (size_t)n * 4, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < n; i++) {
            hits[i].t = h4[i].x; hits[i].u = h4[i].y; hits[i].v = h4[i].z;
            uint32_t prim; memcpy(&prim, &h4[i].w, 4);
            hits[i].primitive = any_hit ? 0xffffffffu : prim; hits[i].instance = hi[i];
        }
    }
    return VPT_OK;
}

#endif  // VPT_LAB

int vpt_lut_calculate(int device, uint32_t kind, uint32_t sx, uint32_t sy, uint32_t sz, uint32_t sample_count, uint32_t time_ms, float* out) {
    // sampleCount / 20 passes (LookupTableCalculator.cpp:97); fewer than one pass would divide the table by zero
    if (!out || kind > VPT_LUT_REFRACT_BELOW || sx == 0 || sy == 0 || sz == 0 || (uint64_t)sx * sy * sz > (1u << 28) || sample_count < 20u)
        return VPT_ERR_INVALID_ARGUMENT;
    if (hipSetDevice(device) != hipSuccess) return VPT_ERR_DEVICE;
    const size_t cells = (size_t)sx * sy * sz;
    float* d = nullptr;
    if (hipMalloc((void**)&d, cells * 4) != hipSuccess) return VPT_ERR_OUT_OF_MEMORY;
    hipStream_t s = nullptr;
    int rc = VPT_OK;
    if (hipStreamCreate(&s) != hipSuccess || hipMemsetAsync(d, 0, cells * 4, s) != hipSuccess) rc = VPT_ERR_DEVICE;
    const uint32_t passes = sample_count / 20u, time_hash = vptfp::pcg_hash(time_ms);
    const uint32_t per_launch = 4096;  // bounds one launch to ~80k samples per cell
    for (uint32_t first = 0; !rc && first < passes; first += per_launch) {
        launch_lut(s, (int)kind, d, sx, sy, sz, sample_count, time_hash, first, std::min(per_launch, passes - first));
        if (hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) rc = VPT_ERR_DEVICE;
    }
    if (!rc && hipMemcpy(out, d, cells * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = VPT_ERR_DEVICE;
    if (!rc) for (size_t i = 0; i < cells; i++) out[i] /= (float)passes;  // LookupTableCalculator.cpp:152-155
    if (s) (void)hipStreamDestroy(s);
    (void)hipFree(d);
    return rc;
}

}  // extern "C"
