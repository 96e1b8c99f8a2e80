// kernels_path.hip — wavefront integrator stages for gfx950.
//
//   primary    bounce 0 fused: camera ray (RayGen.slang:12-64) + everything below for the first hit
//   extend     RayGen.slang:90        persistent-threads closest-hit traversal over the ray queue
//   shade      ClosestHit.slang + Miss.slang: surface, material, NEE sampling, BSDF sampling;
//              emits <=2 shadow rays per path into a wave-compacted shadow queue
//   shadow     RTCommon.slang:47-64   persistent-threads occlusion / light-identity queries
//   accumulate RayGen.slang:92-128    join visibility, luminance clamp, throughput, Russian roulette,
//              retire or regenerate paths, wave-compact survivors into the next queue
//   resolve    RayGen.slang:130-159   running mean over the frames in flight, in frame order
//
// A "wave" is 64 lanes; compaction uses one 64-bit ballot + mbcnt prefix and a single atomic per wave.
#include "kernels.hpp"
#include "shading.hpp"
#include "volume.hpp"
#include "atmosphere.hpp"
#include "traverse.hpp"
#include "wave.hpp"
#include "shade_core.hpp"
#include "vote.hpp"

namespace vpt {

#if VPT_LAB   // round 1's stage kernels (VPT_PIPELINE_STAGED_R1): laboratory build only
// ------------------------------------------------------------------ raygen (round 1's staged pipeline only)
// Scenes whose BVH does not fit in LDS run bounce 0 through the same extend / shade / connect stages as every
// other bounce, so the camera rays are written out as ordinary path records.
__global__ __launch_bounds__(256) void k_raygen(RenderParams P, PathState ps, uint32_t* queue, Counters* ctr, uint32_t n_slots, uint32_t dispatch_base) {
    uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li == 0u) ctr->ray_count[0] = n_slots;  // the counters were zeroed at the start of the batch
    if (li >= n_slots) return;
    uint32_t slot, x, y, f;
    launch_pixel(P, li, dispatch_base, slot, x, y, f);
    uint32_t seed = pcg_hash(P.base_seed + dispatch_base + f);  // PathTracer.cpp:139 with an explicit seed
    Rng r; r.s = y + P.width * x + seed;                        // RayGen.slang:28
    V3 o, d;
    camera_ray(P, r, x, y, o, d);
    ps.A[slot] = f4u(o, r.s);
    ps.B[slot] = f4u(d, 0u);
    ps.T[0][slot] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);  // pathThroughput = 1, payload.PDF = 1
    ps.L[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (P.samples_per_frame > 1) { ps.ACC[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); ps.sidx[slot] = 0u; }
    queue[li] = slot;
}

#endif  // VPT_LAB

// ------------------------------------------------------------------ persistent work fetch
template <bool LDS_SCENE>
__device__ inline void stage_scene(const DeviceScene& sc, float4* lds_nodes, float4* lds_tris) {
    if (LDS_SCENE) {
        const float4* gn = reinterpret_cast<const float4*>(sc.nodes_wide);
        const float4* gt = reinterpret_cast<const float4*>(sc.tris);
        for (uint32_t i = threadIdx.x; i < sc.node_count * 8; i += blockDim.x) lds_nodes[i] = gn[i];
        for (uint32_t i = threadIdx.x; i < sc.tri_count * 3; i += blockDim.x) lds_tris[i] = gt[i];
        __syncthreads();
    }
}
template <bool LDS_SCENE, bool COUNT>
__device__ inline bool trace_any(const DeviceScene& sc, const float4* lds_nodes, const float4* lds_tris, V3 o, V3 d, float tmin,
                                 float tmax, const TravStack& stack, HitRec& h, TravStats& st) {
    if (LDS_SCENE) { LdsSceneSrc src{lds_nodes, lds_tris, sc.strict_hits != 0u}; return trace_closest<COUNT>(src, o, d, tmin, tmax, stack, h, st); }
    GlobalSceneSrc src{sc.nodes, sc.tris, sc.strict_hits != 0u};
    return trace_closest<COUNT>(src, o, d, tmin, tmax, stack, h, st);
}

// Sky visibility / light identity as exact any-hit queries (traverse.hpp).
// rq: USE_RAY_QUERIES (RTCommon.slang:52-63: the direction as it is, TMin 1e-4, TMax 1e6); otherwise RTCommon.slang:64-84: normalised, TMin 1e-5, TMax 1000.
template <bool LDS_SCENE, bool COUNT>
__device__ inline bool sky_visible(const DeviceScene& sc, const float4* lds_nodes, const float4* lds_tris, V3 o, V3 d, const TravStack& stack, TravStats& st, bool rq) {
    const float tmin = rq ? 0.0001f : 0.00001f, tmax = rq ? 1000000.0f : 1000.0f;
    if (!rq) d = normalize(d);
    if (LDS_SCENE) { LdsSceneSrc src{lds_nodes, lds_tris, sc.strict_hits != 0u}; return !trace_occluded<COUNT, false>(src, o, d, tmin, tmax, 0.0f, 0u, stack, st); }
    GlobalSceneSrc src{sc.nodes, sc.tris, sc.strict_hits != 0u};
    return !trace_occluded<COUNT, false>(src, o, d, tmin, tmax, 0.0f, 0u, stack, st);
}
template <bool LDS_SCENE, bool COUNT>
__device__ inline bool light_visible(const DeviceScene& sc, const float4* lds_nodes, const float4* lds_tris, V3 o, V3 d, uint32_t gid, const TravStack& stack, TravStats& st) {
    uint32_t slot = sc.tri_slot_of_gid[gid];
    if (slot == 0xffffffffu) return false;  // the sampled light triangle is a sliver: nothing can hit it
    if (LDS_SCENE) { LdsSceneSrc src{lds_nodes, lds_tris, sc.strict_hits != 0u}; return closest_is<COUNT>(src, o, d, 0.0001f, 1000000.0f, gid, slot, stack, st); }
    GlobalSceneSrc src{sc.nodes, sc.tris, sc.strict_hits != 0u};
    return closest_is<COUNT>(src, o, d, 0.0001f, 1000000.0f, gid, slot, stack, st);
}

#if VPT_LAB
// ------------------------------------------------------------------ extend: closest hit of every queued path (round 1's stage kernels)
template <bool LDS_SCENE, bool COUNT, bool STRICT>
__global__ __launch_bounds__(kTraverseBlock, 8) void k_extend(DeviceScene sc, PathState ps, const uint32_t* queue,
                                                          Counters* ctr, uint32_t parity) {
    sc.strict_hits = STRICT ? 1u : 0u;  // compile-time constant from here on (VPT_FLAG_LOCAL_HITS picks the instantiation)
    extern __shared__ __align__(16) unsigned char smem[];
    const TravStack stack = make_stack(smem, sc.stack_overflow);
    float4* lds_nodes = reinterpret_cast<float4*>(smem + kStackDepth * kTraverseBlock * 4);
    float4* lds_tris = lds_nodes + sc.node_count * 8;
    stage_scene<LDS_SCENE>(sc, lds_nodes, lds_tris);
    const uint32_t n = ctr->ray_count[parity];
    const uint32_t chunk = fetch_chunk(n);
    TravStats st; st.nodes = 0; st.tris = 0;
    while (true) {
        uint32_t base = 0;
        if (lane_id() == 0) base = atomicAdd(&ctr->extend_head, chunk);
        base = __shfl(base, 0);
        if (base >= n) break;
        for (uint32_t k = 0; k < chunk; k += 64) {
            uint32_t i = base + k + lane_id();
            if (i >= n) break;
            uint32_t slot = queue[i];
            float4 a = ps.A[slot], b = ps.B[slot];
            V3 d = normalize(xyz(b));  // RayGen.slang:70
            HitRec h;
            bool found = trace_any<LDS_SCENE, COUNT>(sc, lds_nodes, lds_tris, xyz(a), d, 0.01f, 100000.0f, stack, h, st);
            ps.H[slot] = make_float4(found ? h.t : -1.0f, h.u, h.v, __uint_as_float(h.gid));   // the shade stage addresses the triangle's shading record by global id
            ps.hinst[slot] = h.inst;
        }
    }
    if (COUNT) {
        atomicAdd(&ctr->stat_nodes, (unsigned long long)st.nodes);
        atomicAdd(&ctr->stat_tris, (unsigned long long)st.tris);
    }
}

#endif  // VPT_LAB

// Test hook: the closest-hit traversal on caller-supplied rays.
__global__ __launch_bounds__(kTraverseBlock) void k_trace_rays(DeviceScene sc, const vpt_ray* rays, uint32_t n, vpt_hit* hits) {
    extern __shared__ __align__(16) unsigned char smem[];
    const TravStack stack = make_stack(smem, sc.stack_overflow);
    GlobalSceneSrc src{sc.nodes, sc.tris, sc.strict_hits != 0u};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        vpt_ray r = rays[i];
        HitRec h; TravStats st;
        bool found = trace_closest<false>(src, v3(r.origin[0], r.origin[1], r.origin[2]), v3(r.direction[0], r.direction[1], r.direction[2]),
                                          r.tmin, r.tmax, stack, h, st);
        vpt_hit o; o.t = found ? h.t : -1.0f; o.u = found ? h.u : 0.0f; o.v = found ? h.v : 0.0f; o.primitive = h.prim; o.instance = h.inst;
        hits[i] = o;
    }
}

#if VPT_LAB
// ------------------------------------------------------------------ shade (round 1's stage kernels)
// One path per lane: miss shader or closest-hit shader, then the tail of the reference's bounce loop that
// does not depend on visibility (throughput update, Russian roulette, termination, next-sample
// regeneration).  Survivors are ballot-compacted into the next ray queue; paths with anything pending
// (emission, NEE candidates, end of sample) are compacted into the connect queue — entries that carry
// shadow rays from the front, the others from the back, so a wave of the connect kernel is homogeneous.
// Result bits of shade_path()
constexpr uint32_t kSP_Alive = 1u, kSP_Front = 2u, kSP_Back = 4u;  // bits 3-4: number of shadow rays queued

__device__ __forceinline__ uint32_t shade_path(const DeviceScene& sc, const RenderParams& P, const PathState& ps,
                                               const float4* Tin, float4* Tout, uint32_t slot) {
    float4 a = ps.A[slot], b = ps.B[slot], t = Tin[slot];
    ShadeIn in_;
    in_.h = ps.H[slot];
    in_.inst = in_.h.x < 0.0f ? 0u : ps.hinst[slot];
    in_.rng = __float_as_uint(a.w);
    in_.porg = xyz(a); in_.pdir = xyz(b);
    uint32_t dw = __float_as_uint(b.w);
    in_.depth = dw & 0x7fffffffu; in_.in_medium = (dw >> 31) != 0u;
    in_.thr_prev = xyz(t); in_.prev_pdf = t.w;
    in_.vol_index = -1; in_.vol_t = 0.0f; in_.vdepth = 0u; in_.cchan = -1; in_.atm_comp = -1;
    ShadeOut o;
    shade_core<false>(sc, P, ps, slot, in_, o);
    if (o.alive) {
        ps.A[slot] = f4u(o.new_o, o.rng);
        ps.B[slot] = f4u(o.new_d, o.new_depth | (o.in_medium ? 0x80000000u : 0u));
        Tout[slot] = f4(o.thr, o.new_pdf);
    }
    const V3 tp = in_.thr_prev;
    bool thr_finite = !isinf_(tp.x) && !isinf_(tp.y) && !isinf_(tp.z) && !isnan_(tp.x) && !isnan_(tp.y) && !isnan_(tp.z);
    // 0 * inf = NaN must still reach pathLight, so a non-finite throughput always goes through connect
    bool pending = o.want_sky || o.want_light || o.terminated || o.emitted.x != 0.0f || o.emitted.y != 0.0f || o.emitted.z != 0.0f || !thr_finite;
    if (pending) {
        ps.CE[slot] = f4u(o.emitted, o.cflags);
        if (o.want_sky) {
            ps.CS[slot] = f4(o.csky, 0.0f);
            ps.CSO[slot] = f4(o.sky_o, o.sky_d.x);
            ps.CSD[slot] = make_float4(o.sky_d.y, o.sky_d.z, 0.0f, 0.0f);
        }
        if (o.want_light) {
            ps.CL[slot] = f4u(o.clight, o.light_gid);
            ps.CLO[slot] = f4(o.light_o, o.light_d.x);
            ps.CLD[slot] = make_float4(o.light_d.y, o.light_d.z, 0.0f, 0.0f);
        }
    }
    bool has_rays = o.want_sky || o.want_light;
    return (o.alive ? kSP_Alive : 0u) | ((pending && has_rays) ? kSP_Front : 0u) | ((pending && !has_rays) ? kSP_Back : 0u) |
           (((o.want_sky ? 1u : 0u) + (o.want_light ? 1u : 0u)) << 3);
}

// A block shades tiles of cpt x 256 paths (cpt = 1..4 per thread) and reserves queue space with ONE atomic
// per counter per tile; cpt grows with the queue so a launch issues at most ~8k atomics per counter.

__global__ __launch_bounds__(256, 3) void k_shade(DeviceScene sc, RenderParams P, PathState ps, const uint32_t* queue,
                                               uint32_t* queue_next, uint32_t* cqueue, Counters* ctr, uint32_t parity) {
    __shared__ uint32_t s_cnt[4][4];
    __shared__ uint32_t s_base[4][3];
    const uint32_t n = ctr->ray_count[parity];
    const float4* Tin = ps.T[parity];
    float4* Tout = ps.T[parity ^ 1u];
    const uint32_t wave = threadIdx.x >> 6;
    uint32_t cpt = (n + (1u << 21) - 1u) >> 21;
    cpt = cpt < 1u ? 1u : (cpt > 4u ? 4u : cpt);
    const uint32_t tile_size = cpt * 256u;
    for (uint32_t tile = blockIdx.x * tile_size; tile < n; tile += gridDim.x * tile_size) {
        uint32_t s0 = 0u, s1 = 0u, s2 = 0u, s3 = 0u, res = 0u;  // slots and 5-bit results of this lane's 4 paths
        uint32_t tot_alive = 0u, tot_front = 0u, tot_back = 0u, tot_rays = 0u;  // wave totals
#pragma unroll 1
        for (uint32_t c = 0; c < cpt; c++) {
            uint32_t i = tile + c * 256u + threadIdx.x;
            uint32_t r = 0u, slot = 0u;
            if (i < n) {
                slot = queue[i];
                r = shade_path(sc, P, ps, Tin, Tout, slot);
            }
            s0 = (c == 0u) ? slot : s0; s1 = (c == 1u) ? slot : s1; s2 = (c == 2u) ? slot : s2; s3 = (c == 3u) ? slot : s3;
            res |= r << (c * 5u);
            tot_alive += (uint32_t)__popcll(__ballot((r & kSP_Alive) != 0u));
            tot_front += (uint32_t)__popcll(__ballot((r & kSP_Front) != 0u));
            tot_back += (uint32_t)__popcll(__ballot((r & kSP_Back) != 0u));
            tot_rays += (uint32_t)__popcll(__ballot((r & 8u) != 0u)) + 2u * (uint32_t)__popcll(__ballot((r & 16u) != 0u));
        }
        // block-level reservation: one atomic per counter per 1024 paths
        if (lane_id() == 0) { s_cnt[wave][0] = tot_alive; s_cnt[wave][1] = tot_front; s_cnt[wave][2] = tot_back; s_cnt[wave][3] = tot_rays; }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t sum[4] = {0u, 0u, 0u, 0u};
            uint32_t pre[4][3];
            for (uint32_t w = 0; w < 4; w++)
                for (uint32_t q = 0; q < 4; q++) { if (q < 3) pre[w][q] = sum[q]; sum[q] += s_cnt[w][q]; }
            uint32_t b_alive = sum[0] ? atomicAdd(&ctr->ray_count[parity ^ 1u], sum[0]) : 0u;
            uint32_t b_front = sum[1] ? atomicAdd(&ctr->connect_front, sum[1]) : 0u;
            uint32_t b_back = sum[2] ? atomicAdd(&ctr->connect_back, sum[2]) : 0u;
            if (sum[3]) atomicAdd(&ctr->shadow_rays, sum[3]);
            for (uint32_t w = 0; w < 4; w++) { s_base[w][0] = b_alive + pre[w][0]; s_base[w][1] = b_front + pre[w][1]; s_base[w][2] = b_back + pre[w][2]; }
        }
        __syncthreads();
        uint32_t o_alive = s_base[wave][0], o_front = s_base[wave][1], o_back = s_base[wave][2];
#pragma unroll
        for (uint32_t c = 0; c < 4; c++) {
            uint32_t r = (res >> (c * 5u)) & 31u;
            uint32_t slot = c == 0u ? s0 : (c == 1u ? s1 : (c == 2u ? s2 : s3));
            unsigned long long ma = __ballot((r & kSP_Alive) != 0u), mf = __ballot((r & kSP_Front) != 0u), mb = __ballot((r & kSP_Back) != 0u);
            if (r & kSP_Alive) queue_next[o_alive + lanes_below(ma)] = slot;
            if (r & kSP_Front) cqueue[o_front + lanes_below(mf)] = slot;
            if (r & kSP_Back) cqueue[ps.capacity - 1u - (o_back + lanes_below(mb))] = slot;
            o_alive += (uint32_t)__popcll(ma); o_front += (uint32_t)__popcll(mf); o_back += (uint32_t)__popcll(mb);
        }
    }
}

#endif  // VPT_LAB

// ------------------------------------------------------------------ primary: bounce 0, fully fused
// Bounce 0 is ~60 % of all path-bounces of a frame (every pixel has one; later bounces only see the
// survivors), its rays are coherent, and nothing about it has to be read from memory: the slot id gives
// pixel and frame, hence seed, RNG state and camera ray (RayGen.slang:12-64).  So the first bounce runs as one
// kernel — camera ray, closest hit, miss/closest-hit shader, the <= 2 shadow rays, contribution, Russian
// roulette — and only survivors write their records (A, B, T, L) and enter the wavefront queues.
// Finished paths write just the frame sum.  pathThroughput is 1 and pathLight is 0 on entry.
// The same kernel with FIRST = false runs every later bounce of scenes whose BVH rides in LDS (traversal is
// then a handful of LDS reads, so a separate extend/connect stage would only move records through HBM):
// it reads a queued path's records A, B, T, L, does the whole bounce, and writes them back for survivors.
// PLAIN: the scene class set this instantiation serves — every material's textures are 1x1 and the environment is black (the Cornell
// box; chosen by vpt_set_scene / vpt_set_material, vpt_api.hip scene_is_plain).  The general kernel skips the texture taps and the
// environment sampler through uniform branches; here they are not compiled in at all (a quarter of the general kernel's instructions).
template <bool LDS_SCENE, bool COUNT, bool FIRST, bool VOL, bool STRICT, bool PLAIN = false>
__global__ __launch_bounds__(kTraverseBlock, 3) void k_bounce(DeviceScene sc, RenderParams P, PathState ps, StreamState ss, const uint32_t* queue,
                                                             uint32_t* queue_next, Counters* ctr, uint32_t parity, uint32_t n_slots,
                                                             uint32_t dispatch_base, uint32_t k3) {
    // compile-time constant from here on (VPT_FLAG_LOCAL_HITS picks the instantiation) — except in the media kernels (180 KB of code each, 688-720 B of
    // scratch per lane) and in the fused kernel on a tree in memory (VPT_PIPELINE_FUSED forced on a scene AUTO gives to the streams: half their rate),
    // which read the flag at run time: neither is near the speed of light, so one instantiation serves both hit rules
    if (!VOL && LDS_SCENE) sc.strict_hits = STRICT ? 1u : 0u;
    if (PLAIN) { sc.all_plain = 1u; sc.env_black = 1u; } else sc.all_plain = 0u;   // likewise
    if (FIRST && P.dispatch_base_dev) dispatch_base = *P.dispatch_base_dev;   // a replayed graph: the batch's first dispatch index lives in device memory
    const bool rq = (P.flags & VPT_FLAG_RAY_QUERIES) != 0u;   // USE_RAY_QUERIES (RTCommon.slang:52 / :64): which interval and direction the shadow and distance queries use
    extern __shared__ __align__(16) unsigned char smem[];
    const TravStack stack = make_stack(smem, sc.stack_overflow);
    float4* lds_nodes = reinterpret_cast<float4*>(smem + kStackDepth * kTraverseBlock * 4);
    float4* lds_tris = lds_nodes + sc.node_count * 8;
    stage_scene<LDS_SCENE>(sc, lds_nodes, lds_tris);
    // Queue k3 (= bounce index % 3) is read, queue k3 + 1 appended to, the words of queue k3 + 2 zeroed for the bounce after the
    // next: no reset kernel between two bounces.  A queue's length (holes included) is its static part — one chunk per wave
    // that took part in the producing launch, or nothing when that launch appended exactly — plus the dynamically reserved part.
    const uint32_t kn = (k3 + 1u) % 3u, kz = (k3 + 2u) % 3u;
    const uint32_t n = FIRST ? n_slots : ctr->rc3_static[k3] + ctr->rc3[k3];
    const uint32_t waves = gridDim.x * (kTraverseBlock / 64u);
    const uint32_t need = (n + 63u) / 64u, active = need < waves ? need : waves;
    const bool exact = n < kFusedExactBelow;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctr->rc3_static[kn] = exact ? 0u : active * kAppendChunk;
        ctr->rc3[kz] = 0u; ctr->alive3[kz] = 0u;
    }
    // Long queues: wave-private chunked appends (vote.hpp).  Short ones (< kFusedExactBelow entries): the block's four waves
    // add up their survivors in LDS and reserve them with ONE atomic per 256 paths, exactly — no holes, and few enough atomics
    // for a kernel whose whole launch takes ~0.1 ms at that size (one per wave would saturate the counter, ~88 / us).
    __shared__ uint32_t s_cnt[kTraverseBlock / 64u];
    __shared__ uint32_t s_base[kTraverseBlock / 64u];
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t gw = blockIdx.x * (kTraverseBlock / 64u) + wave;
    if (!exact && gw >= active) return;   // (after the block-wide staging above) this wave owns no chunk and no work
    WaveAppender a_next;
    a_next.init(gw, false, active * kAppendChunk);   // chunked mode: the counter rc3[kn] counts what is reserved beyond the static chunks
    TravStats st, sst; st.nodes = 0; st.tris = 0; sst.nodes = 0; sst.tris = 0;
    uint32_t w_paths = 0u, w_alive = 0u, w_rays = 0u, w_hits = 0u;  // wave totals (uniform)
    // Regrouping (long queues of the plain later bounces): a wave does not run the closest-hit shader on the 64 paths it has just
    // traced.  Its misses are finished at once (the miss shader is short), its hits are parked — hit record, ray and throughput
    // records — in a wave-private ring in LDS, and the closest-hit shader and the shadow queries run on FULL chunks of 64 hits
    // whenever the ring has that many (the partial last chunk is flushed at the end).  Queue holes vanish on the way.  Without it
    // a wave drags its misses and holes through the whole closest-hit shader as idle lanes (Cornell bounces: 41 of 64 lanes
    // active, 49 with it).  Every record is still read once, as a coalesced stream, except pathLight, which a parked hit fetches
    // when its chunk runs.  Results cannot change: every path gets exactly its own records and hit.
    const bool regroup = !FIRST && !VOL && !exact;   // (bounce 0: its camera rays are coherent; regrouping them measured 12 % slower)
    __shared__ uint32_t r_idx[kTraverseBlock / 64u][128], r_prim[kTraverseBlock / 64u][128], r_inst[kTraverseBlock / 64u][128];
    __shared__ float r_t[kTraverseBlock / 64u][128], r_u[kTraverseBlock / 64u][128], r_v[kTraverseBlock / 64u][128];
    __shared__ float4 r_ra[kTraverseBlock / 64u][128], r_rb[kTraverseBlock / 64u][128], r_rt[kTraverseBlock / 64u][128];
    uint32_t hit_head = 0u, hit_count = 0u;   // wave-uniform
    uint32_t tile = blockIdx.x;
    for (;;) {
        {
            const bool tiles_done = tile * kTraverseBlock >= n;
            uint32_t idx = 0u, slot = kHole;
            bool alive = false, hit = false, valid = false;
            uint32_t nrays = 0u;
            ShadeOut o;
            V3 light = v3s(0.0f);
            ShadeIn in_;
            V3 light_prev = v3s(0.0f);
            bool aborted = false;
            const bool pop_hits = regroup && (hit_count >= 64u || (tiles_done && hit_count > 0u));
            if (pop_hits) {   // a chunk of parked hits
                const uint32_t cnt = hit_count < 64u ? hit_count : 64u;
                valid = lane_id() < cnt;
                if (valid) {
                    const uint32_t q = (hit_head + lane_id()) & 127u;
                    idx = r_idx[wave][q];
                    const float4 a = r_ra[wave][q], b = r_rb[wave][q], t = r_rt[wave][q];
                    slot = queue[idx];
                    in_.rng = __float_as_uint(a.w);
                    in_.porg = xyz(a); in_.pdir = xyz(b);
                    const uint32_t dw = __float_as_uint(b.w);
                    in_.depth = dw & 0x7fffffffu; in_.in_medium = (dw >> 31) != 0u;
                    in_.thr_prev = xyz(t); in_.prev_pdf = t.w;
                    light_prev = xyz(ss.RL[parity][idx]);
                    in_.vdepth = 0u; in_.cchan = -1;
                    in_.vol_index = -1; in_.vol_t = 0.0f; in_.atm_comp = -1;
                    hit = true;
                    in_.h = make_float4(r_t[wave][q], r_u[wave][q], r_v[wave][q], __uint_as_float(r_prim[wave][q]));
                    in_.inst = r_inst[wave][q];
                }
                hit_head += cnt; hit_count -= cnt;
            } else if (!tiles_done) {
                idx = tile * kTraverseBlock + threadIdx.x;
                tile += gridDim.x;
                slot = FIRST ? idx : (idx < n ? queue[idx] : kHole);
                valid = idx < n && slot != kHole;
                HitRec hr;
                float4 rec_a = make_float4(0.0f, 0.0f, 0.0f, 0.0f), rec_b = rec_a, rec_t = rec_a;
                if (valid) {
                    if (FIRST) {
                        uint32_t x, y, f;
                        launch_pixel(P, idx, dispatch_base, slot, x, y, f);
                        uint32_t seed = pcg_hash(P.base_seed + dispatch_base + f);  // PathTracer.cpp:139 with an explicit seed
                        Rng r; r.s = y + P.width * x + seed;                        // RayGen.slang:28
                        camera_ray(P, r, x, y, in_.porg, in_.pdir);
                        in_.rng = r.s; in_.depth = 0u; in_.in_medium = false; in_.thr_prev = v3s(1.0f); in_.prev_pdf = 1.0f;
                        in_.vdepth = 0u; in_.cchan = -1;
                        if (P.samples_per_frame > 1) ps.sidx[slot] = 0u;
                    } else {   // the path's records, in queue order
                        const float4 a = ss.RA[parity][idx], b = ss.RB[parity][idx], t = ss.RT[parity][idx];
                        rec_a = a; rec_b = b; rec_t = t;
                        in_.rng = __float_as_uint(a.w);
                        in_.porg = xyz(a); in_.pdir = xyz(b);
                        uint32_t dw = __float_as_uint(b.w);
                        in_.depth = dw & 0x7fffffffu; in_.in_medium = (dw >> 31) != 0u;
                        in_.vdepth = VOL ? ps.vdepth[slot] : 0u;
                        in_.cchan = (VOL && sc.atm_on) ? ps.cchan[slot] : -1;
                        in_.thr_prev = xyz(t); in_.prev_pdf = t.w;
                        light_prev = xyz(ss.RL[parity][idx]);
                    }
                    in_.vol_index = -1; in_.vol_t = 0.0f; in_.atm_comp = -1;
                    // RayGen.slang:76-84: a path whose origin is below the planet's surface leaves the loop at once
                    aborted = VOL && sc.atm_on && atmosphere_height(sc, in_.porg) < 0.0f;
                    if (VOL && !aborted) {  // ScatteredInVolume (RayGen.slang:86): GetDistanceToGeometry uses the payload direction as is,
                                            // TMin 1e-5, TMax 1e6 (RTCommon.slang:86-101)
                        // (without USE_RAY_QUERIES: RTCommon.slang:103-117 — normalised direction, TMax 1000)
                        bool g = trace_any<LDS_SCENE, COUNT>(sc, lds_nodes, lds_tris, in_.porg, rq ? in_.pdir : normalize(in_.pdir), 0.00001f, rq ? 1000000.0f : 1000.0f, stack, hr, st);
                        Rng vr; vr.s = in_.rng;
                        int cc;
                        in_.vol_index = scattered_in_media(sc, in_.porg, in_.pdir, vr, g ? hr.t : -1.0f, (float)in_.depth, in_.cchan, in_.vol_t, in_.atm_comp, cc);
                        if (in_.vol_index == -2) in_.cchan = cc;  // the path now tracks this colour channel only (:242-247)
                        in_.rng = vr.s;
                    }
                    if (!VOL || (!aborted && in_.vol_index == -1))
                        hit = trace_any<LDS_SCENE, COUNT>(sc, lds_nodes, lds_tris, in_.porg, normalize(in_.pdir), 0.01f, 100000.0f, stack, hr, st);
                    in_.h = make_float4(hit ? hr.t : -1.0f, hr.u, hr.v, __uint_as_float(hr.gid));
                    in_.inst = hr.inst;
                }
                if (regroup) {   // park the hits (the ring holds < 64 entries here, so 128 slots are enough); the misses go on below
                    const unsigned long long mh = __ballot(valid && hit);
                    if (valid && hit) {
                        const uint32_t q = (hit_head + hit_count + lanes_below(mh)) & 127u;
                        r_idx[wave][q] = idx; r_t[wave][q] = hr.t; r_u[wave][q] = hr.u; r_v[wave][q] = hr.v; r_prim[wave][q] = hr.gid; r_inst[wave][q] = hr.inst;
                        r_ra[wave][q] = rec_a; r_rb[wave][q] = rec_b; r_rt[wave][q] = rec_t;
                    }
                    hit_count += (uint32_t)__popcll(mh);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    valid = valid && !hit;
                    if (__ballot(valid) == 0ull) continue;
                }
            } else {
                break;
            }
            if (valid) {
                if (VOL && aborted) {
                    o.want_sky = false; o.want_light = false; o.emitted = v3s(0.0f); o.csky = v3s(0.0f); o.clight = v3s(0.0f);
                    o.rng = in_.rng; o.new_depth = in_.depth; o.new_o = in_.porg; o.new_d = in_.pdir; o.new_pdf = in_.prev_pdf; o.bxdf = v3s(1.0f);
                    o.in_medium = in_.in_medium; o.vdepth = in_.vdepth; o.cchan = in_.cchan; o.light_gid = 0xffffffffu; o.light_miss_ok = false;
                } else {
                    shade_core<VOL>(sc, P, ps, slot, in_, o);
                }
            // connect, inline (RayGen.slang:92-102; FIRST: pathThroughput == 1, pathLight == 0)
                V3 E = o.emitted;
                if (!VOL && o.want_sky) {
                    if (sky_visible<LDS_SCENE, COUNT>(sc, lds_nodes, lds_tris, o.sky_o, o.sky_d, stack, sst, rq)) E = E + o.csky;
                    nrays++;
                }
                if (VOL && o.want_sky) {  // the sky term is assembled now: its transmittance draws come after the visibility test
                    nrays++;
                    if (sky_visible<LDS_SCENE, COUNT>(sc, lds_nodes, lds_tris, o.sky_o, o.sky_d, stack, sst, rq)) {
                        Rng tr_rng; tr_rng.s = o.rng;
                        V3 csky;
                        if (o.sky_kind == 2) {        // RayGen.slang:405-424: (phase * T_atm * T_boxes) * (sun / pdf)
                            V3 tr = atmosphere_transmittance(sc, tr_rng, o.sky_o, o.sky_d, o.cchan);
                            tr = tr * volumes_transmittance(sc, tr_rng, o.sky_o, o.sky_d, o.sky_tdepth);
                            csky = (o.sky_f * tr) * (o.sky_rgb / o.sky_w);
                        } else {
                            V3 tr = v3s(volumes_transmittance(sc, tr_rng, o.new_o, o.sky_d, o.sky_tdepth));  // from the new origin (ClosestHit.slang:332-349, RayGen.slang:325-343)
                            if (sc.atm_on) tr = nee_atmosphere_transmittance(sc, tr_rng, tr, o.new_o, o.sky_d, o.cchan);
                            if (o.sky_kind == 0) csky = ((o.sky_f * tr) * o.sky_rgb / o.sky_w) * o.sky_mis;
                            else csky = ((tr * o.sky_f) * (o.sky_rgb / o.sky_w)) * o.sky_mis;
                        }
                        o.rng = tr_rng.s;
                        if (o.sky_add) E = E + csky;
                    }
                }
                if (o.want_light) {
                    bool vis = light_visible<LDS_SCENE, COUNT>(sc, lds_nodes, lds_tris, o.light_o, o.light_d, o.light_gid, stack, sst);
                    if (VOL && !vis && o.light_miss_ok) vis = sky_visible<LDS_SCENE, COUNT>(sc, lds_nodes, lds_tris, o.light_o, o.light_d, stack, sst, true);   // (light rays exist with USE_RAY_QUERIES only)
                    if (VOL) {
                        if (vis) {  // ClosestHit.slang:361-370, RayGen.slang:348-361: the light term with the box transmittance
                            Rng tr_rng; tr_rng.s = o.rng;
                            V3 tr = v3s(volumes_transmittance(sc, tr_rng, o.new_o, o.light_d, o.light_tdepth));
                            o.rng = tr_rng.s;
                            V3 cl = o.light_kind == 0 ? ((o.light_f * tr) * o.light_rgb / o.light_w) * o.light_mis
                                                      : ((tr * o.light_f) * (o.light_rgb / o.light_w)) * o.light_mis;
                            if (o.light_add) E = E + cl;
                        }
                    } else if (vis) {
                        E = E + o.clight;
                    }
                    nrays++;
                }
                const int fin_chan = VOL ? o.cchan : -1;  // the channel this sample is accumulated in (RayGen.slang:118-128)
                if (VOL) shade_tail_media(P, ps, slot, in_.thr_prev, aborted, o);
                V3 contrib = E * in_.thr_prev;
                if (o.cflags & kCF_Clamp) {
                    float lum = dot(contrib, v3(0.212671f, 0.715160f, 0.072169f));
                    contrib = contrib * (P.max_luminance / max_(lum, P.max_luminance));
                }
                light = light_prev + contrib;
                if (VOL && aborted) light = light_prev;  // the loop was left before anything was added
                if (o.terminated) {  // end of a sample: NaN/Inf guard, frame sum (RayGen.slang:116-128)
                    bool ok = !isinf_(light.x) && !isinf_(light.y) && !isinf_(light.z) && !isnan_(light.x) && !isnan_(light.y) && !isnan_(light.z);
                    if (VOL && fin_chan != -1) light = v3(fin_chan == 0 ? light.x : 0.0f, fin_chan == 1 ? light.y : 0.0f, fin_chan == 2 ? light.z : 0.0f);
                    if (FIRST || P.samples_per_frame == 1) {  // first (or only) finalisation of the slot: 0 + pathLight
                        ps.ACC[slot] = ok ? f4(v3s(0.0f) + light, 0.0f) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    } else if (ok) {
                        float4 acc = ps.ACC[slot]; ps.ACC[slot] = f4(xyz(acc) + light, 0.0f);
                    }
                    light = v3s(0.0f);
                } else if (FIRST && P.samples_per_frame > 1) {
                    ps.ACC[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // later finalisations add to it
                }
                alive = o.alive;
                if (alive && VOL) { ps.vdepth[slot] = o.vdepth; if (sc.atm_on) ps.cchan[slot] = o.cchan; }
            }
            // survivors: the queue entry and, with it, the path's records go to the next queue (wave-private chunked append)
            uint32_t pn;
            if (!exact) pn = a_next.append(alive, &ctr->rc3[kn]);
            else {
                const unsigned long long ma = __ballot(alive);
                if (lane_id() == 0) s_cnt[wave] = (uint32_t)__popcll(ma);
                __syncthreads();
                if (threadIdx.x == 0) {
                    uint32_t sum = 0u, pre[kTraverseBlock / 64u];
                    for (uint32_t w = 0; w < kTraverseBlock / 64u; w++) { pre[w] = sum; sum += s_cnt[w]; }
                    const uint32_t b = sum ? atomicAdd(&ctr->rc3[kn], sum) : 0u;
                    for (uint32_t w = 0; w < kTraverseBlock / 64u; w++) s_base[w] = b + pre[w];
                }
                __syncthreads();
                pn = s_base[wave] + lanes_below(ma);
            }
            if (alive) {
                queue_next[pn] = slot;
                ss.RA[parity ^ 1u][pn] = f4u(o.new_o, o.rng);
                ss.RB[parity ^ 1u][pn] = f4u(o.new_d, o.new_depth | (o.in_medium ? 0x80000000u : 0u));
                ss.RT[parity ^ 1u][pn] = f4(o.thr, o.new_pdf);
                ss.RL[parity ^ 1u][pn] = f4(light, 0.0f);
            }
            w_paths += (uint32_t)__popcll(__ballot(valid));
            w_alive += (uint32_t)__popcll(__ballot(alive));
            w_rays += (uint32_t)__popcll(__ballot(nrays >= 1u)) + (uint32_t)__popcll(__ballot(nrays >= 2u));
            w_hits += (uint32_t)__popcll(__ballot(hit));
        }
    }
    if (!exact) for (uint32_t j = lane_id(); j < a_next.tail_count(); j += 64u) queue_next[a_next.tail_first() + j] = kHole;  // the unwritten tail of the wave's last chunk
    if (lane_id() == 0) {
        if (w_alive) atomicAdd(&ctr->alive3[kn], w_alive);
        if (w_paths) atomicAdd(&ctr->stat_closest, (unsigned long long)w_paths);
        if (w_rays) atomicAdd(&ctr->stat_shadow, (unsigned long long)w_rays);
        if (FIRST) {
            if (w_hits) atomicAdd(&ctr->stat_primary_hits, (unsigned long long)w_hits);
            if (w_alive) atomicAdd(&ctr->stat_primary_alive, (unsigned long long)w_alive);
            if (w_rays) atomicAdd(&ctr->stat_primary_rays, (unsigned long long)w_rays);
        }
    }
    if (COUNT) {
        atomicAdd(&ctr->stat_nodes, (unsigned long long)st.nodes);
        atomicAdd(&ctr->stat_tris, (unsigned long long)st.tris);
        atomicAdd(&ctr->stat_shadow_nodes, (unsigned long long)sst.nodes);
        atomicAdd(&ctr->stat_shadow_tris, (unsigned long long)sst.tris);
    }
}

// ------------------------------------------------------------------ the rest of a batch in one launch (streams pipeline, short queues)
// A bounce of the streams pipeline is seven dependent launches.  On a queue of 10^5 paths each of them is bounded below by launch latency and by
// the tail of its persistent grid, not by throughput: a 1-frame batch of the atrium spent 4.9 ms in 56 such launches against 1.4 ms per frame
// inside a 226-frame batch, the glass bust (depth 32: 224 launches) 11.8 ms against 0.68 (profiles/r05_latency.json).  k_finish takes what is left of
// a batch once its queue is short and runs every path to its END: closest hit, miss / closest-hit shader, the <= 2 shadow queries, contribution,
// roulette, next bounce, exactly as k_bounce does per bounce (the same shade_core and connect code on the same values in the same order:
// bit-identical to the streams' stages and to the oracle).  No media, no regeneration of camera paths.
// Round 6: persistent waves.  A lane owns ONE path (records read from the streams at the queue's parity) and keeps it in registers from
// bounce to bounce; a lane whose path has ended takes the next entry of the queue (every wave starts on its own 64 entries, further chunks
// through sctr->finish_head), so a wave works on full lanes until the queue runs dry instead of idling behind its longest path (round 5:
// 10 % lane use, profiles/r05_bust_p2_summary.md).  The three searches of a bounce run on the tree in memory with the wave-level vote of the
// stream kernels (vote.hpp: ONE kind of step per iteration — node step or triangle step — for the lanes that want it) instead of the per-lane
// loops of traverse.hpp; a ray's own sequence of visits, tests and interval updates is unchanged, so hits and visibility are identical.
// STRICT (VPT_FLAG_LOCAL_HITS) keeps the validating per-lane loops.
struct VoteRay {   // one search of a lane
    int cur, sp;
    float best_t, bu, bv;
    uint32_t bslot, bgid;
};
// closest hit (tmin < t < tmax; ties -> smaller global id) of the lanes with `active`; every lane of the wave calls
template <bool COUNT>
__device__ __forceinline__ bool mem_closest_vote(const BvhNode* nodes, const BvhTri* tris, const TreeTop& top, const LaneStack& S, bool active, V3 o, V3 d, float tmin, float tmax,
                                                 HitRec& best, TravStats& st) {
    VoteRay r; r.cur = active ? 0 : kLaneDone; r.sp = 0; r.best_t = tmax; r.bu = 0.0f; r.bv = 0.0f; r.bslot = 0xffffffffu; r.bgid = 0xffffffffu;
    const V3 inv = safe_inverse(d);
    while (true) {
        const bool busy = r.cur < kLaneDone, at_node = busy && r.cur >= 0, at_leaf = busy && r.cur < 0;
        const uint32_t nn = (uint32_t)__popcll(__ballot(at_node)), nl = (uint32_t)__popcll(__ballot(at_leaf));
        if (nn + nl == 0u) break;
        const bool node_wins = 4u * nn > kVoteWeight4 * nl;
        if (node_wins & at_node) {
            if (COUNT) st.nodes++;
            vote_node_step<false, false, false>(nodes, top, S, r.cur, r.sp, o, inv, tmin, r.best_t);
        }
        if (!node_wins & at_leaf) {
            if (COUNT) { st.tris++; vote_tri_step_closest<false, false>(tris, S, r.cur, r.sp, o, d, tmin, tmax, r.best_t, r.bu, r.bv, r.bslot, r.bgid); }   // (the counting build: one triangle per step, the count is what the ray needs)
            else vote_tri2_step_closest(tris, S, r.cur, r.sp, o, d, tmin, tmax, r.best_t, r.bu, r.bv, r.bslot, r.bgid);
        }
    }
    const bool found = r.bslot != 0xffffffffu;
    best.t = r.best_t; best.u = r.bu; best.v = r.bv; best.prim = 0xffffffffu; best.inst = 0xffffffffu; best.gid = 0xffffffffu; best.slot = 0;
    if (found) { best.prim = tris[r.bslot].prim; best.inst = tris[r.bslot].inst; best.gid = r.bgid; best.slot = (int)r.bslot; }
    return found;
}
// any-hit: is some triangle hit with t < tlim, or t == tlim and a smaller global id than `expect` (traverse.hpp trace_occluded_pass)?
template <bool COUNT>
__device__ __forceinline__ bool mem_occluded_vote(const BvhNode* nodes, const BvhTri* tris, const TreeTop& top, const LaneStack& S, bool active, V3 o, V3 d, float tmin, float tmax,
                                                  float tlim, uint32_t expect, TravStats& st) {
    int cur = active ? 0 : kLaneDone, sp = 0;
    bool occluded = false;
    const V3 inv = safe_inverse(d);
    while (true) {
        const bool busy = cur < kLaneDone, at_node = busy && cur >= 0, at_leaf = busy && cur < 0;
        const uint32_t nn = (uint32_t)__popcll(__ballot(at_node)), nl = (uint32_t)__popcll(__ballot(at_leaf));
        if (nn + nl == 0u) break;
        const bool node_wins = 4u * nn > kVoteWeight4 * nl;
        if (node_wins & at_node) {
            if (COUNT) st.nodes++;
            vote_node_step<true>(nodes, top, S, cur, sp, o, inv, tmin, tlim);
        }
        if (!node_wins & at_leaf) {
            if (COUNT) { st.tris++; if (vote_tri_step_any<false>(tris, S, cur, sp, o, d, tmin, tmax, tlim, expect)) occluded = true; }
            else if (vote_tri2_step_any(tris, S, cur, sp, o, d, tmin, tmax, tlim, expect)) occluded = true;
        }
    }
    return occluded;
}
#ifndef VPT_FINISH_MIN_BLOCKS   // (-D override: the A/B builds of tests/tools/ab_variants.sh)
#define VPT_FINISH_MIN_BLOCKS 3
#endif
template <bool COUNT, bool STRICT>
__global__ __launch_bounds__(kTraverseBlock, VPT_FINISH_MIN_BLOCKS) void k_finish(DeviceScene sc, RenderParams P, PathState ps, StreamState ss, const uint32_t* queue, StreamCounters* sctr,
                                                             Counters* ctr, uint32_t parity) {
    sc.all_plain = 0u; sc.strict_hits = STRICT ? 1u : 0u;
    const bool rq = (P.flags & VPT_FLAG_RAY_QUERIES) != 0u;
    extern __shared__ __align__(16) unsigned char smem[];
    const TravStack stack = make_stack(smem, sc.stack_overflow);     // STRICT: the per-lane loops
    const LaneStack S = make_lane_stack(smem, sc.stack_overflow);     // the same LDS rows and spill region, as the vote steps address them
    const BvhNode* const nodes = sc.nodes;
    const BvhTri* const tris = sc.tris;
    const TreeTop top = stage_tree_top(smem, nodes, sc.node_count, !STRICT);   // the any-hit searches read the top of the tree from LDS (vote.hpp)
    const uint32_t n = sctr->queue_len[parity].v;
    TravStats st, sst; st.nodes = 0; st.tris = 0; sst.nodes = 0; sst.tris = 0;
    uint32_t w_paths = 0u, w_rays = 0u, w_taken = 0u;   // wave totals (uniform): closest-hit rays, shadow rays, paths taken over
    // the wave's cursor into the queue, wave-uniform by construction (kernels_trace.hip k_trace_vote)
    const uint32_t n_static = gridDim.x * (kTraverseBlock / 64u) * 64u;
    uint32_t w_next = __builtin_amdgcn_readfirstlane((blockIdx.x * (kTraverseBlock / 64u) + (threadIdx.x >> 6)) * 64u), w_end = w_next + 64u < n ? w_next + 64u : n;
    if (w_next >= n) { w_next = 0u; w_end = 0u; }
    bool exhausted = false;
    // the lane's path between two bounces
    bool has_path = false;
    uint32_t slot = 0u, rng_s = 0u, depth = 0u;
    bool in_medium = false;
    V3 porg = v3s(0.0f), pdir = v3s(0.0f), thr = v3s(1.0f), lightp = v3s(0.0f);
    float pdf = 1.0f;
    for (;;) {
        // ---- refill: free lanes take the next queue entries (a second pass when the wave's chunk ran out half-way)
        if (!exhausted) {
#pragma unroll 1
            for (int pass = 0; pass < 2; pass++) {
                const unsigned long long m_free = __ballot(!has_path);
                if (m_free == 0ull) break;
                if (w_next >= w_end) {
                    if (n_static >= n) exhausted = true;
                    else {
                        uint32_t base = 0u;
                        if (lane_id() == 0u) base = atomicAdd(&sctr->finish_head.v, 64u);
                        base = n_static + __builtin_amdgcn_readfirstlane(base);
                        if (base >= n) exhausted = true;
                        else { w_next = base; w_end = base + 64u < n ? base + 64u : n; }
                    }
                }
                if (exhausted) break;
                const uint32_t idx = w_next + lanes_below(m_free);
                bool took = false;
                if (!has_path && idx < w_end) {
                    slot = queue[idx];
                    if (slot != kHole) {   // a hole: the tail of some wave's last chunk of the queue (vote.hpp WaveAppender)
                        const float4 a = ss.RA[parity][idx], b = ss.RB[parity][idx], t = ss.RT[parity][idx];
                        rng_s = __float_as_uint(a.w); porg = xyz(a); pdir = xyz(b);
                        const uint32_t dw = __float_as_uint(b.w);
                        depth = dw & 0x7fffffffu; in_medium = (dw >> 31) != 0u;
                        thr = xyz(t); pdf = t.w;
                        lightp = xyz(ss.RL[parity][idx]);
                        has_path = true; took = true;
                    }
                }
                w_taken += (uint32_t)__popcll(__ballot(took));
                const uint32_t want = (uint32_t)__popcll(m_free), left = w_end - w_next;
                w_next += want < left ? want : left;
            }
        }
        if (__ballot(has_path) == 0ull) {
            if (exhausted) break;
            continue;
        }
        // ---- one bounce of every lane's path
        HitRec hr;
        bool hit = false;
        if constexpr (STRICT) { if (has_path) hit = trace_any<false, COUNT>(sc, nullptr, nullptr, porg, normalize(pdir), 0.01f, 100000.0f, stack, hr, st); }
        else hit = mem_closest_vote<COUNT>(nodes, tris, top, S, has_path, porg, normalize(pdir), 0.01f, 100000.0f, hr, st);
        w_paths += (uint32_t)__popcll(__ballot(has_path));
        ShadeIn in_;
        ShadeOut o;
        o.want_sky = false; o.want_light = false;
        if (has_path) {
            in_.rng = rng_s; in_.porg = porg; in_.pdir = pdir; in_.depth = depth; in_.in_medium = in_medium; in_.thr_prev = thr; in_.prev_pdf = pdf;
            in_.vdepth = 0u; in_.cchan = -1; in_.vol_index = -1; in_.vol_t = 0.0f; in_.atm_comp = -1;
            in_.h = make_float4(hit ? hr.t : -1.0f, hr.u, hr.v, __uint_as_float(hr.gid));
            in_.inst = hr.inst;
            shade_core<false>(sc, P, ps, slot, in_, o);
        }
        // connect, inline (RayGen.slang:92-102), as k_bounce
        const bool q_sky = has_path && o.want_sky, q_light = has_path && o.want_light;
        bool vis_sky = false, vis_light = false;
        if constexpr (STRICT) {
            if (q_sky) vis_sky = sky_visible<false, COUNT>(sc, nullptr, nullptr, o.sky_o, o.sky_d, stack, sst, rq);
            if (q_light) vis_light = light_visible<false, COUNT>(sc, nullptr, nullptr, o.light_o, o.light_d, o.light_gid, stack, sst);
        } else {
            if (__ballot(q_sky) != 0ull) {   // RTCommon.slang:52-63 (USE_RAY_QUERIES) or :64-84 (sky_visible above)
                const float tmin = rq ? 0.0001f : 0.00001f, tmax = rq ? 1000000.0f : 1000.0f;
                const V3 sd = rq ? o.sky_d : normalize(o.sky_d);
                vis_sky = !mem_occluded_vote<COUNT>(nodes, tris, top, S, q_sky, o.sky_o, sd, tmin, tmax, tmax, 0xffffffffu, sst);
            }
            if (__ballot(q_light) != 0ull) {   // traverse.hpp closest_is: the sampled triangle by its own record first, then the search for anything that beats it
                bool search = false;
                float t_e = 0.0f;
                if (q_light) {
                    const uint32_t lslot = sc.tri_slot_of_gid[o.light_gid];
                    if (lslot != 0xffffffffu) {   // 0xffffffff: the sampled light triangle is a sliver, nothing can hit it
                        const float4* q = reinterpret_cast<const float4*>(tris + lslot);
                        const float4 ta = q[0], tb = q[1], tc = q[2];
                        if (COUNT) sst.tris++;
                        float u, v;
                        search = vptfp::ray_triangle(o.light_o, o.light_d, vptfp::v3(ta.x, ta.y, ta.z), vptfp::v3(ta.w, tb.x, tb.y), vptfp::v3(tb.z, tb.w, tc.x), 0.0001f, 1000000.0f, &t_e, &u, &v);
                    }
                }
                const bool occ = mem_occluded_vote<COUNT>(nodes, tris, top, S, search, o.light_o, o.light_d, 0.0001f, 1000000.0f, t_e, o.light_gid, sst);
                vis_light = search && !occ;
            }
        }
        w_rays += (uint32_t)__popcll(__ballot(q_sky)) + (uint32_t)__popcll(__ballot(q_light));
        if (has_path) {
            V3 E = o.emitted;
            if (q_sky && vis_sky) E = E + o.csky;
            if (q_light && vis_light) E = E + o.clight;
            V3 contrib = E * in_.thr_prev;
            if (o.cflags & kCF_Clamp) {
                float lum = dot(contrib, v3(0.212671f, 0.715160f, 0.072169f));
                contrib = contrib * (P.max_luminance / max_(lum, P.max_luminance));
            }
            V3 light = lightp + contrib;
            if (o.terminated) {  // end of a sample: NaN/Inf guard, frame sum (RayGen.slang:116-128)
                const bool ok = !isinf_(light.x) && !isinf_(light.y) && !isinf_(light.z) && !isnan_(light.x) && !isnan_(light.y) && !isnan_(light.z);
                if (P.samples_per_frame == 1) ps.ACC[slot] = ok ? f4(v3s(0.0f) + light, 0.0f) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                else if (ok) { float4 acc = ps.ACC[slot]; ps.ACC[slot] = f4(xyz(acc) + light, 0.0f); }
                light = v3s(0.0f);   // the pixel's next sample of the frame starts from pathLight = 0
            }
            if (o.alive) { rng_s = o.rng; porg = o.new_o; pdir = o.new_d; depth = o.new_depth; in_medium = o.in_medium; thr = o.thr; pdf = o.new_pdf; lightp = light; }
            else has_path = false;
        }
    }
    // wave totals -> the running statistics
    if (lane_id() == 0u) {
        if (w_paths) { atomicAdd(&ctr->stat_closest, (unsigned long long)w_paths); atomicAdd(&ctr->stat_finish_closest, (unsigned long long)w_paths); }
        if (w_rays) { atomicAdd(&ctr->stat_shadow, (unsigned long long)w_rays); atomicAdd(&ctr->stat_finish_shadow, (unsigned long long)w_rays); }
        if (w_taken) atomicAdd(&ctr->stat_finish_paths, (unsigned long long)w_taken);
    }
    if (COUNT) {
        uint32_t a0 = st.nodes, a1 = st.tris, a2 = sst.nodes, a3 = sst.tris;
        for (int off = 32; off > 0; off >>= 1) { a0 += __shfl_down(a0, off); a1 += __shfl_down(a1, off); a2 += __shfl_down(a2, off); a3 += __shfl_down(a3, off); }
        if (lane_id() == 0u) {
            atomicAdd(&ctr->stat_nodes, (unsigned long long)a0);
            atomicAdd(&ctr->stat_tris, (unsigned long long)a1);
            atomicAdd(&ctr->stat_shadow_nodes, (unsigned long long)a2);
            atomicAdd(&ctr->stat_shadow_tris, (unsigned long long)a3);
        }
    }
}
// Behind k_finish nothing of the batch is alive: the queue words say so (the guarded resolve and the host read them).
__global__ void k_finish_done(StreamCounters* sctr, uint32_t parity) { sctr->alive[parity].v = 0u; sctr->queue_len[parity].v = 0u; sctr->finish_head.v = 0u; }

// ------------------------------------------------------------------ vote-scheduled traversal of the tree in LDS (k_whole)
// The per-lane loops of traverse.hpp make a wave run the node branch AND the leaf branch of every iteration as soon as its lanes stand at
// different places of their trees — on the Cornell box a ray needs 1.9 node visits and 1.4 triangle tests, in no particular order.  These are
// the same searches with the wave-level vote of the stream kernels (vote.hpp): every iteration the lanes that are in the call execute ONE kind of
// step — an inner-node step (fp32 node from LDS, four slab tests, nearest-first order / slot order) or a one-triangle step — chosen by ballot, and
// a lane that wants the other kind waits a turn.  A ray's own sequence of visits, tests and interval updates is exactly that of
// trace_closest_pass / trace_occluded_pass (the state machine is per ray; only the interleaving changes), so hits, visibility and the visit
// counters are identical.  Lanes outside the call (no ray, no shadow ray) are simply not part of the ballots.  The validating (STRICT)
// instantiations keep the per-lane loops.
// MEASURED AND NOT TAKEN (round 5, same box, alternating, three rounds: profiles/r05_whole_vote_ab.json): Cornell 1080p 8277-8281 Msamples/s with it
// against 8376-8378 with the per-lane loops (-1.2 %), images and ray statistics identical; 168 VGPRs either way, 16 instead of 36 B of scratch.
// The tree is twelve triangles under three nodes: a lane's two or three steps are over before a vote per step can pay for itself, and what
// idles the lanes of this kernel (47 of 64 per VALU instruction) is the shader's own branching, not the searches.  Built only with
// -DVPT_WHOLE_VOTE=1 (tests/tools/build_variant.py).
#ifndef VPT_WHOLE_VOTE
#define VPT_WHOLE_VOTE 0
#endif
#ifndef VPT_DIAG_NO_LIGHT_SEARCH
#define VPT_DIAG_NO_LIGHT_SEARCH 0
#endif
constexpr int kWalkDone = 0x7fffffff;
__device__ __forceinline__ int walk_pop(TravStack& stack) { return stack.sp ? (int)stack.pop() : kWalkDone; }
template <bool COUNT>
__device__ __forceinline__ bool lds_closest_vote(const LdsSceneSrc& src, V3 o, V3 d, float tmin, float tmax, TravStack stack, HitRec& best, TravStats& st) {
    best.t = tmax; best.u = 0.0f; best.v = 0.0f; best.prim = 0xffffffffu; best.inst = 0xffffffffu; best.gid = 0xffffffffu; best.slot = 0;
    bool found = false;
    const RaySlab slab = make_slab(o, d);
    stack.sp = 0;
    int cur = 0;   // root is inner node 0
    while (true) {
        const bool at_node = cur >= 0 && cur != kWalkDone, at_leaf = cur < 0;
        const uint32_t nn = (uint32_t)__popcll(__ballot(at_node)), nl = (uint32_t)__popcll(__ballot(at_leaf));
        if (nn + nl == 0u) break;
        const bool node_wins = 4u * nn > kVoteWeight4 * nl;   // a node step retires about twice the work of a triangle step
        if (node_wins & at_node) {
            NodeDataWide n;
            src.node(cur, n);
            if (COUNT) st.nodes++;
            float t0, t1, t2, t3;
            node_entries(n, slab, tmin, best.t, t0, t1, t2, t3);
            int c0 = n.c0, c1 = n.c1, c2 = n.c2, c3 = n.c3;
            cswap(t0, c0, t1, c1); cswap(t2, c2, t3, c3); cswap(t0, c0, t2, c2); cswap(t1, c1, t3, c3); cswap(t1, c1, t2, c2);
            if (t0 < kMissT) {  // nearest child next, the others pushed far -> near
                if (t3 < kMissT) stack.push((uint32_t)c3);
                if (t2 < kMissT) stack.push((uint32_t)c2);
                if (t1 < kMissT) stack.push((uint32_t)c1);
                cur = c0;
            } else cur = walk_pop(stack);
        }
        if (!node_wins & at_leaf) {   // ONE triangle of the lane's leaf
            const uint32_t enc = (uint32_t)(~cur);
            const int first = (int)(enc >> 3);
            const uint32_t more = enc & 7u;
            float4 a, b, c;
            src.tri(first, a, b, c);
            if (COUNT) st.tris++;
            float t, u, v;
            const bool hit = ray_triangle_flat(o, d, vptfp::v3(a.x, a.y, a.z), vptfp::v3(a.w, b.x, b.y), vptfp::v3(b.z, b.w, c.x), tmin, tmax, t, u, v);
            const uint32_t gid = __float_as_uint(c.w);
            if (hit & (!found | (t < best.t) | ((t == best.t) & (gid < best.gid)))) {
                best.t = t; best.u = u; best.v = v; best.prim = __float_as_uint(c.y); best.inst = __float_as_uint(c.z); best.gid = gid;
                found = true;
            }
            if (more) cur = ~(int)((((uint32_t)first + 1u) << 3) | (more - 1u));
            else cur = walk_pop(stack);
        }
    }
    return found;
}
// any-hit: LIGHT = false: occluded <=> some triangle is hit in (tmin, tmax); LIGHT = true: something beats the sampled triangle's hit at t_e (traverse.hpp)
template <bool COUNT, bool LIGHT>
__device__ __forceinline__ bool lds_occluded_vote(const LdsSceneSrc& src, V3 o, V3 d, float tmin, float tmax, float t_e, uint32_t expect, TravStack stack, TravStats& st) {
    const float tlimit = LIGHT ? t_e : tmax;
    const RaySlab slab = make_slab(o, d);
    stack.sp = 0;
    int cur = 0;
    bool occluded = false;
    while (true) {
        const bool at_node = cur >= 0 && cur != kWalkDone, at_leaf = cur < 0;
        const uint32_t nn = (uint32_t)__popcll(__ballot(at_node)), nl = (uint32_t)__popcll(__ballot(at_leaf));
        if (nn + nl == 0u) break;
        const bool node_wins = 4u * nn > kVoteWeight4 * nl;
        if (node_wins & at_node) {
            NodeDataWide n;
            src.node(cur, n);
            if (COUNT) st.nodes++;
            float t0, t1, t2, t3;
            node_entries(n, slab, tmin, tlimit, t0, t1, t2, t3);
            int next = kWalkDone;   // order is irrelevant for an any-hit search: hit children in slot order
            if (t3 < kMissT) next = n.c3;
            if (t2 < kMissT) { if (next != kWalkDone) stack.push((uint32_t)next); next = n.c2; }
            if (t1 < kMissT) { if (next != kWalkDone) stack.push((uint32_t)next); next = n.c1; }
            if (t0 < kMissT) { if (next != kWalkDone) stack.push((uint32_t)next); next = n.c0; }
            cur = next != kWalkDone ? next : walk_pop(stack);
        }
        if (!node_wins & at_leaf) {
            const uint32_t enc = (uint32_t)(~cur);
            const int first = (int)(enc >> 3);
            const uint32_t more = enc & 7u;
            float4 a, b, c;
            src.tri(first, a, b, c);
            if (COUNT) st.tris++;
            float t, u, v;
            const bool hit = ray_triangle_flat(o, d, vptfp::v3(a.x, a.y, a.z), vptfp::v3(a.w, b.x, b.y), vptfp::v3(b.z, b.w, c.x), tmin, tmax, t, u, v);
            const uint32_t gid = __float_as_uint(c.w);
            if (hit & (!LIGHT | (t < t_e) | ((t == t_e) & (gid < expect)))) { occluded = true; cur = kWalkDone; }
            else if (more) cur = ~(int)((((uint32_t)first + 1u) << 3) | (more - 1u));
            else cur = walk_pop(stack);
        }
    }
    return occluded;
}
// sky_visible / light_visible of k_whole's non-validating instantiations (the interval and direction rules of sky_visible above)
template <bool COUNT>
__device__ __forceinline__ bool sky_visible_vote(const float4* lds_nodes, const float4* lds_tris, V3 o, V3 d, const TravStack& stack, TravStats& st, bool rq) {
    const float tmin = rq ? 0.0001f : 0.00001f, tmax = rq ? 1000000.0f : 1000.0f;
    if (!rq) d = normalize(d);
    LdsSceneSrc src{lds_nodes, lds_tris, false};
    return !lds_occluded_vote<COUNT, false>(src, o, d, tmin, tmax, 0.0f, 0u, stack, st);
}
template <bool COUNT>
__device__ __forceinline__ bool light_visible_vote(const DeviceScene& sc, const float4* lds_nodes, const float4* lds_tris, V3 o, V3 d, uint32_t gid, const TravStack& stack, TravStats& st) {
    const uint32_t slot = sc.tri_slot_of_gid[gid];
    if (slot == 0xffffffffu) return false;  // the sampled light triangle is a sliver: nothing can hit it
    LdsSceneSrc src{lds_nodes, lds_tris, false};
    float4 a, b, c;
    src.tri((int)slot, a, b, c);   // traverse.hpp closest_is: the sampled triangle by its own record first, then the search for anything that beats it
    if (COUNT) st.tris++;
    float t_e, u, v;
    if (!vptfp::ray_triangle(o, d, vptfp::v3(a.x, a.y, a.z), vptfp::v3(a.w, b.x, b.y), vptfp::v3(b.z, b.w, c.x), 0.0001f, 1000000.0f, &t_e, &u, &v)) return false;
    return !lds_occluded_vote<COUNT, true>(src, o, d, 0.0001f, 1000000.0f, t_e, gid, stack, st);
}

// ------------------------------------------------------------------ whole paths in one launch
// The reference's RayGen invocation IS a whole path: one thread runs the bounce loop of its pixel's sample to the end
// (RayGen.slang:66-114).  k_whole is that loop on persistent waves, for scenes whose BVH rides in LDS and which have no media:
// a lane keeps its path in registers from bounce to bounce, and a lane whose path has ended takes the next unstarted sample of the
// batch, so a launch runs until the batch's samples are used up and no path record, queue or counter crosses HBM in between —
// only the per-sample frame sums (ACC) do.  One launch per batch instead of max_depth: what a 1-frame batch at 1080p needs (its
// later bounces are launches of 10^5 paths that do not fill the chip, vpt_render_async).
// A wave alternates two steps, each on full lanes:
//   trace   every lane holds a ray — a survivor of the shade step or a fresh camera ray — and finds its closest hit; misses run the
//           miss shader and end there, hits are parked in a wave-private ring in LDS (hit record + the path's registers);
//   shade   once the ring holds 64 hits: closest-hit shader, the <= 2 shadow queries, contribution, Russian roulette for those 64;
//           survivors keep their lanes for the next trace step.
// Per path this is k_bounce's arithmetic in k_bounce's order (same shade_core, same connect code), and which lane or wave runs a
// sample cannot matter: seeds come from (pixel, frame), results go to ACC[slot].  pathLight of a parked hit waits in ACC[slot].
__device__ __forceinline__ V3 whole_finish(const RenderParams& P, const PathState& ps, uint32_t slot, V3 E, V3 thr_prev, V3 light_prev, const ShadeOut& o) {
    V3 contrib = E * thr_prev;
    if (o.cflags & kCF_Clamp) {
        float lum = dot(contrib, v3(0.212671f, 0.715160f, 0.072169f));
        contrib = contrib * (P.max_luminance / max_(lum, P.max_luminance));
    }
    V3 light = light_prev + contrib;
    if (o.terminated) {  // end of the sample: NaN/Inf guard, frame sum (RayGen.slang:116-128)
        bool ok = !isinf_(light.x) && !isinf_(light.y) && !isinf_(light.z) && !isnan_(light.x) && !isnan_(light.y) && !isnan_(light.z);
        ps.ACC[slot] = ok ? f4(v3s(0.0f) + light, 0.0f) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    return light;
}
// Samples are dealt in tiles of 64 (consecutive pixels of a row).  Wave w of W takes tiles w, w + W, ... for the first `static_rounds` rounds
// without an atomic, and the tiles behind them `chunk_tiles` at a time through ctr->extend_head (the host picks both: vpt_api.hip whole_schedule).
// (Measured and not kept: one-wave blocks, which leave the CU as soon as THEIR paths have ended and so let the next frame's launch in earlier —
// a 1-frame launch at 1080p 542 us against 509 us, and slower with two or three frames in flight too: profiles/r04_whole_lanes.json.)
template <bool COUNT, bool STRICT, bool PLAIN>
__global__ __launch_bounds__(kTraverseBlock, 3) void k_whole(DeviceScene sc, RenderParams P, PathState ps, Counters* ctr, uint32_t n_slots, uint32_t dispatch_base,
                                                            uint32_t static_rounds, uint32_t chunk_tiles) {
    sc.strict_hits = STRICT ? 1u : 0u;
    if (PLAIN) { sc.all_plain = 1u; sc.env_black = 1u; } else sc.all_plain = 0u;
    if (P.dispatch_base_dev) dispatch_base = *P.dispatch_base_dev;   // a replayed graph: the batch's first dispatch index lives in device memory
    const bool rq = (P.flags & VPT_FLAG_RAY_QUERIES) != 0u;
    extern __shared__ __align__(16) unsigned char smem[];
    const TravStack stack = make_stack(smem, sc.stack_overflow);
    float4* lds_nodes = reinterpret_cast<float4*>(smem + kStackDepth * kTraverseBlock * 4);
    float4* lds_tris = lds_nodes + sc.node_count * 8;
    stage_scene<true>(sc, lds_nodes, lds_tris);
    constexpr uint32_t kWaves = kTraverseBlock / 64u;
    __shared__ uint32_t r_slot[kWaves][128], r_prim[kWaves][128], r_inst[kWaves][128];
    __shared__ float r_t[kWaves][128], r_u[kWaves][128], r_v[kWaves][128];
    __shared__ float4 r_ra[kWaves][128], r_rb[kWaves][128], r_rt[kWaves][128];
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t n = n_slots;
    // tile cursor, wave-uniform by construction (kernels_trace.hip k_trace_vote)
    const uint32_t n_waves = gridDim.x * kWaves, n_tiles = (n + 63u) / 64u;
    const uint32_t dyn_first = static_rounds * n_waves;   // tiles from here on are taken through the counter
    uint32_t static_left = static_rounds;
    uint32_t next_static = __builtin_amdgcn_readfirstlane(blockIdx.x * kWaves + wave);
    uint32_t w_next = 0u, w_end = 0u;
    uint32_t last_seen = dyn_first;   // how far the tile counter had got when this wave last took from it
    bool exhausted = false;
    uint32_t hit_head = 0u, hit_count = 0u;   // wave-uniform
    TravStats st, sst; st.nodes = 0; st.tris = 0; sst.nodes = 0; sst.tris = 0;
    uint32_t w_paths = 0u, w_rays = 0u, w_hits0 = 0u, w_alive0 = 0u, w_rays0 = 0u, w_parked = 0u;   // wave totals (uniform); *0: bounce 0 only; parked: hits of later bounces (their pathLight waits in ACC)
    // the lane's path between two steps
    bool has_ray = false;
    uint32_t slot = 0u, rng_s = 0u, depth = 0u;
    bool in_medium = false;
    V3 porg = v3s(0.0f), pdir = v3s(0.0f), thr = v3s(1.0f), lightp = v3s(0.0f);
    float pdf = 1.0f;
    for (;;) {
        // ---- shade: a chunk of parked hits (a partial one only when nothing can be added to it any more: no lane holds a ray here)
        if (hit_count >= 64u || (exhausted && hit_count > 0u)) {
            const uint32_t cnt = hit_count < 64u ? hit_count : 64u;
            const bool valid = lane_id() < cnt;
            uint32_t nrays = 0u;
            bool first = false, alive = false;
            if (valid) {
                const uint32_t q = (hit_head + lane_id()) & 127u;
                ShadeIn in_;
                const float4 a = r_ra[wave][q], b = r_rb[wave][q], t = r_rt[wave][q];
                slot = r_slot[wave][q];
                in_.rng = __float_as_uint(a.w);
                in_.porg = xyz(a); in_.pdir = xyz(b);
                const uint32_t dw = __float_as_uint(b.w);
                in_.depth = dw & 0x7fffffffu; in_.in_medium = (dw >> 31) != 0u;
                in_.thr_prev = xyz(t); in_.prev_pdf = t.w;
                in_.vdepth = 0u; in_.cchan = -1; in_.vol_index = -1; in_.vol_t = 0.0f; in_.atm_comp = -1;
                in_.h = make_float4(r_t[wave][q], r_u[wave][q], r_v[wave][q], __uint_as_float(r_prim[wave][q]));
                in_.inst = r_inst[wave][q];
                first = in_.depth == 0u;   // (only a camera ray has depth 0: the in-medium walk that leaves the depth alone starts behind a refraction)
                ShadeOut o;
                shade_core<false, (int)kShadeTextured>(sc, P, ps, slot, in_, o);   // "all of these hit something"
                // pathLight so far: fetched behind the shader (three registers less across its peak), in flight during the shadow queries
                const V3 light_prev = first ? v3s(0.0f) : xyz(ps.ACC[slot]);
                // connect, inline (RayGen.slang:92-102)
                V3 E = o.emitted;
                constexpr bool kVote = VPT_WHOLE_VOTE != 0 && !STRICT;   // vote-scheduled searches on the tree in LDS (above); the validating instantiations keep the per-lane loops
                if (o.want_sky) {
                    bool vis;
                    if constexpr (kVote) vis = sky_visible_vote<COUNT>(lds_nodes, lds_tris, o.sky_o, o.sky_d, stack, sst, rq);
                    else vis = sky_visible<true, COUNT>(sc, lds_nodes, lds_tris, o.sky_o, o.sky_d, stack, sst, rq);
                    if (vis) E = E + o.csky;
                    nrays++;
                }
                if (o.want_light) {
                    bool vis;
#if VPT_DIAG_NO_LIGHT_SEARCH   // measuring build only (tests/tools/build_variant.py): what the kernel costs WITHOUT its light-identity searches (wrong images)
                    vis = true;
#else
                    if constexpr (kVote) vis = light_visible_vote<COUNT>(sc, lds_nodes, lds_tris, o.light_o, o.light_d, o.light_gid, stack, sst);
                    else vis = light_visible<true, COUNT>(sc, lds_nodes, lds_tris, o.light_o, o.light_d, o.light_gid, stack, sst);
#endif
                    if (vis) E = E + o.clight;
                    nrays++;
                }
                const V3 light = whole_finish(P, ps, slot, E, in_.thr_prev, light_prev, o);
                alive = o.alive;
                if (alive) {
                    has_ray = true;
                    rng_s = o.rng; porg = o.new_o; pdir = o.new_d; depth = o.new_depth; in_medium = o.in_medium; thr = o.thr; pdf = o.new_pdf; lightp = light;
                }
            }
            hit_head += cnt; hit_count -= cnt;
            w_rays += (uint32_t)__popcll(__ballot(nrays >= 1u)) + (uint32_t)__popcll(__ballot(nrays >= 2u));
            w_rays0 += (uint32_t)__popcll(__ballot(first && nrays >= 1u)) + (uint32_t)__popcll(__ballot(first && nrays >= 2u));
            w_alive0 += (uint32_t)__popcll(__ballot(first && alive));
        }
        // ---- refill: free lanes take the next unstarted samples (a second pass when the wave's chunk ran out half-way)
        if (!exhausted) {
#pragma unroll 1
            for (int pass = 0; pass < 2; pass++) {
                const unsigned long long m_free = __ballot(!has_ray);
                if (m_free == 0ull) break;
                if (w_next >= w_end) {   // the next tile(s)
                    uint32_t tile, span = 1u;
                    if (static_left != 0u) { tile = next_static; next_static += n_waves; static_left--; }
                    else {
                        // guided (chunk_tiles bit 8): the chunk shrinks with what is left — by this wave's last look at the counter — so that the waves run dry within a tile of each other
                        uint32_t take = chunk_tiles & 0xffu;
                        if (chunk_tiles & 0x100u) {
                            const uint32_t left_tiles = n_tiles > last_seen ? n_tiles - last_seen : 0u, fair = left_tiles / (2u * n_waves);
                            take = fair < 1u ? 1u : (fair < take ? fair : take);
                        }
                        uint32_t k = 0u;
                        if (lane_id() == 0u) k = atomicAdd(&ctr->extend_head, take);
                        tile = dyn_first + __builtin_amdgcn_readfirstlane(k); span = take;
                        last_seen = tile + take;
                    }
                    if (tile >= n_tiles) exhausted = true;
                    else { w_next = tile * 64u; w_end = (tile + span) * 64u < n ? (tile + span) * 64u : n; }
                }
                if (exhausted) break;
                const uint32_t li = w_next + lanes_below(m_free);
                if (!has_ray && li < w_end) {
                    uint32_t x, y, f;
                    launch_pixel(P, li, dispatch_base, slot, x, y, f);
                    const uint32_t seed = pcg_hash(P.base_seed + dispatch_base + f);  // PathTracer.cpp:139 with an explicit seed
                    Rng r; r.s = y + P.width * x + seed;                              // RayGen.slang:28
                    camera_ray(P, r, x, y, porg, pdir);
                    rng_s = r.s; depth = 0u; in_medium = false; thr = v3s(1.0f); pdf = 1.0f; lightp = v3s(0.0f);
                    has_ray = true;
                }
                const uint32_t want = (uint32_t)__popcll(m_free), left = w_end - w_next;
                w_next += want < left ? want : left;
            }
        }
        if (__ballot(has_ray) == 0ull) {
            if (exhausted && hit_count == 0u) break;
            continue;
        }
        // ---- trace: closest hits; park the hits, finish the misses
        {
            HitRec hr;
            bool hit = false;
            if (has_ray) {
                if constexpr (VPT_WHOLE_VOTE != 0 && !STRICT) { LdsSceneSrc src{lds_nodes, lds_tris, false}; hit = lds_closest_vote<COUNT>(src, porg, normalize(pdir), 0.01f, 100000.0f, stack, hr, st); }
                else hit = trace_any<true, COUNT>(sc, lds_nodes, lds_tris, porg, normalize(pdir), 0.01f, 100000.0f, stack, hr, st);
            }
            const unsigned long long mh = __ballot(has_ray && hit);
            if (has_ray && hit) {
                const uint32_t q = (hit_head + hit_count + lanes_below(mh)) & 127u;
                r_slot[wave][q] = slot; r_t[wave][q] = hr.t; r_u[wave][q] = hr.u; r_v[wave][q] = hr.v; r_prim[wave][q] = hr.gid; r_inst[wave][q] = hr.inst;
                r_ra[wave][q] = f4u(porg, rng_s);
                r_rb[wave][q] = f4u(pdir, depth | (in_medium ? 0x80000000u : 0u));
                r_rt[wave][q] = f4(thr, pdf);
                if (depth != 0u) ps.ACC[slot] = f4(lightp, 0.0f);   // pathLight so far (a camera ray's is 0)
            }
            hit_count += (uint32_t)__popcll(mh);
            w_paths += (uint32_t)__popcll(__ballot(has_ray));
            w_hits0 += (uint32_t)__popcll(__ballot(has_ray && hit && depth == 0u));
            w_parked += (uint32_t)__popcll(__ballot(has_ray && hit && depth != 0u));
            if (has_ray && !hit) {   // Miss.slang; the path ends here
                ShadeIn in_;
                in_.rng = rng_s; in_.porg = porg; in_.pdir = pdir; in_.depth = depth; in_.in_medium = in_medium; in_.thr_prev = thr; in_.prev_pdf = pdf;
                in_.vdepth = 0u; in_.cchan = -1; in_.vol_index = -1; in_.vol_t = 0.0f; in_.atm_comp = -1;
                in_.h = make_float4(-1.0f, 0.0f, 0.0f, 0.0f); in_.inst = 0u;
                ShadeOut o;
                shade_core<false, (int)kShadeMiss>(sc, P, ps, slot, in_, o);
                (void)whole_finish(P, ps, slot, o.emitted, thr, lightp, o);
            }
            has_ray = false;
            // nothing of the lane's path is live across the shade step (hits wait in the ring): say so, or its 17 registers stay allocated through shade_core
            porg = v3s(0.0f); pdir = v3s(0.0f); thr = v3s(1.0f); lightp = v3s(0.0f); pdf = 1.0f; rng_s = 0u; depth = 0u; in_medium = false;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    if (lane_id() == 0) {
        if (w_paths) atomicAdd(&ctr->stat_closest, (unsigned long long)w_paths);
        if (w_rays) atomicAdd(&ctr->stat_shadow, (unsigned long long)w_rays);
        if (w_hits0) atomicAdd(&ctr->stat_primary_hits, (unsigned long long)w_hits0);
        if (w_alive0) atomicAdd(&ctr->stat_primary_alive, (unsigned long long)w_alive0);
        if (w_rays0) atomicAdd(&ctr->stat_primary_rays, (unsigned long long)w_rays0);
        if (w_parked) atomicAdd(&ctr->stat_connect, (unsigned long long)w_parked);   // vpt_stats.connect_paths: here, the hits whose pathLight made the round trip through ACC
    }
    if (COUNT) {
        atomicAdd(&ctr->stat_nodes, (unsigned long long)st.nodes);
        atomicAdd(&ctr->stat_tris, (unsigned long long)st.tris);
        atomicAdd(&ctr->stat_shadow_nodes, (unsigned long long)sst.nodes);
        atomicAdd(&ctr->stat_shadow_tris, (unsigned long long)sst.tris);
    }
}

#if VPT_LAB
// ------------------------------------------------------------------ connect (round 1's stage kernels)
// Per pending path: trace its (<= 2) shadow rays (RTCommon.slang:47-64: closest committed hit), join the
// visible NEE contributions with the emission BEFORE the luminance clamp (RayGen.slang:92-102), add to
// pathLight, and at the end of a sample apply the NaN/Inf guard and add to the frame sum (:116-128).
//
// A block works on tiles of up to kConnectTile paths.  The shadow rays of a tile are first listed in LDS (sky
// rays from the front, light rays from the back, one entry = owning path of the tile), then traced ONE RAY PER
// LANE — a path with two rays does not hold a lane twice as long while its neighbours idle, and waves see one
// ray kind — and the visibility bits go back to the owners through LDS, which add the contributions in the
// reference's order.
constexpr uint32_t kConnectTile = 512;
constexpr uint32_t kConnectScratch = kConnectTile * 4 + kConnectTile * 2 * 2 + kConnectTile * 2 + 16;  // slots, ray list, visibility, counters
__device__ inline uint32_t connect_tile(uint32_t n) { return n >= (1u << 19) ? 512u : n >= (1u << 17) ? 256u : n >= (1u << 15) ? 128u : 64u; }

template <bool LDS_SCENE, bool COUNT, bool STRICT>
__global__ __launch_bounds__(kTraverseBlock, 8) void k_connect(DeviceScene sc, RenderParams P, PathState ps, const uint32_t* cqueue,
                                                           Counters* ctr, uint32_t parity) {
    sc.strict_hits = STRICT ? 1u : 0u;
    extern __shared__ __align__(16) unsigned char smem[];
    const TravStack stack = make_stack(smem, sc.stack_overflow);
    unsigned char* scratch = smem + kStackDepth * kTraverseBlock * 4;
    uint32_t* t_slot = reinterpret_cast<uint32_t*>(scratch);                                   // [kConnectTile]
    uint16_t* t_list = reinterpret_cast<uint16_t*>(scratch + kConnectTile * 4);                // [2 * kConnectTile]
    unsigned char* t_vis = scratch + kConnectTile * 8;                                         // [2 * kConnectTile]
    uint32_t* t_misc = reinterpret_cast<uint32_t*>(scratch + kConnectTile * 10);               // tile base, #sky, #light
    float4* lds_nodes = reinterpret_cast<float4*>(scratch + kConnectScratch);
    float4* lds_tris = lds_nodes + sc.node_count * 8;
    stage_scene<LDS_SCENE>(sc, lds_nodes, lds_tris);
    const uint32_t nf = ctr->connect_front, nb = ctr->connect_back, n = nf + nb;
    const float4* Tprev = ps.T[parity];
    const uint32_t tile = connect_tile(n);
    const uint32_t tid = threadIdx.x;
    constexpr uint32_t kOwn = kConnectTile / kTraverseBlock;  // paths a thread owns per tile
    TravStats st; st.nodes = 0; st.tris = 0;
    while (true) {
        __syncthreads();  // the previous tile is fully consumed
        if (tid == 0) { t_misc[0] = atomicAdd(&ctr->connect_head, tile); t_misc[1] = 0u; t_misc[2] = 0u; }
        __syncthreads();
        const uint32_t base = t_misc[0];
        if (base >= n) break;
        uint32_t slot[kOwn], flags[kOwn];
#pragma unroll
        for (uint32_t q = 0; q < kOwn; q++) {
            const uint32_t j = tid + q * kTraverseBlock, i = base + j;
            const bool valid = j < tile && i < n;
            slot[q] = 0u; flags[q] = 0u;
            if (valid) {
                slot[q] = (i < nf) ? cqueue[i] : cqueue[ps.capacity - 1u - (i - nf)];
                flags[q] = __float_as_uint(ps.CE[slot[q]].w) | 0x80000000u;  // bit 31: this thread owns a path here
                t_slot[j] = slot[q];
            }
            const bool sky = (flags[q] & kCF_Sky) != 0u, light = (flags[q] & kCF_Light) != 0u;
            const uint32_t ps_ = wave_append(sky, &t_misc[1]);
            if (sky) t_list[ps_] = (uint16_t)j;
            const uint32_t pl_ = wave_append(light, &t_misc[2]);
            if (light) t_list[2u * kConnectTile - 1u - pl_] = (uint16_t)j;
        }
        __syncthreads();
        const uint32_t ns = t_misc[1], nr = ns + t_misc[2];
        for (uint32_t r = tid; r < nr; r += kTraverseBlock) {
            if (r < ns) {  // ClosestHit.slang:139, 344-353
                const uint32_t j = t_list[r], sl = t_slot[j];
                float4 so = ps.CSO[sl], sd = ps.CSD[sl];
                t_vis[2u * j] = sky_visible<LDS_SCENE, COUNT>(sc, lds_nodes, lds_tris, xyz(so), v3(so.w, sd.x, sd.y), stack, st, (P.flags & VPT_FLAG_RAY_QUERIES) != 0u) ? 1 : 0;
            } else {       // ClosestHit.slang:171-176, 358-370
                const uint32_t j = t_list[2u * kConnectTile - 1u - (r - ns)], sl = t_slot[j];
                float4 lo = ps.CLO[sl], ld = ps.CLD[sl];
                const uint32_t expect = __float_as_uint(ps.CL[sl].w);
                t_vis[2u * j + 1u] = light_visible<LDS_SCENE, COUNT>(sc, lds_nodes, lds_tris, xyz(lo), v3(lo.w, ld.x, ld.y), expect, stack, st) ? 1 : 0;
            }
        }
        __syncthreads();
#pragma unroll
        for (uint32_t q = 0; q < kOwn; q++) {
            if (flags[q] & 0x80000000u) {
                const uint32_t j = tid + q * kTraverseBlock, sl = slot[q], fl = flags[q];
                V3 E = xyz(ps.CE[sl]);
                if ((fl & kCF_Sky) && t_vis[2u * j]) E = E + xyz(ps.CS[sl]);
                if ((fl & kCF_Light) && t_vis[2u * j + 1u]) E = E + xyz(ps.CL[sl]);
                V3 contrib = E * xyz(Tprev[sl]);  // RayGen.slang:92
                if (fl & kCF_Clamp) {
                    float lum = dot(contrib, v3(0.212671f, 0.715160f, 0.072169f));
                    contrib = contrib * (P.max_luminance / max_(lum, P.max_luminance));
                }
                V3 light = xyz(ps.L[sl]) + contrib;
                if (fl & kCF_Finalize) {
                    bool ok = !isinf_(light.x) && !isinf_(light.y) && !isinf_(light.z) && !isnan_(light.x) && !isnan_(light.y) && !isnan_(light.z);
                    if (P.samples_per_frame == 1) {  // the only finalisation of this slot: 0 + pathLight
                        ps.ACC[sl] = ok ? f4(v3s(0.0f) + light, 0.0f) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    } else if (ok) {
                        float4 acc = ps.ACC[sl]; ps.ACC[sl] = f4(xyz(acc) + light, 0.0f);
                    }
                    light = v3s(0.0f);  // the pixel's next sample of the frame starts from pathLight = 0
                }
                ps.L[sl] = f4(light, 0.0f);
            }
        }
    }
    if (COUNT) {
        atomicAdd(&ctr->stat_shadow_nodes, (unsigned long long)st.nodes);
        atomicAdd(&ctr->stat_shadow_tris, (unsigned long long)st.tris);
    }
}

#endif  // VPT_LAB

// ------------------------------------------------------------------ resolve: running mean, frames applied in order
// frame_base = index of the first dispatch of the batch (== FrameCount when ScreenSplitCount is 1).
// `guard`: queue size word that must be 0 (every path of the batch has finished) — the host enqueues the resolve right
// behind the bounces it expects to be the last ones and only then looks at the counter; if paths were still alive the
// launch does nothing and is repeated after more bounces.
__global__ __launch_bounds__(256) void k_resolve(RenderParams P, PathState ps, float4* image, uint32_t frames, uint32_t frame_base, const uint32_t* guard) {
    if (guard && *guard != 0u) return;
    if (P.dispatch_base_dev) frame_base = *P.dispatch_base_dev;   // a replayed graph (see k_bounce)
    uint32_t sp = blockIdx.x * blockDim.x + threadIdx.x;
    if (sp >= P.shard_pixels) return;
    float4 px = image[sp];
    V3 color = v3(px.x, px.y, px.z);
    if (P.split == 1u) {
        for (uint32_t f = 0; f < frames; f++) {
            V3 acc = xyz(ps.ACC[f * P.shard_pixels + sp]) / (float)P.samples_per_frame;
            uint32_t fc = frame_base + f;
            if (fc > 0) color = lerp(color, acc, 1.0f / (float)(fc + 1u));
            else color = acc;
        }
    } else {
        // split-screen (RayGen.slang:16-25, 143-157): dispatch d touches only the pixels of its chunk; the very
        // first dispatch also copies each rendered pixel into its whole S x S cell ("pixels that aren't rendered")
        const uint32_t S = P.split, y = sp / P.width, x = sp - y * P.width;
        for (uint32_t f = 0; f < frames; f++) {
            uint32_t d = frame_base + f, c = d % (S * S), fc = d / (S * S);
            if (x % S == c % S && y % S == c / S) {
                V3 acc = xyz(ps.ACC[f * P.shard_pixels + sp]) / (float)P.samples_per_frame;
                if (fc > 0) color = lerp(color, acc, 1.0f / (float)(fc + 1u));
                else color = acc;
            } else if (d == 0u) {
                uint32_t ax = x - x % S, ay = y - y % S;
                color = xyz(ps.ACC[f * P.shard_pixels + ay * P.width + ax]) / (float)P.samples_per_frame;
            }
        }
    }
    image[sp] = make_float4(color.x, color.y, color.z, 1.0f);
}

#if VPT_LAB
// Start of a bounce (round 1's stage kernels): fold the statistics of the previous one, reset cursors and the output queue sizes.
__global__ void k_prepare(Counters* ctr, uint32_t parity) {
    ctr->stat_closest += ctr->ray_count[parity];
    ctr->stat_shadow += ctr->shadow_rays;
    ctr->stat_connect += ctr->connect_front + ctr->connect_back;
    ctr->shadow_rays = 0u;
    ctr->extend_head = 0u; ctr->connect_head = 0u; ctr->connect_front = 0u; ctr->connect_back = 0u;
    ctr->ray_count[parity ^ 1u] = 0u;
}
__global__ void k_fold(Counters* ctr) {
    ctr->stat_shadow += ctr->shadow_rays; ctr->shadow_rays = 0u;
    ctr->stat_connect += ctr->connect_front + ctr->connect_back; ctr->connect_front = 0u; ctr->connect_back = 0u;
}

#endif  // VPT_LAB

// shard rows <-> full image
__global__ __launch_bounds__(256) void k_scatter_rows(const float4* gathered, float4* full, uint32_t width, uint32_t height,
                                                      uint32_t shard_count, uint32_t shard_stride_px) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= width * height) return;
    uint32_t y = i / width, x = i - y * width;
    uint32_t r = y % shard_count, ys = y / shard_count;
    full[i] = gathered[(size_t)r * shard_stride_px + (size_t)ys * width + x];
}

// ------------------------------------------------------------------ derived scene tables
__global__ __launch_bounds__(256) void k_precompute_materials(DeviceScene sc, uint32_t flags, MatResolved* out, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const vpt_material& m = sc.materials[i];
    MatResolved r;
    V2 uv; uv.x = 0.0f; uv.y = 0.0f;
    material_resolve(sc, m, uv, flags, r);   // a field whose texture is not 1x1 holds a value nobody reads (its flag stays clear)
    auto one = [&](uint32_t t) { return sc.textures[t].w == 1 && sc.textures[t].h == 1; };
    r.flags = (one(m.base_color_texture) ? kMatBase : 0u) | (one(m.roughness_texture) ? kMatRoughness : 0u) |
              (one(m.metallic_texture) ? kMatMetallic : 0u) | (one(m.emissive_texture) ? kMatEmissive : 0u);
    if ((r.flags & (kMatBase | kMatRoughness | kMatMetallic | kMatEmissive)) == (kMatBase | kMatRoughness | kMatMetallic | kMatEmissive)) r.flags |= kMatAllValues;
    if (one(m.normal_texture)) {
        V4 nm = tex_sample(sc, m.normal_texture, 0.0f, 0.0f);
        r.nmap[0] = nm.x * 2.0f - 1.0f; r.nmap[1] = nm.y * 2.0f - 1.0f; r.nmap[2] = nm.z * 2.0f - 1.0f;
        r.flags |= kMatNormal;
    } else { r.nmap[0] = r.nmap[1] = r.nmap[2] = 0.0f; }
    r.pad1 = r.pad2 = 0.0f;
    r.tex[0] = sc.textures[m.normal_texture]; r.tex[1] = sc.textures[m.base_color_texture]; r.tex[2] = sc.textures[m.roughness_texture];
    r.tex[3] = sc.textures[m.metallic_texture]; r.tex[4] = sc.textures[m.emissive_texture];
    out[i] = r;
}
// One LightSampler per emissive mesh (device_types.hpp): the fields SampleEmissiveTriangle reads through four tables, side by side.
__global__ __launch_bounds__(64) void k_precompute_lights(DeviceScene sc, LightSampler* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= sc.emissive_count) return;
    const EmissiveDesc& em = sc.emissive[i];
    const vpt_material& m = sc.materials[em.material];
    LightSampler ls;
    ls.tri_count = em.tri_count; ls.gid_base = sc.instances[em.instance].tri_offset; ls.tri_base = sc.emissive_tri_offset[i];
    ls.tex = sc.textures[m.emissive_texture];
    ls.uniform = (ls.tex.w == 1 && ls.tex.h == 1) ? 1u : 0u;
    ls.emissive_color[0] = m.emissive_color[0]; ls.emissive_color[1] = m.emissive_color[1]; ls.emissive_color[2] = m.emissive_color[2];
    V4 te = tex_sample(sc, m.emissive_texture, 0.0f, 0.0f);
    ls.radiance[0] = m.emissive_color[0] * te.x; ls.radiance[1] = m.emissive_color[1] * te.y; ls.radiance[2] = m.emissive_color[2] * te.z;
    ls.pad0 = ls.pad1 = 0.0f;
    out[i] = ls;
}
__global__ __launch_bounds__(256) void k_precompute_tri_ng(DeviceScene sc, float4* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= sc.tri_count) return;
    const BvhTri& t = sc.tris[i];
    V3 ng = triangle_ng(sc, sc.instances[t.inst], t.prim);
    out[t.gid] = make_float4(ng.x, ng.y, ng.z, 0.0f);
}
__global__ __launch_bounds__(256) void k_precompute_tri_shade(DeviceScene sc, float4* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= sc.tri_count) return;
    const BvhTri& t = sc.tris[i];
    const InstanceDesc in = sc.instances[t.inst];
    const MeshDesc me = sc.meshes[in.mesh];
    const uint32_t* idx = sc.indices + me.index_offset + t.prim * 3;
    const vpt_vertex* vb = sc.vertices + me.vertex_offset;
    float4* q = out + (size_t)t.gid * 8;
    for (int k = 0; k < 3; k++) {
        const float4* v = reinterpret_cast<const float4*>(vb + idx[k]);
        q[2 * k] = v[0]; q[2 * k + 1] = v[1];
    }
    V3 ng = triangle_ng(sc, in, t.prim);
    q[6] = make_float4(ng.x, ng.y, ng.z, 0.0f);
    q[7] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}
__global__ __launch_bounds__(256) void k_precompute_emissive(DeviceScene sc, EmissiveTri* out, uint32_t total) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    uint32_t k = 0;
    while (k + 1 < sc.emissive_count && sc.emissive_tri_offset[k + 1] <= i) k++;
    EmissiveTri t;
    emissive_tri_compute(sc, sc.emissive[k], i - sc.emissive_tri_offset[k], t);
    out[i] = t;
}
// Shade class of every instance (device_types.hpp kShade*), from the resolved material table.
__global__ __launch_bounds__(256) void k_classify_instances(DeviceScene sc, unsigned char* out, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t mi = sc.instances[i].material;
    const vpt_material& m = sc.materials[mi];
    const MatResolved& r = sc.mat_resolved[mi];
    const bool emissive = (r.flags & kMatEmissive) ? (r.emissive[0] > 0.0f || r.emissive[1] > 0.0f || r.emissive[2] > 0.0f)
                                         : (m.emissive_color[0] != 0.0f || m.emissive_color[1] != 0.0f || m.emissive_color[2] != 0.0f);
    uint32_t c = kShadeTextured;
    if (emissive) c = kShadeEmissive;
    else if (m.transmission > 0.0f) c = kShadeGlass;
    else if ((r.flags & (kMatAllValues | kMatNormal)) == (kMatAllValues | kMatNormal)) c = kShadePlain;   // the promise k_shade_stream<kShadePlain> relies on: no texture of this material is ever sampled
    out[i] = (unsigned char)c;
}
void launch_classify_instances(hipStream_t s, const DeviceScene& sc, unsigned char* out, uint32_t n) {
    if (n) hipLaunchKernelGGL(k_classify_instances, dim3((n + 255) / 256), dim3(256), 0, s, sc, out, n);
}
// (the light table reads materials and textures too: every caller that refreshes one refreshes the other)
void launch_precompute_materials(hipStream_t s, const DeviceScene& sc, uint32_t flags, MatResolved* out, uint32_t n) {
    if (n) hipLaunchKernelGGL(k_precompute_materials, dim3((n + 255) / 256), dim3(256), 0, s, sc, flags, out, n);
    if (sc.emissive_count) hipLaunchKernelGGL(k_precompute_lights, dim3((sc.emissive_count + 63) / 64), dim3(64), 0, s, sc, const_cast<LightSampler*>(sc.lights));
}
void launch_precompute_tri_ng(hipStream_t s, const DeviceScene& sc, float4* out) {
    if (sc.tri_count) hipLaunchKernelGGL(k_precompute_tri_ng, dim3((sc.tri_count + 255) / 256), dim3(256), 0, s, sc, out);
}
void launch_precompute_tri_shade(hipStream_t s, const DeviceScene& sc, float4* out) {
    if (sc.tri_count) hipLaunchKernelGGL(k_precompute_tri_shade, dim3((sc.tri_count + 255) / 256), dim3(256), 0, s, sc, out);
}
void launch_precompute_emissive(hipStream_t s, const DeviceScene& sc, EmissiveTri* out, uint32_t total) {
    if (total) hipLaunchKernelGGL(k_precompute_emissive, dim3((total + 255) / 256), dim3(256), 0, s, sc, out, total);
}

// ------------------------------------------------------------------ launch wrappers
static inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// first == true: bounce 0 of n_slots fresh slots (queue unused); otherwise one fused bounce of queue[parity].
void launch_bounce(hipStream_t s, uint32_t blocks, bool lds_scene, bool count, bool first, const DeviceScene& sc, const RenderParams& P,
                   const PathState& ps, const StreamState& ss, const uint32_t* queue, uint32_t* queue_next, Counters* ctr, uint32_t parity, uint32_t n_slots,
                   uint32_t dispatch_base, uint32_t k3, bool plain) {
    size_t lds = traverse_lds_bytes(sc, lds_scene);
    dim3 g(blocks), b(kTraverseBlock);
    if (plain && lds_scene && !count && !sc.strict_hits && sc.volume_count == 0u && !sc.atm_on && sc.env_black) {   // the scene-class instantiation
        if (first) hipLaunchKernelGGL((k_bounce<true, false, true, false, false, true>), g, b, lds, s, sc, P, ps, ss, queue, queue_next, ctr, parity, n_slots, dispatch_base, k3);
        else hipLaunchKernelGGL((k_bounce<true, false, false, false, false, true>), g, b, lds, s, sc, P, ps, ss, queue, queue_next, ctr, parity, n_slots, dispatch_base, k3);
        return;
    }
#define VPT_LAUNCH_BOUNCE_V(L, C, F, V) do { if (sc.strict_hits) hipLaunchKernelGGL((k_bounce<L, C, F, V, true>), g, b, lds, s, sc, P, ps, ss, queue, queue_next, ctr, parity, n_slots, dispatch_base, k3); \
        else hipLaunchKernelGGL((k_bounce<L, C, F, V, false>), g, b, lds, s, sc, P, ps, ss, queue, queue_next, ctr, parity, n_slots, dispatch_base, k3); } while (0)
#define VPT_LAUNCH_BOUNCE(L, C, F) VPT_LAUNCH_BOUNCE_V(L, C, F, false)
#define VPT_LAUNCH_MEDIA(L, F) hipLaunchKernelGGL((k_bounce<L, false, F, true, false>), g, b, lds, s, sc, P, ps, ss, queue, queue_next, ctr, parity, n_slots, dispatch_base, k3)
    if (sc.volume_count > 0u || sc.atm_on) {  // the media variants carry no traversal counters and read VPT_FLAG_LOCAL_HITS at run time
        if (lds_scene) { if (first) VPT_LAUNCH_MEDIA(true, true); else VPT_LAUNCH_MEDIA(true, false); }
        else { if (first) VPT_LAUNCH_MEDIA(false, true); else VPT_LAUNCH_MEDIA(false, false); }
    } else if (lds_scene) {
        if (count) { if (first) VPT_LAUNCH_BOUNCE(true, true, true); else VPT_LAUNCH_BOUNCE(true, true, false); }
        else { if (first) VPT_LAUNCH_BOUNCE(true, false, true); else VPT_LAUNCH_BOUNCE(true, false, false); }
    } else {   // a tree in memory: the hit rule is read at run time (above), and the visit counters always run (two adds per visit in a kernel
               // that is the slow side of an A/B anyway: one instantiation per bounce kind instead of four)
        if (first) hipLaunchKernelGGL((k_bounce<false, true, true, false, false>), g, b, lds, s, sc, P, ps, ss, queue, queue_next, ctr, parity, n_slots, dispatch_base, k3);
        else hipLaunchKernelGGL((k_bounce<false, true, false, false, false>), g, b, lds, s, sc, P, ps, ss, queue, queue_next, ctr, parity, n_slots, dispatch_base, k3);
    }
#undef VPT_LAUNCH_MEDIA
#undef VPT_LAUNCH_BOUNCE
#undef VPT_LAUNCH_BOUNCE_V
}
// Whole paths in one launch (k_whole): LDS-resident scenes without media, one sample per pixel and frame.
void launch_whole(hipStream_t s, uint32_t blocks, bool count, const DeviceScene& sc, const RenderParams& P, const PathState& ps, Counters* ctr, uint32_t n_slots,
                  uint32_t dispatch_base, bool plain, uint32_t static_rounds, uint32_t chunk_tiles) {
    const size_t lds = traverse_lds_bytes(sc, true);
    const dim3 g(blocks), b(kTraverseBlock);
#define VPT_LW(C, S, PL) hipLaunchKernelGGL((k_whole<C, S, PL>), g, b, lds, s, sc, P, ps, ctr, n_slots, dispatch_base, static_rounds, chunk_tiles)
    if (plain && !count && !sc.strict_hits && sc.env_black) VPT_LW(false, false, true);
    else if (sc.strict_hits) { if (count) VPT_LW(true, true, false); else VPT_LW(false, true, false); }
    else if (count) VPT_LW(true, false, false);
    else VPT_LW(false, false, false);
#undef VPT_LW
}
void launch_finish(hipStream_t s, uint32_t blocks, bool count, const DeviceScene& sc, const RenderParams& P, const PathState& ps, const StreamState& ss, const uint32_t* queue,
                   StreamCounters* sctr, Counters* ctr, uint32_t parity) {
    const size_t lds = kVoteLdsBytes;   // traversal stacks + the LDS copy of the tree top (vote.hpp)
    const dim3 g(blocks), b(kTraverseBlock);
    if (sc.strict_hits) { if (count) hipLaunchKernelGGL((k_finish<true, true>), g, b, lds, s, sc, P, ps, ss, queue, sctr, ctr, parity); else hipLaunchKernelGGL((k_finish<false, true>), g, b, lds, s, sc, P, ps, ss, queue, sctr, ctr, parity); }
    else if (count) hipLaunchKernelGGL((k_finish<true, false>), g, b, lds, s, sc, P, ps, ss, queue, sctr, ctr, parity);
    else hipLaunchKernelGGL((k_finish<false, false>), g, b, lds, s, sc, P, ps, ss, queue, sctr, ctr, parity);
    hipLaunchKernelGGL(k_finish_done, dim3(1), dim3(1), 0, s, sctr, parity);
}
int finish_blocks_per_cu(const DeviceScene& sc) {
    (void)sc;
    int a = 0, b = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, k_finish<false, false>, kTraverseBlock, kVoteLdsBytes);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, k_finish<false, true>, kTraverseBlock, kVoteLdsBytes);
    const int nb = a < b ? a : b;
    return nb > 0 ? nb : 1;
}
int whole_blocks_per_cu(const DeviceScene& sc, bool plain) {
    int nb = 0;
    const size_t lds = traverse_lds_bytes(sc, true);
    if (plain) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_whole<false, false, true>, kTraverseBlock, lds);
    else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_whole<false, false, false>, kTraverseBlock, lds);
    return nb > 0 ? nb : 1;
}
int bounce_blocks_per_cu(bool lds_scene, const DeviceScene& sc, bool plain) {
    int nb = 0;
    size_t lds = traverse_lds_bytes(sc, lds_scene);
    if (lds_scene && plain) {   // the smaller of the two instantiations a batch launches
        int a = 0, b = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, k_bounce<true, false, true, false, false, true>, kTraverseBlock, lds);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, k_bounce<true, false, false, false, false, true>, kTraverseBlock, lds);
        nb = a < b ? a : b;
        return nb > 0 ? nb : 1;
    }
    if (lds_scene) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_bounce<true, false, false, false, false>, kTraverseBlock, lds);
    else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_bounce<false, true, false, false, false>, kTraverseBlock, lds);
    return nb > 0 ? nb : 1;
}
#if VPT_LAB
void launch_raygen(hipStream_t s, const RenderParams& P, const PathState& ps, uint32_t* queue, Counters* ctr, uint32_t n_slots, uint32_t dispatch_base) {
    hipLaunchKernelGGL(k_raygen, dim3(cdiv(n_slots, 256)), dim3(256), 0, s, P, ps, queue, ctr, n_slots, dispatch_base);
}
void launch_prepare(hipStream_t s, Counters* ctr, uint32_t parity) { hipLaunchKernelGGL(k_prepare, dim3(1), dim3(1), 0, s, ctr, parity); }
void launch_fold(hipStream_t s, Counters* ctr) { hipLaunchKernelGGL(k_fold, dim3(1), dim3(1), 0, s, ctr); }
#endif

size_t traverse_lds_bytes(const DeviceScene& sc, bool lds_scene) {
    size_t b = (size_t)kStackDepth * kTraverseBlock * 4;
    if (lds_scene) b += (size_t)sc.node_count * sizeof(BvhNodeWide) + (size_t)sc.tri_count * sizeof(BvhTri);
    return b;
}
#if VPT_LAB
void launch_extend(hipStream_t s, uint32_t blocks, bool lds_scene, bool count, const DeviceScene& sc, const PathState& ps,
                   const uint32_t* queue, Counters* ctr, uint32_t parity) {
    size_t lds = traverse_lds_bytes(sc, lds_scene);
    if (lds_scene) {
        if (count) { if (sc.strict_hits) hipLaunchKernelGGL((k_extend<true, true, true>), dim3(blocks), dim3(kTraverseBlock), lds, s, sc, ps, queue, ctr, parity); else hipLaunchKernelGGL((k_extend<true, true, false>), dim3(blocks), dim3(kTraverseBlock), lds, s, sc, ps, queue, ctr, parity); }
        else { if (sc.strict_hits) hipLaunchKernelGGL((k_extend<true, false, true>), dim3(blocks), dim3(kTraverseBlock), lds, s, sc, ps, queue, ctr, parity); else hipLaunchKernelGGL((k_extend<true, false, false>), dim3(blocks), dim3(kTraverseBlock), lds, s, sc, ps, queue, ctr, parity); }
    } else {
        if (count) { if (sc.strict_hits) hipLaunchKernelGGL((k_extend<false, true, true>), dim3(blocks), dim3(kTraverseBlock), lds, s, sc, ps, queue, ctr, parity); else hipLaunchKernelGGL((k_extend<false, true, false>), dim3(blocks), dim3(kTraverseBlock), lds, s, sc, ps, queue, ctr, parity); }
        else { if (sc.strict_hits) hipLaunchKernelGGL((k_extend<false, false, true>), dim3(blocks), dim3(kTraverseBlock), lds, s, sc, ps, queue, ctr, parity); else hipLaunchKernelGGL((k_extend<false, false, false>), dim3(blocks), dim3(kTraverseBlock), lds, s, sc, ps, queue, ctr, parity); }
    }
}
void launch_connect(hipStream_t s, uint32_t blocks, bool lds_scene, bool count, const DeviceScene& sc, const RenderParams& P,
                    const PathState& ps, const uint32_t* cqueue, Counters* ctr, uint32_t parity) {
    size_t lds = traverse_lds_bytes(sc, lds_scene) + kConnectScratch;
    if (lds_scene) {
        if (count) { if (sc.strict_hits) hipLaunchKernelGGL((k_connect<true, true, true>), dim3(blocks), dim3(kTraverseBlock), lds, s, sc, P, ps, cqueue, ctr, parity); else hipLaunchKernelGGL((k_connect<true, true, false>), dim3(blocks), dim3(kTraverseBlock), lds, s, sc, P, ps, cqueue, ctr, parity); }
        else { if (sc.strict_hits) hipLaunchKernelGGL((k_connect<true, false, true>), dim3(blocks), dim3(kTraverseBlock), lds, s, sc, P, ps, cqueue, ctr, parity); else hipLaunchKernelGGL((k_connect<true, false, false>), dim3(blocks), dim3(kTraverseBlock), lds, s, sc, P, ps, cqueue, ctr, parity); }
    } else {
        if (count) { if (sc.strict_hits) hipLaunchKernelGGL((k_connect<false, true, true>), dim3(blocks), dim3(kTraverseBlock), lds, s, sc, P, ps, cqueue, ctr, parity); else hipLaunchKernelGGL((k_connect<false, true, false>), dim3(blocks), dim3(kTraverseBlock), lds, s, sc, P, ps, cqueue, ctr, parity); }
        else { if (sc.strict_hits) hipLaunchKernelGGL((k_connect<false, false, true>), dim3(blocks), dim3(kTraverseBlock), lds, s, sc, P, ps, cqueue, ctr, parity); else hipLaunchKernelGGL((k_connect<false, false, false>), dim3(blocks), dim3(kTraverseBlock), lds, s, sc, P, ps, cqueue, ctr, parity); }
    }
}
void launch_shade(hipStream_t s, uint32_t blocks, const DeviceScene& sc, const RenderParams& P, const PathState& ps,
                  const uint32_t* queue, uint32_t* queue_next, uint32_t* cqueue, Counters* ctr, uint32_t parity) {
    hipLaunchKernelGGL(k_shade, dim3(blocks), dim3(256), 0, s, sc, P, ps, queue, queue_next, cqueue, ctr, parity);
}
#endif  // VPT_LAB
void launch_resolve(hipStream_t s, const RenderParams& P, const PathState& ps, float* image, uint32_t frames, uint32_t frame_base, const uint32_t* guard) {
    hipLaunchKernelGGL(k_resolve, dim3(cdiv(P.shard_pixels, 256)), dim3(256), 0, s, P, ps, reinterpret_cast<float4*>(image), frames, frame_base, guard);
}
void launch_trace_rays(hipStream_t s, uint32_t blocks, const DeviceScene& sc, const vpt_ray* rays, uint32_t n, vpt_hit* hits) {
    uint32_t g = cdiv(n, kTraverseBlock);
    hipLaunchKernelGGL(k_trace_rays, dim3(g < blocks ? g : blocks), dim3(kTraverseBlock), (size_t)kStackDepth * kTraverseBlock * 4, s, sc, rays, n, hits);
}
void launch_scatter_rows(hipStream_t s, const float* gathered, float* full, uint32_t w, uint32_t h, uint32_t shard_count, uint32_t stride_px) {
    hipLaunchKernelGGL(k_scatter_rows, dim3(cdiv(w * h, 256)), dim3(256), 0, s, reinterpret_cast<const float4*>(gathered),
                       reinterpret_cast<float4*>(full), w, h, shard_count, stride_px);
}
size_t stack_overflow_bytes(uint32_t blocks) { return (size_t)blocks * kTraverseBlock * kStackOverflow * 4; }
#if VPT_LAB
int traverse_blocks_per_cu(bool lds_scene, const DeviceScene& sc) {
    int nb = 0;
    size_t lds = traverse_lds_bytes(sc, lds_scene) + kConnectScratch;
    if (lds_scene) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_connect<true, false, false>, kTraverseBlock, lds);
    else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_connect<false, false, false>, kTraverseBlock, lds);
    return nb > 0 ? nb : 1;
}
int shade_blocks_per_cu() {
    int nb = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_shade, 256, 0);
    return nb > 0 ? nb : 1;
}
#endif

}  // namespace vpt
