// kernels_stream.hip — the staged pipeline's shade / join stages on compact streams (the shadow stage's kernel, k_trace_shadow,
// lives with the other traversal kernels in kernels_trace.hip).
//
//   shade   ClosestHit.slang + Miss.slang + the visibility-independent tail of the bounce loop (shade_core.hpp), one path
//           per lane, persistent.  What a path leaves behind goes into streams by wave-private chunked appends (vote.hpp):
//           its slot into the next ray queue if it lives on, a pending record (emission | flags, the two NEE contributions,
//           the pre-update throughput) if anything has to be joined, and its <= 2 shadow rays into the sky-ray and
//           light-ray streams — so a wave of the shadow stage sees rays of ONE kind, read as coalesced 32-byte records.
//   join    RayGen.slang:92-128: visible NEE contributions joined with the emission BEFORE the luminance clamp, pathLight,
//           NaN guard and frame sum at the end of a sample.
//
// The ray records (origin | rng, direction | depth, throughput | pdf) MOVE with a path's queue entry; its frame sum, pathLight and
// medium stay addressed by slot, and every value is computed per path, so nothing here can change a bit of the image.
#include "kernels.hpp"
#include "shade_core.hpp"
#include "vote.hpp"

namespace vpt {


// ------------------------------------------------------------------ classify: the shade queue, sorted by shade class
// The extend stage leaves one class byte per queue entry (kShade*: miss | plain | textured | glass | emissive; 0xff for a
// hole).  Each block takes tiles of 4096 entries, counts the classes with wave ballots, reserves its part of every class queue
// with ONE atomic per class, and writes the entries out in order: dense queues, one per class, in which a wave of the shade
// stage finds paths of one class only.  The last block to finish lays out the streams the class launches append to.
constexpr uint32_t kClassifyItems = 16, kClassifyTile = 256 * kClassifyItems;

struct ClassQueues { uint32_t* q[kShadeClasses]; };

__global__ __launch_bounds__(256) void k_classify(const uint32_t* queue, const unsigned char* cls, ClassQueues cq, StreamCounters* sc, uint32_t parity, uint32_t shade_waves) {
    __shared__ uint32_t s_cnt[4][kShadeClasses];
    __shared__ uint32_t s_base[4][kShadeClasses];
    const uint32_t n = sc->queue_len[parity].v;
    const uint32_t wave = threadIdx.x >> 6;
    for (uint32_t tile = blockIdx.x * kClassifyTile; tile < n; tile += gridDim.x * kClassifyTile) {
        uint32_t kc[kClassifyItems];
        uint32_t cnt[kShadeClasses];
#pragma unroll
        for (uint32_t c = 0; c < kShadeClasses; c++) cnt[c] = 0u;
#pragma unroll
        for (uint32_t it = 0; it < kClassifyItems; it++) {
            const uint32_t i = tile + it * 256u + threadIdx.x;
            kc[it] = (i < n && queue[i] != kHole) ? (uint32_t)cls[i] : 0xffu;
#pragma unroll
            for (uint32_t c = 0; c < kShadeClasses; c++) cnt[c] += (uint32_t)__popcll(__ballot(kc[it] == c));  // wave-uniform
        }
        __syncthreads();  // the previous tile's s_base is consumed
        if (lane_id() == 0u) {
#pragma unroll
            for (uint32_t c = 0; c < kShadeClasses; c++) s_cnt[wave][c] = cnt[c];
        }
        __syncthreads();
        if (threadIdx.x < kShadeClasses) {
            const uint32_t c = threadIdx.x;
            const uint32_t t0 = s_cnt[0][c], t1 = s_cnt[1][c], t2 = s_cnt[2][c], t3 = s_cnt[3][c], tot = t0 + t1 + t2 + t3;
            const uint32_t b = tot ? atomicAdd(&sc->class_len[c].v, tot) : 0u;
            s_base[0][c] = b; s_base[1][c] = b + t0; s_base[2][c] = b + t0 + t1; s_base[3][c] = b + t0 + t1 + t2;
        }
        __syncthreads();
        uint32_t run[kShadeClasses];
#pragma unroll
        for (uint32_t c = 0; c < kShadeClasses; c++) run[c] = s_base[wave][c];
#pragma unroll
        for (uint32_t it = 0; it < kClassifyItems; it++) {
#pragma unroll
            for (uint32_t c = 0; c < kShadeClasses; c++) {
                const unsigned long long m = __ballot(kc[it] == c);
                if (kc[it] == c) cq.q[c][run[c] + lanes_below(m)] = tile + it * 256u + threadIdx.x;   // the entry's POSITION in the ray queue
                run[c] += (uint32_t)__popcll(m);
            }
        }
    }
    // ---- the last block to get here lays out the class launches' appends (class_len is final then)
    __shared__ uint32_t s_last;
    __syncthreads();
    if (threadIdx.x == 0u) {
        __threadfence();
        s_last = atomicAdd(&sc->classify_done, 1u) == gridDim.x - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (s_last && threadIdx.x == 0u) {
        uint32_t reserved = 0u;  // every stream the shade launches append to starts with the static chunks of the chunked launches
        for (uint32_t c = 0; c < kShadeClasses; c++) {
            const uint32_t nc = atomicAdd(&sc->class_len[c].v, 0u);
            const uint32_t need = (nc + 63u) / 64u, active = need < shade_waves ? need : shade_waves;
            const uint32_t exact = nc < kAppendExactBelow ? 1u : 0u;
            sc->class_active[c] = active; sc->class_exact[c] = exact; sc->class_base[c] = reserved;
            if (!exact) reserved += active * kAppendChunk;
        }
        sc->queue_len[parity ^ 1u].v = reserved; sc->pend_len.v = reserved; sc->sky_len.v = reserved; sc->light_len.v = reserved;
        sc->sky_head.v = 0u; sc->light_head.v = 0u;
        __threadfence();
    }
}

// ------------------------------------------------------------------ shade
#ifndef VPT_SHADE_MIN_BLOCKS
#define VPT_SHADE_MIN_BLOCKS 3   // blocks of four waves per CU the register allocation aims at (168 registers per lane); 2 and 4 measured: profiles/r05_shade_occupancy_ab.log
#endif
template <int CLS>
__global__ __launch_bounds__(256, VPT_SHADE_MIN_BLOCKS) void k_shade_stream(DeviceScene sc, RenderParams P, PathState ps, StreamState ss, const uint32_t* queue, const uint32_t* order,
                                                         uint32_t* queue_next, Counters* ctr, StreamCounters* sctr, uint32_t parity, uint32_t cls) {
    const uint32_t n = sctr->class_len[cls].v;    // this class's queue: dense, written by k_classify
    const uint32_t active = sctr->class_active[cls];   // min(waves of the grid, ceil(n / 64))
    const uint32_t gw = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));   // (said to be uniform: what derives from it — the appenders, the cursor — then lives in scalar registers)
    if (gw >= active) return;                     // this wave owns no chunk of any stream and takes no work
    const uint32_t chunk = fetch_chunk(n);
    WaveAppender a_next, a_pend, a_sky, a_light;
    const bool exact = sctr->class_exact[cls] != 0u;
    const uint32_t first_chunk = sctr->class_base[cls] / kAppendChunk + gw;   // this launch's static chunks follow the earlier launches'
    a_next.init(first_chunk, exact); a_pend.init(first_chunk, exact); a_sky.init(first_chunk, exact); a_light.init(first_chunk, exact);
    uint32_t w_paths = 0u, w_rays = 0u, w_pend = 0u, w_alive = 0u;  // wave totals (uniform)
    // Regrouping (unsorted mode): the wave first looks at the hit records of its next 64 entries and parks their queue positions
    // in two wave-private LDS rings, hits (with the hit record) and misses; the shaders then run on FULL chunks of 64 hits or 64
    // misses, whichever ring has filled up, and the partial chunks are flushed when the input is exhausted.  Holes vanish on the
    // way.  Nothing moves in memory and every path still gets exactly its own records, so the image cannot change; what changes is
    // that misses and holes no longer idle through the closest-hit shader (glass bust: most paths leave the bust into the sky).
    constexpr bool kRegroup = CLS == kShadeAny;
    __shared__ uint32_t r_q[4][128], r_m[4][128];
    __shared__ float4 r_h[4][128];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t hit_head = 0u, hit_count = 0u, miss_head = 0u, miss_count = 0u;   // wave-uniform
    uint32_t pos = gw * 64u, end = pos + 64u;      // the wave's static first 64 entries, then chunks through the cursor
    bool done = false;
    while (true) {
        {
            uint32_t qi = 0u, slot = kHole;
            bool valid = false;
            float4 hrec = make_float4(-1.0f, 0.0f, 0.0f, 0.0f);
            const bool pop_hits = kRegroup && (hit_count >= 64u || (done && hit_count > 0u));
            const bool pop_miss = kRegroup && !pop_hits && (miss_count >= 64u || (done && miss_count > 0u));
            if (pop_hits || pop_miss) {
                const uint32_t cnt = pop_hits ? (hit_count < 64u ? hit_count : 64u) : (miss_count < 64u ? miss_count : 64u);
                valid = lane_id() < cnt;
                if (valid) {
                    if (pop_hits) { const uint32_t q = (hit_head + lane_id()) & 127u; qi = r_q[wave][q]; hrec = r_h[wave][q]; }
                    else qi = r_m[wave][(miss_head + lane_id()) & 127u];
                    slot = queue[qi];
                }
                if (pop_hits) { hit_head += cnt; hit_count -= cnt; } else { miss_head += cnt; miss_count -= cnt; }
            } else if (!done) {
                if (pos >= end || pos >= n) {   // next chunk of the queue
                    if (active * 64u >= n) { done = true; continue; }
                    uint32_t nb = 0u;
                    if (lane_id() == 0u) nb = atomicAdd(&sctr->class_head[cls].v, chunk);
                    pos = active * 64u + __builtin_amdgcn_readfirstlane(nb);
                    end = pos + chunk;
                    if (pos >= n) done = true;
                    continue;
                }
                const uint32_t i = pos + lane_id();
                pos += 64u;
                // sorted mode: `order` holds positions in the ray queue (k_classify); unsorted: the queue is worked through in order
                qi = i < n ? (order ? order[i] : i) : 0u;
                slot = i < n ? queue[qi] : kHole;
                valid = slot != kHole;
                if (valid) hrec = ld_stream(&ss.SH[qi]);
                if (kRegroup) {   // each ring holds < 64 entries here, so 128 slots are enough
                    const bool is_hit = valid && !(hrec.x < 0.0f);
                    const unsigned long long mh = __ballot(is_hit), mm = __ballot(valid && !is_hit);
                    if (is_hit) { const uint32_t q = (hit_head + hit_count + lanes_below(mh)) & 127u; r_q[wave][q] = qi; r_h[wave][q] = hrec; }
                    if (valid && !is_hit) r_m[wave][(miss_head + miss_count + lanes_below(mm)) & 127u] = qi;
                    hit_count += (uint32_t)__popcll(mh); miss_count += (uint32_t)__popcll(mm);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    continue;
                }
            } else {
                break;
            }
            bool alive = false, pending = false, want_sky = false, want_light = false;
            ShadeOut o;
            V3 thr_prev = v3s(0.0f);
            if (valid) {
                const float4 a = ld_stream(&ss.RA[parity][qi]), b = ld_stream(&ss.RB[parity][qi]), t = ld_stream(&ss.RT[parity][qi]);
                ShadeIn in_;
                in_.h = hrec;
                in_.inst = in_.h.x < 0.0f ? 0u : __builtin_nontemporal_load(&ss.SHI[qi]);
                in_.rng = __float_as_uint(a.w);
                in_.porg = xyz(a); in_.pdir = xyz(b);
                const uint32_t dw = __float_as_uint(b.w);
                in_.depth = dw & 0x7fffffffu; in_.in_medium = (dw >> 31) != 0u;
                in_.thr_prev = xyz(t); in_.prev_pdf = t.w;
                in_.vol_index = -1; in_.vol_t = 0.0f; in_.vdepth = 0u; in_.cchan = -1; in_.atm_comp = -1;
                shade_core<false, CLS>(sc, P, ps, slot, in_, o);
                thr_prev = in_.thr_prev;
                alive = o.alive; want_sky = o.want_sky; want_light = o.want_light;
                const bool thr_finite = !isinf_(thr_prev.x) && !isinf_(thr_prev.y) && !isinf_(thr_prev.z) && !isnan_(thr_prev.x) && !isnan_(thr_prev.y) && !isnan_(thr_prev.z);
                // 0 * inf = NaN must still reach pathLight, so a non-finite throughput always goes through the join
                pending = want_sky || want_light || o.terminated || o.emitted.x != 0.0f || o.emitted.y != 0.0f || o.emitted.z != 0.0f || !thr_finite;
            }
            const uint32_t p_next = a_next.append(alive, &sctr->queue_len[parity ^ 1u].v);
            if (alive) {   // the survivor's records move to where its queue entry goes — pathLight included: the join stage then
                           // updates it in stream order (two coalesced streams) instead of a scattered read-modify-write by slot
                queue_next[p_next] = slot;
                st_stream(&ss.RA[parity ^ 1u][p_next], f4u(o.new_o, o.rng));
                st_stream(&ss.RB[parity ^ 1u][p_next], f4u(o.new_d, o.new_depth | (o.in_medium ? 0x80000000u : 0u)));
                st_stream(&ss.RT[parity ^ 1u][p_next], f4(o.thr, o.new_pdf));
                st_stream(&ss.RL[parity ^ 1u][p_next], ld_stream(&ss.RL[parity][qi]));
            }
            const uint32_t p_sky = a_sky.append(want_sky, &sctr->sky_len.v);
            if (want_sky) {
                st_stream(&ss.SKO[p_sky], f4(o.sky_o, o.sky_d.x));
                st_stream(&ss.SKD[p_sky], make_float4(o.sky_d.y, o.sky_d.z, __uint_as_float(0xffffffffu), 0.0f));
            }
            const uint32_t p_light = a_light.append(want_light, &sctr->light_len.v);
            if (want_light) {
                st_stream(&ss.LTO[p_light], f4(o.light_o, o.light_d.x));
                st_stream(&ss.LTD[p_light], make_float4(o.light_d.y, o.light_d.z, __uint_as_float(o.light_gid), 0.0f));
            }
            const uint32_t p_pend = a_pend.append(pending, &sctr->pend_len.v);
            if (pending) {   // where the join finds the path's pathLight (and its slot, in the queue): its entry in the NEXT queue if it
                             // lives on (the pixel's next sample of this frame included: samples_per_frame > 1), else its entry in this one
                const bool lives_on = alive;
                st_stream(&ss.PE[p_pend], f4u(o.emitted, o.cflags | (lives_on ? kCF_Alive : 0u)));
                st_stream(&ss.PS[p_pend], f4u(o.csky, p_sky));
                st_stream(&ss.PL[p_pend], f4u(o.clight, p_light));
                st_stream(&ss.PT[p_pend], f4u(thr_prev, lives_on ? p_next : qi));
            }
            w_paths += (uint32_t)__popcll(__ballot(valid));
            w_alive += (uint32_t)__popcll(__ballot(alive));
            w_rays += (uint32_t)__popcll(__ballot(want_sky)) + (uint32_t)__popcll(__ballot(want_light));
            w_pend += (uint32_t)__popcll(__ballot(pending));
        }
    }
    // the unwritten tails of this wave's last chunks become holes
    for (uint32_t j = lane_id(); j < a_next.tail_count(); j += 64u) queue_next[a_next.tail_first() + j] = kHole;
    for (uint32_t j = lane_id(); j < a_pend.tail_count(); j += 64u) ss.PT[a_pend.tail_first() + j] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(kHole));
    for (uint32_t j = lane_id(); j < a_sky.tail_count(); j += 64u) ss.SKD[a_sky.tail_first() + j] = make_float4(0.0f, 0.0f, __uint_as_float(kRayHole), 0.0f);
    for (uint32_t j = lane_id(); j < a_light.tail_count(); j += 64u) ss.LTD[a_light.tail_first() + j] = make_float4(0.0f, 0.0f, __uint_as_float(kRayHole), 0.0f);
    if (lane_id() == 0u) {
        if (w_alive) atomicAdd(&sctr->alive[parity ^ 1u].v, w_alive);
        if (w_paths) atomicAdd(&ctr->stat_closest, (unsigned long long)w_paths);
        if (w_rays) atomicAdd(&ctr->stat_shadow, (unsigned long long)w_rays);
        if (w_pend) atomicAdd(&ctr->stat_connect, (unsigned long long)w_pend);
    }
}

// ------------------------------------------------------------------ join
#ifndef VPT_JOIN_UNROLL   // (-D override: tests/tools/build_variant.py)
#define VPT_JOIN_UNROLL 4
#endif
constexpr int kJoinUnroll = VPT_JOIN_UNROLL;
__global__ __launch_bounds__(256) void k_join(RenderParams P, PathState ps, StreamState ss, const StreamCounters* sctr, const uint32_t* queue, const uint32_t* queue_next,
                                              uint32_t parity) {
    const uint32_t n = sctr->pend_len.v;
    // Six dependent fetches per entry (pending record -> its flags -> the two NEE records -> their visibility bytes -> pathLight): a
    // thread walks kJoinUnroll entries in step, phase by phase, so that four of each are in flight per lane (memory-level
    // parallelism is what bounds this kernel: it does no arithmetic to speak of).  Entries are independent: a path has one pending
    // entry per bounce, and the scattered frame sums are touched once per slot per bounce.
    constexpr int K = kJoinUnroll;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t j0 = blockIdx.x * blockDim.x + threadIdx.x; j0 < n; j0 += stride * K) {
        float4 pt[K], pe[K], s4[K], l4[K], lp[K];
        uint32_t pos[K], fl[K], sl[K];
        bool on[K], vs[K], vl[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            const uint32_t j = j0 + (uint32_t)k * stride;
            on[k] = j < n;
            pt[k] = on[k] ? ss.PT[j] : make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(kHole));
        }
#pragma unroll
        for (int k = 0; k < K; k++) {
            pos[k] = __float_as_uint(pt[k].w);   // the path's entry in the next queue (it lives on) or in this one (it ended)
            on[k] = on[k] && pos[k] != kHole;
            pe[k] = on[k] ? ss.PE[j0 + (uint32_t)k * stride] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
#pragma unroll
        for (int k = 0; k < K; k++) {
            const uint32_t j = j0 + (uint32_t)k * stride;
            fl[k] = __float_as_uint(pe[k].w);
            vs[k] = on[k] && (fl[k] & kCF_Sky); vl[k] = on[k] && (fl[k] & kCF_Light);
            if (vs[k]) s4[k] = ss.PS[j];
            if (vl[k]) l4[k] = ss.PL[j];
            // pathLight travels with the path's queue entry (k_shade_stream): pending entries and queue entries were appended by the
            // same waves in the same order, so these accesses are streams too
            if (on[k]) {
                lp[k] = (fl[k] & kCF_Alive) ? ss.RL[parity ^ 1u][pos[k]] : ss.RL[parity][pos[k]];
                if (fl[k] & kCF_Finalize) sl[k] = (fl[k] & kCF_Alive) ? queue_next[pos[k]] : queue[pos[k]];
            }
        }
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (vs[k]) vs[k] = ss.vis_sky[__float_as_uint(s4[k].w)] != 0;
            if (vl[k]) vl[k] = ss.vis_light[__float_as_uint(l4[k].w)] != 0;
        }
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (!on[k]) continue;
            V3 E = xyz(pe[k]);
            if (vs[k]) E = E + xyz(s4[k]);
            if (vl[k]) E = E + xyz(l4[k]);
            V3 contrib = E * xyz(pt[k]);  // RayGen.slang:92
            if (fl[k] & kCF_Clamp) {
                float lum = dot(contrib, v3(0.212671f, 0.715160f, 0.072169f));
                contrib = contrib * (P.max_luminance / max_(lum, P.max_luminance));
            }
            V3 light = xyz(lp[k]) + contrib;
            if (fl[k] & kCF_Finalize) {  // end of a sample: NaN/Inf guard, frame sum (RayGen.slang:116-128)
                bool ok = !isinf_(light.x) && !isinf_(light.y) && !isinf_(light.z) && !isnan_(light.x) && !isnan_(light.y) && !isnan_(light.z);
                if (P.samples_per_frame == 1) {  // the only finalisation of this slot: 0 + pathLight
                    ps.ACC[sl[k]] = ok ? f4(v3s(0.0f) + light, 0.0f) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                } else if (ok) {
                    float4 acc = ps.ACC[sl[k]]; ps.ACC[sl[k]] = f4(xyz(acc) + light, 0.0f);
                }
                light = v3s(0.0f);  // the pixel's next sample of the frame (samples_per_frame > 1) starts from pathLight = 0
            }
            if (fl[k] & kCF_Alive) ss.RL[parity ^ 1u][pos[k]] = f4(light, 0.0f);   // a path that ended has no use for it any more
        }
    }
}

// Camera rays of a batch as the first ray queue and its records (RayGen.slang:12-64; the staged pipeline runs bounce 0 through
// the same stages as every other bounce).
__global__ __launch_bounds__(256) void k_raygen_stream(RenderParams P, PathState ps, StreamState ss, uint32_t* queue, uint32_t n_slots, uint32_t dispatch_base, uint32_t media) {
    const uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= n_slots) return;
    if (P.dispatch_base_dev) dispatch_base = *P.dispatch_base_dev;   // a replayed graph (kernels_path.hip k_bounce)
    uint32_t slot, x, y, f;
    launch_pixel(P, li, dispatch_base, slot, x, y, f);
    const uint32_t seed = pcg_hash(P.base_seed + dispatch_base + f);  // PathTracer.cpp:139 with an explicit seed
    Rng r; r.s = y + P.width * x + seed;                              // RayGen.slang:28
    V3 o, d;
    camera_ray(P, r, x, y, o, d);
    ss.RA[0][li] = f4u(o, r.s);
    ss.RB[0][li] = f4u(d, 0u);
    ss.RT[0][li] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);  // pathThroughput = 1, payload.PDF = 1
    ss.RL[0][li] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // pathLight = 0
    if (P.samples_per_frame > 1) { ps.ACC[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); ps.sidx[slot] = 0u; }
    if (media) { ps.vdepth[slot] = 0u; ps.cchan[slot] = -1; }   // payload.VolumeDepth, payload.ColorChannel (RayGen.slang:54-62)
    queue[li] = slot;
}
void launch_raygen_stream(hipStream_t s, const RenderParams& P, const PathState& ps, const StreamState& ss, uint32_t* queue, uint32_t n_slots, uint32_t dispatch_base, bool media) {
    hipLaunchKernelGGL(k_raygen_stream, dim3((n_slots + 255u) / 256u), dim3(256), 0, s, P, ps, ss, queue, n_slots, dispatch_base, media ? 1u : 0u);
}

// Start of a batch: the ray queue raygen filled (samples [0, n_first) of the batch's n_total; the rest are started by refills).
__global__ void k_stream_begin(StreamCounters* sc, uint32_t n_first, uint32_t n_total) {
    sc->queue_len[0].v = n_first; sc->alive[0].v = n_first;
    sc->queue_len[1].v = 0u; sc->alive[1].v = 0u;
    sc->refill_next = n_first; sc->refill_total = n_total; sc->refill_count = 0u;
}

// ---- Path regeneration by refill (vpt_config.resident_frames = K: at most cap = K frames of paths are resident in a batch of F > K).
// The reference's RayGen thread starts its pixel's next sample when a path has ended (RayGen.slang:28-33).  Here that happens at the
// QUEUE level: once the shade stage of a bounce has written the next ray queue, the room the ended paths left (cap - live paths) is
// filled with the next unstarted samples of the batch — a contiguous run of sample ids, so a block of coherent camera rays appended
// behind the survivors, written by the camera-ray code of k_raygen_stream.  The shade kernels know nothing of it.  (Round 4 regenerated
// per LANE inside the shade stage: the fresh camera rays were scattered through the queue in runs of a few entries — incoherent for the
// extend stage — and every shade instantiation carried the hook: -4 ... -19 % on the streams, profiles/r04_frames_sweep.json.)
// A sample's seed depends on (pixel, frame) and its result lands in ACC[its slot]: which launch starts it cannot matter.
__global__ void k_refill_plan(StreamCounters* sc, uint32_t parity_next, uint32_t cap) {
    const uint32_t alive = sc->alive[parity_next].v, len = sc->queue_len[parity_next].v;
    const uint32_t left = sc->refill_total - sc->refill_next;
    uint32_t add = alive < cap ? cap - alive : 0u;
    if (add > left) add = left;
    sc->refill_entry = len; sc->refill_first = sc->refill_next; sc->refill_count = add;
    sc->refill_next += add;
    sc->queue_len[parity_next].v = len + add; sc->alive[parity_next].v = alive + add;
}
__global__ __launch_bounds__(256) void k_refill_stream(RenderParams P, PathState ps, StreamState ss, uint32_t* queue, const StreamCounters* sc, uint32_t parity_next, uint32_t dispatch_base) {
    const uint32_t count = sc->refill_count, e0 = sc->refill_entry, s0 = sc->refill_first;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < count; j += gridDim.x * blockDim.x) {
        const uint32_t slot = s0 + j, e = e0 + j;
        uint32_t x, y, f;
        pixel_of_slot(P, slot, x, y, f);
        const uint32_t seed = pcg_hash(P.base_seed + dispatch_base + f);  // PathTracer.cpp:139 with an explicit seed
        Rng r; r.s = y + P.width * x + seed;                              // RayGen.slang:28
        V3 o, d;
        camera_ray(P, r, x, y, o, d);
        st_stream(&ss.RA[parity_next][e], f4u(o, r.s));
        st_stream(&ss.RB[parity_next][e], f4u(d, 0u));
        st_stream(&ss.RT[parity_next][e], make_float4(1.0f, 1.0f, 1.0f, 1.0f));
        st_stream(&ss.RL[parity_next][e], make_float4(0.0f, 0.0f, 0.0f, 0.0f));
        if (P.samples_per_frame > 1) { ps.ACC[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); ps.sidx[slot] = 0u; }
        queue[e] = slot;
    }
}
void launch_refill(hipStream_t s, uint32_t blocks, const RenderParams& P, const PathState& ps, const StreamState& ss, uint32_t* queue_next, StreamCounters* sc, uint32_t parity_next,
                   uint32_t cap, uint32_t dispatch_base) {
    hipLaunchKernelGGL(k_refill_plan, dim3(1), dim3(1), 0, s, sc, parity_next, cap);
    hipLaunchKernelGGL(k_refill_stream, dim3(blocks), dim3(256), 0, s, P, ps, ss, queue_next, sc, parity_next, dispatch_base);
}
// Unsorted mode: the whole ray queue is "class 0" of one shade launch running the general code (k_shade_stream<kShadeAny>).
__global__ void k_layout_single(StreamCounters* sc, uint32_t parity, uint32_t shade_waves) {
    const uint32_t n = sc->queue_len[parity].v;
    const uint32_t need = (n + 63u) / 64u, active = need < shade_waves ? need : shade_waves;
    const uint32_t exact = n < kAppendExactBelow ? 1u : 0u;
    sc->class_len[0].v = n; sc->class_active[0] = active; sc->class_exact[0] = exact; sc->class_base[0] = 0u;
    const uint32_t reserved = exact ? 0u : active * kAppendChunk;
    sc->queue_len[parity ^ 1u].v = reserved; sc->pend_len.v = reserved; sc->sky_len.v = reserved; sc->light_len.v = reserved;
    sc->sky_head.v = 0u; sc->light_head.v = 0u;
}

// Start of a bounce: cursors and class queue lengths to zero (the stream lengths are set by k_classify's last block).
__global__ void k_prepare_stream(StreamCounters* sc, uint32_t parity) {
    sc->alive[parity ^ 1u].v = 0u;
    sc->extend_head.v = 0u; sc->shade_head.v = 0u;   // (shade_head: the distance stage's cursor in media batches; the shadow-ray cursors are reset where the streams are laid out: the previous bounce's shadow kernels may still run)
    for (uint32_t c = 0; c < kShadeClasses; c++) { sc->class_len[c].v = 0u; sc->class_head[c].v = 0u; }
    sc->classify_done = 0u;
}

// ------------------------------------------------------------------ launch
void launch_stream_begin(hipStream_t s, StreamCounters* sc, uint32_t n_first, uint32_t n_total) { hipLaunchKernelGGL(k_stream_begin, dim3(1), dim3(1), 0, s, sc, n_first, n_total); }
void launch_prepare_stream(hipStream_t s, StreamCounters* sc, uint32_t parity) { hipLaunchKernelGGL(k_prepare_stream, dim3(1), dim3(1), 0, s, sc, parity); }
void launch_classify(hipStream_t s, const uint32_t* queue, const unsigned char* cls, uint32_t* const* class_queue, StreamCounters* sc, uint32_t parity,
                     uint32_t max_entries, uint32_t shade_waves) {
    ClassQueues cq;
    for (uint32_t c = 0; c < kShadeClasses; c++) cq.q[c] = class_queue[c];
    uint32_t blocks = (max_entries + kClassifyTile - 1u) / kClassifyTile;
    blocks = blocks < 1u ? 1u : (blocks > 2048u ? 2048u : blocks);
    hipLaunchKernelGGL(k_classify, dim3(blocks), dim3(256), 0, s, queue, cls, cq, sc, parity, shade_waves);
}
void launch_layout_single(hipStream_t s, StreamCounters* sc, uint32_t parity, uint32_t shade_waves) {
    hipLaunchKernelGGL(k_layout_single, dim3(1), dim3(1), 0, s, sc, parity, shade_waves);
}
void launch_shade_stream(hipStream_t s, uint32_t blocks, uint32_t cls, bool sorted, const DeviceScene& sc, const RenderParams& P, const PathState& ps, const StreamState& ss,
                         const uint32_t* queue, const uint32_t* order, uint32_t* queue_next, Counters* ctr, StreamCounters* sctr, uint32_t parity) {
    const dim3 g(blocks), b(256);
    if (!sorted) hipLaunchKernelGGL((k_shade_stream<kShadeAny>), g, b, 0, s, sc, P, ps, ss, queue, nullptr, queue_next, ctr, sctr, parity, 0u);
    else if (cls == kShadeMiss) hipLaunchKernelGGL((k_shade_stream<(int)kShadeMiss>), g, b, 0, s, sc, P, ps, ss, queue, order, queue_next, ctr, sctr, parity, cls);
    else if (cls == kShadePlain) hipLaunchKernelGGL((k_shade_stream<(int)kShadePlain>), g, b, 0, s, sc, P, ps, ss, queue, order, queue_next, ctr, sctr, parity, cls);
    else hipLaunchKernelGGL((k_shade_stream<(int)kShadeTextured>), g, b, 0, s, sc, P, ps, ss, queue, order, queue_next, ctr, sctr, parity, cls);  // textured, glass, emissive: the general hit code
}
void launch_join(hipStream_t s, uint32_t blocks, const RenderParams& P, const PathState& ps, const StreamState& ss, const StreamCounters* sctr, const uint32_t* queue,
                 const uint32_t* queue_next, uint32_t parity) {
    hipLaunchKernelGGL(k_join, dim3(blocks), dim3(256), 0, s, P, ps, ss, sctr, queue, queue_next, parity);
}
int join_blocks_per_cu() {   // a streaming kernel: as many waves as fit (its grid-stride loop takes any grid)
    int nb = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_join, 256, 0);
    return nb > 0 ? nb : 1;
}
int shade_stream_blocks_per_cu() {
    int nb = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_shade_stream<kShadeAny>, 256, 0);
    return nb > 0 ? nb : 1;
}
}  // namespace vpt
