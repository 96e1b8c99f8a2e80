// kernels_media.hip — participating media (box volumes, atmosphere) on the staged pipeline's streams, for scenes whose BVH lives
// in memory (RayGen.slang:162-470, Volume.slang:190-288; the LDS-resident scenes keep the fused per-bounce kernel k_bounce<VOL>).
//
// A bounce with media has data dependencies the surface-only stages do not: the scatter decision needs the distance to the
// geometry (its own closest-hit query, RTCommon.slang:86-117), and the transmittance of a NEE sample is TRACKED — it draws random
// numbers — only once the sample's shadow ray is known to be clear, so everything drawn after it (throughput, roulette, the next
// camera sample) depends on two visibility bits.  Per bounce, on one HIP stream:
//
//   distance   k_trace_vote (closest) on the ray queue with the payload direction as it is, [1e-5, 1e6]       -> SH.x
//   scatter    k_media_scatter: ScatteredInVolume (free flight per box, delta-tracked atmosphere collision)   -> MS (per queue entry)
//   extend     k_trace_vote (closest), normalised direction, [0.01, 1e5]                                      -> SH, SHI
//   shade      k_shade_media: shade_core<VOL> (scatter event | closest-hit | miss shader) up to the visibility tests;
//              shadow rays into the sky / light streams, everything the tail needs into MP (per queue entry)
//   shadow     k_trace_shadow x 2 (a light ray that may also count a clean miss as visible carries that in LTD.w)
//   tail       k_media_tail: transmittance draws of the visible samples, contribution, throughput, roulette, frame sum,
//              survivors appended to the next queue with their records (RA, RB, RT, RL)
//
// Every value is computed by the functions the fused kernel calls (shade_core.hpp, volume.hpp, atmosphere.hpp) in the same
// order on the same random stream, so the image is bit-identical to the fused pipeline's and to the oracle's.
#include "kernels.hpp"
#include "shade_core.hpp"
#include "vote.hpp"

namespace vpt {

namespace {
// MP[3].w
constexpr uint32_t kMF_Sky = 1u, kMF_Light = 2u, kMF_SkyAdd = 4u, kMF_LightAdd = 8u, kMF_SkyKindShift = 4u, kMF_LightKind = 64u, kMF_InMedium = 128u,
                   kMF_Aborted = 256u;

// the wave's next 64 queue entries: its static first 64, then chunks through the cursor (as k_shade_stream)
struct WaveCursor {
    uint32_t pos, end, n, active, chunk;
    bool done;
    __device__ __forceinline__ void init(uint32_t gw, uint32_t n_, uint32_t active_) {
        n = n_; active = active_; chunk = fetch_chunk(n_); pos = gw * 64u; end = pos + 64u; done = false;
    }
    // returns false when the queue is exhausted; otherwise `i` is this lane's entry (may be >= n: no entry)
    __device__ __forceinline__ bool next(uint32_t* head, uint32_t& i) {
        while (pos >= end || pos >= n) {
            if (done || active * 64u >= n) { done = true; return false; }
            uint32_t nb = 0u;
            if (lane_id() == 0u) nb = atomicAdd(head, chunk);
            pos = active * 64u + __builtin_amdgcn_readfirstlane(nb);
            end = pos + chunk;
            if (pos >= n) { done = true; return false; }
        }
        i = pos + lane_id();
        pos += 64u;
        return true;
    }
};
}  // namespace

// ------------------------------------------------------------------ scatter decision (RayGen.slang:76-88, 162-262)
__global__ __launch_bounds__(256) void k_media_scatter(DeviceScene sc, PathState ps, StreamState ss, MediaState ms, const uint32_t* queue, const StreamCounters* sctr,
                                                       uint32_t parity) {
    const uint32_t n = sctr->queue_len[parity].v;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t slot = queue[i];
        if (slot == kHole) continue;
        const float4 a = ss.RA[parity][i], b = ss.RB[parity][i];
        const V3 porg = xyz(a), pdir = xyz(b);
        const uint32_t depth = __float_as_uint(b.w) & 0x7fffffffu;
        int cchan = sc.atm_on ? ps.cchan[slot] : -1;
        // RayGen.slang:76-84: a path whose origin is below the planet's surface leaves the loop at once
        const bool aborted = sc.atm_on && atmosphere_height(sc, porg) < 0.0f;
        int vol_index = -1, atm_comp = -1;
        float vol_t = 0.0f;
        Rng vr; vr.s = __float_as_uint(a.w);
        if (!aborted) {   // ScatteredInVolume against GetDistanceToGeometry (the distance stage: payload direction as is, TMin 1e-5, TMax 1e6)
            const float t = ss.SH[i].x;
            int cc;
            vol_index = scattered_in_media(sc, porg, pdir, vr, t < 0.0f ? -1.0f : t, (float)depth, cchan, vol_t, atm_comp, cc);
            if (vol_index == -2) cchan = cc;   // the path now tracks this colour channel only (:242-247)
        }
        ms.MS[i] = make_float4(__int_as_float(vol_index), vol_t, __uint_as_float((uint32_t)(atm_comp + 1) | (aborted ? 16u : 0u) | ((uint32_t)(cchan + 1) << 8)),
                               __uint_as_float(vr.s));
    }
}

// ------------------------------------------------------------------ shade: everything up to the visibility tests
__global__ __launch_bounds__(256, 3) void k_shade_media(DeviceScene sc, RenderParams P, PathState ps, StreamState ss, MediaState ms, const uint32_t* queue, Counters* ctr,
                                                        StreamCounters* sctr, uint32_t parity) {
    const uint32_t n = sctr->class_len[0].v, active = sctr->class_active[0];
    const uint32_t gw = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (gw >= active) return;
    const bool exact = sctr->class_exact[0] != 0u;
    WaveAppender a_sky, a_light;
    a_sky.init(gw, exact); a_light.init(gw, exact);
    WaveCursor cur; cur.init(gw, n, active);
    uint32_t w_paths = 0u, w_rays = 0u;
    uint32_t i;
    while (cur.next(&sctr->class_head[0].v, i)) {
        const uint32_t slot = i < n ? queue[i] : kHole;
        const bool valid = slot != kHole;
        ShadeOut o;
        bool aborted = false;
        o.want_sky = false; o.want_light = false;
        if (valid) {
            const float4 a = ld_stream(&ss.RA[parity][i]), b = ld_stream(&ss.RB[parity][i]), t = ld_stream(&ss.RT[parity][i]), m = ms.MS[i];
            ShadeIn in_;
            in_.porg = xyz(a); in_.pdir = xyz(b);
            const uint32_t dw = __float_as_uint(b.w);
            in_.depth = dw & 0x7fffffffu; in_.in_medium = (dw >> 31) != 0u;
            in_.thr_prev = xyz(t); in_.prev_pdf = t.w;
            in_.vdepth = ps.vdepth[slot];
            const uint32_t mw = __float_as_uint(m.z);
            in_.vol_index = __float_as_int(m.x); in_.vol_t = m.y; in_.atm_comp = (int)(mw & 15u) - 1; in_.cchan = (int)(mw >> 8) - 1;
            in_.rng = __float_as_uint(m.w);
            aborted = (mw & 16u) != 0u;
            const bool traced = !aborted && in_.vol_index == -1;   // only then did the fused kernel trace the extension ray at all
            in_.h = traced ? ld_stream(&ss.SH[i]) : make_float4(-1.0f, 0.0f, 0.0f, 0.0f);
            in_.inst = (traced && !(in_.h.x < 0.0f)) ? __builtin_nontemporal_load(&ss.SHI[i]) : 0u;
            if (aborted) {   // the loop is left before anything happens: nothing is added, nothing drawn (k_bounce does the same)
                o.emitted = v3s(0.0f); o.csky = v3s(0.0f); o.clight = v3s(0.0f);
                o.rng = in_.rng; o.new_depth = in_.depth; o.new_o = in_.porg; o.new_d = in_.pdir; o.new_pdf = in_.prev_pdf; o.bxdf = v3s(1.0f);
                o.in_medium = in_.in_medium; o.vdepth = in_.vdepth; o.cchan = in_.cchan; o.light_gid = 0xffffffffu; o.light_miss_ok = false;
                o.sky_add = false; o.light_add = false; o.sky_kind = 0; o.light_kind = 0;
                o.sky_f = o.sky_rgb = o.light_f = o.light_rgb = v3s(0.0f); o.sky_w = o.sky_mis = o.light_w = o.light_mis = 1.0f; o.sky_tdepth = o.light_tdepth = 0.0f;
            } else {
                shade_core<true>(sc, P, ps, slot, in_, o);
            }
        }
        const uint32_t p_sky = a_sky.append(o.want_sky, &sctr->sky_len.v);
        if (o.want_sky) {
            st_stream(&ss.SKO[p_sky], f4(o.sky_o, o.sky_d.x));
            st_stream(&ss.SKD[p_sky], make_float4(o.sky_d.y, o.sky_d.z, __uint_as_float(0xffffffffu), 0.0f));
        }
        const uint32_t p_light = a_light.append(o.want_light, &sctr->light_len.v);
        if (o.want_light) {   // LTD.w = 1: the sample also counts as visible when the ray hits nothing at all (RayGen.slang:296-299)
            st_stream(&ss.LTO[p_light], f4(o.light_o, o.light_d.x));
            st_stream(&ss.LTD[p_light], make_float4(o.light_d.y, o.light_d.z, __uint_as_float(o.light_gid), o.light_miss_ok ? 1.0f : 0.0f));
        }
        if (valid) {
            const uint32_t fl = (o.want_sky ? kMF_Sky : 0u) | (o.want_light ? kMF_Light : 0u) | (o.sky_add ? kMF_SkyAdd : 0u) | (o.light_add ? kMF_LightAdd : 0u) |
                                ((uint32_t)o.sky_kind << kMF_SkyKindShift) | (o.light_kind ? kMF_LightKind : 0u) | (o.in_medium ? kMF_InMedium : 0u) | (aborted ? kMF_Aborted : 0u);
            ms.MP[0][i] = f4u(o.emitted, o.rng);
            ms.MP[1][i] = f4u(o.new_o, o.new_depth);
            ms.MP[2][i] = f4(o.new_d, o.new_pdf);
            ms.MP[3][i] = f4u(o.bxdf, fl);
            if (o.want_sky) { ms.MP[4][i] = f4(o.sky_f, o.sky_w); ms.MP[5][i] = f4(o.sky_rgb, o.sky_mis); }
            if (o.want_light) { ms.MP[6][i] = f4(o.light_f, o.light_w); ms.MP[7][i] = f4(o.light_rgb, o.light_mis); }
            ms.MP[8][i] = make_float4(o.sky_tdepth, o.light_tdepth, __uint_as_float(o.vdepth), __int_as_float(o.cchan));
            ms.MP[9][i] = make_float4(__uint_as_float(p_sky), __uint_as_float(p_light), 0.0f, 0.0f);
        }
        w_paths += (uint32_t)__popcll(__ballot(valid));
        w_rays += (uint32_t)__popcll(__ballot(o.want_sky)) + (uint32_t)__popcll(__ballot(o.want_light));
    }
    for (uint32_t j = lane_id(); j < a_sky.tail_count(); j += 64u) ss.SKD[a_sky.tail_first() + j] = make_float4(0.0f, 0.0f, __uint_as_float(kRayHole), 0.0f);
    for (uint32_t j = lane_id(); j < a_light.tail_count(); j += 64u) ss.LTD[a_light.tail_first() + j] = make_float4(0.0f, 0.0f, __uint_as_float(kRayHole), 0.0f);
    if (lane_id() == 0u) {
        if (w_paths) atomicAdd(&ctr->stat_closest, (unsigned long long)w_paths);
        if (w_rays) atomicAdd(&ctr->stat_shadow, (unsigned long long)w_rays);
    }
}

// ------------------------------------------------------------------ tail: what depends on the visibility of the NEE samples
__global__ __launch_bounds__(256, 3) void k_media_tail(DeviceScene sc, RenderParams P, PathState ps, StreamState ss, MediaState ms, const uint32_t* queue, uint32_t* queue_next,
                                                       Counters* ctr, StreamCounters* sctr, uint32_t parity) {
    const uint32_t n = sctr->class_len[0].v, active = sctr->class_active[1];
    const uint32_t gw = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (gw >= active) return;
    const bool exact = sctr->class_exact[0] != 0u;
    WaveAppender a_next;
    a_next.init(gw, exact);
    WaveCursor cur; cur.init(gw, n, active);
    uint32_t w_alive = 0u, w_pend = 0u;
    uint32_t i;
    while (cur.next(&sctr->class_head[1].v, i)) {
        const uint32_t slot = i < n ? queue[i] : kHole;
        const bool valid = slot != kHole;
        bool alive = false;
        ShadeOut o;
        V3 light = v3s(0.0f);
        if (valid) {
            const float4 m0 = ms.MP[0][i], m1 = ms.MP[1][i], m2 = ms.MP[2][i], m3 = ms.MP[3][i], m8 = ms.MP[8][i], m9 = ms.MP[9][i];
            const uint32_t fl = __float_as_uint(m3.w);
            const bool aborted = (fl & kMF_Aborted) != 0u;
            o.emitted = xyz(m0); o.rng = __float_as_uint(m0.w);
            o.new_o = xyz(m1); o.new_depth = __float_as_uint(m1.w);
            o.new_d = xyz(m2); o.new_pdf = m2.w;
            o.bxdf = xyz(m3); o.in_medium = (fl & kMF_InMedium) != 0u;
            o.sky_tdepth = m8.x; o.light_tdepth = m8.y; o.vdepth = __float_as_uint(m8.z); o.cchan = __float_as_int(m8.w);
            const V3 thr_prev = xyz(ld_stream(&ss.RT[parity][i]));
            const V3 light_prev = xyz(ld_stream(&ss.RL[parity][i]));
            V3 E = o.emitted;
            if (fl & kMF_Sky) {   // the sky term is assembled now: its transmittance draws come after the visibility test
                const uint32_t p_sky = __float_as_uint(m9.x);
                if (ss.vis_sky[p_sky]) {
                    const float4 so = ss.SKO[p_sky], sd = ss.SKD[p_sky], m4 = ms.MP[4][i], m5 = ms.MP[5][i];
                    const V3 sky_o = xyz(so), sky_d = v3(so.w, sd.x, sd.y), sky_f = xyz(m4), sky_rgb = xyz(m5);
                    const float sky_w = m4.w, sky_mis = m5.w;
                    const int sky_kind = (int)((fl >> kMF_SkyKindShift) & 3u);
                    Rng tr_rng; tr_rng.s = o.rng;
                    V3 csky;
                    if (sky_kind == 2) {        // RayGen.slang:405-424: (phase * T_atm * T_boxes) * (sun / pdf)
                        V3 tr = atmosphere_transmittance(sc, tr_rng, sky_o, sky_d, o.cchan);
                        tr = tr * volumes_transmittance(sc, tr_rng, sky_o, sky_d, o.sky_tdepth);
                        csky = (sky_f * tr) * (sky_rgb / sky_w);
                    } else {
                        V3 tr = v3s(volumes_transmittance(sc, tr_rng, o.new_o, sky_d, o.sky_tdepth));  // from the new origin (ClosestHit.slang:332-349, RayGen.slang:325-343)
                        if (sc.atm_on) tr = nee_atmosphere_transmittance(sc, tr_rng, tr, o.new_o, sky_d, o.cchan);
                        if (sky_kind == 0) csky = ((sky_f * tr) * sky_rgb / sky_w) * sky_mis;
                        else csky = ((tr * sky_f) * (sky_rgb / sky_w)) * sky_mis;
                    }
                    o.rng = tr_rng.s;
                    if (fl & kMF_SkyAdd) E = E + csky;
                }
            }
            if (fl & kMF_Light) {
                const uint32_t p_light = __float_as_uint(m9.y);
                if (ss.vis_light[p_light]) {  // ClosestHit.slang:361-370, RayGen.slang:348-361: the light term with the box transmittance
                    const float4 lo = ss.LTO[p_light], ld = ss.LTD[p_light], m6 = ms.MP[6][i], m7 = ms.MP[7][i];
                    const V3 light_d = v3(lo.w, ld.x, ld.y), light_f = xyz(m6), light_rgb = xyz(m7);
                    const float light_w = m6.w, light_mis = m7.w;
                    Rng tr_rng; tr_rng.s = o.rng;
                    const V3 tr = v3s(volumes_transmittance(sc, tr_rng, o.new_o, light_d, o.light_tdepth));
                    o.rng = tr_rng.s;
                    const V3 cl = (fl & kMF_LightKind) == 0u ? ((light_f * tr) * light_rgb / light_w) * light_mis : ((tr * light_f) * (light_rgb / light_w)) * light_mis;
                    if (fl & kMF_LightAdd) E = E + cl;
                }
            }
            const int fin_chan = o.cchan;  // the channel this sample is accumulated in (RayGen.slang:118-128)
            shade_tail_media(P, ps, slot, thr_prev, aborted, o);
            V3 contrib = E * thr_prev;
            if (o.cflags & kCF_Clamp) {
                float lum = dot(contrib, v3(0.212671f, 0.715160f, 0.072169f));
                contrib = contrib * (P.max_luminance / max_(lum, P.max_luminance));
            }
            light = light_prev + contrib;
            if (aborted) light = light_prev;  // the loop was left before anything was added
            if (o.terminated) {  // end of a sample: NaN/Inf guard, frame sum (RayGen.slang:116-128)
                bool ok = !isinf_(light.x) && !isinf_(light.y) && !isinf_(light.z) && !isnan_(light.x) && !isnan_(light.y) && !isnan_(light.z);
                if (fin_chan != -1) light = v3(fin_chan == 0 ? light.x : 0.0f, fin_chan == 1 ? light.y : 0.0f, fin_chan == 2 ? light.z : 0.0f);
                if (P.samples_per_frame == 1) {  // the only finalisation of the slot: 0 + pathLight
                    ps.ACC[slot] = ok ? f4(v3s(0.0f) + light, 0.0f) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                } else if (ok) {                 // k_raygen_stream zeroed the frame sum
                    float4 acc = ps.ACC[slot]; ps.ACC[slot] = f4(xyz(acc) + light, 0.0f);
                }
                light = v3s(0.0f);
            }
            alive = o.alive;
            if (alive) { ps.vdepth[slot] = o.vdepth; if (sc.atm_on) ps.cchan[slot] = o.cchan; }
        }
        const uint32_t pn = a_next.append(alive, &sctr->queue_len[parity ^ 1u].v);
        if (alive) {
            queue_next[pn] = slot;
            st_stream(&ss.RA[parity ^ 1u][pn], f4u(o.new_o, o.rng));
            st_stream(&ss.RB[parity ^ 1u][pn], f4u(o.new_d, o.new_depth | (o.in_medium ? 0x80000000u : 0u)));
            st_stream(&ss.RT[parity ^ 1u][pn], f4(o.thr, o.new_pdf));
            st_stream(&ss.RL[parity ^ 1u][pn], f4(light, 0.0f));
        }
        w_alive += (uint32_t)__popcll(__ballot(alive));
        w_pend += (uint32_t)__popcll(__ballot(valid));
    }
    for (uint32_t j = lane_id(); j < a_next.tail_count(); j += 64u) queue_next[a_next.tail_first() + j] = kHole;
    if (lane_id() == 0u) {
        if (w_alive) atomicAdd(&sctr->alive[parity ^ 1u].v, w_alive);
        if (w_pend) atomicAdd(&ctr->stat_connect, (unsigned long long)w_pend);
    }
}

// Stream lengths of a media bounce: the shade launch appends to the sky / light ray streams, the tail launch to the next queue.
__global__ void k_layout_media(StreamCounters* sc, uint32_t parity, uint32_t shade_waves, uint32_t tail_waves) {
    const uint32_t n = sc->queue_len[parity].v;
    const uint32_t need = (n + 63u) / 64u, a_s = need < shade_waves ? need : shade_waves, a_t = need < tail_waves ? need : tail_waves;
    const uint32_t exact = n < kAppendExactBelow ? 1u : 0u;
    sc->class_len[0].v = n; sc->class_exact[0] = exact; sc->class_base[0] = 0u;
    sc->class_active[0] = a_s; sc->class_active[1] = a_t;
    sc->sky_len.v = exact ? 0u : a_s * kAppendChunk; sc->light_len.v = exact ? 0u : a_s * kAppendChunk;
    sc->queue_len[parity ^ 1u].v = exact ? 0u : a_t * kAppendChunk;
    sc->pend_len.v = 0u;
    sc->sky_head.v = 0u; sc->light_head.v = 0u;
}

void launch_media_scatter(hipStream_t s, uint32_t blocks, const DeviceScene& sc, const PathState& ps, const StreamState& ss, const MediaState& ms, const uint32_t* queue,
                          const StreamCounters* sctr, uint32_t parity) {
    hipLaunchKernelGGL(k_media_scatter, dim3(blocks), dim3(256), 0, s, sc, ps, ss, ms, queue, sctr, parity);
}
void launch_layout_media(hipStream_t s, StreamCounters* sctr, uint32_t parity, uint32_t shade_waves, uint32_t tail_waves) {
    hipLaunchKernelGGL(k_layout_media, dim3(1), dim3(1), 0, s, sctr, parity, shade_waves, tail_waves);
}
void launch_shade_media(hipStream_t s, uint32_t blocks, const DeviceScene& sc, const RenderParams& P, const PathState& ps, const StreamState& ss, const MediaState& ms,
                        const uint32_t* queue, Counters* ctr, StreamCounters* sctr, uint32_t parity) {
    hipLaunchKernelGGL(k_shade_media, dim3(blocks), dim3(256), 0, s, sc, P, ps, ss, ms, queue, ctr, sctr, parity);
}
void launch_media_tail(hipStream_t s, uint32_t blocks, const DeviceScene& sc, const RenderParams& P, const PathState& ps, const StreamState& ss, const MediaState& ms,
                       const uint32_t* queue, uint32_t* queue_next, Counters* ctr, StreamCounters* sctr, uint32_t parity) {
    hipLaunchKernelGGL(k_media_tail, dim3(blocks), dim3(256), 0, s, sc, P, ps, ss, ms, queue, queue_next, ctr, sctr, parity);
}
int shade_media_blocks_per_cu() {
    int nb = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_shade_media, 256, 0);
    return nb > 0 ? nb : 1;
}
int media_tail_blocks_per_cu() {
    int nb = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_media_tail, 256, 0);
    return nb > 0 ? nb : 1;
}

}  // namespace vpt
