// shade_core.hpp — the miss / closest-hit shader and the visibility-independent tail of the reference's bounce loop for ONE
// path, on values held in registers (ClosestHit.slang:20-378, Miss.slang:8-77, RayGen.slang:104-113, and the volume /
// atmosphere scatter events of RayGen.slang:265-470).  Shared by the fused per-bounce kernels (kernels_path.hip) and the
// staged shade stage (kernels_path.hip k_shade, kernels_stream.hip); the callers own every load / store of path records.
#pragma once
#include "shading.hpp"
#include "volume.hpp"
#include "atmosphere.hpp"
#include "wave.hpp"

namespace vpt {

// ------------------------------------------------------------------ small helpers
__device__ inline float4 f4(V3 v, float w) { return make_float4(v.x, v.y, v.z, w); }
__device__ inline float4 f4u(V3 v, uint32_t w) { return make_float4(v.x, v.y, v.z, __uint_as_float(w)); }
__device__ inline V3 xyz(float4 v) { return v3(v.x, v.y, v.z); }
__device__ inline void pixel_of_slot(const RenderParams& P, uint32_t slot, uint32_t& x, uint32_t& y, uint32_t& f) {
    f = slot / P.shard_pixels;
    uint32_t sp = slot - f * P.shard_pixels;
    uint32_t ys = sp / P.width;
    x = sp - ys * P.width;
    y = P.shard_rank + P.shard_count * ys;
}

// Launch index -> (slot, pixel, dispatch-in-batch).  With ScreenSplitCount S == 1 a dispatch covers the whole shard
// and the launch index IS the slot.  With S > 1 dispatch k of the batch covers the pixels of chunk
// c = (dispatch_base + k) % S^2 only: LaunchID * S + (c % S, c / S)  (RayGen.slang:16-25); its launch grid is
// exactly the in-bounds part, and P.launch_off[k] is where it starts in the batch-wide launch index space.
__device__ inline void launch_pixel(const RenderParams& P, uint32_t li, uint32_t dispatch_base, uint32_t& slot, uint32_t& x, uint32_t& y, uint32_t& f) {
    if (P.split == 1u) { slot = li; pixel_of_slot(P, li, x, y, f); return; }
    uint32_t k = 0;
    while (P.launch_off[k + 1] <= li) k++;
    uint32_t r = li - P.launch_off[k];
    uint32_t c = (dispatch_base + k) % (P.split * P.split);
    uint32_t cx = c % P.split, cy = c / P.split;
    uint32_t lw = (P.width - cx + P.split - 1u) / P.split;
    uint32_t ly = r / lw, lx = r - ly * lw;
    x = lx * P.split + cx; y = ly * P.split + cy; f = k;
    slot = k * P.shard_pixels + y * P.width + x;
}

struct ShadeIn {
    uint32_t rng;
    V3 porg, pdir;      // payload.Origin / payload.Direction
    uint32_t depth;     // payload.Depth
    bool in_medium;     // payload.InMedium
    V3 thr_prev;        // pathThroughput before this bounce
    float prev_pdf;     // payload.PDF of the previous bounce
    float4 h;           // hit record t,u,v | PrimitiveIndex (t < 0: miss)
    uint32_t inst;      // InstanceIndex
    int vol_index;      // >= 0: the path scattered in this box before reaching the geometry; -2: in the atmosphere (media kernels only)
    float vol_t;        //       at this distance along payload.Direction
    uint32_t vdepth;    // payload.VolumeDepth
    int cchan;          // payload.ColorChannel (for an atmosphere event: the channel the collision was sampled for)
    int atm_comp;       // atmosphere event: 0 Rayleigh, 1 Mie, 2 ozone
};
struct ShadeOut {
    bool alive, terminated, want_sky, want_light, in_medium;
    bool light_miss_ok;  // volume NEE compares a MISS as "hit (0, 0)" (RayGen.slang:296-299): visible if nothing is hit and the sample is triangle 0
    uint32_t rng, new_depth, cflags, light_gid, vdepth;
    V3 new_o, new_d, thr;
    float new_pdf;
    V3 emitted, csky, clight, sky_o, sky_d, light_o, light_d;
    // media kernels only: shade_core<true> stops before the throughput / roulette tail, because with an atmosphere the
    // sky sample's transmittance is tracked (draws random numbers) only once the shadow ray is known to be clear
    V3 bxdf;            // payload.BxDF
    V3 sky_f, sky_rgb;  // sky NEE ingredients: BSDF or colour*phase towards the sample, sample radiance
    float sky_tdepth, sky_w, sky_mis;  // rayDepth argument of the box transmittance, sample pdf, MIS weight (1 for the atmosphere event)
    V3 light_f, light_rgb;           // the same for the emissive-mesh sample (its transmittance may draw too: heterogeneous boxes)
    float light_tdepth, light_w, light_mis;
    int light_kind;                  // 0 surface, 1 box scatter event
    bool light_add;
    int sky_kind;       // 0 surface, 1 box scatter event, 2 atmosphere scatter event (the three expressions differ in association)
    bool sky_add;       // false: trace and track (the draws count) but add nothing (ozone collision)
    int cchan;          // payload.ColorChannel after this bounce
};

// The miss / closest-hit shader and the visibility-independent tail of the bounce loop for ONE path, on
// values held in registers (the callers own every load/store of the path records).
// CLS: what the caller knows about every path it hands over (device_types.hpp kShade*): kShadeMiss — none hit anything;
// kShadePlain — all hit a material whose textures are all 1x1 (the texture code drops out; results are those of the general
// code, which would take the same branches); any other hit class — all hit something; kShadeAny — nothing.
template <bool VOL, int CLS = kShadeAny>
__device__ __forceinline__ void shade_core(const DeviceScene& sc, const RenderParams& P, const PathState& ps, uint32_t slot,
                                           const ShadeIn& in_, ShadeOut& out) {
    bool alive = false, want_sky = false, want_light = false, light_miss_ok = false;
    // Without USE_RAY_QUERIES (RTCommon.slang:64-84) the light-identity compare of an emissive-mesh NEE sample reads a payload word nothing has
    // written: pinned as "never equal" (oracle.cpp does_ray_intersect), i.e. the sample is drawn (its random numbers count) and never visible.
    // Its shadow ray is therefore not even queued: no result of it could matter.
    const bool light_rays = (P.flags & VPT_FLAG_RAY_QUERIES) != 0u;
    if (VOL) {
        out.sky_add = true; out.sky_kind = 0; out.sky_tdepth = 0.0f; out.sky_w = 1.0f; out.sky_mis = 1.0f; out.sky_f = v3s(0.0f); out.sky_rgb = v3s(0.0f);
        out.light_add = true; out.light_kind = 0; out.light_tdepth = 0.0f; out.light_w = 1.0f; out.light_mis = 1.0f; out.light_f = v3s(0.0f); out.light_rgb = v3s(0.0f);
    }
    // media whose transmittance is TRACKED (random draws) rather than evaluated: every unobscured NEE sample is tracked, used or not
    const bool tracked = VOL && (sc.atm_on || sc.hetero);
    uint32_t vdepth = VOL ? in_.vdepth : 0u;
    const float4 h = in_.h;
    Rng rng; rng.s = in_.rng;
    const V3 porg = in_.porg, pdir = in_.pdir;
    const uint32_t depth = in_.depth;
    bool in_medium = in_.in_medium;
    const V3 thr_prev = in_.thr_prev;
    const float prev_pdf = in_.prev_pdf;
    V3 emitted = v3s(0.0f), csky = v3s(0.0f), clight = v3s(0.0f);
    V3 sky_o = v3s(0.0f), sky_d = v3s(0.0f), light_o = v3s(0.0f), light_d = v3s(0.0f);
    uint32_t light_gid = 0xffffffffu;
    V3 new_o = porg, new_d = pdir, bxdf = v3s(1.0f);
    float new_pdf = prev_pdf;
    uint32_t new_depth = depth;
    if (VOL && in_.vol_index >= 0) {
        // ---- EvaluateVolumeScatteringEvent, RayGen.slang:265-380 (no atmosphere, no temperature grid)
        const vpt_volume& v = sc.volumes[in_.vol_index];
        new_o = porg + pdir * in_.vol_t;
        emitted = ld3(v.emissive_color) + temperature_emission(sc, v, rng, new_o);  // RayGen.slang:268
        V3 to_sky = v3s(0.0f); V4 sky = v4(0.0f, 0.0f, 0.0f, 0.0f);
        if (P.flags & VPT_FLAG_SKY_MIS) {
            sample_sky(sc, P, rng, to_sky, sky);
            sky.x *= P.sky_intensity; sky.y *= P.sky_intensity; sky.z *= P.sky_intensity;
        }
        V3 to_light = v3s(0.0f); V4 lc = v4(0.0f, 0.0f, 0.0f, 0.0f);
        if (P.flags & VPT_FLAG_MESH_MIS) sample_emissive(sc, rng, new_o, to_light, lc, light_gid);
        const V3 nd = volume_scatter_direction(sc.phase, v, pdir, rng, vdepth);
        const float ph = volume_phase(sc.phase, v, pdir, nd, vdepth);
        // NEE: the shadow rays start AT the scatter point (no offset); the sky term is assembled after the visibility test
        if ((P.flags & VPT_FLAG_SKY_MIS) && sky.w > 0.0f) {
            float ps_ = volume_phase(sc.phase, v, pdir, to_sky, vdepth);
            if (ps_ > 0.0f || tracked) {  // tracked transmittance draws whenever the sample is unobscured (:316-343)
                out.sky_f = ld3(v.color) * ps_; out.sky_tdepth = (float)vdepth;
                out.sky_rgb = v3(sky.x, sky.y, sky.z); out.sky_w = sky.w; out.sky_mis = power_heuristics(sky.w, ps_); out.sky_kind = 1;
                out.sky_add = ps_ > 0.0f;
                want_sky = true; sky_o = new_o; sky_d = to_sky;
            }
        }
        if ((P.flags & VPT_FLAG_MESH_MIS) && lc.w > 0.0f && light_rays) {
            float pl = volume_phase(sc.phase, v, pdir, to_light, vdepth);
            if (pl > 0.0f || tracked) {
                out.light_f = ld3(v.color) * pl; out.light_tdepth = (float)(vdepth + 1u);  // :353 passes VolumeDepth + 1
                out.light_rgb = v3(lc.x, lc.y, lc.z); out.light_w = lc.w; out.light_mis = power_heuristics(lc.w, pl); out.light_kind = 1;
                out.light_add = pl > 0.0f;
                want_light = true; light_o = new_o; light_d = to_light;
                light_miss_ok = light_gid == 0u;
            }
        }
        new_d = nd;
        bxdf = ld3(v.color) * ph; new_pdf = ph;
        new_depth = depth + 1u;
        vdepth = vdepth + 1u;
    } else if (VOL && in_.vol_index == -2) {
        // ---- EvaluateAtmosphereScatteringEvent, RayGen.slang:382-470
        const int comp = in_.atm_comp;
        new_o = porg + in_.vol_t * pdir;
        V3 nd;
        if (comp == 0) nd = sample_rayleigh(rng, pdir);
        else if (comp == 1) nd = sample_hg(rng, pdir, 0.85f);
        else nd = pdir;
        if (P.flags & VPT_FLAG_SKY_MIS) {
            V3 to_sky; V4 cp;
            sample_sky(sc, P, rng, to_sky, cp);
            cp.x *= P.sky_intensity; cp.y *= P.sky_intensity; cp.z *= P.sky_intensity;
            // the sun term needs the shadow ray first: its transmittance is tracked only when the ray is clear (:398-409)
            out.sky_rgb = v3(cp.x, cp.y, cp.z); out.sky_w = cp.w; out.sky_mis = 1.0f; out.sky_kind = 2;
            out.sky_tdepth = (float)vdepth;
            sky_o = new_o; sky_d = to_sky;
            if (comp == 0) {
                out.sky_f = v3s(rayleigh_phase(pdir, to_sky)); want_sky = true;
                bxdf = v3s(rayleigh_phase(pdir, nd)); new_pdf = rayleigh_phase(pdir, nd);
            } else if (comp == 1) {
                out.sky_f = v3s(phase_hg(pdir, to_sky, 0.85f)); want_sky = true;
                float att = VPT_C_MIE_ABSORPTION / VPT_C_MIE;
                bxdf = v3s(phase_hg(pdir, nd, 0.85f) * (1.0f - att)); new_pdf = phase_hg(pdir, nd, 0.85f);
            } else {
                // ozone only absorbs; the reference still traces the shadow ray and tracks the transmittance (random draws)
                out.sky_f = v3s(0.0f); want_sky = true; out.sky_add = false;
                bxdf = v3s(0.0f); new_pdf = 1.0f;
            }
        } else {
            if (comp == 0) { bxdf = v3s(rayleigh_phase(pdir, nd)); new_pdf = rayleigh_phase(pdir, nd); }
            else {  // Mie AND ozone (`componentHit == 0 ... else`, :455-466)
                float att = VPT_C_MIE_ABSORPTION / VPT_C_MIE;
                bxdf = v3s(phase_mie(pdir, nd) * att); new_pdf = phase_hg(pdir, nd, 0.85f);
            }
        }
        new_d = nd;
        new_depth = depth + 1u;
    } else if (VOL && sc.atm_on && h.x < 0.0f) {
        new_depth = kMaxDepthMarker;  // Miss.slang:11-14: with an atmosphere the sky is in-scattered sunlight only
    } else if (CLS == kShadeMiss || (CLS == kShadeAny && h.x < 0.0f)) {
        // ---- Miss.slang:8-77
        V4 cp = v4(0.0f, 0.0f, 0.0f, 1.0f);
        if (((P.flags & VPT_FLAG_SHOW_ENV_DIRECTLY) || depth > 0) && sc.env_black) {
            cp = v4(0.0f, 0.0f, 0.0f, 0.0f);  // an all-zero env map returns exactly 0 (pdf included) for any direction
        } else if ((P.flags & VPT_FLAG_SHOW_ENV_DIRECTLY) || depth > 0) {
            V3 d = rotate_sc(pdir, v3(1.0f, 0.0f, 0.0f), P.sky_rot[4], P.sky_rot[5]);   // rotate(.., -(sky_altitude / 180 * pi))
            d = rotate_sc(d, v3(0.0f, 1.0f, 0.0f), P.sky_rot[6], P.sky_rot[7]);         // rotate(.., -(sky_azimuth / 180 * pi))
            V2 uv = direction_to_uv(d);
            cp = env_sample(sc, uv.x, uv.y);
        }
        emitted = v3(cp.x, cp.y, cp.z) * P.sky_intensity;
        if (P.flags & VPT_FLAG_FURNACE) emitted = v3s(1.0f);
        if ((P.flags & VPT_FLAG_SKY_MIS) && depth > 0) emitted = emitted * power_heuristics(prev_pdf, cp.w);
        new_depth = kMaxDepthMarker;  // payload.BxDF / PDF stay stale; the path ends here
    } else {
        // ---- ClosestHit.slang:20-378
        V3 rd = normalize(pdir);  // WorldRayDirection()
        uint32_t inst_id = in_.inst;
        const InstanceDesc& in = sc.instances[inst_id];
        const vpt_material& mat = sc.materials[in.material];
        MatResolved mr = sc.mat_resolved[in.material];
        if (CLS == (int)kShadePlain || sc.all_plain) mr.flags = 63u;   // what the class (or the scene: k_bounce<PLAIN>) promises, as a compile-time fact: no texel fetch is compiled in
        const bool geo_only = (P.flags & VPT_FLAG_GEOMETRY_NORMALS) != 0;
        SurfaceFrame s;
        surface_geom(sc, s, in, __float_as_uint(h.w), h.y, h.z, rd, geo_only);
        MatTaps taps;
        material_issue(sc.texels, mr, s.uv, geo_only, taps);   // every texel this hit needs, in flight together
        surface_frame(s, rd, geo_only, mr, taps.normal);
        Bsdf bs; V3 mcol; float mdens, maniso, arot;
        bsdf_init(bs, sc, mat, mr, taps, s.inside, P.flags, mcol, mdens, maniso, arot);
        bool is_light = bs.emissive.x > 0.0f || bs.emissive.y > 0.0f || bs.emissive.z > 0.0f;
        rotate_tangents(s, mr.rot_sin, mr.rot_cos);  // AnisotropyRotation has no texture: the table entry is always valid
        bool scattered = false;
        if (in_medium) {  // :80-116
            float pm_aniso = ps.maniso[slot];
            if (pm_aniso != 1.0f) {
                float4 m = ps.M[slot];
                float gd = length(porg - s.pos);
                float sd = -log_(rng.uf()) / m.w;
                if (sd < gd) {
                    new_o = porg + (sd * pdir);
                    new_d = sample_hg(rng, pdir, pm_aniso);
                    bxdf = xyz(m);     // payload.BxDF = MediumColor; PDF stays stale, depth unchanged
                    scattered = true;
                }
            }
        }
        if (!scattered) {
            // sky NEE sample (:125-148) — 3 draws
            V3 to_sky = v3s(0.0f); V4 sky = v4(0.0f, 0.0f, 0.0f, 0.0f);
            if (P.flags & VPT_FLAG_SKY_MIS) {
                if (VOL) sample_sky(sc, P, rng, to_sky, sky); else sample_env(sc, P, rng, to_sky, sky);
                sky.x *= P.sky_intensity; sky.y *= P.sky_intensity; sky.z *= P.sky_intensity;  // applied twice upstream (quirk 1)
            }
            // emissive-mesh NEE sample (:155-184) — 4 draws unless this is an emitter
            V3 to_light = v3s(0.0f); V4 lc = v4(0.0f, 0.0f, 0.0f, 0.0f);
            if ((P.flags & VPT_FLAG_MESH_MIS) && !is_light) sample_emissive(sc, rng, s.pos, to_light, lc, light_gid);
            // BSDF sampling (:190-201; Material.slang:94-165)
            V3 V = s.world_to_tangent(normalize(-rd));
            V3 H = ggx_sample(rng, V, bs.ax, bs.ay);
            float Fs = bs.fresnel(dot(V, H));
            float x1 = rng.uf();
            V3 L; bool refr = false;
            if (x1 < bs.pm) { L = normalize(reflect(-V, H)); }
            else if (x1 < bs.pm + bs.pd) {
                if (rng.uf() < Fs) L = normalize(reflect(-V, H));
                else L = normalize(random_sphere(rng) + v3(0.0f, 0.0f, 1.0f));
            } else {
                if (rng.uf() < Fs) L = normalize(reflect(-V, H));
                else { L = normalize(refract(-V, H, bs.eta)); refr = true; }
            }
            bool valid_dir = !((L.z < 0.0f && !refr) || (refr && L.z >= 0.0f));
            // the two energy-compensation taps depend on V only: fetch once for all evaluations
            float ec_r = 1.0f, ec_g = 1.0f;
            if (bs.ec) {
                ec_r = lut_sample(bs.lut_r, 64, 64, 32, V.z, bs.roughness, bs.anisotropy * 32.0f);
                // the glass tap only ever scales terms weighted by pg; with pg == 0 those terms are (finite)*0 = 0
                // whatever the tap is, so it is skipped (tables are finite)
                if (bs.pg != 0.0f)
                    ec_g = lut_sample(bs.eta > 1.0f ? bs.lut_i : bs.lut_o, 128, 128, 32, pow_(V.z, 1.0f / 2.0f), bs.roughness,
                                      (clamp_(bs.ior, 1.0001f, 2.0f) - 1.0f) * 32.0f);
            }
            Eval se; se.f = v3s(0.0f); se.pdf = 0.0f;
            V3 Ls = v3s(0.0f);
            const float gv = bs.smith(V);  // G1(V): shared by every evaluation of this hit
            bs.set_view(V, ec_r, ec_g);
    if (valid_dir) { se = bs.eval(V, L, ec_r, ec_g, gv); Ls = L; }
            bool was_refracted = Ls.z < 0.0f;
            V3 scatter_world = s.tangent_to_world(Ls);
            if (!was_refracted && dot(scatter_world, s.Ng) < 0.0f) { se.pdf = 0.0f; se.f = v3s(0.0f); }
            if (was_refracted && s.inside) { in_medium = false; }
            else if (was_refracted && !s.inside) {
                in_medium = true;
                ps.M[slot] = make_float4(mcol.x, mcol.y, mcol.z, mdens);
                ps.maniso[slot] = maniso;
            }
            // emission with MIS against light sampling (:265-317)
            if (P.flags & VPT_FLAG_MESH_MIS) {
                if (depth == 0 && is_light) emitted = emitted + bs.emissive;
                else if (is_light) {
                    float d2 = dot(s.pos - porg, s.pos - porg);
                    float ct = fabs_(dot(s.N, normalize(porg - s.pos)));
                    uint32_t tc = 0;
                    float area = 0.0f;  // of the hit triangle in world space: the light table holds exactly that value
                    for (uint32_t k = 0; k < sc.emissive_count; k++)
                        if (sc.emissive[k].instance == inst_id) {
                            tc = sc.emissive[k].tri_count;
                            area = sc.emissive_tri[sc.emissive_tri_offset[k] + (__float_as_uint(h.w) - in.tri_offset)].area;   // h.w = global id; the table is per mesh triangle
                            break;
                        }
                    float lp = (1.0f / (float)sc.emissive_count) * (1.0f / (float)tc) * (1.0f / area) * (d2 / ct);
                    lp = max_(lp, P.emissive_pdf_bias);
                    emitted = emitted + bs.emissive * power_heuristics(prev_pdf, lp);
                }
            } else {
                emitted = emitted + bs.emissive;
            }
            // NEE contributions, evaluated speculatively; the connect stage decides whether they count
            // (EvaluateBSDF draws no random numbers, so evaluating before the visibility test is equivalent)
            new_o = s.pos + s.N * (was_refracted ? -1e-3f : 1e-3f);  // volumes shadow NEE from the NEW origin (:332-333, 364)
            // with an atmosphere every unobscured sky sample has its transmittance tracked (draws), used or not (ClosestHit.slang:330-349)
            if ((P.flags & VPT_FLAG_SKY_MIS) && (sky.w > 0.0f || tracked)) {
                Eval e = bs.eval(V, s.world_to_tangent(to_sky), ec_r, ec_g, gv);
                if (e.pdf > 0.0f || tracked) {
                    if (VOL) {  // assembled after the visibility test
                        out.sky_f = e.f; out.sky_tdepth = 0.0f;
                        out.sky_rgb = v3(sky.x, sky.y, sky.z); out.sky_w = sky.w; out.sky_mis = power_heuristics(sky.w, e.pdf); out.sky_kind = 0;
                        out.sky_add = sky.w > 0.0f && e.pdf > 0.0f;
                    } else {
                        csky = (e.f * v3(sky.x, sky.y, sky.z) / sky.w) * power_heuristics(sky.w, e.pdf);
                    }
                    want_sky = true; sky_o = s.pos + s.N * 1e-5f; sky_d = to_sky;
                }
            }
            if ((P.flags & VPT_FLAG_MESH_MIS) && !is_light && lc.w > 0.0f && light_rays) {
                Eval e = bs.eval(V, s.world_to_tangent(to_light), ec_r, ec_g, gv);
                if (e.pdf > 0.0f) {
                    if (VOL) {
                        out.light_f = e.f; out.light_tdepth = 0.0f; out.light_rgb = v3(lc.x, lc.y, lc.z); out.light_w = lc.w;
                        out.light_mis = power_heuristics(lc.w, e.pdf); out.light_kind = 0;
                    } else {
                        clight = (e.f * v3(lc.x, lc.y, lc.z) / lc.w) * power_heuristics(lc.w, e.pdf);
                    }
                    want_light = true; light_o = s.pos + to_light * 1e-2f; light_d = to_light;
                }
            }
            new_d = scatter_world;
            bxdf = se.f; new_pdf = se.pdf;
            new_depth = (se.pdf <= 0.0f) ? (kMaxDepthMarker + depth) : (depth + 1u);  // :374-376
        }
    }
    if (VOL) {  // media kernels: the caller resolves visibility, then runs shade_tail_media()
        out.want_sky = want_sky; out.want_light = want_light; out.in_medium = in_medium;
        out.rng = rng.s; out.new_depth = new_depth; out.light_gid = light_gid; out.light_miss_ok = light_miss_ok; out.vdepth = vdepth;
        out.new_o = new_o; out.new_d = new_d; out.new_pdf = new_pdf; out.bxdf = bxdf; out.cchan = in_.cchan;
        out.emitted = emitted; out.csky = csky; out.clight = clight;
        out.sky_o = sky_o; out.sky_d = sky_d; out.light_o = light_o; out.light_d = light_d;
        return;
    }
    // ---- RayGen.slang:104-113: throughput, Russian roulette (drawn on every iteration), loop condition
    V3 thr = thr_prev * (bxdf / new_pdf);
    float p = min_(max_(thr.x, max_(thr.y, thr.z)), 1.0f);
    float u = rng.uf();
    bool terminated = p < u;
    if (!terminated) thr = thr / p;
    if (!(new_depth < P.max_depth)) terminated = true;
    uint32_t cflags = (want_sky ? kCF_Sky : 0u) | (want_light ? kCF_Light : 0u) | (new_depth != 1u ? kCF_Clamp : 0u);
    if (terminated) {
        cflags |= kCF_Finalize;
        if (P.samples_per_frame > 1) {
            uint32_t sample = ps.sidx[slot] + 1u;
            if (sample < P.samples_per_frame) {  // next sample of the pixel continues the RNG stream (RayGen.slang:33)
                uint32_t x, y, f;
                pixel_of_slot(P, slot, x, y, f);
                camera_ray(P, rng, x, y, new_o, new_d);
                thr = v3s(1.0f); new_pdf = 1.0f; new_depth = 0u; in_medium = false; vdepth = 0u;
                ps.sidx[slot] = sample;
                alive = true;
            }
        }
    } else {
        alive = true;
    }
    out.alive = alive; out.terminated = terminated; out.want_sky = want_sky; out.want_light = want_light; out.in_medium = in_medium;
    out.rng = rng.s; out.new_depth = new_depth; out.cflags = cflags; out.light_gid = light_gid;
    out.light_miss_ok = light_miss_ok; out.vdepth = vdepth;
    out.new_o = new_o; out.new_d = new_d; out.thr = thr; out.new_pdf = new_pdf;
    out.emitted = emitted; out.csky = csky; out.clight = clight;
    out.sky_o = sky_o; out.sky_d = sky_d; out.light_o = light_o; out.light_d = light_d;
}

// The tail of the bounce loop for the media kernels (RayGen.slang:104-129), on the state shade_core<true> left in `o`.
// `aborted`: the loop was left by `break` before anything happened (origin below the planet's surface, :76-84).
__device__ __forceinline__ void shade_tail_media(const RenderParams& P, const PathState& ps, uint32_t slot, V3 thr_prev, bool aborted, ShadeOut& o) {
    Rng rng; rng.s = o.rng;
    V3 thr = thr_prev;
    bool terminated = true;
    if (!aborted) {
        thr = thr_prev * (o.bxdf / o.new_pdf);
        float p = min_(max_(thr.x, max_(thr.y, thr.z)), 1.0f);
        float u = rng.uf();
        terminated = p < u;
        if (!terminated) thr = thr / p;
        if (!(o.new_depth < P.max_depth)) terminated = true;
    }
    bool alive = false;
    uint32_t cflags = (o.new_depth != 1u ? kCF_Clamp : 0u);
    if (terminated) {
        cflags |= kCF_Finalize;
        if (P.samples_per_frame > 1) {
            uint32_t sample = ps.sidx[slot] + 1u;
            if (sample < P.samples_per_frame) {
                uint32_t x, y, f;
                pixel_of_slot(P, slot, x, y, f);
                camera_ray(P, rng, x, y, o.new_o, o.new_d);
                thr = v3s(1.0f); o.new_pdf = 1.0f; o.new_depth = 0u; o.in_medium = false; o.vdepth = 0u; o.cchan = -1;
                ps.sidx[slot] = sample;
                alive = true;
            }
        }
    } else {
        alive = true;
    }
    o.alive = alive; o.terminated = terminated; o.cflags = cflags; o.rng = rng.s; o.thr = thr;
}

}  // namespace vpt
