// kernels_path.hip — wavefront integrator stages for gfx950.
//
//   raygen     RayGen.slang:12-64     pixel/frame -> camera ray, path state, full queue
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
#include "traverse.hpp"

namespace vpt {

__device__ inline uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ inline uint32_t lanes_below(unsigned long long mask) {  // popcount of mask bits below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
// One atomic per wave: returns this lane's slot in the output stream (valid where pred).
__device__ inline uint32_t wave_append(bool pred, uint32_t* counter) {
    unsigned long long mask = __ballot(pred);
    uint32_t total = (uint32_t)__popcll(mask);
    uint32_t base = 0;
    if (total) {
        uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
        if (lane_id() == leader) base = atomicAdd(counter, total);
        base = __shfl(base, (int)leader);
    }
    return base + lanes_below(mask);
}

// ------------------------------------------------------------------ raygen
__global__ __launch_bounds__(256) void k_raygen(RenderParams P, PathState ps, uint32_t* queue, uint32_t n_slots,
                                                uint32_t dispatch_base) {
    uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_slots) return;
    uint32_t f = slot / P.shard_pixels, sp = slot - f * P.shard_pixels;
    uint32_t ys = sp / P.width, x = sp - ys * P.width;
    uint32_t y = P.shard_rank + P.shard_count * ys;
    uint32_t seed = pcg_hash(P.base_seed + dispatch_base + f);  // PathTracer.cpp:139 with an explicit seed
    Rng r; r.s = y + P.width * x + seed;                        // RayGen.slang:28
    V3 o, d;
    camera_ray(P, r, x, y, o, d);
    ps.rng[slot] = r.s;
    ps.ox[slot] = o.x; ps.oy[slot] = o.y; ps.oz[slot] = o.z;
    ps.dx[slot] = d.x; ps.dy[slot] = d.y; ps.dz[slot] = d.z;
    ps.tx[slot] = 1.0f; ps.ty[slot] = 1.0f; ps.tz[slot] = 1.0f;
    ps.lx[slot] = 0.0f; ps.ly[slot] = 0.0f; ps.lz[slot] = 0.0f;
    ps.bx[slot] = 1.0f; ps.by[slot] = 1.0f; ps.bz[slot] = 1.0f;
    ps.pdf[slot] = 1.0f;
    ps.depth[slot] = 0u;
    ps.medium_flag[slot] = 0u;
    ps.ax[slot] = 0.0f; ps.ay[slot] = 0.0f; ps.az[slot] = 0.0f;
    queue[slot] = slot;
}

// ------------------------------------------------------------------ persistent traversal kernels
constexpr uint32_t kFetch = 256;  // rays per queue fetch per wave (4 x 64): one atomic per 256 rays

template <bool LDS_SCENE>
__device__ inline void stage_scene(const DeviceScene& sc, float4* lds_nodes, float4* lds_tris) {
    if (LDS_SCENE) {
        const float4* gn = reinterpret_cast<const float4*>(sc.nodes);
        const float4* gt = reinterpret_cast<const float4*>(sc.tris);
        for (uint32_t i = threadIdx.x; i < sc.node_count * 4; i += blockDim.x) lds_nodes[i] = gn[i];
        for (uint32_t i = threadIdx.x; i < sc.tri_count * 3; i += blockDim.x) lds_tris[i] = gt[i];
        __syncthreads();
    }
}

template <bool LDS_SCENE, bool COUNT>
__global__ __launch_bounds__(kTraverseBlock) void k_extend(DeviceScene sc, PathState ps, const uint32_t* queue,
                                                          Counters* ctr, uint32_t parity) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t* stack = reinterpret_cast<uint32_t*>(smem) + threadIdx.x;
    float4* lds_nodes = reinterpret_cast<float4*>(smem + kStackDepth * kTraverseBlock * 4);
    float4* lds_tris = lds_nodes + sc.node_count * 4;
    stage_scene<LDS_SCENE>(sc, lds_nodes, lds_tris);
    const uint32_t n = ctr->ray_count[parity];
    TravStats st; st.nodes = 0; st.tris = 0;
    while (true) {
        uint32_t base = 0;
        if (lane_id() == 0) base = atomicAdd(&ctr->extend_head, kFetch);
        base = __shfl(base, 0);
        if (base >= n) break;
        for (uint32_t k = 0; k < kFetch; k += 64) {
            uint32_t i = base + k + lane_id();
            if (i >= n) break;
            uint32_t slot = queue[i];
            V3 o = v3(ps.ox[slot], ps.oy[slot], ps.oz[slot]);
            V3 d = normalize(v3(ps.dx[slot], ps.dy[slot], ps.dz[slot]));  // RayGen.slang:70
            HitRec h;
            bool found;
            if (LDS_SCENE) { LdsSceneSrc src{lds_nodes, lds_tris}; found = trace_closest<COUNT>(src, o, d, 0.01f, 100000.0f, stack, kTraverseBlock, h, st); }
            else { GlobalSceneSrc src{sc.nodes, sc.tris}; found = trace_closest<COUNT>(src, o, d, 0.01f, 100000.0f, stack, kTraverseBlock, h, st); }
            ps.ht[slot] = found ? h.t : -1.0f;
            ps.hu[slot] = h.u; ps.hv[slot] = h.v;
            ps.hprim[slot] = h.prim; ps.hinst[slot] = h.inst;
        }
    }
    if (COUNT) {
        atomicAdd(&ctr->stat_nodes, (unsigned long long)st.nodes);
        atomicAdd(&ctr->stat_tris, (unsigned long long)st.tris);
    }
}

template <bool LDS_SCENE, bool COUNT>
__global__ __launch_bounds__(kTraverseBlock) void k_shadow(DeviceScene sc, PathState ps, const ShadowRay* rays, Counters* ctr) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t* stack = reinterpret_cast<uint32_t*>(smem) + threadIdx.x;
    float4* lds_nodes = reinterpret_cast<float4*>(smem + kStackDepth * kTraverseBlock * 4);
    float4* lds_tris = lds_nodes + sc.node_count * 4;
    stage_scene<LDS_SCENE>(sc, lds_nodes, lds_tris);
    const uint32_t n = ctr->shadow_count;
    TravStats st; st.nodes = 0; st.tris = 0;
    while (true) {
        uint32_t base = 0;
        if (lane_id() == 0) base = atomicAdd(&ctr->shadow_head, kFetch);
        base = __shfl(base, 0);
        if (base >= n) break;
        for (uint32_t k = 0; k < kFetch; k += 64) {
            uint32_t i = base + k + lane_id();
            if (i >= n) break;
            const float4* rp = reinterpret_cast<const float4*>(rays + i);
            float4 r0 = rp[0], r1 = rp[1];
            uint32_t sk = __float_as_uint(r0.w), expect = __float_as_uint(r1.w);
            V3 o = v3(r0.x, r0.y, r0.z), d = v3(r1.x, r1.y, r1.z);  // direction not re-normalised (RTCommon.slang:55)
            HitRec h;
            bool found;
            if (LDS_SCENE) { LdsSceneSrc src{lds_nodes, lds_tris}; found = trace_closest<COUNT>(src, o, d, 0.0001f, 1000000.0f, stack, kTraverseBlock, h, st); }
            else { GlobalSceneSrc src{sc.nodes, sc.tris}; found = trace_closest<COUNT>(src, o, d, 0.0001f, 1000000.0f, stack, kTraverseBlock, h, st); }
            uint32_t slot = sk & 0x7fffffffu;
            if (sk >> 31) { if (found && h.gid == expect) atomicOr(ps.vis + slot, 2u); }  // ClosestHit.slang:173-176
            else { if (!found) atomicOr(ps.vis + slot, 1u); }                              // ClosestHit.slang:139
        }
    }
    if (COUNT) {
        atomicAdd(&ctr->stat_shadow_nodes, (unsigned long long)st.nodes);
        atomicAdd(&ctr->stat_shadow_tris, (unsigned long long)st.tris);
    }
}

// Test hook: the extend traversal on caller-supplied rays.
__global__ __launch_bounds__(kTraverseBlock) void k_trace_rays(DeviceScene sc, const vpt_ray* rays, uint32_t n, vpt_hit* hits) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t* stack = reinterpret_cast<uint32_t*>(smem) + threadIdx.x;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    vpt_ray r = rays[i];
    HitRec h; TravStats st;
    GlobalSceneSrc src{sc.nodes, sc.tris};
    bool found = trace_closest<false>(src, v3(r.origin[0], r.origin[1], r.origin[2]), v3(r.direction[0], r.direction[1], r.direction[2]),
                                      r.tmin, r.tmax, stack, kTraverseBlock, h, st);
    vpt_hit o; o.t = found ? h.t : -1.0f; o.u = found ? h.u : 0.0f; o.v = found ? h.v : 0.0f; o.primitive = h.prim; o.instance = h.inst;
    hits[i] = o;
}

// ------------------------------------------------------------------ shade
__global__ __launch_bounds__(256) void k_shade(DeviceScene sc, RenderParams P, PathState ps, const uint32_t* queue,
                                               ShadowRay* shadow, Counters* ctr, uint32_t parity) {
    const uint32_t n = ctr->ray_count[parity];
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool active = i < n;
    uint32_t slot = active ? queue[i] : 0u;
    bool want_sky = false, want_light = false;
    V3 sky_o = v3s(0.0f), sky_d = v3s(0.0f), light_o = v3s(0.0f), light_d = v3s(0.0f);
    uint32_t light_gid = 0xffffffffu;
    if (active) {
        Rng rng; rng.s = ps.rng[slot];
        V3 porg = v3(ps.ox[slot], ps.oy[slot], ps.oz[slot]);  // payload.Origin (previous vertex)
        V3 pdir = v3(ps.dx[slot], ps.dy[slot], ps.dz[slot]);  // payload.Direction
        uint32_t depth = ps.depth[slot];
        float prev_pdf = ps.pdf[slot];
        float ht = ps.ht[slot];
        V3 emitted = v3s(0.0f), csky = v3s(0.0f), clight = v3s(0.0f);
        if (ht < 0.0f) {
            // ---- Miss.slang:8-77
            V4 cp = v4(0.0f, 0.0f, 0.0f, 1.0f);
            if ((P.flags & VPT_FLAG_SHOW_ENV_DIRECTLY) || depth > 0) {
                V3 d = rotate(pdir, v3(1.0f, 0.0f, 0.0f), -(P.sky_altitude / 180.0f * VPT_PI));
                d = rotate(d, v3(0.0f, 1.0f, 0.0f), -(P.sky_azimuth / 180.0f * VPT_PI));
                V2 uv = direction_to_uv(d);
                cp = env_sample(sc, uv.x, uv.y);
            }
            emitted = v3(cp.x, cp.y, cp.z) * P.sky_intensity;
            if (P.flags & VPT_FLAG_FURNACE) emitted = v3s(1.0f);
            if ((P.flags & VPT_FLAG_SKY_MIS) && depth > 0) emitted = emitted * power_heuristics(prev_pdf, cp.w);
            ps.depth[slot] = kMaxDepthMarker;
        } else {
            // ---- ClosestHit.slang:20-378
            V3 rd = normalize(pdir);  // WorldRayDirection()
            uint32_t inst_id = ps.hinst[slot];
            const InstanceDesc& in = sc.instances[inst_id];
            const vpt_material& mat = sc.materials[in.material];
            SurfaceFrame s;
            surface_init(sc, s, in, ps.hprim[slot], ps.hu[slot], ps.hv[slot], rd, mat.normal_texture,
                         (P.flags & VPT_FLAG_GEOMETRY_NORMALS) != 0);
            Bsdf b; V3 mcol; float mdens, maniso, arot;
            bsdf_init(sc, b, mat, s.uv, s.inside, P.flags, mcol, mdens, maniso, arot);
            bool is_light = b.emissive.x > 0.0f || b.emissive.y > 0.0f || b.emissive.z > 0.0f;
            rotate_tangents(s, arot);
            uint32_t mflag = ps.medium_flag[slot];
            bool scattered = false;
            if (mflag & 1u) {  // :80-116
                float pm_aniso = ps.maniso[slot];
                if (pm_aniso != 1.0f) {
                    float gd = length(porg - s.pos);
                    float sd = -log_(rng.uf()) / ps.mdensity[slot];
                    if (sd < gd) {
                        V3 no = porg + (sd * pdir);
                        V3 nd = sample_hg(rng, pdir, pm_aniso);
                        ps.ox[slot] = no.x; ps.oy[slot] = no.y; ps.oz[slot] = no.z;
                        ps.dx[slot] = nd.x; ps.dy[slot] = nd.y; ps.dz[slot] = nd.z;
                        ps.bx[slot] = ps.mcr[slot]; ps.by[slot] = ps.mcg[slot]; ps.bz[slot] = ps.mcb[slot];
                        scattered = true;  // PDF stays stale, depth unchanged, nothing emitted
                    }
                }
            }
            if (!scattered) {
                // sky NEE sample (:125-148) — 3 draws
                V3 to_sky = v3s(0.0f); V4 sky = v4(0.0f, 0.0f, 0.0f, 0.0f);
                if (P.flags & VPT_FLAG_SKY_MIS) {
                    sample_env(sc, P, rng, to_sky, sky);
                    sky.x *= P.sky_intensity; sky.y *= P.sky_intensity; sky.z *= P.sky_intensity;  // applied twice upstream (quirk 1)
                }
                // emissive-mesh NEE sample (:155-184) — 4 draws unless this is an emitter
                V3 to_light = v3s(0.0f); V4 lc = v4(0.0f, 0.0f, 0.0f, 0.0f);
                if ((P.flags & VPT_FLAG_MESH_MIS) && !is_light) sample_emissive(sc, rng, s.pos, to_light, lc, light_gid);
                // BSDF sampling (:190-201; Material.slang:94-165)
                V3 V = s.world_to_tangent(normalize(-rd));
                V3 H = ggx_sample(rng, V, b.ax, b.ay);
                float Fs = b.fresnel(dot(V, H));
                float x1 = rng.uf();
                V3 L; bool refr = false;
                if (x1 < b.pm) { L = normalize(reflect(-V, H)); }
                else if (x1 < b.pm + b.pd) {
                    if (rng.uf() < Fs) L = normalize(reflect(-V, H));
                    else L = normalize(random_sphere(rng) + v3(0.0f, 0.0f, 1.0f));
                } else {
                    if (rng.uf() < Fs) L = normalize(reflect(-V, H));
                    else { L = normalize(refract(-V, H, b.eta)); refr = true; }
                }
                bool valid_dir = !((L.z < 0.0f && !refr) || (refr && L.z >= 0.0f));
                // the two energy-compensation taps depend on V only: fetch once for all evaluations
                float ec_r = 1.0f, ec_g = 1.0f;
                if (b.ec) {
                    ec_r = lut_sample(b.lut_r, 64, 64, 32, V.z, b.roughness, b.anisotropy * 32.0f);
                    ec_g = lut_sample(b.eta > 1.0f ? b.lut_i : b.lut_o, 128, 128, 32, pow_(V.z, 1.0f / 2.0f), b.roughness,
                                      (clamp_(b.ior, 1.0001f, 2.0f) - 1.0f) * 32.0f);
                }
                Eval se; se.f = v3s(0.0f); se.pdf = 0.0f;
                V3 Ls = v3s(0.0f);
                if (valid_dir) { se = b.eval(V, L, ec_r, ec_g); Ls = L; }
                bool was_refracted = Ls.z < 0.0f;
                V3 scatter_world = s.tangent_to_world(Ls);
                if (!was_refracted && dot(scatter_world, s.Ng) < 0.0f) { se.pdf = 0.0f; se.f = v3s(0.0f); }
                if (was_refracted && s.inside) { mflag &= ~1u; }
                else if (was_refracted && !s.inside) {
                    mflag |= 1u;
                    ps.mcr[slot] = mcol.x; ps.mcg[slot] = mcol.y; ps.mcb[slot] = mcol.z;
                    ps.maniso[slot] = maniso; ps.mdensity[slot] = mdens;
                }
                // emission with MIS against light sampling (:265-317)
                if (P.flags & VPT_FLAG_MESH_MIS) {
                    if (depth == 0 && is_light) emitted = emitted + b.emissive;
                    else if (is_light) {
                        V3 a = mat_point(in.xform, s.p1), bb = mat_point(in.xform, s.p2), cc = mat_point(in.xform, s.p3);
                        float area = length(cross(bb - a, cc - a)) * 0.5f;
                        float d2 = dot(s.pos - porg, s.pos - porg);
                        float ct = fabs_(dot(s.N, normalize(porg - s.pos)));
                        uint32_t tc = 0;
                        for (uint32_t k = 0; k < sc.emissive_count; k++)
                            if (sc.emissive[k].instance == inst_id) { tc = sc.emissive[k].tri_count; break; }
                        float lp = (1.0f / (float)sc.emissive_count) * (1.0f / (float)tc) * (1.0f / area) * (d2 / ct);
                        lp = max_(lp, P.emissive_pdf_bias);
                        emitted = emitted + b.emissive * power_heuristics(prev_pdf, lp);
                    }
                } else {
                    emitted = emitted + b.emissive;
                }
                // NEE contributions, evaluated speculatively; the shadow stage decides whether they count
                // (EvaluateBSDF draws no random numbers, so evaluating before the visibility test is equivalent)
                if ((P.flags & VPT_FLAG_SKY_MIS) && sky.w > 0.0f) {
                    Eval e = b.eval(V, s.world_to_tangent(to_sky), ec_r, ec_g);
                    if (e.pdf > 0.0f) {
                        csky = (e.f * v3(sky.x, sky.y, sky.z) / sky.w) * power_heuristics(sky.w, e.pdf);
                        want_sky = true; sky_o = s.pos + s.N * 1e-5f; sky_d = to_sky;
                    }
                }
                if ((P.flags & VPT_FLAG_MESH_MIS) && !is_light && lc.w > 0.0f) {
                    Eval e = b.eval(V, s.world_to_tangent(to_light), ec_r, ec_g);
                    if (e.pdf > 0.0f) {
                        clight = (e.f * v3(lc.x, lc.y, lc.z) / lc.w) * power_heuristics(lc.w, e.pdf);
                        want_light = true; light_o = s.pos + to_light * 1e-2f; light_d = to_light;
                    }
                }
                V3 no = s.pos + s.N * (was_refracted ? -1e-3f : 1e-3f);
                ps.ox[slot] = no.x; ps.oy[slot] = no.y; ps.oz[slot] = no.z;
                ps.dx[slot] = scatter_world.x; ps.dy[slot] = scatter_world.y; ps.dz[slot] = scatter_world.z;
                ps.bx[slot] = se.f.x; ps.by[slot] = se.f.y; ps.bz[slot] = se.f.z;
                ps.pdf[slot] = se.pdf;
                ps.medium_flag[slot] = mflag;
                ps.depth[slot] = (se.pdf <= 0.0f) ? (kMaxDepthMarker + depth) : (depth + 1u);  // :374-376
            }
        }
        ps.rng[slot] = rng.s;
        ps.ex[slot] = emitted.x; ps.ey[slot] = emitted.y; ps.ez[slot] = emitted.z;
        ps.skx[slot] = csky.x; ps.sky[slot] = csky.y; ps.skz[slot] = csky.z;
        ps.lgx[slot] = clight.x; ps.lgy[slot] = clight.y; ps.lgz[slot] = clight.z;
        ps.vis[slot] = 0u;
    }
    // wave-level compaction of the (<=2 per path) shadow rays into one stream
    uint32_t is = wave_append(want_sky, &ctr->shadow_count);
    if (want_sky) {
        float4* q = reinterpret_cast<float4*>(shadow + is);
        q[0] = make_float4(sky_o.x, sky_o.y, sky_o.z, __uint_as_float(slot));
        q[1] = make_float4(sky_d.x, sky_d.y, sky_d.z, __uint_as_float(0u));
    }
    uint32_t il = wave_append(want_light, &ctr->shadow_count);
    if (want_light) {
        float4* q = reinterpret_cast<float4*>(shadow + il);
        q[0] = make_float4(light_o.x, light_o.y, light_o.z, __uint_as_float(slot | 0x80000000u));
        q[1] = make_float4(light_d.x, light_d.y, light_d.z, __uint_as_float(light_gid));
    }
}

// ------------------------------------------------------------------ accumulate (+ compaction, + path regeneration)
__global__ __launch_bounds__(256) void k_accumulate(RenderParams P, PathState ps, const uint32_t* queue_in, uint32_t* queue_out,
                                                    Counters* ctr, uint32_t parity) {
    const uint32_t n = ctr->ray_count[parity];
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool active = i < n;
    uint32_t slot = active ? queue_in[i] : 0u;
    bool alive = false;
    if (active) {
        uint32_t vis = ps.vis[slot];
        V3 E = v3(ps.ex[slot], ps.ey[slot], ps.ez[slot]);
        if (vis & 1u) E = E + v3(ps.skx[slot], ps.sky[slot], ps.skz[slot]);  // ClosestHit.slang:344-353
        if (vis & 2u) E = E + v3(ps.lgx[slot], ps.lgy[slot], ps.lgz[slot]);  // ClosestHit.slang:358-370
        V3 thr = v3(ps.tx[slot], ps.ty[slot], ps.tz[slot]);
        V3 light = v3(ps.lx[slot], ps.ly[slot], ps.lz[slot]);
        uint32_t depth = ps.depth[slot];
        V3 contrib = E * thr;  // RayGen.slang:92
        if (depth != 1u) {
            float lum = dot(contrib, v3(0.212671f, 0.715160f, 0.072169f));
            contrib = contrib * (P.max_luminance / max_(lum, P.max_luminance));
        }
        light = light + contrib;
        thr = thr * (v3(ps.bx[slot], ps.by[slot], ps.bz[slot]) / ps.pdf[slot]);
        float p = min_(max_(thr.x, max_(thr.y, thr.z)), 1.0f);
        Rng rng; rng.s = ps.rng[slot];
        float u = rng.uf();  // drawn on every iteration, terminal ones included
        bool terminated = (p < u);
        if (!terminated) thr = thr / p;
        if (!(depth < P.max_depth)) terminated = true;
        if (!terminated) {
            alive = true;
            ps.tx[slot] = thr.x; ps.ty[slot] = thr.y; ps.tz[slot] = thr.z;
            ps.lx[slot] = light.x; ps.ly[slot] = light.y; ps.lz[slot] = light.z;
        } else {
            // sample finished: NaN/Inf guard, per-frame sum (RayGen.slang:116-128)
            bool ok = !isinf_(light.x) && !isinf_(light.y) && !isinf_(light.z) && !isnan_(light.x) && !isnan_(light.y) && !isnan_(light.z);
            if (ok) { ps.ax[slot] += light.x; ps.ay[slot] += light.y; ps.az[slot] += light.z; }
            uint32_t mflag = ps.medium_flag[slot];
            uint32_t sample = (mflag >> 8) + 1u;
            if (sample < P.samples_per_frame) {
                // next sample of the same pixel continues the same RNG stream (RayGen.slang:33)
                uint32_t sp = slot % P.shard_pixels;
                uint32_t ys = sp / P.width, x = sp - ys * P.width;
                uint32_t y = P.shard_rank + P.shard_count * ys;
                V3 o, d;
                camera_ray(P, rng, x, y, o, d);
                ps.ox[slot] = o.x; ps.oy[slot] = o.y; ps.oz[slot] = o.z;
                ps.dx[slot] = d.x; ps.dy[slot] = d.y; ps.dz[slot] = d.z;
                ps.tx[slot] = 1.0f; ps.ty[slot] = 1.0f; ps.tz[slot] = 1.0f;
                ps.lx[slot] = 0.0f; ps.ly[slot] = 0.0f; ps.lz[slot] = 0.0f;
                ps.bx[slot] = 1.0f; ps.by[slot] = 1.0f; ps.bz[slot] = 1.0f;
                ps.pdf[slot] = 1.0f;
                ps.depth[slot] = 0u;
                ps.medium_flag[slot] = sample << 8;
                alive = true;
            }
        }
        ps.rng[slot] = rng.s;
    }
    uint32_t o = wave_append(alive, &ctr->ray_count[parity ^ 1u]);
    if (alive) queue_out[o] = slot;
}

// ------------------------------------------------------------------ resolve: running mean, frames applied in order
__global__ __launch_bounds__(256) void k_resolve(RenderParams P, PathState ps, float4* image, uint32_t frames, uint32_t frame_base) {
    uint32_t sp = blockIdx.x * blockDim.x + threadIdx.x;
    if (sp >= P.shard_pixels) return;
    float4 px = image[sp];
    V3 color = v3(px.x, px.y, px.z);
    for (uint32_t f = 0; f < frames; f++) {
        uint32_t slot = f * P.shard_pixels + sp;
        V3 acc = v3(ps.ax[slot], ps.ay[slot], ps.az[slot]) / (float)P.samples_per_frame;
        uint32_t fc = frame_base + f;
        if (fc > 0) color = lerp(color, acc, 1.0f / (float)(fc + 1u));
        else color = acc;
    }
    image[sp] = make_float4(color.x, color.y, color.z, 1.0f);
}

__global__ void k_prepare(Counters* ctr, uint32_t parity) {
    ctr->extend_head = 0u; ctr->shadow_head = 0u; ctr->shadow_count = 0u; ctr->ray_count[parity ^ 1u] = 0u;
}

// shard rows <-> full image
__global__ __launch_bounds__(256) void k_scatter_rows(const float4* gathered, float4* full, uint32_t width, uint32_t height,
                                                      uint32_t shard_count, uint32_t shard_stride_px) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= width * height) return;
    uint32_t y = i / width, x = i - y * width;
    uint32_t r = y % shard_count, ys = y / shard_count;
    full[i] = gathered[(size_t)r * shard_stride_px + (size_t)ys * width + x];
}

// ------------------------------------------------------------------ launch wrappers
static inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

void launch_raygen(hipStream_t s, const RenderParams& P, const PathState& ps, uint32_t* queue, uint32_t n_slots, uint32_t dispatch_base) {
    hipLaunchKernelGGL(k_raygen, dim3(cdiv(n_slots, 256)), dim3(256), 0, s, P, ps, queue, n_slots, dispatch_base);
}
void launch_prepare(hipStream_t s, Counters* ctr, uint32_t parity) { hipLaunchKernelGGL(k_prepare, dim3(1), dim3(1), 0, s, ctr, parity); }

size_t traverse_lds_bytes(const DeviceScene& sc, bool lds_scene) {
    size_t b = (size_t)kStackDepth * kTraverseBlock * 4;
    if (lds_scene) b += (size_t)sc.node_count * 64 + (size_t)sc.tri_count * 48;
    return b;
}
void launch_extend(hipStream_t s, uint32_t blocks, bool lds_scene, bool count, const DeviceScene& sc, const PathState& ps,
                   const uint32_t* queue, Counters* ctr, uint32_t parity) {
    size_t lds = traverse_lds_bytes(sc, lds_scene);
    if (lds_scene) {
        if (count) hipLaunchKernelGGL((k_extend<true, true>), dim3(blocks), dim3(kTraverseBlock), lds, s, sc, ps, queue, ctr, parity);
        else hipLaunchKernelGGL((k_extend<true, false>), dim3(blocks), dim3(kTraverseBlock), lds, s, sc, ps, queue, ctr, parity);
    } else {
        if (count) hipLaunchKernelGGL((k_extend<false, true>), dim3(blocks), dim3(kTraverseBlock), lds, s, sc, ps, queue, ctr, parity);
        else hipLaunchKernelGGL((k_extend<false, false>), dim3(blocks), dim3(kTraverseBlock), lds, s, sc, ps, queue, ctr, parity);
    }
}
void launch_shadow(hipStream_t s, uint32_t blocks, bool lds_scene, bool count, const DeviceScene& sc, const PathState& ps,
                   const ShadowRay* rays, Counters* ctr) {
    size_t lds = traverse_lds_bytes(sc, lds_scene);
    if (lds_scene) {
        if (count) hipLaunchKernelGGL((k_shadow<true, true>), dim3(blocks), dim3(kTraverseBlock), lds, s, sc, ps, rays, ctr);
        else hipLaunchKernelGGL((k_shadow<true, false>), dim3(blocks), dim3(kTraverseBlock), lds, s, sc, ps, rays, ctr);
    } else {
        if (count) hipLaunchKernelGGL((k_shadow<false, true>), dim3(blocks), dim3(kTraverseBlock), lds, s, sc, ps, rays, ctr);
        else hipLaunchKernelGGL((k_shadow<false, false>), dim3(blocks), dim3(kTraverseBlock), lds, s, sc, ps, rays, ctr);
    }
}
void launch_shade(hipStream_t s, uint32_t n_upper, const DeviceScene& sc, const RenderParams& P, const PathState& ps,
                  const uint32_t* queue, ShadowRay* shadow, Counters* ctr, uint32_t parity) {
    hipLaunchKernelGGL(k_shade, dim3(cdiv(n_upper, 256)), dim3(256), 0, s, sc, P, ps, queue, shadow, ctr, parity);
}
void launch_accumulate(hipStream_t s, uint32_t n_upper, const RenderParams& P, const PathState& ps, const uint32_t* qin,
                       uint32_t* qout, Counters* ctr, uint32_t parity) {
    hipLaunchKernelGGL(k_accumulate, dim3(cdiv(n_upper, 256)), dim3(256), 0, s, P, ps, qin, qout, ctr, parity);
}
void launch_resolve(hipStream_t s, const RenderParams& P, const PathState& ps, float* image, uint32_t frames, uint32_t frame_base) {
    hipLaunchKernelGGL(k_resolve, dim3(cdiv(P.shard_pixels, 256)), dim3(256), 0, s, P, ps, reinterpret_cast<float4*>(image), frames, frame_base);
}
void launch_trace_rays(hipStream_t s, const DeviceScene& sc, const vpt_ray* rays, uint32_t n, vpt_hit* hits) {
    hipLaunchKernelGGL(k_trace_rays, dim3(cdiv(n, kTraverseBlock)), dim3(kTraverseBlock), (size_t)kStackDepth * kTraverseBlock * 4, s, sc, rays, n, hits);
}
void launch_scatter_rows(hipStream_t s, const float* gathered, float* full, uint32_t w, uint32_t h, uint32_t shard_count, uint32_t stride_px) {
    hipLaunchKernelGGL(k_scatter_rows, dim3(cdiv(w * h, 256)), dim3(256), 0, s, reinterpret_cast<const float4*>(gathered),
                       reinterpret_cast<float4*>(full), w, h, shard_count, stride_px);
}
int traverse_blocks_per_cu(bool lds_scene, const DeviceScene& sc) {
    int nb = 0;
    size_t lds = traverse_lds_bytes(sc, lds_scene);
    if (lds_scene) hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_extend<true, false>, kTraverseBlock, lds);
    else hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_extend<false, false>, kTraverseBlock, lds);
    return nb > 0 ? nb : 1;
}

}  // namespace vpt
