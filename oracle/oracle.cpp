// oracle.cpp — CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
// library.  The product (vulkan-path-tracer_amd/) never includes, links or calls it.
//
// What it restates (all paths relative to /root/reference/PathTracer):
//   Shaders/Sampler.slang:4-166,286-422   RNG + sampling routines
//   Shaders/RTCommon.slang:5-136          payload, heuristics, DirectionToUV, Rotate
//   Shaders/Surface.slang                 hit frame
//   Shaders/Material.slang                BSDF sample/eval
//   Shaders/ClosestHit.slang, Miss.slang  shading, NEE, MIS
//   Shaders/RayGen.slang:9-160            per-pixel sample loop, accumulation
//   Shaders/PostProcess/*.slang + PostProcessor.cpp:128-246   bloom chain + tonemap
//   PathTracer.cpp:449-469 (emissive list), 1161-1296 (env importance/alias/pdf)
//   Shaders/LookupReflect.slang, LookupRefract.slang   (known-answer test vs Assets/LookupTables)
//
// PARITY PINNING.  The reference cannot be built or run here (Vulkan RT + absent VulkanHelper
// submodule + Slang compiler), has no tests and no golden images, so this oracle is pinned
// against the only reference-produced numbers in the tree: Assets/LookupTables/*.bin (generated
// by the reference's own Material/Sampler code; tests/test_oracle_kat.py reproduces table cells
// by Monte Carlo through THIS file's BSDF functions), the PCG known answers derived from
// Sampler.slang:4-9, and furnace-mode energy conservation; every numeric constant of the hot-path shaders is
// audited against this file and the product's sources (tests/test_reference_constants.py).  Beyond reference-produced numbers, the
// algorithmic content is held against INDEPENDENT float64 restatements written from the Slang
// sources, not from this file: the refraction tables' cells (tests/test_oracle_lut_fp64.py),
// EvaluateBSDF and the VNDF / SampleBSDF draws (tests/test_oracle_bsdf_fp64.py), the bloom chain and
// tonemap (tests/test_oracle_post.py), and the whole per-sample integrator — RayGen, ClosestHit, Miss,
// Surface, emissive-triangle NEE, MIS, clamp, roulette (tests/ref_integrator64.py,
// tests/test_oracle_integrator_fp64.py: per-sample values agree to 2e-4 .. 1e-3 on every sample compared).
// The second restatement covers textures, environment importance sampling (alias map included), the medium inside a glass
// mesh, homogeneous box volumes with all three phase functions and the atmosphere; density grids are outside it.  BVH traversal, scene import and the
// elementary fp32 functions live in the Vulkan driver / VulkanHelper / Slang: parity for those
// is UNPINNED; they follow include/vpt_fp32.h on both sides.
//
// Structure is deliberately the reference's: one scalar "megakernel" per pixel with a simple
// median-split BVH (or brute force) — nothing here is shared with the HIP wavefront design.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/vpt.h"
#include "../include/vpt_fp32.h"

using namespace vptfp;

namespace {

// Defines.slang
const float M_PI_F = 3.1415926535897F;
const float M_2_PI_F = 6.2831853071795F;
const float M_1_OVER_PI_F = 0.3183098861837F;
const uint32_t MAX_DEPTH_C = 1000000u;

struct Rng {  // Sampler.slang:21-100
    uint32_t s;
    uint32_t pcg() { s = pcg_hash(s); return s; }
    float uf() { return u32_to_unit(pcg()); }
};

struct Tex {
    int w, h, c;
    std::vector<uint8_t> d;
};

struct Tri {  // world-space triangle for intersection only
    V3 v0, e1, e2;
    uint32_t prim, inst;
    bool skip;  // sliver (vpt_fp32.h triangle_degenerate): never intersected
};

struct Node {
    V3 bmin, bmax;
    int left, right;   // children, or -1
    int first, count;  // leaf range
};

struct Emissive {  // PathTracer.h:321-328
    uint32_t mesh, material, tri_count, instance;
    float xform[16];
};

struct Alias { uint32_t alias; float importance; };  // Bindings.slang:1-5

struct Counters {
    uint64_t closest = 0, shadow = 0, nodes = 0, tris = 0, samples = 0;
};

struct Oracle {
    uint32_t W = 0, H = 0;
    std::vector<std::vector<vpt_vertex>> mverts;
    std::vector<std::vector<uint32_t>> mindices;
    std::vector<vpt_material> materials;
    std::vector<vpt_instance> instances;
    std::vector<float> inv3;  // 9 per instance
    std::vector<Tex> textures;
    std::vector<float> env;  // RGBA32F with pdf in alpha
    uint32_t envW = 0, envH = 0;
    std::vector<Alias> alias;
    std::vector<float> lutR, lutO, lutI;
    std::vector<Emissive> emissive;
    uint32_t emissive_tris = 0;
    std::vector<Tri> tris;
    std::vector<Node> nodes;
    std::vector<int> tri_order;
    float view_inv[16], proj_inv[16];
    vpt_params P;
    std::vector<float> image;  // RGBA32F
    uint64_t dispatch_count = 0;
    uint32_t frame_count = 0, samples_accum = 0;
    bool brute_force = false;
    Counters ctr;
    std::vector<vpt_volume> volumes;  // uVolumes (Volume.slang:9), homogeneous only
    bool atm_on = false;              // ENABLE_ATMOSPHERE
    struct DensityGrid { uint32_t dim[3]; float max_density; std::vector<float> values, block_max; };
    std::vector<DensityGrid> grids;   // uNanoVDBBuffersDensity / uVolumeMaxDensities, densified
    vpt_atmosphere atm;
    uint32_t phase = VPT_PHASE_HENYEY_GREENSTEIN;  // PHASE_FUNCTION_* define (PathTracer.h:219)
};

// ------------------------------------------------------------------ software samplers
// Vulkan linear filtering at mip 0; texel_coords() is the shared fp32 contract.
V4 fetch_tex(const Tex& t, int x, int y) {
    V4 r;
    if (t.c == 4) {
        const uint8_t* p = &t.d[(size_t)(y * t.w + x) * 4];
        r.x = (float)p[0] / 255.0f; r.y = (float)p[1] / 255.0f; r.z = (float)p[2] / 255.0f; r.w = (float)p[3] / 255.0f;
    } else {
        float v = (float)t.d[(size_t)y * t.w + x] / 255.0f;
        r.x = v; r.y = 0.0f; r.z = 0.0f; r.w = 1.0f;  // R8 view: (r,0,0,1)
    }
    return r;
}
V4 lerp4(V4 a, V4 b, float t) {
    V4 r; r.x = lerp(a.x, b.x, t); r.y = lerp(a.y, b.y, t); r.z = lerp(a.z, b.z, t); r.w = lerp(a.w, b.w, t);
    return r;
}
V4 sample_tex(const Tex& t, float u, float v) {  // LINEAR / REPEAT (PathTracer.cpp:84-91)
    int x0, x1, y0, y1; float fx, fy;
    texel_coords(u, t.w, true, &x0, &x1, &fx);
    texel_coords(v, t.h, true, &y0, &y1, &fy);
    V4 a = lerp4(fetch_tex(t, x0, y0), fetch_tex(t, x1, y0), fx);
    V4 b = lerp4(fetch_tex(t, x0, y1), fetch_tex(t, x1, y1), fx);
    return lerp4(a, b, fy);
}
V4 fetch_f4(const float* img, int w, int x, int y) {
    const float* p = img + ((size_t)y * w + x) * 4;
    V4 r; r.x = p[0]; r.y = p[1]; r.z = p[2]; r.w = p[3];
    return r;
}
V4 sample_f4(const float* img, int w, int h, float u, float v, bool repeat) {
    int x0, x1, y0, y1; float fx, fy;
    texel_coords(u, w, repeat, &x0, &x1, &fx);
    texel_coords(v, h, repeat, &y0, &y1, &fy);
    V4 a = lerp4(fetch_f4(img, w, x0, y0), fetch_f4(img, w, x1, y0), fx);
    V4 b = lerp4(fetch_f4(img, w, x0, y1), fetch_f4(img, w, x1, y1), fx);
    return lerp4(a, b, fy);
}
// R32F 2D array, LINEAR, CLAMP_TO_EDGE (PathTracer.cpp:93-94); layer = RNE(clamp(layer)).
float sample_lut(const std::vector<float>& lut, int sx, int sy, int sz, float u, float v, float layer) {
    int x0, x1, y0, y1; float fx, fy;
    texel_coords(u, sx, false, &x0, &x1, &fx);
    texel_coords(v, sy, false, &y0, &y1, &fy);
    int l = lut_layer(layer, sz);
    const float* p = &lut[(size_t)l * sx * sy];
    float a = lerp(p[y0 * sx + x0], p[y0 * sx + x1], fx);
    float b = lerp(p[y1 * sx + x0], p[y1 * sx + x1], fx);
    return lerp(a, b, fy);
}

// ------------------------------------------------------------------ BVH (oracle-private: median split) + brute force
struct Hit { float t, u, v; uint32_t prim, inst; int gid; };

void tri_bounds(const Tri& t, V3& lo, V3& hi) {
    V3 a = t.v0, b = t.v0 + t.e1, c = t.v0 + t.e2;
    lo = v3(std::min(a.x, std::min(b.x, c.x)), std::min(a.y, std::min(b.y, c.y)), std::min(a.z, std::min(b.z, c.z)));
    hi = v3(std::max(a.x, std::max(b.x, c.x)), std::max(a.y, std::max(b.y, c.y)), std::max(a.z, std::max(b.z, c.z)));
}

int build_node(Oracle& o, int first, int count) {
    Node n;
    n.bmin = v3s(1e30f); n.bmax = v3s(-1e30f);
    V3 cmin = v3s(1e30f), cmax = v3s(-1e30f);
    for (int i = first; i < first + count; i++) {
        V3 lo, hi; tri_bounds(o.tris[o.tri_order[i]], lo, hi);
        n.bmin = v3(std::min(n.bmin.x, lo.x), std::min(n.bmin.y, lo.y), std::min(n.bmin.z, lo.z));
        n.bmax = v3(std::max(n.bmax.x, hi.x), std::max(n.bmax.y, hi.y), std::max(n.bmax.z, hi.z));
        V3 c = (lo + hi) * 0.5f;
        cmin = v3(std::min(cmin.x, c.x), std::min(cmin.y, c.y), std::min(cmin.z, c.z));
        cmax = v3(std::max(cmax.x, c.x), std::max(cmax.y, c.y), std::max(cmax.z, c.z));
    }
    // generous conservative padding: the oracle only needs never to cull a triangle the shared
    // ray_triangle() would accept.
    float m = std::max(std::max(fabsf(n.bmin.x), fabsf(n.bmax.x)),
                       std::max(std::max(fabsf(n.bmin.y), fabsf(n.bmax.y)), std::max(fabsf(n.bmin.z), fabsf(n.bmax.z))));
    float pad = 1e-4f * m + 1e-6f;
    n.bmin = n.bmin - v3s(pad); n.bmax = n.bmax + v3s(pad);
    n.left = n.right = -1; n.first = first; n.count = count;
    int id = (int)o.nodes.size();
    o.nodes.push_back(n);
    if (count > 4) {
        V3 e = cmax - cmin;
        int ax = (e.x >= e.y && e.x >= e.z) ? 0 : (e.y >= e.z ? 1 : 2);
        int mid = first + count / 2;
        auto key = [&](int ti) {
            V3 lo, hi; tri_bounds(o.tris[ti], lo, hi);
            V3 c = lo + hi;
            return ax == 0 ? c.x : (ax == 1 ? c.y : c.z);
        };
        std::nth_element(o.tri_order.begin() + first, o.tri_order.begin() + mid, o.tri_order.begin() + first + count,
                         [&](int a, int b) { float ka = key(a), kb = key(b); return ka < kb || (ka == kb && a < b); });
        int l = build_node(o, first, mid - first);
        int r = build_node(o, mid, first + count - mid);
        o.nodes[id].left = l; o.nodes[id].right = r; o.nodes[id].count = 0;
    }
    return id;
}

inline void consider(const Oracle& o, int gid, V3 org, V3 dir, float tmin, float tmax, Hit& best, Counters* c) {
    const Tri& tr = o.tris[gid];
    if (tr.skip) return;  // sliver: not intersectable (vpt_fp32.h triangle_degenerate)
    float t, u, v;
    if (c) c->tris++;
    if (ray_triangle(org, dir, tr.v0, tr.e1, tr.e2, tmin, tmax, &t, &u, &v) &&
        (!(o.P.flags & VPT_FLAG_LOCAL_HITS) || hit_is_local(org, dir, tr.v0, tr.e1, tr.e2, t))) {
        if (best.gid < 0 || t < best.t || (t == best.t && gid < best.gid)) {
            best.t = t; best.u = u; best.v = v; best.prim = tr.prim; best.inst = tr.inst; best.gid = gid;
        }
    }
}

// Debug ray log (orc_ray_log_*): every closest_hit() call of the calling thread while enabled.
struct LoggedRay { float o[3], tmin, d[3], tmax; float t; int gid; };
static thread_local std::vector<LoggedRay>* g_ray_log = nullptr;
bool closest_hit_impl(const Oracle& o, V3 org, V3 dir, float tmin, float tmax, Hit& best, Counters* c);
// Closest hit with tmin < t < tmax; ties -> smaller global triangle id.
bool closest_hit(const Oracle& o, V3 org, V3 dir, float tmin, float tmax, Hit& best, Counters* c) {
    bool r = closest_hit_impl(o, org, dir, tmin, tmax, best, c);
    if (g_ray_log) g_ray_log->push_back({{org.x, org.y, org.z}, tmin, {dir.x, dir.y, dir.z}, tmax, r ? best.t : -1.0f, r ? best.gid : -1});
    return r;
}
bool closest_hit_impl(const Oracle& o, V3 org, V3 dir, float tmin, float tmax, Hit& best, Counters* c) {
    best.gid = -1; best.t = tmax;
    if (o.brute_force || o.nodes.empty()) {
        for (int g = 0; g < (int)o.tris.size(); g++) consider(o, g, org, dir, tmin, tmax, best, c);
        return best.gid >= 0;
    }
    int stack[128]; int sp = 0; stack[sp++] = 0;
    double inv[3] = {1.0 / (double)dir.x, 1.0 / (double)dir.y, 1.0 / (double)dir.z};
    double og[3] = {org.x, org.y, org.z};
    while (sp) {
        const Node& n = o.nodes[stack[--sp]];
        if (c) c->nodes++;
        double lo = tmin, hi = (best.gid >= 0) ? (double)best.t : (double)tmax;
        const float bl[3] = {n.bmin.x, n.bmin.y, n.bmin.z}, bh[3] = {n.bmax.x, n.bmax.y, n.bmax.z};
        bool miss = false;
        for (int a = 0; a < 3; a++) {
            double t0 = ((double)bl[a] - og[a]) * inv[a], t1 = ((double)bh[a] - og[a]) * inv[a];
            if (t0 != t0 || t1 != t1) {  // 0*inf: origin on a slab plane of a parallel ray -> inside iff within slab
                if (og[a] < bl[a] || og[a] > bh[a]) { miss = true; }
                continue;
            }
            if (t0 > t1) std::swap(t0, t1);
            lo = std::max(lo, t0); hi = std::min(hi, t1);
        }
        if (miss || lo > hi * (1.0 + 1e-6) + 1e-9) continue;
        if (n.left < 0) {
            for (int i = n.first; i < n.first + n.count; i++) consider(o, o.tri_order[i], org, dir, tmin, tmax, best, c);
        } else {
            stack[sp++] = n.left; stack[sp++] = n.right;
        }
    }
    return best.gid >= 0;
}

// RTCommon.slang:47-64 (USE_RAY_QUERIES path): TMin 1e-4, TMax 1e6, closest committed hit.
// Without USE_RAY_QUERIES (RTCommon.slang:64-84, MissShadow.slang:4-9): TraceRay with ACCEPT_FIRST_HIT_AND_END_SEARCH | SKIP_CLOSEST_HIT_SHADER and
// the shadow miss shader, on the NORMALISED direction with TMin 1e-5, TMax 1000; "intersects" = payload.Depth stayed 0 = some triangle was hit.
// Nothing on that path writes payload.TriangleIdx / InstanceIdx (the closest-hit shader is skipped, MissShadow leaves them alone), so the
// callers' light-identity compare reads an UNDEFINED word upstream.  Pinned here as "never equal to a sampled light": an emissive-mesh NEE
// sample is never visible in this mode (its random draws still happen).  DESIGN.md §6.
bool does_ray_intersect(const Oracle& o, V3 org, V3 dir, uint32_t& tri, uint32_t& inst, Counters* c) {
    tri = 0; inst = 0;
    Hit h;
    if (c) c->shadow++;
    if (!(o.P.flags & VPT_FLAG_RAY_QUERIES)) {
        tri = 0xffffffffu; inst = 0xffffffffu;
        return closest_hit(o, org, normalize(dir), 0.00001f, 1000.0f, h, c);
    }
    if (closest_hit(o, org, dir, 0.0001f, 1000000.0f, h, c)) { tri = h.prim; inst = h.inst; return true; }
    return false;
}

// ------------------------------------------------------------------ Payload (RTCommon.slang:5-35, surface subset)
struct Payload {
    V3 origin, direction, bxdf;
    float pdf;
    V3 emitted;
    uint32_t depth;
    Rng rng;
    bool in_medium;
    float medium_density, medium_anisotropy;
    V3 medium_color, medium_emissive;
    uint32_t volume_depth;  // RTCommon.slang:34
    int color_channel;      // RTCommon.slang:31: -1 until the path scatters in the atmosphere
};

float power_heuristics(float a, float b) {  // RTCommon.slang:124-127
    return pow_(a, 2.0f) / (pow_(a, 2.0f) + pow_(b, 2.0f));
}
V2 direction_to_uv(V3 v) {  // RTCommon.slang:129-136; asin argument clamped (contract: no NaN coords)
    float gamma = asin_(clamp_(v.y, -1.0f, 1.0f));
    float theta = atan2_(v.x, -v.z);
    V2 uv; uv.x = theta * M_1_OVER_PI_F * 0.5f + 0.5f; uv.y = gamma * M_1_OVER_PI_F + 0.5f;
    return uv;
}

// ------------------------------------------------------------------ Surface (Surface.slang:26-147)
struct Surface {
    V3 pos; V2 uv;
    V3 N, T, B, Ng;
    vpt_vertex va, vb, vc;
    bool inside;
    V3 tangent_to_world(V3 v) const { return normalize((v.x * T + v.y * B) + v.z * N); }
    V3 world_to_tangent(V3 v) const { return normalize(v3(dot(v, T), dot(v, B), dot(v, N))); }
};
inline V3 P3(const float* p) { return v3(p[0], p[1], p[2]); }

void surface_init(const Oracle& o, Surface& s, uint32_t inst, uint32_t mesh, uint32_t prim, V3 bary, V3 raydir,
                  const Tex& normal_tex) {
    const std::vector<uint32_t>& idx = o.mindices[mesh];
    const std::vector<vpt_vertex>& vs = o.mverts[mesh];
    s.va = vs[idx[prim * 3 + 0]]; s.vb = vs[idx[prim * 3 + 1]]; s.vc = vs[idx[prim * 3 + 2]];
    const float* M = o.instances[inst].transform;
    const float* I = &o.inv3[(size_t)inst * 9];
    V3 p = (P3(s.va.position) * bary.x + P3(s.vb.position) * bary.y) + P3(s.vc.position) * bary.z;
    s.pos = mat_point(M, p);
    s.uv.x = (s.va.texcoord[0] * bary.x + s.vb.texcoord[0] * bary.y) + s.vc.texcoord[0] * bary.z;
    s.uv.y = (s.va.texcoord[1] * bary.x + s.vb.texcoord[1] * bary.y) + s.vc.texcoord[1] * bary.z;
    s.Ng = normalize(cross(P3(s.vb.position) - P3(s.va.position), P3(s.vc.position) - P3(s.va.position)));
    s.Ng = normalize(rowvec_mat3(s.Ng, I));
    bool geo_only = (o.P.flags & VPT_FLAG_GEOMETRY_NORMALS) != 0;
    if (geo_only) {
        s.N = s.Ng;
    } else {
        s.N = normalize((P3(s.va.normal) * bary.x + P3(s.vb.normal) * bary.y) + P3(s.vc.normal) * bary.z);
        s.N = normalize(rowvec_mat3(s.N, I));
    }
    V3 view = -raydir;
    if (dot(s.Ng, view) < 0.0f) { s.N = -s.N; s.Ng = -s.Ng; s.inside = true; } else { s.inside = false; }
    V3 up = fabs_(s.N.z) < 0.9999999f ? v3(0, 0, 1) : v3(1, 0, 0);
    s.T = normalize(cross(up, s.N));
    s.B = normalize(cross(s.N, s.T));
    if (!geo_only) {
        V4 nm = sample_tex(normal_tex, s.uv.x, s.uv.y);
        V3 nv = v3(nm.x * 2.0f - 1.0f, nm.y * 2.0f - 1.0f, nm.z * 2.0f - 1.0f);
        s.N = s.tangent_to_world(nv);
    }
    if (dot(s.N, view) < 0.0f) {
        float eps = 0.01f;
        s.N = normalize(s.N - view * (dot(s.N, view) - eps));
    }
    V3 pr = normalize(reflect(-view, s.N));
    if (dot(pr, s.Ng) < 0.0f) {
        float eps = 0.1f;
        float dp = dot(s.N, s.Ng);
        s.N = normalize(s.N + s.Ng * (eps + dp));
    }
    s.T = normalize(cross(s.N, up));
    s.B = normalize(cross(s.N, s.T));
}
void rotate_tangents(Surface& s, float deg) {  // Surface.slang:129-136
    float rot = deg * (M_PI_F / 180.0f);
    float sn, cs; sincos_(rot, &sn, &cs);
    s.T = (s.T * cs + cross(s.N, s.T) * sn) + (s.N * dot(s.N, s.T)) * (1.0f - cs);
    s.B = cross(s.T, s.N);
}

// ------------------------------------------------------------------ Material (Material.slang)
struct Eval { V3 bxdf; float pdf; };
struct Mat {
    vpt_material p;  // Properties (modified copy)
    float eta, ax, ay;
    const Oracle* o;
    bool ec;  // USE_ENERGY_COMPENSATION

    float dielectric_fresnel(float c) const {  // Material.slang:434-449
        float st2 = eta * eta * (1.0f - c * c);
        if (st2 > 1.0f) return 1.0f;
        float ct = sqrt_(max_(1.0f - st2, 0.0f));
        float rs = (eta * ct - c) / (eta * ct + c);
        float rp = (eta * c - ct) / (eta * c + ct);
        return 0.5f * (rs * rs + rp * rp);
    }
    float schlick(float vh) const { float m = clamp_(1.0f - vh, 0.0f, 1.0f); float m2 = m * m; return m2 * m2 * m; }
    float ggx_d(V3 h) const {  // 394-404
        float ax2 = ax * ax, ay2 = ay * ay;
        return 1.0f / (M_PI_F * ax * ay * pow_((h.x * h.x) / ax2 + (h.y * h.y) / ay2 + h.z * h.z, 2.0f));
    }
    float lambda(V3 v) const {  // 406-418
        float vz2 = fabs_(v.z) * fabs_(v.z);
        float ax2 = ax * ax, ay2 = ay * ay;
        float nom = -1.0f + sqrt_(1.0f + (ax2 * (v.x * v.x) + ay2 * (v.y * v.y)) / vz2);
        return nom / 2.0f;
    }
    float smith(V3 v) const { return 1.0f / (1.0f + lambda(v)); }

    Eval eval_reflection(V3 V, V3 L, V3 F) const {  // 331-351
        Eval e; e.bxdf = v3s(0.0f); e.pdf = 0.0f;
        if (L.z <= 1e-5f) return e;
        V3 H = normalize(V + L);
        float VdotH = dot(V, H);
        float D = ggx_d(H);
        float GV = smith(V), GL = smith(L);
        e.pdf = (GV * max_(VdotH, 0.0f) * D / V.z) / (4.0f * VdotH);
        e.bxdf = ((F * D) * GV) * GL / (4.0f * V.z);
        return e;
    }
    Eval eval_refraction(V3 V, V3 L, V3 F) const {  // 359-387
        Eval e; e.bxdf = v3s(0.0f); e.pdf = 0.0f;
        if (L.z >= 1e-5f) return e;
        V3 H = normalize(V * eta + L);
        if (H.z < 0.0f) H = -H;
        float VdotH = dot(V, H), LdotH = dot(L, H);
        float D = ggx_d(H);
        float GV = smith(V), GL = smith(L);
        float G = GV * GL;
        float den = LdotH + eta * VdotH;
        float den2 = den * den;
        float eta2 = eta * eta;
        float jac = (eta2 * fabs_(LdotH)) / den2;
        e.pdf = (GV * fabs_(VdotH) * D / V.z) * jac;
        e.bxdf = (((F * D) * G) * eta2 / den2) * (fabs_(VdotH) * fabs_(LdotH) / fabs_(V.z));
        return e;
    }
    float lut_reflect(V3 V) const { return sample_lut(o->lutR, 64, 64, 32, V.z, p.roughness, p.anisotropy * 32.0f); }
    Eval eval_diffuse(V3, V3 L) const {  // 256-264
        Eval e;
        float pdf = L.z * M_1_OVER_PI_F;
        e.bxdf = (P3(p.base_color) * M_1_OVER_PI_F) * L.z;
        e.pdf = pdf * (L.z > 0.0f ? 1.0f : 0.0f);
        return e;
    }
    Eval eval_metallic(V3 V, V3 L) const {  // 266-283
        V3 H = normalize(V + L);
        V3 F = lerp(P3(p.base_color), P3(p.specular_color), schlick(dot(V, H)));
        Eval e = eval_reflection(V, L, F);
        if (ec) {
            float c = lut_reflect(V);
            c = (1.0f - c) / c;
            e.bxdf = (v3s(1.0f) + P3(p.base_color) * c) * e.bxdf;
        }
        return e;
    }
    Eval eval_dielectric_reflection(V3 V, V3 L) const {  // 285-298
        Eval e = eval_reflection(V, L, P3(p.specular_color));
        if (ec) { float c = lut_reflect(V); e.bxdf = e.bxdf / c; }
        return e;
    }
    void lobe_probs(float& pm, float& pd, float& pg) const {
        pm = p.metallic;
        pd = (1.0f - p.metallic) * (1.0f - p.transmission);
        pg = (1.0f - p.metallic) * p.transmission;
        float sum = pm + pd + pg;
        pm /= sum; pd /= sum; pg /= sum;
    }
    Eval eval_bsdf(V3 V, V3 L) const {  // 167-254
        float pm, pd, pg; lobe_probs(pm, pd, pg);
        bool refracted = L.z < 0.0f;
        V3 H; bool valid_refr = false;
        if (refracted) {
            H = normalize(V * eta + L);
            if (H.z < 0.0f) H = -H;
            float VdotH = dot(V, H), LdotH = dot(L, H);
            valid_refr = (VdotH > 0.0f && LdotH < 0.0f) || (VdotH < 0.0f && LdotH > 0.0f);
        } else {
            H = normalize(V + L);
        }
        float F = dielectric_fresnel(fabs_(dot(V, H)));
        Eval r; r.bxdf = v3s(0.0f); r.pdf = 0.0f;
        float gec = 0.0f;
        if (ec) {
            bool inside = eta > 1.0f;
            float layer = (clamp_(p.ior, 1.0001f, 2.0f) - 1.0f) * 32.0f;
            gec = sample_lut(inside ? o->lutI : o->lutO, 128, 128, 32, pow_(V.z, 1.0f / 2.0f), p.roughness, layer);
        }
        if (!refracted) {
            Eval m = eval_metallic(V, L);
            r.bxdf = r.bxdf + m.bxdf * pm; r.pdf += m.pdf * pm;
            Eval d = eval_diffuse(V, L);
            r.bxdf = r.bxdf + d.bxdf * pd * (1.0f - F); r.pdf += d.pdf * pd * (1.0f - F);
            Eval s = eval_dielectric_reflection(V, L);
            r.bxdf = r.bxdf + s.bxdf * pd * F; r.pdf += s.pdf * pd * F;
            Eval g = eval_reflection(V, L, P3(p.specular_color));
            if (ec && gec > 0.01f) g.bxdf = g.bxdf / gec;
            r.bxdf = r.bxdf + g.bxdf * pg * F; r.pdf += g.pdf * pg * F;
        }
        if (refracted && valid_refr) {
            Eval g = eval_refraction(V, L, P3(p.base_color));
            if (ec && gec > 0.01f) g.bxdf = g.bxdf / gec;
            r.bxdf = r.bxdf + g.bxdf * pg * (1.0f - F); r.pdf += g.pdf * pg * (1.0f - F);
        }
        return r;
    }
};

void material_init(const Oracle& o, Mat& m, const vpt_material& src, const Surface& s) {  // Material.slang:39-87
    m.p = src; m.o = &o; m.ec = (o.P.flags & VPT_FLAG_ENERGY_COMPENSATION) != 0;
    V4 tb = sample_tex(o.textures[m.p.base_color_texture], s.uv.x, s.uv.y);
    m.p.ior = max_(m.p.ior, 1.000001f);
    m.p.base_color[0] *= pow_(tb.x, 2.2f); m.p.base_color[1] *= pow_(tb.y, 2.2f); m.p.base_color[2] *= pow_(tb.z, 2.2f);
    m.p.roughness *= sample_tex(o.textures[m.p.roughness_texture], s.uv.x, s.uv.y).x;
    m.p.metallic *= sample_tex(o.textures[m.p.metallic_texture], s.uv.x, s.uv.y).x;
    V4 te = sample_tex(o.textures[m.p.emissive_texture], s.uv.x, s.uv.y);
    m.p.emissive_color[0] *= te.x; m.p.emissive_color[1] *= te.y; m.p.emissive_color[2] *= te.z;
    float aspect = sqrt_(1.0f - sqrt_(m.p.anisotropy) * 0.9f);
    m.ax = max_(0.00001f, m.p.roughness / aspect);
    m.ay = max_(0.00001f, m.p.roughness * aspect);
    m.eta = s.inside ? m.p.ior : 1.0f / m.p.ior;
    if (o.P.flags & VPT_FLAG_FURNACE) {
        for (int i = 0; i < 3; i++) {
            m.p.base_color[i] = 1.0f; m.p.emissive_color[i] = 0.0f; m.p.specular_color[i] = 1.0f;
            m.p.medium_color[i] = 1.0f; m.p.medium_emissive_color[i] = 0.0f;
        }
    }
}

// ------------------------------------------------------------------ Sampler routines (Sampler.slang)
V2 random_circle(Rng& r) {  // 102-112
    float u1 = r.uf(), u2 = r.uf();
    float theta = 2.0f * M_PI_F * u1;
    float rad = sqrt_(u2);
    float s, c; sincos_(theta, &s, &c);
    V2 o; o.x = rad * c; o.y = rad * s; return o;
}
V3 random_sphere(Rng& r) {  // 114-133
    float u1 = r.uf(), u2 = r.uf();
    float theta = 2.0f * M_PI_F * u1;
    float z = 1.0f - 2.0f * u2;
    float rad = sqrt_(1.0f - z * z);
    float s, c; sincos_(theta, &s, &c);
    return v3(rad * c, rad * s, z);
}
V3 ggx_sample(Rng& r, V3 Ve, float ax, float ay) {  // 141-166
    float u1 = r.uf(), u2 = r.uf();
    V3 Vh = normalize(v3(ax * Ve.x, ay * Ve.y, fabs_(Ve.z)));
    float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
    V3 T1 = lensq > 0.0f ? v3(-Vh.y, Vh.x, 0.0f) * (1.0f / sqrt_(lensq)) : v3(1, 0, 0);
    V3 T2 = cross(Vh, T1);
    float rad = sqrt_(u1);
    float phi = 2.0f * M_PI_F * u2;
    float sp, cp; sincos_(phi, &sp, &cp);
    float t1 = rad * cp, t2 = rad * sp;
    float s = 0.5f * (1.0f + Vh.z);
    t2 = (1.0f - s) * sqrt_(1.0f - t1 * t1) + s * t2;
    V3 Nh = (t1 * T1 + t2 * T2) + sqrt_(max_(0.0f, 1.0f - t1 * t1 - t2 * t2)) * Vh;
    return normalize(v3(ax * Nh.x, ay * Nh.y, max_(0.0f, Nh.z)));
}
V3 sample_hg(Rng& r, V3 dir, float G) {  // 168-193
    float r1 = r.uf(), r2 = r.uf();
    float ct;
    if (fabs_(G) < 1e-5f) {
        ct = 2.0f * r1 - 1.0f;
    } else {
        float sq = (1.0f - G * G) / (1.0f - G + 2.0f * G * r1);
        ct = (1.0f + G * G - sq * sq) / (2.0f * G);
    }
    float phi = 2.0f * M_PI_F * r2;
    float st = sqrt_(1.0f - ct * ct);
    float sp, cp; sincos_(phi, &sp, &cp);
    V3 nd = v3(st * cp, st * sp, ct);
    V3 up = fabs_(dir.y) < 0.9999999f ? v3(0, 1, 0) : v3(0, 0, 1);
    V3 t = normalize(cross(up, dir));
    V3 b = cross(dir, t);
    return normalize((nd.x * t + nd.y * b) + nd.z * dir);
}
void importance_sample_env(const Oracle& o, Rng& r, V3& to_light, V4& out) {  // 286-346
    float x0 = r.uf(), x1 = r.uf(), x2 = r.uf();
    uint32_t w = o.envW, h = o.envH;
    uint32_t size = w * h;
    uint32_t idx = std::min((uint32_t)(x0 * (float)size), size - 1);
    Alias e = o.alias[idx];
    uint32_t env_idx;
    if (x1 < e.importance) { env_idx = idx; x1 /= e.importance; }
    else { env_idx = e.alias; x1 = (x1 - e.importance) / (1.0f - e.importance); }
    uint32_t px = env_idx % w, py = env_idx / w;
    float u = ((float)px + x1) / (float)w;
    float phi = u * (2.0f * M_PI_F) - M_PI_F;
    float sp, cp; sincos_(phi, &sp, &cp);
    float step = M_PI_F / (float)h;
    float theta0 = (float)py * step;
    float ct = cos_(theta0) * (1.0f - x2) + cos_(theta0 + step) * x2;
    float theta = acos_(clamp_(ct, -1.0f, 1.0f));
    float st = sin_(theta);
    float v = theta * M_1_OVER_PI_F;
    to_light = v3(sp * st, -ct, (-cp) * st);
    float az = o.P.sky_azimuth / 180.0f * M_PI_F, al = o.P.sky_altitude / 180.0f * M_PI_F;
    to_light = rotate(to_light, v3(0, 1, 0), az);
    to_light = rotate(to_light, v3(1, 0, 0), al);
    out = sample_f4(o.env.data(), (int)w, (int)h, u, v, true);
    out.x *= o.P.sky_intensity; out.y *= o.P.sky_intensity; out.z *= o.P.sky_intensity;
}
void sample_emissive(const Oracle& o, Rng& r, V3 pos, V3& to_light, V4& cpdf, uint32_t& tri, uint32_t& inst) {  // 348-422
    tri = 0xffffffffu; inst = 0xffffffffu;
    uint32_t n = (uint32_t)o.emissive.size();
    if (n == 0) { to_light = v3s(0.0f); cpdf.x = cpdf.y = cpdf.z = cpdf.w = 0.0f; return; }
    uint32_t mi = std::min((uint32_t)floor_(r.uf() * (float)n), n - 1);
    const Emissive& em = o.emissive[mi];
    inst = em.instance;
    uint32_t ti = std::min((uint32_t)floor_(r.uf() * (float)em.tri_count), em.tri_count - 1);
    tri = ti;
    const std::vector<uint32_t>& idx = o.mindices[em.mesh];
    const std::vector<vpt_vertex>& vs = o.mverts[em.mesh];
    const vpt_vertex &a = vs[idx[ti * 3]], &b = vs[idx[ti * 3 + 1]], &c = vs[idx[ti * 3 + 2]];
    V3 p0 = mat_point(em.xform, P3(a.position)), p1 = mat_point(em.xform, P3(b.position)), p2 = mat_point(em.xform, P3(c.position));
    float x0 = r.uf(), x1 = r.uf();
    float su = sqrt_(x0);
    float b0 = 1.0f - su, b1 = x1 * su, b2 = 1.0f - b0 - b1;
    V3 tp = (b0 * p0 + b1 * p1) + b2 * p2;
    float uu = (b0 * a.texcoord[0] + b1 * b.texcoord[0]) + b2 * c.texcoord[0];
    float vv = (b0 * a.texcoord[1] + b1 * b.texcoord[1]) + b2 * c.texcoord[1];
    to_light = normalize(tp - pos);
    V3 nrm = normalize(cross(p2 - p0, p1 - p0));
    float area = length(cross(p1 - p0, p2 - p0)) * 0.5f;
    float d2 = dot(tp - pos, tp - pos);
    float ct = fabs_(dot(nrm, to_light));
    cpdf.w = d2 / ((float)n * (float)em.tri_count * area * ct);
    const vpt_material& m = o.materials[em.material];
    V4 te = sample_tex(o.textures[m.emissive_texture], uu, vv);
    cpdf.x = m.emissive_color[0] * te.x; cpdf.y = m.emissive_color[1] * te.y; cpdf.z = m.emissive_color[2] * te.z;
}

// ------------------------------------------------------------------ Atmosphere (Atmosphere.slang, Sampler.slang:431-462, RTCommon.slang:174-211)
const float C_RAYLEIGH[3] = {5.802f * 1e-6f, 13.558f * 1e-6f, 33.100f * 1e-6f};  // Atmosphere.slang:7-11
const float C_MIE_SCATTERING = 3.996f * 1e-6f, C_MIE_ABSORPTION = 4.40f * 1e-6f;
const float C_MIE = C_MIE_SCATTERING + C_MIE_ABSORPTION;
const float C_OZONE[3] = {0.650f * 1e-6f, 1.881f * 1e-6f, 0.085f * 1e-6f};
V2 intersect_sphere(V3 org, V3 dir, V3 center, float radius) {  // RTCommon.slang:174-192
    org = org - center;
    float a = dot(dir, dir);
    float b = 2.0f * dot(org, dir);
    float c = dot(org, org) - radius * radius;
    float disc = b * b - 4.0f * a * c;
    V2 r;
    if (disc < 0.0f) { r.x = -1.0f; r.y = -1.0f; return r; }
    r.x = (-b - sqrt_(disc)) / (2.0f * a);
    r.y = (-b + sqrt_(disc)) / (2.0f * a);
    return r;
}
float rayleigh_phase(V3 V, V3 L) { float ct = dot(V, L); return (3.0f / (16.0f * M_PI_F)) * (1.0f + ct * ct); }  // RTCommon.slang:197-201
float phase_mie(V3 V, V3 L) {  // RTCommon.slang:204-211 with g = 0.85
    float ct = dot(V, L);
    float g = min_(0.85f, 0.9381f);
    float k = 1.55f * g - 0.55f * g * g * g;
    float kc = k * ct;
    return (1.0f - k * k) / ((4.0f * M_PI_F) * (1.0f - kc) * (1.0f - kc));
}
V3 sample_rayleigh(Rng& r, V3 dir) {  // Sampler.slang:195-215
    float r1 = r.uf(), r2 = r.uf();
    float u = -pow_(2.0f * (2.0f * r1 - 1.0f) + sqrt_(4.0f * pow_(2.0f * r1 - 1.0f, 2.0f) + 1.0f), 1.0f / 3.0f);
    float ct = u - (1.0f / u);
    float phi = 2.0f * M_PI_F * r2;
    float st = sqrt_(1.0f - ct * ct);
    float sp, cp; sincos_(phi, &sp, &cp);
    V3 nd = v3(st * cp, st * sp, ct);
    V3 up = fabs_(dir.y) < 0.9999999f ? v3(0, 1, 0) : v3(0, 0, 1);
    V3 t = normalize(cross(up, dir));
    V3 b = cross(dir, t);
    return normalize((nd.x * t + nd.y * b) + nd.z * dir);
}
void sample_sun_disk(const Oracle& o, Rng& r, float sun_theta, V3& to_light, V4& cpdf) {  // Sampler.slang:431-462
    float az = o.P.sky_azimuth / 180.0f * M_PI_F, al = o.P.sky_altitude / 180.0f * M_PI_F;
    V3 sun = rotate(v3(0.0f, 0.0f, -1.0f), v3(1.0f, 0.0f, 0.0f), al);
    sun = rotate(sun, v3(0.0f, 1.0f, 0.0f), az);
    float ctm = cos_(sun_theta);
    float phi = 2.0f * M_PI_F * r.uf();
    float ct = lerp(ctm, 1.0f, r.uf());
    float st = sqrt_(1.0f - ct * ct);
    float sp, cp; sincos_(phi, &sp, &cp);
    V3 local = v3(cp * st, sp * st, ct);
    V3 w = normalize(sun);
    V3 up = fabs_(w.z) < 0.999f ? v3(0, 0, 1) : v3(1, 0, 0);
    V3 u = normalize(cross(up, w));
    V3 v = cross(w, u);
    to_light = (u * local.x + v * local.y) + w * local.z;
    float solid = 2.0f * M_PI_F * (1.0f - ctm);
    cpdf.w = 1.0f / solid;
    V3 c = (2e5f * P3(o.atm.sun_color)) * o.P.sky_intensity;
    cpdf.x = c.x; cpdf.y = c.y; cpdf.z = c.z;
}
// Sampler::ImportanceSampleSky, Sampler.slang:464-476
void importance_sample_sky(const Oracle& o, Rng& r, V3& to_light, V4& out) {
    if (o.atm_on) sample_sun_disk(o, r, 0.004675f, to_light, out);
    else importance_sample_env(o, r, to_light, out);
}
float atmosphere_height(const Oracle& o, V3 p) { return length(p - P3(o.atm.planet_position)) - o.atm.planet_radius; }  // Atmosphere.slang:13-16
float rayleigh_density(const Oracle& o, float h) { return exp_(-h / o.atm.rayleigh_density_falloff); }
float mie_density(const Oracle& o, float h) { return exp_(-h / o.atm.mie_density_falloff); }
float ozone_density(const Oracle& o, float h) { return exp_(-(fabs_(h - o.atm.ozone_peak) / o.atm.ozone_density_falloff)); }
struct AtmCoef { float ray, mie, ozo, majorant; };
AtmCoef atmosphere_coefficients(const Oracle& o, int ch) {  // Atmosphere.slang:52-60 == 151-155
    AtmCoef k;
    k.ray = C_RAYLEIGH[ch] * o.atm.rayleigh_multiplier[ch];
    k.mie = C_MIE * o.atm.mie_multiplier[ch];
    k.ozo = C_OZONE[ch] * o.atm.ozone_multiplier[ch];
    k.majorant = (rayleigh_density(o, 0.0f) * k.ray + mie_density(o, 0.0f) * k.mie) + ozone_density(o, o.atm.ozone_peak) * k.ozo;
    return k;
}
// CalculateTransmittanceThroughAtmosphere, Atmosphere.slang:33-107: float3 with only `ch` set (ratio tracking + roulette)
V3 atmosphere_transmittance(const Oracle& o, Rng& r, V3 org, V3 dir, int ch) {
    V2 planet = intersect_sphere(org, dir, P3(o.atm.planet_position), o.atm.planet_radius);
    if (planet.y > 0.0f) return v3s(0.0f);
    V2 at = intersect_sphere(org, dir, P3(o.atm.planet_position), o.atm.planet_radius + o.atm.atmosphere_height);
    float tmin = max_(at.x, 0.0f), tmax = at.y;
    if (tmax < 0.0f) return v3s(1.0f);
    AtmCoef k = atmosphere_coefficients(o, ch);
    if (k.majorant <= 0.0f) return v3s(1.0f);
    float t = 0.0f, tr = 1.0f;
    for (int i = 0; i < 1000; i++) {
        float dt = -log_(1.0f - r.uf()) / k.majorant;
        t += dt;
        if (t >= tmax - tmin) break;
        float h = atmosphere_height(o, org + dir * (t + tmin));
        if (h < 0.0f) break;
        float dr = rayleigh_density(o, h) * k.ray, dm = mie_density(o, h) * k.mie, dz = ozone_density(o, h) * k.ozo;
        tr *= 1.0f - (dr + dm + dz) / k.majorant;
        float p = tr;
        if (r.uf() > p) { tr = 0.0f; break; }
        tr /= p;
    }
    V3 out = v3s(0.0f);
    if (ch == 0) out.x = tr; else if (ch == 1) out.y = tr; else out.z = tr;
    return out;
}
// SampleAtmosphereScatterDistance, Atmosphere.slang:117-201: delta tracking; comp 0 Rayleigh, 1 Mie, 2 ozone, -1 none
float atmosphere_scatter_distance(const Oracle& o, Rng& r, V3 org, V3 dir, int ch, int& comp) {
    V2 at = intersect_sphere(org, dir, P3(o.atm.planet_position), o.atm.planet_radius + o.atm.atmosphere_height);
    float tmin_a = max_(at.x, 0.0f), tmax_a = at.y;
    comp = -1;
    V2 planet = intersect_sphere(org, dir, P3(o.atm.planet_position), o.atm.planet_radius);
    float tmin_p = planet.x;
    if (tmax_a < 0.0f) return -1.0f;
    AtmCoef k = atmosphere_coefficients(o, ch);
    if (k.majorant <= 0.0f) return -1.0f;
    float t = tmin_a;
    for (int i = 0; i < 1000; i++) {
        float dt = -log_(1.0f - r.uf()) / k.majorant;
        t += dt;
        if (t >= tmax_a) break;
        if (tmin_p > 0.0f && t >= tmin_p) break;
        float h = atmosphere_height(o, org + dir * t);
        float dr = rayleigh_density(o, h) * k.ray, dm = mie_density(o, h) * k.mie, dz = ozone_density(o, h) * k.ozo;
        float dens = (dr + dm) + dz;
        if (dens / k.majorant < r.uf()) continue;
        float pr = dr / dens, pm = dm / dens;
        float x = r.uf();
        if (x <= pr) comp = 0; else if (x <= pr + pm) comp = 1; else comp = 2;
        return t;
    }
    return -1.0f;
}
// The per-channel product the surface / volume NEE applies to a sky sample (ClosestHit.slang:335-349 == RayGen.slang:328-343)
V3 nee_atmosphere_transmittance(const Oracle& o, Rng& r, V3 tr, V3 org, V3 dir, int color_channel) {
    if (color_channel == -1) {
        tr.x *= atmosphere_transmittance(o, r, org, dir, 0).x;
        tr.y *= atmosphere_transmittance(o, r, org, dir, 1).y;
        tr.z *= atmosphere_transmittance(o, r, org, dir, 2).z;
        return tr;
    }
    return tr * atmosphere_transmittance(o, r, org, dir, color_channel);
}

// ------------------------------------------------------------------ Volumes (Volume.slang / RayGen.slang:162-380, homogeneous boxes)
V3 sample_draine(Rng& r, V3 dir, float g, float a) {  // Sampler.slang:217-266
    float r1 = r.uf(), r2 = r.uf();
    float ct;
    if (fabs_(g) < 1e-5f) {
        ct = 2.0f * r1 - 1.0f;
    } else if (fabs_(a) < 1e-5f) {
        float sq = (1.0f - g * g) / (1.0f - g + 2.0f * g * r1);
        ct = (1.0f + g * g - sq * sq) / (2.0f * g);
    } else {
        const float g2 = g * g, g3 = g * g2, g4 = g2 * g2, g6 = g2 * g4;
        const float pgp1_2 = (1.0f + g2) * (1.0f + g2);
        const float T1a = -a + a * g4;
        const float T1a3 = T1a * T1a * T1a;
        const float T2 = -1296.0f * (-1.0f + g2) * (a - a * g2) * (T1a) * (4.0f * g2 + a * pgp1_2);
        const float T3 = 3.0f * g2 * (1.0f + g * (-1.0f + 2.0f * r1)) + a * (2.0f + g2 + g3 * (1.0f + 2.0f * g2) * (-1.0f + 2.0f * r1));
        const float T4a = 432.0f * T1a3 + T2 + 432.0f * (a - a * g2) * T3 * T3;
        const float T4b = -144.0f * a * g2 + 288.0f * a * g4 - 144.0f * a * g6;
        const float T4b3 = T4b * T4b * T4b;
        const float T4 = T4a + sqrt_(-4.0f * T4b3 + T4a * T4a);
        const float T4p3 = pow_(T4, 1.0f / 3.0f);
        const float T6 = (2.0f * T1a + (48.0f * pow_(2.0f, 1.0f / 3.0f) * (-(a * g2) + 2.0f * a * g4 - a * g6)) / T4p3 + T4p3 / (3.0f * pow_(2.0f, 1.0f / 3.0f))) / (a - a * g2);
        const float T5 = 6.0f * (1.0f + g2) + T6;
        ct = (1.0f + g2 - pow_(-0.5f * sqrt_(T5) + sqrt_(6.0f * (1.0f + g2) - (8.0f * T3) / (a * (-1.0f + g2) * sqrt_(T5)) - T6) / 2.0f, 2.0f)) / (2.0f * g);
    }
    float phi = 2.0f * M_PI_F * r2;
    float st = sqrt_(1.0f - ct * ct);
    float sp, cp; sincos_(phi, &sp, &cp);
    V3 nd = v3(st * cp, st * sp, ct);
    V3 up = fabs_(dir.y) < 0.9999999f ? v3(0, 1, 0) : v3(0, 0, 1);
    V3 t = normalize(cross(up, dir));
    V3 b = cross(dir, t);
    return normalize((nd.x * t + nd.y * b) + nd.z * dir);
}
struct HgDraineFit { float ghg, gd, alpha_d, w_d; };
HgDraineFit hg_draine_fit(float d) {  // Sampler.slang:271-274 == Volume.slang:397-400
    HgDraineFit f;
    f.ghg = exp_(-(0.0990567f / (d - 1.67154f)));
    f.gd = exp_(-(2.20679f / (d + 3.91029f)) - 0.428934f);
    f.alpha_d = exp_(3.62489f - (8.29288f / (d + 5.52825f)));
    f.w_d = exp_(-(0.599085f / (d - 0.641583f)) - 0.665888f);
    return f;
}
V3 sample_hg_plus_draine(Rng& r, V3 dir, float d, uint32_t depth) {  // Sampler.slang:268-284
    HgDraineFit f = hg_draine_fit(d);
    float ghg = pow_(max_(f.ghg, 0.0f), 1.0f + (float)depth);
    float gd = pow_(max_(f.gd, 0.0f), 1.0f + (float)depth);
    float u = r.uf();
    if (u < f.w_d) return sample_hg(r, dir, ghg);
    return sample_draine(r, dir, gd, f.alpha_d);
}
float phase_hg(V3 V, V3 L, float g) {  // RTCommon.slang:213-220
    if (g == 0.0f) return 1.0f / (4.0f * M_PI_F);
    float ct = dot(V, L);
    return (1.0f / (4.0f * M_PI_F)) * ((1.0f - g * g) / pow_(1.0f + g * g - 2.0f * g * ct, 1.5f));
}
float phase_draine(V3 V, V3 L, float g, float a) {  // RTCommon.slang:222-227
    float ct = dot(V, L);
    return ((1.0f - g * g) * (1.0f + a * ct * ct)) / (4.0f * (1.0f + (a * (1.0f + 2.0f * g * g)) / 3.0f) * M_PI_F * pow_(1.0f + g * g - 2.0f * g * ct, 1.5f));
}
struct VolIsect { float tn, tf; };
VolIsect ray_aabb(V3 org, V3 dir, const float* bmin, const float* bmax) {  // Volume.slang:183-207
    V3 inv = v3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
    V3 t0 = (P3(bmin) - org) * inv, t1 = (P3(bmax) - org) * inv;
    V3 ts = v3(min_(t0.x, t1.x), min_(t0.y, t1.y), min_(t0.z, t1.z));
    V3 tb = v3(max_(t0.x, t1.x), max_(t0.y, t1.y), max_(t0.z, t1.z));
    VolIsect r;
    r.tn = max_(max_(ts.x, ts.y), max_(ts.x, ts.z));
    r.tf = min_(min_(tb.x, tb.y), min_(tb.x, tb.z));
    if (r.tf < 0.0f || r.tn > r.tf) { r.tn = -1.0f; r.tf = -1.0f; }
    return r;
}
float effective_anisotropy(const vpt_volume& v, float depth) {  // Volume.slang:141-147
    if (v.approximated_scattering != 0) {
        float s = v.anisotropy > 0.0f ? 1.0f : (v.anisotropy < 0.0f ? -1.0f : 0.0f);
        return pow_(fabs_(v.anisotropy), 1.0f + depth) * s;
    }
    return v.anisotropy;
}
V3 volume_scatter_direction(const Oracle& o, const vpt_volume& v, V3 dir, Rng& r, uint32_t depth) {  // Volume.slang:350-368
    if (o.phase == VPT_PHASE_HENYEY_GREENSTEIN) return sample_hg(r, dir, effective_anisotropy(v, (float)depth));
    if (o.phase == VPT_PHASE_DRAINE) return sample_draine(r, dir, effective_anisotropy(v, (float)depth), v.alpha);
    return sample_hg_plus_draine(r, dir, v.droplet_size, depth);
}
float volume_phase(const Oracle& o, const vpt_volume& v, V3 V, V3 L, uint32_t depth) {  // Volume.slang:370-388, 395-406
    if (o.phase == VPT_PHASE_HENYEY_GREENSTEIN) return phase_hg(V, L, effective_anisotropy(v, (float)depth));
    if (o.phase == VPT_PHASE_DRAINE) return phase_draine(V, L, effective_anisotropy(v, (float)depth), v.alpha);
    HgDraineFit f = hg_draine_fit(v.droplet_size);
    return lerp(phase_hg(V, L, f.ghg), phase_draine(V, L, f.gd, f.alpha_d), f.w_d);
}
// ---- heterogeneous boxes: density from a dense grid (the reference's NanoVDB tree, densified) --------------------
float effective_density(const vpt_volume& v, float base, float depth) {  // Volume.slang:149-156
    if (v.approximated_scattering != 0) return base * pow_(v.approximated_scattering_falloff, depth);
    return base;
}
// SampleNanoVDBBuffer, Volume.slang:69-117, on a dense grid whose index box is [0, dim): normalise in the box, flip y,
// floor to a voxel, jitter by -1..1 voxels per axis (three raw PCG draws), clamp, value / max * sharpness in [0, 1].
float sample_density_grid(const Oracle& o, const vpt_volume& v, Rng& r, V3 x) {
    const Oracle::DensityGrid& g = o.grids[v.density_data_index];
    V3 n = (x - P3(v.corner_min)) / (P3(v.corner_max) - P3(v.corner_min));
    n.y = 1.0f - n.y;
    V3 gp = n * v3((float)g.dim[0], (float)g.dim[1], (float)g.dim[2]);
    int cx = f2i_clamped(floor_(gp.x), -1, (int)g.dim[0]), cy = f2i_clamped(floor_(gp.y), -1, (int)g.dim[1]), cz = f2i_clamped(floor_(gp.z), -1, (int)g.dim[2]);
    r.s = pcg_hash(r.s); cx += (int)(r.s % 3u) - 1;
    r.s = pcg_hash(r.s); cy += (int)(r.s % 3u) - 1;
    r.s = pcg_hash(r.s); cz += (int)(r.s % 3u) - 1;
    cx = std::min(std::max(cx, 0), (int)g.dim[0] - 1); cy = std::min(std::max(cy, 0), (int)g.dim[1] - 1); cz = std::min(std::max(cz, 0), (int)g.dim[2] - 1);
    float value = g.values[(size_t)cx + (size_t)cy * g.dim[0] + (size_t)cz * g.dim[0] * g.dim[1]];
    return clamp_(value / g.max_density * v.grid_sharpness, 0.0f, 1.0f);
}
struct VolBlock { int index; V3 lo, hi; };
struct VolTrav { V3 block_size; float eps, t_enter, t_exit; };
VolTrav make_traversal(const vpt_volume& v, VolIsect is) {  // Volume.slang:119-127
    VolTrav c;
    V3 ext = P3(v.corner_max) - P3(v.corner_min);
    c.block_size = ext / v3s(32.0f);
    c.eps = 0.0001f * max_(ext.x, max_(ext.y, ext.z));
    c.t_enter = max_(is.tn, 0.0f);
    c.t_exit = is.tf;
    return c;
}
VolBlock block_info(const vpt_volume& v, V3 pos, const VolTrav& c) {  // Volume.slang:129-147
    V3 rel = (pos - P3(v.corner_min)) / (P3(v.corner_max) - P3(v.corner_min));
    int ix = f2i_clamped(rel.x * 32.0f, 0, 31), iy = f2i_clamped(rel.y * 32.0f, 0, 31), iz = f2i_clamped(rel.z * 32.0f, 0, 31);
    VolBlock b;
    b.index = ix + iy * 32 + iz * 32 * 32;
    b.lo = P3(v.corner_min) + c.block_size * v3((float)ix, (float)iy, (float)iz);
    b.hi = b.lo + c.block_size;
    return b;
}
VolIsect ray_aabb3(V3 org, V3 dir, V3 lo, V3 hi) { float a[3] = {lo.x, lo.y, lo.z}, b[3] = {hi.x, hi.y, hi.z}; return ray_aabb(org, dir, a, b); }
V3 blackbody(float kelvin) {  // RTCommon.slang:139-172
    float temp = kelvin / 100.0f;
    float r, g, b;
    if (temp <= 66.0f) r = 255.0f; else r = 329.698727446f * pow_(temp - 60.0f, -0.1332047592f);
    if (temp <= 66.0f) g = 99.4708025861f * log_(temp) - 161.1195681661f; else g = 288.1221695283f * pow_(temp - 60.0f, -0.0755148492f);
    if (temp >= 66.0f) b = 255.0f; else if (temp <= 19.0f) b = 0.0f; else b = 138.5177312231f * log_(temp - 10.0f) - 305.0447927307f;
    V3 c = v3(r, g, b) / 255.0f;
    return v3(clamp_(c.x, 0.0f, 1.0f), clamp_(c.y, 0.0f, 1.0f), clamp_(c.z, 0.0f, 1.0f));
}
// GetEmissionFromTemperatureAtPoint, Volume.slang:233-258 (reads the volume's density grid: see vpt.h)
V3 temperature_emission(const Oracle& o, const vpt_volume& v, Rng& r, V3 x) {
    if (!v.has_temperature_data) return v3s(0.0f);
    float tn = sample_density_grid(o, v, r, x);
    V3 color;
    if (v.use_blackbody) color = blackbody(tn * (float)(v.kelvin_max - v.kelvin_min) + (float)v.kelvin_min);
    else color = P3(v.temperature_color);
    float intensity = pow_(tn, v.temperature_gamma) * v.temperature_scale;
    return intensity * v3(pow_(color.x, v.emissive_color_gamma), pow_(color.y, v.emissive_color_gamma), pow_(color.z, v.emissive_color_gamma));
}
// ProcessHeterogeneousVolumeScattering, Volume.slang:299-348: delta tracking block by block
float heterogeneous_scatter(const Oracle& o, const vpt_volume& v, V3 org, V3 dir, Rng& r, float depth, VolIsect is) {
    const Oracle::DensityGrid& g = o.grids[v.density_data_index];
    VolTrav c = make_traversal(v, is);
    VolBlock b = block_info(v, org + dir * (c.t_enter + c.eps), c);
    float t = 0.0f;
    for (int i = 0; i < 10000; i++) {
        V3 cur = org + dir * (c.t_enter + t + c.eps);
        VolIsect bi = ray_aabb3(cur, dir, b.lo, b.hi);
        float maxd = effective_density(v, g.block_max[b.index] * v.density, depth);
        float sd = -log_(r.uf()) / maxd;
        if (bi.tf <= 0.0f) {
            t += c.eps;
            if (c.t_enter + t > c.t_exit) return -1.0f;
            b = block_info(v, org + dir * (c.t_enter + t + c.eps), c);
            continue;
        }
        float to_exit = bi.tf - max_(bi.tn, 0.0f);
        if (sd > to_exit) {
            t += to_exit + c.eps;
            if (c.t_enter + t > c.t_exit) return -1.0f;
            b = block_info(v, org + dir * (c.t_enter + t + c.eps), c);
            continue;
        }
        t += sd;
        if (c.t_enter + t > c.t_exit) return -1.0f;
        V3 pos = org + dir * (c.t_enter + t);
        float dens = effective_density(v, sample_density_grid(o, v, r, pos) * v.density, depth);
        if (dens / maxd < r.uf()) continue;
        return c.t_enter + t;
    }
    return -1.0f;
}
// ProcessHeterogeneousVolumeTransmittance, Volume.slang:448-520: ratio tracking + roulette, block by block
float heterogeneous_transmittance(const Oracle& o, const vpt_volume& v, Rng& r, V3 org, V3 dir, float depth, VolIsect is) {
    const Oracle::DensityGrid& g = o.grids[v.density_data_index];
    VolTrav c = make_traversal(v, is);
    VolBlock b = block_info(v, org + dir * (c.t_enter + c.eps), c);
    float tr = 1.0f, t = 0.0f;
    for (int j = 0; j < 1000; j++) {
        V3 cur = org + dir * (c.t_enter + t + c.eps);
        VolIsect bi = ray_aabb3(cur, dir, b.lo, b.hi);
        float maxd = effective_density(v, g.block_max[b.index] * v.density, depth);
        float sd = -log_(r.uf()) / maxd;
        if (bi.tf <= 0.0f) {
            t += c.eps;
            if (c.t_enter + t > c.t_exit) break;
            b = block_info(v, org + dir * (c.t_enter + t + c.eps), c);
            continue;
        }
        float to_exit = bi.tf - max_(bi.tn, 0.0f);
        if (sd > to_exit) {
            t += to_exit + c.eps;
            if (c.t_enter + t > c.t_exit) break;
            b = block_info(v, org + dir * (c.t_enter + t + c.eps), c);
            continue;
        }
        t += sd;
        if (c.t_enter + t > c.t_exit) break;
        V3 pos = org + dir * (c.t_enter + t);
        float dens = effective_density(v, sample_density_grid(o, v, r, pos) * v.density, depth);
        tr *= 1.0f - (dens / maxd);
        float p = tr;
        if (r.uf() > p) return 0.0f;
        tr /= p;
    }
    return tr;
}
// Volume::CalculateVolumesTransmittance, Volume.slang:419-446: Beer-Lambert for homogeneous boxes, tracked (random
// draws) for heterogeneous ones
float volumes_transmittance(const Oracle& o, Rng& r, V3 org, V3 dir, float depth) {
    float tr = 1.0f;
    for (const vpt_volume& v : o.volumes) {
        VolIsect is = ray_aabb(org, dir, v.corner_min, v.corner_max);
        is.tn = max_(is.tn, 0.0f);
        if (v.density_data_index >= 0 && is.tf >= 0.0f) {
            tr *= heterogeneous_transmittance(o, v, r, org, dir, depth, is);
            if (tr <= 0.0f) return 0.0f;
        } else {
            float len = is.tf - is.tn;
            if (len > 0.0f) tr *= exp_(-v.density * len);
        }
    }
    return clamp_(tr, 0.0f, 1.0f);
}
// Volume::DoesRayScatterInVolume, Volume.slang:261-297
float does_ray_scatter(const Oracle& o, const vpt_volume& v, V3 org, V3 dir, Rng& r, float depth, float ignore_if_farther) {
    VolIsect is = ray_aabb(org, dir, v.corner_min, v.corner_max);
    if (is.tf < 0.0f) return -1.0f;
    if (ignore_if_farther >= 0.0f && is.tn > ignore_if_farther) return -1.0f;
    float inside = is.tf - max_(is.tn, 0.0f);
    if (inside <= 0.0f) return -1.0f;
    if (v.density_data_index >= 0) return heterogeneous_scatter(o, v, org, dir, r, depth, is);
    float sd = -log_(r.uf()) / v.density;  // Sampler.slang:425-428
    if (sd < inside) return max_(is.tn, 0.0f) + sd;
    return -1.0f;
}
void sample_emissive(const Oracle& o, Rng& r, V3 pos, V3& to_light, V4& cpdf, uint32_t& tri, uint32_t& inst);
void importance_sample_env(const Oracle& o, Rng& r, V3& to_light, V4& out);
// EvaluateVolumeScatteringEvent, RayGen.slang:265-380
void volume_scatter_event(const Oracle& o, Payload& p, float sd, int vi, Counters* c) {
    const vpt_volume& v = o.volumes[vi];
    p.origin = p.origin + p.direction * sd;
    p.emitted = P3(v.emissive_color) + temperature_emission(o, v, p.rng, p.origin);  // :268
    V3 to_sky = v3s(0.0f); V4 sky; sky.x = sky.y = sky.z = sky.w = 0.0f;
    if (o.P.flags & VPT_FLAG_SKY_MIS) {
        importance_sample_sky(o, p.rng, to_sky, sky);
        sky.x *= o.P.sky_intensity; sky.y *= o.P.sky_intensity; sky.z *= o.P.sky_intensity;
        uint32_t t0, t1;
        if (does_ray_intersect(o, p.origin, to_sky, t0, t1, c)) sky.x = sky.y = sky.z = sky.w = 0.0f;
    }
    V3 to_light = v3s(0.0f); V4 lc; lc.x = lc.y = lc.z = lc.w = 0.0f;
    if (o.P.flags & VPT_FLAG_MESH_MIS) {
        uint32_t lt, li;
        sample_emissive(o, p.rng, p.origin, to_light, lc, lt, li);
        if (lc.w > 0.0f) {  // with pdf 0 the sample is discarded at :354 whatever the visibility test says
            uint32_t ht, hi;
            does_ray_intersect(o, p.origin, to_light, ht, hi, c);  // a miss reports (0, 0), which is compared like a hit (:296-299)
            if (ht != lt || hi != li) lc.x = lc.y = lc.z = lc.w = 0.0f;
        }
    }
    V3 nd = volume_scatter_direction(o, v, p.direction, p.rng, p.volume_depth);
    float ph = volume_phase(o, v, p.direction, nd, p.volume_depth);
    V3 sbxdf = P3(v.color) * ph;
    if ((o.P.flags & VPT_FLAG_SKY_MIS) && sky.w > 0.0f) {
        float ps = volume_phase(o, v, p.direction, to_sky, p.volume_depth);
        V3 tr = v3s(volumes_transmittance(o, p.rng, p.origin, to_sky, (float)p.volume_depth));
        if (o.atm_on) tr = nee_atmosphere_transmittance(o, p.rng, tr, p.origin, to_sky, p.color_channel);  // :328-343
        V3 bx = P3(v.color) * ps;
        if (ps > 0.0f) p.emitted = p.emitted + ((tr * bx) * (v3(sky.x, sky.y, sky.z) / sky.w)) * power_heuristics(sky.w, ps);
    }
    if ((o.P.flags & VPT_FLAG_MESH_MIS) && lc.w > 0.0f) {
        float pl = volume_phase(o, v, p.direction, to_light, p.volume_depth);
        V3 tr = v3s(volumes_transmittance(o, p.rng, p.origin, to_light, (float)(p.volume_depth + 1u)));
        V3 bx = P3(v.color) * pl;
        if (pl > 0.0f) p.emitted = p.emitted + ((tr * bx) * (v3(lc.x, lc.y, lc.z) / lc.w)) * power_heuristics(lc.w, pl);
    }
    p.direction = nd;
    p.bxdf = sbxdf; p.pdf = ph;
    p.depth++;
    p.volume_depth++;
}
// EvaluateAtmosphereScatteringEvent, RayGen.slang:382-470
void atmosphere_scatter_event(const Oracle& o, Payload& p, float sd, int comp, Counters* c) {
    p.origin = p.origin + sd * p.direction;
    V3 nd;
    if (comp == 0) nd = sample_rayleigh(p.rng, p.direction);
    else if (comp == 1) nd = sample_hg(p.rng, p.direction, 0.85f);
    else nd = p.direction;
    if (o.P.flags & VPT_FLAG_SKY_MIS) {
        V3 to_sky; V4 cp;
        importance_sample_sky(o, p.rng, to_sky, cp);
        cp.x *= o.P.sky_intensity; cp.y *= o.P.sky_intensity; cp.z *= o.P.sky_intensity;
        uint32_t t0, t1;
        bool obscured = does_ray_intersect(o, p.origin, to_sky, t0, t1, c);
        V3 tr = v3s(1.0f);
        if (!obscured) {
            tr = atmosphere_transmittance(o, p.rng, p.origin, to_sky, p.color_channel);
            tr = tr * volumes_transmittance(o, p.rng, p.origin, to_sky, (float)p.volume_depth);
        } else {
            tr = v3s(0.0f);
        }
        V3 sun = v3(cp.x, cp.y, cp.z) / cp.w;
        if (comp == 0) {
            p.emitted = p.emitted + (rayleigh_phase(p.direction, to_sky) * tr) * sun;
            p.bxdf = v3s(rayleigh_phase(p.direction, nd)); p.pdf = rayleigh_phase(p.direction, nd);
        } else if (comp == 1) {
            p.emitted = p.emitted + (phase_hg(p.direction, to_sky, 0.85f) * tr) * sun;
            float att = C_MIE_ABSORPTION / C_MIE;
            p.bxdf = v3s(phase_hg(p.direction, nd, 0.85f) * (1.0f - att)); p.pdf = phase_hg(p.direction, nd, 0.85f);
        } else {
            p.bxdf = v3s(0.0f); p.pdf = 1.0f;
        }
    } else {
        if (comp == 0) {
            p.bxdf = v3s(rayleigh_phase(p.direction, nd)); p.pdf = rayleigh_phase(p.direction, nd);
        } else {  // Mie AND ozone (quirk: `componentHit == 0 ... else`)
            float att = C_MIE_ABSORPTION / C_MIE;
            p.bxdf = v3s(phase_mie(p.direction, nd) * att); p.pdf = phase_hg(p.direction, nd, 0.85f);
        }
    }
    p.direction = nd;
    p.depth++;
}
// ScatteredInVolume, RayGen.slang:162-263
bool scattered_in_volume(const Oracle& o, Payload& p, Counters* c) {
    const int n = (int)o.volumes.size();
    float dist[VPT_MAX_VOLUMES]; int idx[VPT_MAX_VOLUMES];
    for (int i = 0; i < n; i++) {
        VolIsect is = ray_aabb(p.origin, p.direction, o.volumes[i].corner_min, o.volumes[i].corner_max);
        dist[i] = max_(0.0f, is.tn); idx[i] = i;
    }
    for (int i = 0; i < n; i++)
        for (int j = i + 1; j < n; j++)
            if (dist[j] < dist[i]) { std::swap(dist[i], dist[j]); std::swap(idx[i], idx[j]); }
    float dgeo = -1.0f;  // GetDistanceToGeometry, RTCommon.slang:86-101: the payload direction as is; without USE_RAY_QUERIES (:103-117) normalised, TMax 1000
    Hit h;
    if (o.P.flags & VPT_FLAG_RAY_QUERIES) { if (closest_hit(o, p.origin, p.direction, 0.00001f, 1000000.0f, h, c)) dgeo = h.t; }
    else if (closest_hit(o, p.origin, normalize(p.direction), 0.00001f, 1000.0f, h, c)) dgeo = h.t;
    float sd = -1.0f; int sv = -1;
    for (int i = 0; i < n; i++) {
        float t = does_ray_scatter(o, o.volumes[idx[i]], p.origin, p.direction, p.rng, (float)p.depth, sd);
        if (t >= 0.0f && (t < sd || sd < 0.0f)) { sd = t; sv = idx[i]; }
    }
    int cc = p.color_channel, comp = -1;
    if (o.atm_on) {  // :212-236
        if (cc == -1) { float pick = p.rng.uf(); cc = pick < 0.33333f ? 0 : (pick < 0.66666f ? 1 : 2); }
        float ad = atmosphere_scatter_distance(o, p.rng, p.origin, p.direction, cc, comp);
        if (ad >= 0.0f && (ad < sd || sd < 0.0f)) { sd = ad; sv = -2; }
    }
    if (sd >= 0.0f && (dgeo < 0.0f || sd < dgeo)) {
        if (sv == -2) { p.color_channel = cc; atmosphere_scatter_event(o, p, sd, comp, c); }
        else volume_scatter_event(o, p, sd, sv, c);
        return true;
    }
    return false;
}

// ------------------------------------------------------------------ SampleBSDF (Material.slang:94-165)
struct BSample { V3 L, bxdf; float pdf; };
BSample sample_bsdf(const Mat& m, Rng& r, V3 V, V3 H) {
    float pm, pd, pg; m.lobe_probs(pm, pd, pg);
    float F = m.dielectric_fresnel(dot(V, H));
    float x1 = r.uf();
    V3 L; bool refracted = false;
    if (x1 < pm) {
        L = normalize(reflect(-V, H));
    } else if (x1 < pm + pd) {
        if (r.uf() < F) L = normalize(reflect(-V, H));
        else L = normalize(random_sphere(r) + v3(0, 0, 1));
    } else {
        if (r.uf() < F) L = normalize(reflect(-V, H));
        else { L = normalize(refract(-V, H, m.eta)); refracted = true; }
    }
    BSample s;
    if ((L.z < 0.0f && !refracted) || (refracted && L.z >= 0.0f)) { s.L = v3s(0.0f); s.bxdf = v3s(0.0f); s.pdf = 0.0f; return s; }
    Eval e = m.eval_bsdf(V, L);
    s.L = L; s.bxdf = e.bxdf; s.pdf = e.pdf;
    return s;
}

// ------------------------------------------------------------------ Miss (Miss.slang:8-77)
void miss_shader(const Oracle& o, Payload& p) {
    if (o.atm_on) { p.depth = MAX_DEPTH_C; return; }  // Miss.slang:11-14: the sky is in-scattered sunlight only
    V4 cp;
    bool show = (o.P.flags & VPT_FLAG_SHOW_ENV_DIRECTLY) != 0;
    if (show || p.depth > 0) {
        float az = o.P.sky_azimuth / 180.0f * M_PI_F, al = o.P.sky_altitude / 180.0f * M_PI_F;
        V3 d = rotate(p.direction, v3(1, 0, 0), -al);
        d = rotate(d, v3(0, 1, 0), -az);
        V2 uv = direction_to_uv(d);
        cp = sample_f4(o.env.data(), (int)o.envW, (int)o.envH, uv.x, uv.y, true);
    } else {
        cp.x = cp.y = cp.z = 0.0f; cp.w = 1.0f;
    }
    p.emitted = v3(cp.x, cp.y, cp.z) * o.P.sky_intensity;
    if (o.P.flags & VPT_FLAG_FURNACE) p.emitted = v3s(1.0f);
    if ((o.P.flags & VPT_FLAG_SKY_MIS) && p.depth > 0) p.emitted = p.emitted * power_heuristics(p.pdf, cp.w);
    p.depth = MAX_DEPTH_C;
}

// ------------------------------------------------------------------ ClosestHit (ClosestHit.slang:20-378)
void closest_hit_shader(const Oracle& o, Payload& p, V3 raydir, const Hit& hit, Counters* c) {
    p.emitted = v3s(0.0f);
    V3 bary = v3(1.0f - hit.u - hit.v, hit.u, hit.v);
    uint32_t inst = hit.inst;
    uint32_t mat_idx = o.instances[inst].material_index, mesh_idx = o.instances[inst].mesh_index;
    Surface s;
    surface_init(o, s, inst, mesh_idx, hit.prim, bary, raydir, o.textures[o.materials[mat_idx].normal_texture]);
    Mat m;
    material_init(o, m, o.materials[mat_idx], s);
    bool is_light = m.p.emissive_color[0] > 0.0f || m.p.emissive_color[1] > 0.0f || m.p.emissive_color[2] > 0.0f;
    rotate_tangents(s, m.p.anisotropy_rotation);

    if (p.in_medium) {  // 80-116
        float gd = length(p.origin - s.pos);
        if (p.medium_anisotropy == 1.0f) {
            // Beer-law branch writes payload.BxDF which line 323 overwrites: dead.
        } else {
            float sd = -log_(p.rng.uf()) / p.medium_density;
            if (sd < gd) {
                p.origin = p.origin + (sd * p.direction);
                p.direction = sample_hg(p.rng, p.direction, p.medium_anisotropy);
                p.bxdf = p.medium_color;
                return;  // PDF left stale, depth not incremented
            }
        }
    }

    V3 to_sky, to_sky_t = v3s(0.0f); V4 sky; sky.x = sky.y = sky.z = sky.w = 0.0f;
    bool can_sky = false;
    if (o.P.flags & VPT_FLAG_SKY_MIS) {  // 125-148
        importance_sample_sky(o, p.rng, to_sky, sky);
        sky.x *= o.P.sky_intensity; sky.y *= o.P.sky_intensity; sky.z *= o.P.sky_intensity;
        to_sky_t = s.world_to_tangent(to_sky);
        uint32_t t0, t1;
        can_sky = !does_ray_intersect(o, s.pos + s.N * 1e-5f, to_sky, t0, t1, c);
        if (!can_sky) sky.x = sky.y = sky.z = sky.w = 0.0f;
    }
    V3 to_light = v3s(0.0f), to_light_t = v3s(0.0f); V4 lc; lc.x = lc.y = lc.z = lc.w = 0.0f;
    bool can_light = false;
    if ((o.P.flags & VPT_FLAG_MESH_MIS) && !is_light) {  // 155-184
        uint32_t lt, li;
        sample_emissive(o, p.rng, s.pos, to_light, lc, lt, li);
        if (lc.w > 0.0f) {
            to_light_t = s.world_to_tangent(to_light);
            uint32_t ht, hi;
            bool found = does_ray_intersect(o, s.pos + to_light * 1e-2f, to_light, ht, hi, c);
            can_light = found ? (lt == ht && li == hi) : false;
            if (!can_light) lc.x = lc.y = lc.z = lc.w = 0.0f;
        }
    }
    V3 V = normalize(-raydir);
    V = s.world_to_tangent(V);
    V3 H = ggx_sample(p.rng, V, m.ax, m.ay);
    BSample bs = sample_bsdf(m, p.rng, V, H);
    bool was_refracted = bs.L.z < 0.0f;
    V3 scatter_world = s.tangent_to_world(bs.L);
    if (!was_refracted && dot(scatter_world, s.Ng) < 0.0f) { bs.pdf = 0.0f; bs.bxdf = v3s(0.0f); }
    if (was_refracted && s.inside) {
        p.in_medium = false;
    } else if (was_refracted && !s.inside) {
        p.in_medium = true;
        p.medium_color = P3(m.p.medium_color); p.medium_emissive = P3(m.p.medium_emissive_color);
        p.medium_anisotropy = m.p.medium_anisotropy; p.medium_density = m.p.medium_density;
    }
    Eval sky_e; sky_e.bxdf = v3s(0.0f); sky_e.pdf = 0.0f;
    if ((o.P.flags & VPT_FLAG_SKY_MIS) && can_sky) sky_e = m.eval_bsdf(V, to_sky_t);
    Eval light_e; light_e.bxdf = v3s(0.0f); light_e.pdf = 0.0f;
    if ((o.P.flags & VPT_FLAG_MESH_MIS) && can_light && !is_light) light_e = m.eval_bsdf(V, to_light_t);

    V3 Le = P3(m.p.emissive_color);
    if (o.P.flags & VPT_FLAG_MESH_MIS) {  // 265-311
        if (p.depth == 0 && is_light) {
            p.emitted = p.emitted + Le;
        } else if (is_light) {
            const float* M = o.instances[inst].transform;
            V3 a = mat_point(M, P3(s.va.position)), b = mat_point(M, P3(s.vb.position)), cc = mat_point(M, P3(s.vc.position));
            float area = length(cross(b - a, cc - a)) * 0.5f;
            float d2 = dot(s.pos - p.origin, s.pos - p.origin);
            float ct = fabs_(dot(s.N, normalize(p.origin - s.pos)));
            uint32_t tc = 0;  // emissiveMesh is uninitialised upstream if not found; 0 here
            for (size_t i = 0; i < o.emissive.size(); i++)
                if (o.emissive[i].instance == inst) { tc = o.emissive[i].tri_count; break; }
            float lp = (1.0f / (float)o.emissive.size()) * (1.0f / (float)tc) * (1.0f / area) * (d2 / ct);
            lp = max_(lp, o.P.emissive_pdf_bias);
            p.emitted = p.emitted + Le * power_heuristics(p.pdf, lp);
        }
    } else {
        p.emitted = p.emitted + Le;
    }
    p.origin = s.pos + s.N * (was_refracted ? -1e-3f : 1e-3f);
    p.direction = scatter_world;
    p.bxdf = bs.bxdf; p.pdf = bs.pdf;
    if (o.P.flags & VPT_FLAG_SKY_MIS) {  // 323-355
        if (can_sky) {
            V3 tr = v3s(volumes_transmittance(o, p.rng, p.origin, to_sky, 0.0f));  // :332-333, from the NEW origin; 1 without volumes
            if (o.atm_on) tr = nee_atmosphere_transmittance(o, p.rng, tr, p.origin, to_sky, p.color_channel);  // :335-349
            if (sky.w > 0.0f && sky_e.pdf > 0.0f)
                p.emitted = p.emitted + (sky_e.bxdf * tr * v3(sky.x, sky.y, sky.z) / sky.w) * power_heuristics(sky.w, sky_e.pdf);
        }
    }
    if ((o.P.flags & VPT_FLAG_MESH_MIS) && !is_light && can_light && lc.w > 0.0f && light_e.pdf > 0.0f) {
        V3 tr = v3s(volumes_transmittance(o, p.rng, p.origin, to_light, 0.0f));  // :364
        p.emitted = p.emitted + (light_e.bxdf * tr * v3(lc.x, lc.y, lc.z) / lc.w) * power_heuristics(lc.w, light_e.pdf);
    }
    bool invalid = bs.pdf <= 0.0f;
    p.depth = invalid ? (MAX_DEPTH_C + p.depth) : (p.depth + 1);
}

// ------------------------------------------------------------------ RayGen (RayGen.slang:9-160)
void raygen_pixel(Oracle& o, uint32_t lx, uint32_t ly, uint32_t frame_count, uint32_t seed, uint32_t chunk, Counters& c) {
    uint32_t S = o.P.screen_chunk_count;
    uint32_t x = lx * S + chunk % S, y = ly * S + chunk / S;
    if (x >= o.W || y >= o.H) return;
    Payload p;
    p.rng.s = y + o.W * x + seed;
    float* px = &o.image[((size_t)y * o.W + x) * 4];
    V3 prev = v3(px[0], px[1], px[2]);
    V3 acc = v3s(0.0f);
    for (uint32_t i = 0; i < o.P.samples_per_frame; i++) {
        float j0 = p.rng.uf(), j1 = p.rng.uf();
        float cx = ((float)x + 0.5f) + (j0 * (0.5f - -0.5f) + -0.5f);
        float cy = ((float)y + 0.5f) + (j1 * (0.5f - -0.5f) + -0.5f);
        float ux = cx / (float)o.W, uy = cy / (float)o.H;
        float dx = ux * 2.0f - 1.0f, dy = uy * 2.0f - 1.0f;
        V4 o4; o4.x = 0; o4.y = 0; o4.z = 0; o4.w = 1;
        V4 org4 = mat_v4(o.view_inv, o4);
        V3 origin = v3(org4.x, org4.y, org4.z);
        V4 t4; t4.x = dx; t4.y = dy; t4.z = 1.0f; t4.w = 1.0f;
        V4 tg = mat_v4(o.proj_inv, t4);
        V3 tn = normalize(v3(tg.x, tg.y, tg.z));
        V4 d4; d4.x = tn.x; d4.y = tn.y; d4.z = tn.z; d4.w = 0.0f;
        V4 dd = mat_v4(o.view_inv, d4);
        V3 direction = v3(dd.x, dd.y, dd.z);
        V3 focus = origin + direction * max_(o.P.focus_distance, 0.001f);
        V2 rc = random_circle(p.rng);
        float rox = rc.x * 0.5f * o.P.dof_strength, roy = rc.y * 0.5f * o.P.dof_strength;
        V3 right = v3(o.view_inv[0], o.view_inv[1], o.view_inv[2]);
        V3 upv = v3(o.view_inv[4], o.view_inv[5], o.view_inv[6]);
        origin = origin + (rox * right + roy * upv);
        direction = normalize(focus - origin);

        p.depth = 0; p.origin = origin; p.direction = direction; p.bxdf = v3s(1.0f); p.pdf = 1.0f;
        p.emitted = v3s(0.0f); p.in_medium = false; p.volume_depth = 0; p.color_channel = -1;
        p.medium_density = 0.0f; p.medium_anisotropy = 0.0f; p.medium_color = v3s(0.0f); p.medium_emissive = v3s(0.0f);
        V3 thr = v3s(1.0f), light = v3s(0.0f);
        for (; p.depth < o.P.max_depth;) {
            V3 rd = normalize(p.direction);
            p.emitted = v3s(0.0f);
            Hit h;
            c.closest++;  // one per loop iteration (the GPU counts path-bounces)
            // RayGen.slang:86-90; without volumes ScatteredInVolume only makes the unused distance query (a9: dropped)
            if (o.atm_on && atmosphere_height(o, p.origin) < 0.0f) break;  // RayGen.slang:76-84: below the planet's surface
            bool scattered = (!o.volumes.empty() || o.atm_on) && scattered_in_volume(o, p, &c);
            if (scattered) {}
            else if (closest_hit(o, p.origin, rd, 0.01f, 100000.0f, h, &c)) closest_hit_shader(o, p, rd, h, &c);
            else miss_shader(o, p);
            V3 contrib = p.emitted * thr;
            if (p.depth != 1) {
                float lum = dot(contrib, v3(0.212671f, 0.715160f, 0.072169f));
                float scale = o.P.max_luminance / max_(lum, o.P.max_luminance);
                contrib = contrib * scale;
            }
            light = light + contrib;
            thr = thr * (p.bxdf / p.pdf);
            float pr = max_(thr.x, max_(thr.y, thr.z));
            pr = min_(pr, 1.0f);
            if (pr < p.rng.uf()) break;
            thr = thr / pr;
        }
        c.samples++;
        bool ok = !isinf_(light.x) && !isinf_(light.y) && !isinf_(light.z) && !isnan_(light.x) && !isnan_(light.y) && !isnan_(light.z);
        if (ok) {  // RayGen.slang:116-129: a split path carries one colour channel
            if (p.color_channel == -1) acc = acc + light;
            else if (p.color_channel == 0) acc.x += light.x;
            else if (p.color_channel == 1) acc.y += light.y;
            else acc.z += light.z;
        }
    }
    acc = acc / (float)o.P.samples_per_frame;
    V3 color;
    if (frame_count > 0) {
        float a = 1.0f / (float)(frame_count + 1);
        color = lerp(prev, acc, a);
    } else {
        color = acc;
    }
    if (frame_count == 0 && chunk == 0) {
        for (uint32_t i = 0; i < S; i++)
            for (uint32_t j = 0; j < S; j++) {
                uint32_t qx = x + i, qy = y + j;
                if (qx < o.W && qy < o.H) {
                    float* q = &o.image[((size_t)qy * o.W + qx) * 4];
                    q[0] = color.x; q[1] = color.y; q[2] = color.z; q[3] = 1.0f;
                }
            }
    }
    px[0] = color.x; px[1] = color.y; px[2] = color.z; px[3] = 1.0f;
}

// ------------------------------------------------------------------ host prep (PathTracer.cpp)
void build_env(Oracle& o, const float* rgba, uint32_t w, uint32_t h) {  // 1161-1296
    o.envW = w; o.envH = h;
    uint64_t size = (uint64_t)w * h;
    o.env.assign(rgba, rgba + size * 4);
    std::vector<float> imp(size);
    float cos0 = 1.0f;
    const float step_phi = 2.0f * 3.14159265358979323846f / (float)w;
    const float step_theta = 3.14159265358979323846f / (float)h;
    for (uint32_t y = 0; y < h; y++) {
        float theta1 = (float)(y + 1) * step_theta;
        float cos1 = cos_(theta1);
        float area = (cos0 - cos1) * step_phi;
        cos0 = cos1;
        for (uint32_t x = 0; x < w; x++) {
            size_t i = (size_t)y * w + x;
            imp[i] = area * std::max(o.env[i * 4], std::max(o.env[i * 4 + 1], o.env[i * 4 + 2]));
        }
    }
    o.alias.resize(size);
    float sum = 0.0f;
    for (uint64_t i = 0; i < size; i++) sum = sum + imp[i];  // std::accumulate, sequential fp32
    float average = sum / (float)size;
    for (uint64_t i = 0; i < size; i++) {
        o.alias[i].importance = (average == 0.0f) ? 0.0f : imp[i] / average;
        o.alias[i].alias = (uint32_t)i;
    }
    std::vector<uint32_t> part(size + 1, 0u);  // +1: upstream writes partitionTable[size] when every texel is "low"
    uint32_t low = 0, high = (uint32_t)size;
    for (uint32_t i = 0; i < size; i++) {
        if (o.alias[i].importance < 1.0f) { low++; part[low] = i; }  // pre-increment quirk: slot 0 never written
        else { high--; part[high] = i; }
    }
    for (low = 0; low < high && high < size; low++) {
        uint32_t li = part[low], hi = part[high];
        o.alias[li].alias = hi;
        float diff = 1.0f - o.alias[li].importance;
        o.alias[hi].importance -= diff;
        if (o.alias[hi].importance < 1.0f) high++;
    }
    for (uint64_t i = 0; i < size; i++) {
        float mx = std::max(o.env[i * 4], std::max(o.env[i * 4 + 1], o.env[i * 4 + 2]));
        o.env[i * 4 + 3] = (sum == 0.0f) ? 0.0f : mx / sum;
    }
}

void build_emissive(Oracle& o) {  // 449-469
    o.emissive.clear(); o.emissive_tris = 0;
    for (uint32_t i = 0; i < o.instances.size(); i++) {
        const vpt_material& m = o.materials[o.instances[i].material_index];
        if (m.emissive_color[0] != 0.0f || m.emissive_color[1] != 0.0f || m.emissive_color[2] != 0.0f) {
            Emissive e;
            e.mesh = o.instances[i].mesh_index; e.material = o.instances[i].material_index;
            e.tri_count = (uint32_t)(o.mindices[e.mesh].size() / 3); e.instance = i;
            memcpy(e.xform, o.instances[i].transform, 64);
            o.emissive.push_back(e); o.emissive_tris += e.tri_count;
        }
    }
}

void build_tris(Oracle& o) {
    o.tris.clear();
    for (uint32_t i = 0; i < o.instances.size(); i++) {
        uint32_t mesh = o.instances[i].mesh_index;
        const float* M = o.instances[i].transform;
        uint32_t nt = (uint32_t)(o.mindices[mesh].size() / 3);
        for (uint32_t t = 0; t < nt; t++) {
            V3 a = mat_point(M, P3(o.mverts[mesh][o.mindices[mesh][t * 3]].position));
            V3 b = mat_point(M, P3(o.mverts[mesh][o.mindices[mesh][t * 3 + 1]].position));
            V3 c = mat_point(M, P3(o.mverts[mesh][o.mindices[mesh][t * 3 + 2]].position));
            Tri tr; tr.v0 = a; tr.e1 = b - a; tr.e2 = c - a; tr.prim = t; tr.inst = i; tr.skip = triangle_degenerate(tr.e1, tr.e2);
            o.tris.push_back(tr);
        }
    }
    o.tri_order.resize(o.tris.size());
    for (size_t i = 0; i < o.tris.size(); i++) o.tri_order[i] = (int)i;
    o.nodes.clear();
    if (!o.tris.empty()) build_node(o, 0, (int)o.tris.size());
}

// ------------------------------------------------------------------ post (PostProcess/*.slang, PostProcessor.cpp:128-246)
struct Img { int w, h; std::vector<float> d; };
inline V3 ld3(const Img& im, int x, int y) { const float* p = &im.d[((size_t)y * im.w + x) * 4]; return v3(p[0], p[1], p[2]); }
inline void st4(Img& im, int x, int y, V3 c) { float* p = &im.d[((size_t)y * im.w + x) * 4]; p[0] = c.x; p[1] = c.y; p[2] = c.z; p[3] = 1.0f; }
inline int iclamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

V3 aces_fitted(V3 c) {  // Tonemap.slang:20-55
    V3 a = v3((0.59719f * c.x + 0.35458f * c.y) + 0.04823f * c.z, (0.07600f * c.x + 0.90834f * c.y) + 0.01566f * c.z,
              (0.02840f * c.x + 0.13383f * c.y) + 0.83777f * c.z);
    V3 n = a * (a + v3s(0.0245786f)) - v3s(0.000090537f);
    V3 d = a * (0.983729f * a + v3s(0.4329510f)) + v3s(0.238081f);
    V3 r = n / d;
    V3 q = v3((1.60475f * r.x + -0.53108f * r.y) + -0.07367f * r.z, (-0.10208f * r.x + 1.10813f * r.y) + -0.00605f * r.z,
              (-0.00327f * r.x + -0.07276f * r.y) + 1.07602f * r.z);
    return v3(saturate_(q.x), saturate_(q.y), saturate_(q.z));
}

}  // namespace

// ====================================================================== C entry points (ctypes)
extern "C" {

void* orc_create(const vpt_scene_desc* sc, uint32_t w, uint32_t h) {
    Oracle* o = new Oracle();
    o->W = w; o->H = h;
    o->image.assign((size_t)w * h * 4, 0.0f);
    for (uint32_t i = 0; i < sc->mesh_count; i++) {
        o->mverts.emplace_back(sc->meshes[i].vertices, sc->meshes[i].vertices + sc->meshes[i].vertex_count);
        o->mindices.emplace_back(sc->meshes[i].indices, sc->meshes[i].indices + sc->meshes[i].index_count);
    }
    o->materials.assign(sc->materials, sc->materials + sc->material_count);
    o->instances.assign(sc->instances, sc->instances + sc->instance_count);
    o->inv3.resize((size_t)sc->instance_count * 9);
    for (uint32_t i = 0; i < sc->instance_count; i++) inverse3x3_from_mat4(sc->instances[i].transform, &o->inv3[(size_t)i * 9]);
    for (uint32_t i = 0; i < sc->texture_count; i++) {
        Tex t; t.w = (int)sc->textures[i].width; t.h = (int)sc->textures[i].height; t.c = (int)sc->textures[i].channels;
        t.d.assign(sc->textures[i].data, sc->textures[i].data + (size_t)t.w * t.h * t.c);
        o->textures.push_back(t);
    }
    build_env(*o, sc->env_rgba, sc->env_width, sc->env_height);
    o->lutR.assign(sc->lut_reflection, sc->lut_reflection + 64 * 64 * 32);
    o->lutO.assign(sc->lut_refraction_outside, sc->lut_refraction_outside + 128 * 128 * 32);
    o->lutI.assign(sc->lut_refraction_inside, sc->lut_refraction_inside + 128 * 128 * 32);
    build_emissive(*o);
    build_tris(*o);
    float id[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    memcpy(o->view_inv, id, 64); memcpy(o->proj_inv, id, 64);
    o->P.samples_per_frame = 1; o->P.max_samples = 5000; o->P.max_depth = 200; o->P.max_luminance = 500.0f;
    o->P.focus_distance = 1.0f; o->P.dof_strength = 0.0f; o->P.sky_azimuth = 0.0f; o->P.sky_altitude = 0.0f;
    o->P.sky_intensity = 1.0f; o->P.screen_chunk_count = 1; o->P.emissive_pdf_bias = 0.0f;
    o->P.flags = VPT_FLAGS_DEFAULT; o->P.base_seed = 1;
    return o;
}
void orc_destroy(void* h) { delete (Oracle*)h; }
void orc_set_camera(void* h, const float* vi, const float* pi) { Oracle* o = (Oracle*)h; memcpy(o->view_inv, vi, 64); memcpy(o->proj_inv, pi, 64); }
void orc_reset(void* h) { Oracle* o = (Oracle*)h; o->frame_count = 0; o->dispatch_count = 0; o->samples_accum = 0; }
void orc_set_params(void* h, const vpt_params* p) { Oracle* o = (Oracle*)h; o->P = *p; orc_reset(h); }
void orc_set_material(void* h, uint32_t idx, const vpt_material* m) { Oracle* o = (Oracle*)h; o->materials[idx] = *m; build_emissive(*o); orc_reset(h); }
int orc_set_volumes(void* h, const vpt_volume* v, uint32_t n) {
    Oracle* o = (Oracle*)h;
    if (n > VPT_MAX_VOLUMES) return -1;
    for (uint32_t i = 0; i < n; i++) if (v[i].density_data_index < -1 || v[i].density_data_index >= (int)o->grids.size() || (v[i].has_temperature_data && v[i].density_data_index < 0)) return -1;
    o->volumes.assign(v, v + n); orc_reset(h);
    return 0;
}
// AddDensityDataToVolume, PathTracer.cpp:1390-1442 on a dense grid: max density, 32^3 block maxima of density/max (y flipped)
int orc_add_density_grid(void* h, uint32_t dx, uint32_t dy, uint32_t dz, const float* d) {
    Oracle* o = (Oracle*)h;
    Oracle::DensityGrid g;
    g.dim[0] = dx; g.dim[1] = dy; g.dim[2] = dz;
    g.values.assign(d, d + (size_t)dx * dy * dz);
    float mx = 0.0f;
    for (float v : g.values) mx = std::max(mx, v);
    g.max_density = mx;
    g.block_max.assign(32768, 0.0f);
    for (uint32_t z = 0; z < dz; z++)
        for (uint32_t y = 0; y < dy; y++)
            for (uint32_t x = 0; x < dx; x++) {
                float raw = g.values[(size_t)x + (size_t)(dy - 1 - y) * dx + (size_t)z * dx * dy];
                float dens = clamp_(raw / mx, 0.0f, 1.0f);
                uint32_t bi = ((x * 32u) / dx) + ((y * 32u) / dy) * 32u + ((z * 32u) / dz) * 1024u;
                if (g.block_max[bi] < dens) g.block_max[bi] = dens;
            }
    o->grids.push_back(std::move(g));
    return (int)o->grids.size() - 1;
}
void orc_clear_density_grids(void* h) { Oracle* o = (Oracle*)h; o->grids.clear(); orc_reset(h); }
void orc_set_atmosphere(void* h, const vpt_atmosphere* a) {
    Oracle* o = (Oracle*)h;
    o->atm_on = a != nullptr;
    if (a) o->atm = *a;
    orc_reset(h);
}
void orc_set_phase_function(void* h, uint32_t phase) { Oracle* o = (Oracle*)h; o->phase = phase; orc_reset(h); }
void orc_set_brute_force(void* h, int on) { ((Oracle*)h)->brute_force = on != 0; }

// PathTracer::PathTrace x dispatches (PathTracer.cpp:122-156) with Seed = PCGHash(base_seed + dispatch).
int orc_render(void* h, uint32_t dispatches, int threads) {
    Oracle* o = (Oracle*)h;
    int done = 0;
    for (uint32_t d = 0; d < dispatches; d++) {
        if (o->samples_accum >= o->P.max_samples) { done = 1; break; }
        uint32_t S = o->P.screen_chunk_count;
        uint32_t seed = pcg_hash(o->P.base_seed + (uint32_t)o->dispatch_count);
        uint32_t chunk = (uint32_t)(o->dispatch_count % (S * S));
        uint32_t lw = (o->W + S - 1) / S, lh = (o->H + S - 1) / S;
        uint32_t fc = o->frame_count;
        std::vector<Counters> tc(threads > 0 ? threads : 1);
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads > 0 ? threads : 1)
        for (int ly = 0; ly < (int)lh; ly++) {
            int tid = 0;
#ifdef _OPENMP
            tid = omp_get_thread_num();
#endif
            for (uint32_t lx = 0; lx < lw; lx++) raygen_pixel(*o, lx, (uint32_t)ly, fc, seed, chunk, tc[tid]);
        }
        for (auto& c : tc) { o->ctr.closest += c.closest; o->ctr.shadow += c.shadow; o->ctr.nodes += c.nodes; o->ctr.tris += c.tris; o->ctr.samples += c.samples; }
        o->dispatch_count++;
        o->frame_count = (uint32_t)(o->dispatch_count / (S * S));
        o->samples_accum = o->frame_count * o->P.samples_per_frame;
    }
    return done;
}
void orc_get_radiance(void* h, float* out) { Oracle* o = (Oracle*)h; memcpy(out, o->image.data(), o->image.size() * 4); }
void orc_set_radiance(void* h, const float* in, uint32_t frame_count) {
    Oracle* o = (Oracle*)h; memcpy(o->image.data(), in, o->image.size() * 4);
    o->frame_count = frame_count; o->dispatch_count = (uint64_t)frame_count * o->P.screen_chunk_count * o->P.screen_chunk_count;
    o->samples_accum = frame_count * o->P.samples_per_frame;
}
void orc_get_counters(void* h, uint64_t* out5) { Oracle* o = (Oracle*)h; out5[0] = o->ctr.closest; out5[1] = o->ctr.shadow; out5[2] = o->ctr.nodes; out5[3] = o->ctr.tris; out5[4] = o->ctr.samples; }
void orc_get_scene_info(void* h, uint32_t* out4) { Oracle* o = (Oracle*)h; out4[0] = (uint32_t)o->tris.size(); out4[1] = (uint32_t)o->nodes.size(); out4[2] = (uint32_t)o->emissive.size(); out4[3] = o->emissive_tris; }
void orc_get_env_tables(void* h, uint32_t* alias_out, float* importance_out, float* pdf_out) {
    Oracle* o = (Oracle*)h;
    for (size_t i = 0; i < o->alias.size(); i++) { alias_out[i] = o->alias[i].alias; importance_out[i] = o->alias[i].importance; pdf_out[i] = o->env[i * 4 + 3]; }
}

void orc_trace_rays(void* h, const vpt_ray* rays, uint32_t n, vpt_hit* hits) {
    Oracle* o = (Oracle*)h;
#pragma omp parallel for schedule(dynamic, 1024)
    for (int i = 0; i < (int)n; i++) {
        Hit b;
        V3 org = v3(rays[i].origin[0], rays[i].origin[1], rays[i].origin[2]);
        V3 dir = v3(rays[i].direction[0], rays[i].direction[1], rays[i].direction[2]);
        if (closest_hit(*o, org, dir, rays[i].tmin, rays[i].tmax, b, nullptr)) {
            hits[i].t = b.t; hits[i].u = b.u; hits[i].v = b.v; hits[i].primitive = b.prim; hits[i].instance = b.inst;
        } else {
            hits[i].t = -1.0f; hits[i].u = 0.0f; hits[i].v = 0.0f; hits[i].primitive = 0xffffffffu; hits[i].instance = 0xffffffffu;
        }
    }
}

// PostProcessor::PostProcess on an arbitrary RGBA32F image.
void orc_postprocess(const float* rgba, uint32_t w, uint32_t h, const vpt_post_params* pp, uint32_t flags, uint8_t* out8, float* bloom0_out) {
    std::vector<Img> mips;
    {
        int cw = (int)w, ch = (int)h;
        for (int i = 0; i < 10; i++) {  // PostProcessor.cpp:136-157
            Img im; im.w = cw; im.h = ch; im.d.assign((size_t)cw * ch * 4, 0.0f);
            mips.push_back(std::move(im));
            if (cw % 2 != 0) cw -= 1;
            if (ch % 2 != 0) ch -= 1;
            cw /= 2; ch /= 2;
            if (cw < 2 || ch < 2) break;
        }
    }
    int mip_count = iclamp((int)pp->mip_count, 1, (int)mips.size());
    Img in; in.w = (int)w; in.h = (int)h; in.d.assign(rgba, rgba + (size_t)w * h * 4);
    // pass 0: soft threshold (BloomDownSample.slang:32-45)
    for (int y = 0; y < in.h; y++)
        for (int x = 0; x < in.w; x++) {
            V3 c = ld3(in, x, y);
            float br = dot(c, v3(0.2126f, 0.7152f, 0.0722f));
            float f = smoothstep(pp->bloom_threshold - pp->falloff_range, pp->bloom_threshold + pp->falloff_range, br);
            st4(mips[0], x, y, c * f);
        }
    // down passes i >= 1 (BloomDownSample.slang:46-63): 16 taps, divided by 25
    for (int i = 1; i < mip_count; i++) {
        const Img& src = mips[i - 1]; Img& dst = mips[i];
        for (int y = 0; y < dst.h; y++)
            for (int x = 0; x < dst.w; x++) {
                V3 c = v3s(0.0f);
                for (int a = -2; a < 2; a++)
                    for (int b = -2; b < 2; b++)
                        c = c + ld3(src, iclamp(x * 2 + a, 0, src.w - 1), iclamp(y * 2 + b, 0, src.h - 1));
                c = c / 25.0f;  // pow(range*2+1, 2) with range = 2 folds to 25
                c = c * pp->bloom_strength;
                st4(dst, x, y, c);
            }
    }
    // up passes (BloomUpSample.slang:30-48), k = mip-1 .. 1: coarse k added into finer k-1
    for (int i = mip_count - 1; i > 0; i--) {
        const Img& src = mips[i]; Img& dst = mips[i - 1];
        for (int y = 0; y < dst.h; y++)
            for (int x = 0; x < dst.w; x++) {
                V3 c = v3s(0.0f);
                for (int a = -2; a < 2; a++)
                    for (int b = -2; b < 2; b++)
                        c = c + ld3(src, iclamp(x / 2 + a + 1, 0, src.w - 1), iclamp(y / 2 + b + 1, 0, src.h - 1));
                c = c / 25.0f;  // pow(range*2+1, 2) with range = 2 folds to 25
                c = c * pp->bloom_strength;
                st4(dst, x, y, c + ld3(dst, x, y));
            }
    }
    if (bloom0_out) memcpy(bloom0_out, mips[0].d.data(), mips[0].d.size() * 4);
    // tonemap (Tonemap.slang:159-176)
    bool linear = (flags & VPT_FLAG_TONEMAP_LINEAR_BLOOM_TAP) != 0;
    for (int y = 0; y < (int)h; y++)
        for (int x = 0; x < (int)w; x++) {
            V3 c = ld3(in, x, y);
            float u = (float)x / (float)w, v = (float)y / (float)h;
            V3 bl;
            if (linear) { V4 s = sample_f4(mips[0].d.data(), (int)w, (int)h, u, v, false); bl = v3(s.x, s.y, s.z); }
            else {
                int tx = iclamp((int)floor_(u * (float)w), 0, (int)w - 1), ty = iclamp((int)floor_(v * (float)h), 0, (int)h - 1);
                bl = ld3(mips[0], tx, ty);
            }
            c = c + bl;
            c = c * pp->exposure;
            float ig = 1.0f / pp->gamma;
            c = v3(pow_(c.x, ig), pow_(c.y, ig), pow_(c.z, ig));
            c = aces_fitted(c);
            uint8_t* q = &out8[((size_t)y * w + x) * 4];
            q[0] = unorm8(c.x); q[1] = unorm8(c.y); q[2] = unorm8(c.z); q[3] = 255;
        }
}

// ---------------------------------------------------------------- known-answer tests
uint32_t orc_pcg_hash(uint32_t x) { return pcg_hash(x); }
float orc_uniform_float(uint32_t h) { return u32_to_unit(h); }
// elementary functions of the fp32 contract, for accuracy tests against libm (fn: 0 sin 1 cos 2 log 3 exp 4 asin 5 acos 6 atan2(x,y) 7 pow(x,y))
void orc_fp32_eval(int fn, const float* x, const float* y, float* out, uint32_t n) {
    for (uint32_t i = 0; i < n; i++) {
        switch (fn) {
            case 0: out[i] = sin_(x[i]); break;
            case 1: out[i] = cos_(x[i]); break;
            case 2: out[i] = log_(x[i]); break;
            case 3: out[i] = exp_(x[i]); break;
            case 4: out[i] = asin_(x[i]); break;
            case 5: out[i] = acos_(x[i]); break;
            case 6: out[i] = atan2_(x[i], y[i]); break;
            default: out[i] = pow_(x[i], y[i]); break;
        }
    }
}

// Leaf primitives of the fp32 contract that the HIP kernels and this oracle SHARE (include/vpt_fp32.h), exposed one by one so
// that tests/test_fp32_contract.py can hold them against independent float64 / numpy restatements — HIP-vs-oracle parity
// cannot see a bug in code both sides compile.  fn: 0 ray_triangle (in: o3 d3 v0 3 e1 3 e2 3 tmin tmax; out: hit t u v),
// 1 texel_coords (in: u size repeat; out: i0 i1 w), 2 lut_layer (in: layer layers; out: index), 3 refract (in: i3 n3 eta;
// out: 3), 4 smoothstep (in: e0 e1 x), 5 reflect (in: i3 n3; out: 3), 6 normalize (in: 3; out: 3), 7 unorm8 (in: c),
// 8 hit_is_local (in: o3 d3 v0 3 e1 3 e2 3 t; out: 0/1), 9 triangle_degenerate (in: e1 3 e2 3; out: 0/1),
// 10 unorm8_to_float (in: byte as float; the HIP texel fetch uses it, this oracle divides by 255).
void orc_leaf_eval(int fn, const float* in, float* out, uint32_t n) {
    for (uint32_t i = 0; i < n; i++) {
        switch (fn) {
            case 0: {
                const float* a = in + (size_t)i * 17; float t = 0, u = 0, v = 0;
                bool h = ray_triangle(v3(a[0], a[1], a[2]), v3(a[3], a[4], a[5]), v3(a[6], a[7], a[8]), v3(a[9], a[10], a[11]), v3(a[12], a[13], a[14]), a[15], a[16], &t, &u, &v);
                out[i * 4] = h ? 1.0f : 0.0f; out[i * 4 + 1] = t; out[i * 4 + 2] = u; out[i * 4 + 3] = v; break;
            }
            case 1: { const float* a = in + (size_t)i * 3; int i0, i1; float w; texel_coords(a[0], (int)a[1], a[2] != 0.0f, &i0, &i1, &w); out[i * 3] = (float)i0; out[i * 3 + 1] = (float)i1; out[i * 3 + 2] = w; break; }
            case 2: out[i] = (float)lut_layer(in[i * 2], (int)in[i * 2 + 1]); break;
            case 3: { const float* a = in + (size_t)i * 7; V3 r = refract(v3(a[0], a[1], a[2]), v3(a[3], a[4], a[5]), a[6]); out[i * 3] = r.x; out[i * 3 + 1] = r.y; out[i * 3 + 2] = r.z; break; }
            case 4: out[i] = smoothstep(in[i * 3], in[i * 3 + 1], in[i * 3 + 2]); break;
            case 5: { const float* a = in + (size_t)i * 6; V3 r = reflect(v3(a[0], a[1], a[2]), v3(a[3], a[4], a[5])); out[i * 3] = r.x; out[i * 3 + 1] = r.y; out[i * 3 + 2] = r.z; break; }
            case 6: { const float* a = in + (size_t)i * 3; V3 r = normalize(v3(a[0], a[1], a[2])); out[i * 3] = r.x; out[i * 3 + 1] = r.y; out[i * 3 + 2] = r.z; break; }
            case 7: out[i] = (float)unorm8(in[i]); break;
            case 8: { const float* a = in + (size_t)i * 16; out[i] = hit_is_local(v3(a[0], a[1], a[2]), v3(a[3], a[4], a[5]), v3(a[6], a[7], a[8]), v3(a[9], a[10], a[11]), v3(a[12], a[13], a[14]), a[15]) ? 1.0f : 0.0f; break; }
            case 10: out[i] = unorm8_to_float((uint32_t)in[i]); break;
            default: { const float* a = in + (size_t)i * 6; out[i] = triangle_degenerate(v3(a[0], a[1], a[2]), v3(a[3], a[4], a[5])) ? 1.0f : 0.0f; break; }
        }
    }
}

// LookupReflect.slang:25-85 for one table cell (x,y,z) of a (sx,sy,sz) table; nsamples MC samples.
float orc_lut_reflect_cell(uint32_t x, uint32_t y, uint32_t z, uint32_t sx, uint32_t sy, uint32_t sz, uint32_t nsamples, uint32_t seed) {
    Rng r; r.s = y + x * x + seed;
    float vc = clamp_((float)x / (float)sx, 0.05f, 0.999f);
    float rough = clamp_((float)y / (float)sy, 0.0001f, 1.0f);
    float aniso = (float)z / (float)sz;
    float aspect = sqrt_(1.0f - sqrt_(aniso) * 0.9f);
    Mat m; memset(&m.p, 0, sizeof(m.p));
    m.ax = max_(0.0001f, rough / aspect); m.ay = max_(0.0001f, rough * aspect);
    m.p.anisotropy = aniso; m.p.roughness = rough; m.eta = 1.0f; m.o = nullptr; m.ec = false;
    double fin = 0.0;
    for (uint32_t i = 0; i < nsamples; i++) {
        float mag = sqrt_(1.0f - vc * vc);
        float phi = r.uf() * M_2_PI_F;
        float s, c; sincos_(phi, &s, &c);
        V3 V = normalize(v3(mag * c, mag * s, vc));
        V3 H = ggx_sample(r, V, m.ax, m.ay);
        V3 L = normalize(reflect(-V, H));
        if (L.z <= 0.0f) continue;
        Eval e = m.eval_reflection(V, L, v3s(1.0f));
        if (e.pdf <= 0.0f) continue;
        if (isnan_(e.bxdf.x) || isinf_(e.bxdf.x)) continue;
        fin += (double)(e.bxdf.x / e.pdf);
    }
    return (float)(fin / (double)nsamples);
}
// LookupRefract.slang:23-103; above != 0 -> ABOVE_SURFACE (Eta = 1/ior).
float orc_lut_refract_cell(uint32_t x, uint32_t y, uint32_t z, uint32_t sx, uint32_t sy, uint32_t sz, int above, uint32_t nsamples, uint32_t seed) {
    Rng r; r.s = y + x * x + seed;
    float vc = clamp_(pow_((float)x / ((float)sx - 1.0f), 2.0f), 0.01f, 0.9999f);
    float rough = clamp_((float)y / ((float)sy - 1.0f), 0.01f, 1.0f);
    float ior = 1.0f + clamp_((float)z / ((float)sz - 1.0f), 0.0001f, 1.0f);
    Mat m; memset(&m.p, 0, sizeof(m.p));
    m.ax = rough; m.ay = rough; m.p.roughness = rough; m.p.ior = ior; m.eta = above ? (1.0f / ior) : ior; m.o = nullptr; m.ec = false;
    double fin = 0.0;
    for (uint32_t i = 0; i < nsamples; i++) {
        float mag = sqrt_(1.0f - vc * vc);
        float phi = r.uf() * M_2_PI_F;
        float s, c; sincos_(phi, &s, &c);
        V3 V = normalize(v3(mag * c, mag * s, vc));
        V3 H = ggx_sample(r, V, m.ax, m.ay);
        float F = m.dielectric_fresnel(fabs_(dot(V, H)));
        float val = 0.0f;
        if (r.uf() < F) {
            V3 L = normalize(reflect(-V, H));
            if (L.z > 0.0f) { Eval e = m.eval_reflection(V, L, v3s(1.0f)); if (e.pdf > 0.0f && !isnan_(e.bxdf.x) && !isinf_(e.bxdf.x)) val += e.bxdf.x / e.pdf; }
        } else {
            V3 L = normalize(refract(-V, H, m.eta));
            if (L.z < 0.0f) { Eval e = m.eval_refraction(V, L, v3s(1.0f)); if (e.pdf > 0.0f && !isnan_(e.bxdf.x) && !isinf_(e.bxdf.x)) val += e.bxdf.x / e.pdf; }
        }
        if (!isnan_(val) && !isinf_(val)) fin += (double)val;
    }
    return (float)(fin / (double)nsamples);
}

// LookupTableCalculator::CalculateTable (LookupTableCalculator.cpp:44-157) for a list of cells, with the exact
// arithmetic structure of the reference: sampleCount/20 passes; pass i reseeds the cell with
// Sampler(y + x*x + PCG(i*2 + sampleCount + PCG(time_ms))) (:99-103; the reference's time_ms is a wall-clock
// reading per pass, here one fixed value), adds finalValue/20 (fp32, LookupReflect.slang:60-89 /
// LookupRefract.slang:53-102) and the sum is divided by the pass count at the end (:152-155).
// kind 0 reflect, 1 refract ABOVE_SURFACE, 2 refract BELOW_SURFACE.  cells[i] = x + y*sx + z*sx*sy.
void orc_lut_cells(uint32_t kind, uint32_t sx, uint32_t sy, uint32_t sz, uint32_t sample_count, uint32_t time_ms,
                   const uint32_t* cells, uint32_t n, float* out, int threads) {
    const uint32_t passes = sample_count / 20u, th = pcg_hash(time_ms);
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads > 0 ? threads : 1)
    for (int64_t ci = 0; ci < (int64_t)n; ci++) {
        const uint32_t index = cells[ci], x = index % sx, y = (index / sx) % sy, z = index / (sx * sy);
        Mat m; memset(&m.p, 0, sizeof(m.p)); m.o = nullptr; m.ec = false;
        float vc;
        if (kind == 0) {
            vc = clamp_((float)x / (float)sx, 0.05f, 0.999f);
            float rough = clamp_((float)y / (float)sy, 0.0001f, 1.0f);
            float aniso = (float)z / (float)sz;
            float aspect = sqrt_(1.0f - sqrt_(aniso) * 0.9f);
            m.ax = max_(0.0001f, rough / aspect); m.ay = max_(0.0001f, rough * aspect);
            m.p.anisotropy = aniso; m.p.roughness = rough; m.eta = 1.0f;
        } else {
            vc = clamp_(pow_((float)x / ((float)sx - 1.0f), 2.0f), 0.01f, 0.9999f);
            float rough = clamp_((float)y / ((float)sy - 1.0f), 0.01f, 1.0f);
            float ior = 1.0f + clamp_((float)z / ((float)sz - 1.0f), 0.0001f, 1.0f);
            m.ax = rough; m.ay = rough; m.p.roughness = rough; m.p.ior = ior; m.eta = kind == 1 ? (1.0f / ior) : ior;
        }
        float cell = 0.0f;
        for (uint32_t i = 0; i < passes; i++) {
            Rng r; r.s = y + x * x + pcg_hash(i * 2u + sample_count + th);
            float fin = 0.0f;
            for (uint32_t k = 0; k < 20u; k++) {
                float mag = sqrt_(1.0f - vc * vc);
                float phi = r.uf() * M_2_PI_F;
                float s, c; sincos_(phi, &s, &c);
                V3 V = normalize(v3(mag * c, mag * s, vc));
                V3 H = ggx_sample(r, V, m.ax, m.ay);
                if (kind == 0) {
                    V3 L = normalize(reflect(-V, H));
                    if (L.z <= 0.0f) continue;
                    Eval e = m.eval_reflection(V, L, v3s(1.0f));
                    if (e.pdf <= 0.0f) continue;
                    if (isnan_(e.bxdf.x) || isinf_(e.bxdf.x)) continue;
                    fin += e.bxdf.x / e.pdf;
                } else {
                    float F = m.dielectric_fresnel(fabs_(dot(V, H)));
                    float val = 0.0f;
                    if (r.uf() < F) {
                        V3 L = normalize(reflect(-V, H));
                        if (L.z > 0.0f) { Eval e = m.eval_reflection(V, L, v3s(1.0f)); if (e.pdf > 0.0f && !isnan_(e.bxdf.x) && !isinf_(e.bxdf.x)) val += e.bxdf.x / e.pdf; }
                    } else {
                        V3 L = normalize(refract(-V, H, m.eta));
                        if (L.z < 0.0f) { Eval e = m.eval_refraction(V, L, v3s(1.0f)); if (e.pdf > 0.0f && !isnan_(e.bxdf.x) && !isinf_(e.bxdf.x)) val += e.bxdf.x / e.pdf; }
                    }
                    if (!isnan_(val) && !isinf_(val)) fin += val;
                }
            }
            cell += fin / 20.0f;
        }
        out[ci] = cell / (float)passes;
    }
}

// Monte-Carlo means of the two atmosphere estimators for one ray (test hook): out[0] = E[ratio-tracked
// transmittance of channel ch] (Atmosphere.slang:33-107), out[1] = fraction of delta-tracking runs that leave the
// atmosphere without a collision (:117-201).  Both estimate exp(-optical depth) of the same ray.
void orc_atmosphere_estimators(const vpt_atmosphere* a, const float* org, const float* dir, int ch, uint32_t seed, uint32_t n, float* out) {
    Oracle o; o.atm_on = true; o.atm = *a;
    Rng r; r.s = seed;
    double tr = 0.0, esc = 0.0;
    for (uint32_t i = 0; i < n; i++) {
        V3 t = atmosphere_transmittance(o, r, P3(org), P3(dir), ch);
        tr += ch == 0 ? t.x : (ch == 1 ? t.y : t.z);
        int comp;
        if (atmosphere_scatter_distance(o, r, P3(org), P3(dir), ch, comp) < 0.0f) esc += 1.0;
    }
    out[0] = (float)(tr / n); out[1] = (float)(esc / n);
}

// Debug hook: the frame sample (accumulatedLight / SampleCount) of single pixels for frames first..first+n-1, i.e. what
// dispatch k alone contributes to pixel (x, y); seeds as in orc_render.  out[(p * n + k) * 3 + c].
void orc_pixel_samples(void* h, const uint32_t* xs, const uint32_t* ys, uint32_t npix, uint32_t first, uint32_t n, float* out) {
    Oracle* o = (Oracle*)h;
    Counters c{};
    for (uint32_t p = 0; p < npix; p++)
        for (uint32_t k = 0; k < n; k++) {
            float* px = &o->image[((size_t)ys[p] * o->W + xs[p]) * 4];
            float keep[4] = {px[0], px[1], px[2], px[3]};
            raygen_pixel(*o, xs[p], ys[p], 0u, pcg_hash(o->P.base_seed + first + k), 0u, c);  // frame_count 0: the pixel becomes the sample itself
            for (int q = 0; q < 3; q++) out[((size_t)p * n + k) * 3 + q] = px[q];
            for (int q = 0; q < 4; q++) px[q] = keep[q];
        }
}

// Debug hook: all rays (closest and shadow queries, in call order) of one pixel's sample of dispatch `frame`:
// 10 floats per ray {o.xyz, tmin, d.xyz, tmax, t or -1, global triangle id or -1}; returns the ray count (<= cap).
uint32_t orc_pixel_rays(void* h, uint32_t x, uint32_t y, uint32_t frame, float* out, uint32_t cap) {
    Oracle* o = (Oracle*)h;
    std::vector<LoggedRay> log;
    g_ray_log = &log;
    float rgb[3];
    orc_pixel_samples(h, &x, &y, 1, frame, 1, rgb);
    g_ray_log = nullptr;
    uint32_t n = (uint32_t)std::min<size_t>(log.size(), cap);
    for (uint32_t i = 0; i < n; i++) {
        const LoggedRay& r = log[i];
        float* q = out + (size_t)i * 10;
        q[0] = r.o[0]; q[1] = r.o[1]; q[2] = r.o[2]; q[3] = r.tmin; q[4] = r.d[0]; q[5] = r.d[1]; q[6] = r.d[2]; q[7] = r.tmax; q[8] = r.t; q[9] = (float)r.gid;
    }
    (void)o;
    return n;
}

// Debug hook: the flattened world-space triangles {v0, e1, e2, prim, inst, gid} (12 dwords each), as the BVH builders see them.
uint32_t orc_get_triangles(void* h, float* out, uint32_t cap) {
    Oracle* o = (Oracle*)h;
    uint32_t n = (uint32_t)std::min<size_t>(o->tris.size(), cap);
    for (uint32_t i = 0; out && i < n; i++) {
        const Tri& t = o->tris[i];
        float* q = out + (size_t)i * 12;
        q[0] = t.v0.x; q[1] = t.v0.y; q[2] = t.v0.z; q[3] = t.e1.x; q[4] = t.e1.y; q[5] = t.e1.z; q[6] = t.e2.x; q[7] = t.e2.y; q[8] = t.e2.z;
        uint32_t ids[3] = {t.prim, t.inst, i};
        memcpy(q + 9, ids, 12);
    }
    return (uint32_t)o->tris.size();
}

// Test hooks on the BSDF restatement (Material.slang:94-449), for closed-form pins.  The material is taken as is (no
// textures: Material.Initialize with all-white 1x1 textures), hit from outside; energy compensation off (it needs a scene's LUTs).
static void hook_material(Mat& m, const vpt_material* src) {
    m.p = *src; m.o = nullptr; m.ec = false;
    m.p.ior = max_(m.p.ior, 1.000001f);
    float aspect = sqrt_(1.0f - sqrt_(m.p.anisotropy) * 0.9f);
    m.ax = max_(0.00001f, m.p.roughness / aspect);
    m.ay = max_(0.00001f, m.p.roughness * aspect);
    m.eta = 1.0f / m.p.ior;
}
// D(h) of GGXDistributionAnisotropic (Material.slang:394-404) for n directions h (xyz triples)
void orc_ggx_d(const vpt_material* mat, const float* h, uint32_t n, float* out) {
    Mat m; hook_material(m, mat);
    for (uint32_t i = 0; i < n; i++) out[i] = m.ggx_d(v3(h[i * 3], h[i * 3 + 1], h[i * 3 + 2]));
}
// EvaluateBSDF(V, L) (Material.slang:167-254) for n directions L: out[i*4..] = f.rgb (cos included), pdf
void orc_bsdf_eval(const vpt_material* mat, const float* V, const float* L, uint32_t n, float* out) {
    Mat m; hook_material(m, mat);
    V3 v = v3(V[0], V[1], V[2]);
    for (uint32_t i = 0; i < n; i++) {
        Eval e = m.eval_bsdf(v, v3(L[i * 3], L[i * 3 + 1], L[i * 3 + 2]));
        out[i * 4] = e.bxdf.x; out[i * 4 + 1] = e.bxdf.y; out[i * 4 + 2] = e.bxdf.z; out[i * 4 + 3] = e.pdf;
    }
}
// The same with energy compensation on and the three lookup tables given (reflection 64x64x32, refraction from outside / inside
// 128x128x32), hit from outside (inside == 0) or inside: what tests/test_oracle_bsdf_fp64.py holds against a float64 restatement.
void orc_bsdf_eval_ec(const vpt_material* mat, const float* V, const float* L, uint32_t n, const float* lut_r, const float* lut_o, const float* lut_i,
                      int inside, float* out) {
    static Oracle* holder = nullptr;   // only the tables are read through it
    if (!holder) holder = new Oracle();
    holder->lutR.assign(lut_r, lut_r + 64 * 64 * 32);
    holder->lutO.assign(lut_o, lut_o + 128 * 128 * 32);
    holder->lutI.assign(lut_i, lut_i + 128 * 128 * 32);
    Mat m; hook_material(m, mat);
    m.o = holder; m.ec = true;
    if (inside) m.eta = m.p.ior;
    V3 v = v3(V[0], V[1], V[2]);
    for (uint32_t i = 0; i < n; i++) {
        Eval e = m.eval_bsdf(v, v3(L[i * 3], L[i * 3 + 1], L[i * 3 + 2]));
        out[i * 4] = e.bxdf.x; out[i * 4 + 1] = e.bxdf.y; out[i * 4 + 2] = e.bxdf.z; out[i * 4 + 3] = e.pdf;
    }
}
// n draws of the VNDF + SampleBSDF (Sampler.slang:141-166, Material.slang:94-165) from one RNG stream: out[i*7..] = L.xyz, f.rgb, pdf
void orc_bsdf_sample(const vpt_material* mat, const float* V, uint32_t seed, uint32_t n, float* out) {
    Mat m; hook_material(m, mat);
    V3 v = v3(V[0], V[1], V[2]);
    Rng r; r.s = seed;
    for (uint32_t i = 0; i < n; i++) {
        V3 H = ggx_sample(r, v, m.ax, m.ay);
        BSample b = sample_bsdf(m, r, v, H);
        float* q = out + (size_t)i * 7;
        q[0] = b.L.x; q[1] = b.L.y; q[2] = b.L.z; q[3] = b.bxdf.x; q[4] = b.bxdf.y; q[5] = b.bxdf.z; q[6] = b.pdf;
    }
}
// The same draws with energy compensation on and the three lookup tables given (orc_bsdf_eval_ec).
void orc_bsdf_sample_ec(const vpt_material* mat, const float* V, uint32_t seed, uint32_t n, const float* lut_r, const float* lut_o, const float* lut_i, float* out) {
    static Oracle* holder = nullptr;
    if (!holder) holder = new Oracle();
    holder->lutR.assign(lut_r, lut_r + 64 * 64 * 32);
    holder->lutO.assign(lut_o, lut_o + 128 * 128 * 32);
    holder->lutI.assign(lut_i, lut_i + 128 * 128 * 32);
    Mat m; hook_material(m, mat);
    m.o = holder; m.ec = true;
    V3 v = v3(V[0], V[1], V[2]);
    Rng r; r.s = seed;
    for (uint32_t i = 0; i < n; i++) {
        V3 H = ggx_sample(r, v, m.ax, m.ay);
        BSample b = sample_bsdf(m, r, v, H);
        float* q = out + (size_t)i * 7;
        q[0] = b.L.x; q[1] = b.L.y; q[2] = b.L.z; q[3] = b.bxdf.x; q[4] = b.bxdf.y; q[5] = b.bxdf.z; q[6] = b.pdf;
    }
}

}  // extern "C"
