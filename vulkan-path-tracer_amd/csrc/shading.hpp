// shading.hpp — device-side surface / material / sampling routines of the shade stage.
// Reference semantics: Shaders/Surface.slang, Material.slang, Sampler.slang, RTCommon.slang
// (file:line cited per function).  All arithmetic goes through include/vpt_fp32.h and is compiled
// with -ffp-contract=off, so expression order here is part of the parity contract.
#pragma once
#include "device_types.hpp"

namespace vpt {
using namespace vptfp;

// Defines.slang:1-7 (as fp32)
#define VPT_PI 3.1415926535897F
#define VPT_2PI 6.2831853071795F
#define VPT_1_OVER_PI 0.3183098861837F
constexpr uint32_t kMaxDepthMarker = 1000000u;  // Defines.slang:16 MAX_DEPTH

struct Rng {  // Sampler.slang:21-43
    uint32_t s;
    __device__ inline float uf() { s = pcg_hash(s); return u32_to_unit(s); }
};

__device__ inline V3 ld3(const float* p) { return v3(p[0], p[1], p[2]); }
__device__ inline V4 v4(float x, float y, float z, float w) { V4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
__device__ inline V4 lerp4(V4 a, V4 b, float t) { return v4(lerp(a.x, b.x, t), lerp(a.y, b.y, t), lerp(a.z, b.z, t), lerp(a.w, b.w, t)); }

// ------------------------------------------------------------------ software samplers (LINEAR, mip 0)
__device__ inline V4 tex_fetch(const uint8_t* texels, const TexDesc& t, int x, int y) {
    if (t.c == 4) {
        uchar4 p = *reinterpret_cast<const uchar4*>(texels + t.offset + ((size_t)y * t.w + x) * 4);
        return v4(unorm8_to_float(p.x), unorm8_to_float(p.y), unorm8_to_float(p.z), unorm8_to_float(p.w));   // == p / 255.0f, bit for bit
    }
    float v = unorm8_to_float(texels[t.offset + (size_t)y * t.w + x]);
    return v4(v, 0.0f, 0.0f, 1.0f);
}
// uTextureSampler: LINEAR / REPEAT (PathTracer.cpp:84-91)
__device__ inline V4 tex_sample(const DeviceScene& sc, uint32_t ti, float u, float v) {
    TexDesc t = sc.textures[ti];
    if (t.w == 1 && t.h == 1) return tex_fetch(sc.texels, t, 0, 0);  // == the filtered result, bit for bit
    int x0, x1, y0, y1; float fx, fy;
    texel_coords(u, (int)t.w, true, &x0, &x1, &fx);
    texel_coords(v, (int)t.h, true, &y0, &y1, &fy);
    V4 a = lerp4(tex_fetch(sc.texels, t, x0, y0), tex_fetch(sc.texels, t, x1, y0), fx);
    V4 b = lerp4(tex_fetch(sc.texels, t, x0, y1), tex_fetch(sc.texels, t, x1, y1), fx);
    return lerp4(a, b, fy);
}
// The same lookup in two halves, so that a hit can have the taps of all its textures in flight at once: tex_issue computes the four
// tap addresses and starts the loads, tex_finish decodes and filters (the operations of tex_sample in its order; a 1x1 texture has all
// four taps on its one texel and both weights 0, which filters to that texel bit for bit).
struct TexTaps { uint32_t r00, r10, r01, r11, c; float fx, fy; };
__device__ inline void tex_issue(const uint8_t* texels, const TexDesc& t, float u, float v, TexTaps& k) {
    int x0, x1, y0, y1;
    texel_coords(u, (int)t.w, true, &x0, &x1, &k.fx);
    texel_coords(v, (int)t.h, true, &y0, &y1, &k.fy);
    const uint32_t row0 = (uint32_t)y0 * t.w, row1 = (uint32_t)y1 * t.w;
    k.c = t.c;
    if (t.c == 4) {
        const uint32_t* p = reinterpret_cast<const uint32_t*>(texels + t.offset);
        k.r00 = p[row0 + (uint32_t)x0]; k.r10 = p[row0 + (uint32_t)x1]; k.r01 = p[row1 + (uint32_t)x0]; k.r11 = p[row1 + (uint32_t)x1];
    } else {
        const uint8_t* p = texels + t.offset;
        k.r00 = p[row0 + (uint32_t)x0]; k.r10 = p[row0 + (uint32_t)x1]; k.r01 = p[row1 + (uint32_t)x0]; k.r11 = p[row1 + (uint32_t)x1];
    }
}
// (a one-channel tap was loaded as a zero-extended byte: its bytes 1-3 are 0, which decode to the 0.0f the one-channel texel has there;
// only alpha differs — so one decode serves both kinds, and a wave that holds both issues it once)
__device__ inline V4 tex_decode(uint32_t r, uint32_t c) {
    return v4(unorm8_to_float(r & 255u), unorm8_to_float((r >> 8) & 255u), unorm8_to_float((r >> 16) & 255u), c == 4 ? unorm8_to_float(r >> 24) : 1.0f);
}
__device__ inline V4 tex_finish(const TexTaps& k) {
    V4 a = lerp4(tex_decode(k.r00, k.c), tex_decode(k.r10, k.c), k.fx);
    V4 b = lerp4(tex_decode(k.r01, k.c), tex_decode(k.r11, k.c), k.fx);
    return lerp4(a, b, k.fy);
}
__device__ inline V4 env_sample(const DeviceScene& sc, float u, float v) {
    int w = (int)sc.env_w, h = (int)sc.env_h;
    const float4* e = reinterpret_cast<const float4*>(sc.env);
    int x0, x1, y0, y1; float fx, fy;
    texel_coords(u, w, true, &x0, &x1, &fx);
    texel_coords(v, h, true, &y0, &y1, &fy);
    float4 p00 = e[(size_t)y0 * w + x0], p10 = e[(size_t)y0 * w + x1], p01 = e[(size_t)y1 * w + x0], p11 = e[(size_t)y1 * w + x1];
    V4 a = lerp4(v4(p00.x, p00.y, p00.z, p00.w), v4(p10.x, p10.y, p10.z, p10.w), fx);
    V4 b = lerp4(v4(p01.x, p01.y, p01.z, p01.w), v4(p11.x, p11.y, p11.z, p11.w), fx);
    return lerp4(a, b, fy);
}
// uLookupTableSampler: LINEAR / CLAMP_TO_EDGE on an R32F 2D array (PathTracer.cpp:93-94)
__device__ inline float lut_sample(const float* lut, int sx, int sy, int sz, float u, float v, float layer) {
    int x0, x1, y0, y1; float fx, fy;
    texel_coords(u, sx, false, &x0, &x1, &fx);
    texel_coords(v, sy, false, &y0, &y1, &fy);
    const float* p = lut + (size_t)lut_layer(layer, sz) * sx * sy;
    float a = lerp(p[y0 * sx + x0], p[y0 * sx + x1], fx);
    float b = lerp(p[y1 * sx + x0], p[y1 * sx + x1], fx);
    return lerp(a, b, fy);
}

// ------------------------------------------------------------------ RTCommon.slang
__device__ inline float power_heuristics(float a, float b) {  // :124-127
    float a2 = pow_(a, 2.0f);
    return a2 / (a2 + pow_(b, 2.0f));
}
__device__ inline V2 direction_to_uv(V3 v) {  // :129-136
    float gamma = asin_(clamp_(v.y, -1.0f, 1.0f));
    float theta = atan2_(v.x, -v.z);
    V2 uv; uv.x = theta * VPT_1_OVER_PI * 0.5f + 0.5f; uv.y = gamma * VPT_1_OVER_PI + 0.5f;
    return uv;
}

// ------------------------------------------------------------------ Surface.slang:26-147
struct SurfaceFrame {
    V3 pos, N, T, B, Ng;
    V2 uv;
    V3 p1, p2, p3;  // object-space positions of the hit triangle (light-MIS area, ClosestHit.slang:276-287)
    bool inside;
    __device__ inline V3 tangent_to_world(V3 v) const { return normalize((v.x * T + v.y * B) + v.z * N); }
    __device__ inline V3 world_to_tangent(V3 v) const { return normalize(v3(dot(v, T), dot(v, B), dot(v, N))); }
};

// Geometric normal of one triangle in world space (Surface.slang:48-49); evaluated once per triangle by
// k_precompute_tri_ng and read back per hit.
__device__ inline V3 triangle_ng(const DeviceScene& sc, const InstanceDesc& in, uint32_t prim) {
    const MeshDesc me = sc.meshes[in.mesh];
    const uint32_t* idx = sc.indices + me.index_offset + prim * 3;
    const vpt_vertex* vb = sc.vertices + me.vertex_offset;
    V3 p1 = v3(vb[idx[0]].position[0], vb[idx[0]].position[1], vb[idx[0]].position[2]);
    V3 p2 = v3(vb[idx[1]].position[0], vb[idx[1]].position[1], vb[idx[1]].position[2]);
    V3 p3 = v3(vb[idx[2]].position[0], vb[idx[2]].position[1], vb[idx[2]].position[2]);
    V3 ng = normalize(cross(p2 - p1, p3 - p1));
    return normalize(rowvec_mat3(ng, in.inv3));
}

// SurfaceFrame in two halves around the texel fetches: surface_geom reads the triangle and interpolates (the uv is known after it),
// surface_frame applies the normal map and the reference's two normal corrections.
__device__ inline void surface_geom(const DeviceScene& sc, SurfaceFrame& s, const InstanceDesc& in, uint32_t gid, float hu, float hv, V3 raydir, bool geo_only) {
    // the triangle's three vertices and its geometric normal from its de-indexed record (k_precompute_tri_shade): one 128-byte line
    // (the hit record carries the GLOBAL triangle id, so this fetch does not wait for the instance record)
    const float4* q = sc.tri_shade + (size_t)gid * 8;
    const float4 a0 = q[0], a1 = q[1], b0 = q[2], b1 = q[3], c0 = q[4], c1 = q[5], g6 = q[6];
    s.p1 = v3(a0.x, a0.y, a0.z); s.p2 = v3(b0.x, b0.y, b0.z); s.p3 = v3(c0.x, c0.y, c0.z);
    V3 n1 = v3(a0.w, a1.x, a1.y), n2 = v3(b0.w, b1.x, b1.y), n3 = v3(c0.w, c1.x, c1.y);
    float bx = 1.0f - hu - hv, by = hu, bz = hv;  // ClosestHit.slang:45
    s.pos = mat_point(in.xform, (s.p1 * bx + s.p2 * by) + s.p3 * bz);
    s.uv.x = (a1.z * bx + b1.z * by) + c1.z * bz;
    s.uv.y = (a1.w * bx + b1.w * by) + c1.w * bz;
    s.Ng = v3(g6.x, g6.y, g6.z);
    if (geo_only) {
        s.N = s.Ng;
    } else {
        s.N = normalize((n1 * bx + n2 * by) + n3 * bz);
        s.N = normalize(rowvec_mat3(s.N, in.inv3));
    }
    V3 view = -raydir;
    if (dot(s.Ng, view) < 0.0f) { s.N = -s.N; s.Ng = -s.Ng; s.inside = true; } else { s.inside = false; }
}
__device__ inline void surface_frame(SurfaceFrame& s, V3 raydir, bool geo_only, const MatResolved& mr, const TexTaps& normal_taps) {
    V3 view = -raydir;
    V3 up = fabs_(s.N.z) < 0.9999999f ? v3(0.0f, 0.0f, 1.0f) : v3(1.0f, 0.0f, 0.0f);
    s.T = normalize(cross(up, s.N));
    s.B = normalize(cross(s.N, s.T));
    if (!geo_only) {
        V3 nv;
        if (mr.flags & kMatNormal) nv = v3(mr.nmap[0], mr.nmap[1], mr.nmap[2]);
        else { V4 nm = tex_finish(normal_taps); nv = v3(nm.x * 2.0f - 1.0f, nm.y * 2.0f - 1.0f, nm.z * 2.0f - 1.0f); }
        s.N = s.tangent_to_world(nv);
    }
    float nv = dot(s.N, view);
    if (nv < 0.0f) s.N = normalize(s.N - view * (nv - 0.01f));
    V3 pr = normalize(reflect(-view, s.N));
    if (dot(pr, s.Ng) < 0.0f) s.N = normalize(s.N + s.Ng * (0.1f + dot(s.N, s.Ng)));
    s.T = normalize(cross(s.N, up));
    s.B = normalize(cross(s.N, s.T));
}
__device__ inline void rotate_tangents(SurfaceFrame& s, float sn, float cs) {  // Surface.slang:129-136, sincos per material
    s.T = (s.T * cs + cross(s.N, s.T) * sn) + (s.N * dot(s.N, s.T)) * (1.0f - cs);
    s.B = cross(s.T, s.N);
}

// ------------------------------------------------------------------ Material.slang
struct Eval { V3 f; float pdf; };

struct Bsdf {
    V3 base, spec, emissive;
    float metallic, roughness, ior, transmission, anisotropy;
    float eta, ax, ay;
    float pm, pd, pg;  // normalised lobe probabilities (Material.slang:97-106 == 170-179)
    const float *lut_r, *lut_o, *lut_i;
    bool ec;
    // What every evaluation of one hit shares (set_view): the expressions below are the ones eval() used to evaluate per call, with
    // the same operands — each is now evaluated once per hit instead of once per evaluated direction (up to three).
    float inv_4vz;       // 1 / (4 V.z): the divisor of EvaluateReflection's f, applied as in V3 operator/ (multiply by the reciprocal)
    float ec_c, inv_ec_r, inv_ec_g;  // (1 - ec_r) / ec_r, 1 / ec_r, 1 / ec_g
    __device__ inline void set_view(V3 V, float ec_r, float ec_g) {
        inv_4vz = 1.0f / (4.0f * V.z);
        ec_c = (1.0f - ec_r) / ec_r; inv_ec_r = 1.0f / ec_r; inv_ec_g = 1.0f / ec_g;
    }

    __device__ inline float fresnel(float c) const {  // :434-449
        float st2 = eta * eta * (1.0f - c * c);
        if (st2 > 1.0f) return 1.0f;
        float ct = sqrt_(max_(1.0f - st2, 0.0f));
        float rs = (eta * ct - c) / (eta * ct + c);
        float rp = (eta * c - ct) / (eta * c + ct);
        return 0.5f * (rs * rs + rp * rp);
    }
    __device__ inline float ggx_d(V3 h) const {  // :394-404
        float ax2 = ax * ax, ay2 = ay * ay;
        return 1.0f / (VPT_PI * ax * ay * pow_((h.x * h.x) / ax2 + (h.y * h.y) / ay2 + h.z * h.z, 2.0f));
    }
    __device__ inline float smith(V3 v) const {  // :406-423
        float vz2 = fabs_(v.z) * fabs_(v.z);
        float lam = (-1.0f + sqrt_(1.0f + ((ax * ax) * (v.x * v.x) + (ay * ay) * (v.y * v.y)) / vz2)) / 2.0f;
        return 1.0f / (1.0f + lam);
    }
    // EvaluateReflection, :331-351, split into the part that does not depend on the Fresnel colour
    // (H, D, G1(L), pdf — identical for the metallic, dielectric-specular and glass-reflection lobes of one
    // direction pair) and the colour part.  gv = G1(V) depends on V only and is shared by every
    // evaluation of a hit.  Each expression is evaluated exactly as upstream writes it, just once.
    struct ReflCommon { float D, GL, pdf; bool valid; };
    __device__ inline ReflCommon reflection_common(V3 V, V3 L, V3 H, float gv) const {
        ReflCommon c; c.D = 0.0f; c.GL = 0.0f; c.pdf = 0.0f; c.valid = !(L.z <= 1e-5f);
        if (!c.valid) return c;
        float VdotH = dot(V, H);
        c.D = ggx_d(H);
        c.GL = smith(L);
        c.pdf = (gv * max_(VdotH, 0.0f) * c.D / V.z) / (4.0f * VdotH);
        return c;
    }
    __device__ inline V3 reflection_f(const ReflCommon& c, V3 V, V3 F, float gv) const {
        if (!c.valid) return v3s(0.0f);
        return (((F * c.D) * gv) * c.GL) * inv_4vz;   // == ... / (4.0f * V.z)
    }
    __device__ inline Eval refraction(V3 V, V3 L, V3 F, float gv) const {  // :359-387
        Eval e; e.f = v3s(0.0f); e.pdf = 0.0f;
        if (L.z >= 1e-5f) return e;
        V3 H = normalize(V * eta + L);
        if (H.z < 0.0f) H = -H;
        float VdotH = dot(V, H), LdotH = dot(L, H);
        float D = ggx_d(H);
        float GL = smith(L);
        float G = gv * GL;
        float den = LdotH + eta * VdotH;
        float den2 = den * den, eta2 = eta * eta;
        float jac = (eta2 * fabs_(LdotH)) / den2;
        e.pdf = (gv * fabs_(VdotH) * D / V.z) * jac;
        e.f = (((F * D) * G) * eta2 / den2) * (fabs_(VdotH) * fabs_(LdotH) / fabs_(V.z));
        return e;
    }
    // EvaluateBSDF, :167-254.  ec_r / ec_g (the two LUT taps) and gv = G1(V) depend on V only, so the
    // caller computes them once per hit and passes them to all (up to three) evaluations.
    __device__ inline Eval eval(V3 V, V3 L, float ec_r, float ec_g, float gv) const {
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
        float F = fresnel(fabs_(dot(V, H)));
        Eval r; r.f = v3s(0.0f); r.pdf = 0.0f;
        if (!refracted) {
            ReflCommon rc = reflection_common(V, L, H, gv);  // H == normalize(V + L) in every reflection lobe
            // metallic (:266-283)
            {
                float m = clamp_(1.0f - dot(V, H), 0.0f, 1.0f);
                float m2 = m * m;
                V3 ef = reflection_f(rc, V, lerp(base, spec, m2 * m2 * m), gv);
                if (ec) ef = (v3s(1.0f) + base * ec_c) * ef;
                r.f = r.f + ef * pm; r.pdf += rc.pdf * pm;
            }
            // diffuse (:256-264)
            {
                float dpdf = (L.z * VPT_1_OVER_PI) * (L.z > 0.0f ? 1.0f : 0.0f);
                V3 df = (base * VPT_1_OVER_PI) * L.z;
                r.f = r.f + df * pd * (1.0f - F); r.pdf += dpdf * pd * (1.0f - F);
            }
            // dielectric specular (:285-298) and glass reflection (:225-238): both EvaluateReflection(SpecularColor)
            {
                V3 ef = reflection_f(rc, V, spec, gv);
                V3 sf = ef;
                if (ec) sf = sf * inv_ec_r;   // == sf / ec_r
                r.f = r.f + sf * pd * F; r.pdf += rc.pdf * pd * F;
                V3 gf = ef;
                if (ec && ec_g > 0.01f) gf = gf * inv_ec_g;   // == gf / ec_g
                r.f = r.f + gf * pg * F; r.pdf += rc.pdf * pg * F;
            }
        }
        if (refracted && valid_refr) {  // :240-252
            Eval e = refraction(V, L, base, gv);
            if (ec && ec_g > 0.01f) e.f = e.f * inv_ec_g;   // == e.f / ec_g
            r.f = r.f + e.f * pg * (1.0f - F); r.pdf += e.pdf * pg * (1.0f - F);
        }
        return r;
    }
};

// Material.Initialize, :39-87 (+ FURNACE_TEST_MODE override :78-86): the texture-dependent part.  Called per
// hit with the hit's uv, or once per material (uv irrelevant) by k_precompute_materials when every value
// texture is 1x1.
__device__ inline void material_resolve(const DeviceScene& sc, const vpt_material& m, V2 uv, uint32_t flags, MatResolved& r) {
    V4 tb = tex_sample(sc, m.base_color_texture, uv.x, uv.y);
    r.ior = max_(m.ior, 1.000001f);
    r.inv_ior = 1.0f / r.ior;
    r.base[0] = m.base_color[0] * pow_(tb.x, 2.2f); r.base[1] = m.base_color[1] * pow_(tb.y, 2.2f); r.base[2] = m.base_color[2] * pow_(tb.z, 2.2f);
    r.roughness = m.roughness * tex_sample(sc, m.roughness_texture, uv.x, uv.y).x;
    r.metallic = m.metallic * tex_sample(sc, m.metallic_texture, uv.x, uv.y).x;
    V4 te = tex_sample(sc, m.emissive_texture, uv.x, uv.y);
    r.emissive[0] = m.emissive_color[0] * te.x; r.emissive[1] = m.emissive_color[1] * te.y; r.emissive[2] = m.emissive_color[2] * te.z;
    float aspect = sqrt_(1.0f - sqrt_(m.anisotropy) * 0.9f);
    r.aspect = aspect;
    r.ax = max_(0.00001f, r.roughness / aspect);
    r.ay = max_(0.00001f, r.roughness * aspect);
    if (flags & VPT_FLAG_FURNACE) { r.base[0] = r.base[1] = r.base[2] = 1.0f; r.emissive[0] = r.emissive[1] = r.emissive[2] = 0.0f; }
    r.pm = r.metallic;
    r.pd = (1.0f - r.metallic) * (1.0f - m.transmission);
    r.pg = (1.0f - r.metallic) * m.transmission;
    float sum = r.pm + r.pd + r.pg;
    r.pm /= sum; r.pd /= sum; r.pg /= sum;
    sincos_(m.anisotropy_rotation * (VPT_PI / 180.0f), &r.rot_sin, &r.rot_cos);
}
// The same per hit, for the fields whose texture is not 1x1 (MatResolved.flags says which are already valid): material_issue starts
// the texel loads of those textures (and of the normal map) together, material_finish evaluates material_resolve's expressions on them.
struct MatTaps { TexTaps normal, base, rough, metal, emis; };
__device__ inline void material_issue(const uint8_t* texels, const MatResolved& pre, V2 uv, bool geo_only, MatTaps& k) {
    if (!geo_only && !(pre.flags & kMatNormal)) tex_issue(texels, pre.tex[0], uv.x, uv.y, k.normal);
    if (pre.flags & kMatAllValues) return;
    if (!(pre.flags & kMatBase)) tex_issue(texels, pre.tex[1], uv.x, uv.y, k.base);
    if (!(pre.flags & kMatRoughness)) tex_issue(texels, pre.tex[2], uv.x, uv.y, k.rough);
    if (!(pre.flags & kMatMetallic)) tex_issue(texels, pre.tex[3], uv.x, uv.y, k.metal);
    if (!(pre.flags & kMatEmissive)) tex_issue(texels, pre.tex[4], uv.x, uv.y, k.emis);
}
__device__ inline void material_finish(const vpt_material& m, const MatTaps& k, uint32_t flags, MatResolved& r) {
    if (r.flags & kMatAllValues) return;
    if (!(r.flags & kMatBase)) {
        V4 tb = tex_finish(k.base);
        r.base[0] = m.base_color[0] * pow_(tb.x, 2.2f); r.base[1] = m.base_color[1] * pow_(tb.y, 2.2f); r.base[2] = m.base_color[2] * pow_(tb.z, 2.2f);
        if (flags & VPT_FLAG_FURNACE) r.base[0] = r.base[1] = r.base[2] = 1.0f;
    }
    if (!(r.flags & kMatRoughness)) {
        r.roughness = m.roughness * tex_finish(k.rough).x;
        r.ax = max_(0.00001f, r.roughness / r.aspect);
        r.ay = max_(0.00001f, r.roughness * r.aspect);
    }
    if (!(r.flags & kMatMetallic)) {
        r.metallic = m.metallic * tex_finish(k.metal).x;
        r.pm = r.metallic;
        r.pd = (1.0f - r.metallic) * (1.0f - m.transmission);
        r.pg = (1.0f - r.metallic) * m.transmission;
        float sum = r.pm + r.pd + r.pg;
        r.pm /= sum; r.pd /= sum; r.pg /= sum;
    }
    if (!(r.flags & kMatEmissive)) {
        V4 te = tex_finish(k.emis);
        r.emissive[0] = m.emissive_color[0] * te.x; r.emissive[1] = m.emissive_color[1] * te.y; r.emissive[2] = m.emissive_color[2] * te.z;
        if (flags & VPT_FLAG_FURNACE) r.emissive[0] = r.emissive[1] = r.emissive[2] = 0.0f;
    }
}
__device__ inline void bsdf_init(Bsdf& b, const DeviceScene& sc, const vpt_material& m, const MatResolved& pre, const MatTaps& taps, bool inside,
                                 uint32_t flags, V3& medium_color, float& medium_density, float& medium_aniso, float& aniso_rotation) {
    MatResolved r = pre;
    material_finish(m, taps, flags, r);
    b.ior = r.ior;
    b.base = v3(r.base[0], r.base[1], r.base[2]);
    b.roughness = r.roughness; b.metallic = r.metallic;
    b.emissive = v3(r.emissive[0], r.emissive[1], r.emissive[2]);
    b.spec = ld3(m.specular_color);
    b.transmission = m.transmission;
    b.anisotropy = m.anisotropy;
    b.ax = r.ax; b.ay = r.ay;
    b.eta = inside ? r.ior : r.inv_ior;
    medium_color = ld3(m.medium_color);
    medium_density = m.medium_density; medium_aniso = m.medium_anisotropy; aniso_rotation = m.anisotropy_rotation;
    if (flags & VPT_FLAG_FURNACE) { b.spec = v3s(1.0f); medium_color = v3s(1.0f); }
    b.ec = (flags & VPT_FLAG_ENERGY_COMPENSATION) != 0;
    b.lut_r = sc.lut_r; b.lut_o = sc.lut_o; b.lut_i = sc.lut_i;
    b.pm = r.pm; b.pd = r.pd; b.pg = r.pg;
}

// ------------------------------------------------------------------ Sampler.slang
__device__ inline V2 random_circle(Rng& r) {  // :102-112
    float u1 = r.uf(), u2 = r.uf();
    float s, c; sincos_(2.0f * VPT_PI * u1, &s, &c);
    float rad = sqrt_(u2);
    V2 o; o.x = rad * c; o.y = rad * s; return o;
}
__device__ inline V3 random_sphere(Rng& r) {  // :114-133
    float u1 = r.uf(), u2 = r.uf();
    float s, c; sincos_(2.0f * VPT_PI * u1, &s, &c);
    float z = 1.0f - 2.0f * u2;
    float rad = sqrt_(1.0f - z * z);
    return v3(rad * c, rad * s, z);
}
__device__ inline V3 ggx_sample(Rng& r, V3 Ve, float ax, float ay) {  // :141-166 (Heitz 2018 VNDF)
    float u1 = r.uf(), u2 = r.uf();
    V3 Vh = normalize(v3(ax * Ve.x, ay * Ve.y, fabs_(Ve.z)));
    float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
    V3 T1 = lensq > 0.0f ? v3(-Vh.y, Vh.x, 0.0f) * (1.0f / sqrt_(lensq)) : v3(1.0f, 0.0f, 0.0f);
    V3 T2 = cross(Vh, T1);
    float rad = sqrt_(u1);
    float sp, cp; sincos_(2.0f * VPT_PI * u2, &sp, &cp);
    float t1 = rad * cp, t2 = rad * sp;
    float s = 0.5f * (1.0f + Vh.z);
    t2 = (1.0f - s) * sqrt_(1.0f - t1 * t1) + s * t2;
    V3 Nh = (t1 * T1 + t2 * T2) + sqrt_(max_(0.0f, 1.0f - t1 * t1 - t2 * t2)) * Vh;
    return normalize(v3(ax * Nh.x, ay * Nh.y, max_(0.0f, Nh.z)));
}
__device__ inline V3 sample_hg(Rng& r, V3 dir, float G) {  // :168-193
    float r1 = r.uf(), r2 = r.uf();
    float ct;
    if (fabs_(G) < 1e-5f) ct = 2.0f * r1 - 1.0f;
    else { float sq = (1.0f - G * G) / (1.0f - G + 2.0f * G * r1); ct = (1.0f + G * G - sq * sq) / (2.0f * G); }
    float sp, cp; sincos_(2.0f * VPT_PI * r2, &sp, &cp);
    float st = sqrt_(1.0f - ct * ct);
    V3 nd = v3(st * cp, st * sp, ct);
    V3 up = fabs_(dir.y) < 0.9999999f ? v3(0.0f, 1.0f, 0.0f) : v3(0.0f, 0.0f, 1.0f);
    V3 t = normalize(cross(up, dir));
    V3 b = cross(dir, t);
    return normalize((nd.x * t + nd.y * b) + nd.z * dir);
}
// ImportanceSampleEnvMap, :286-346 (3 draws)
// `need_direction`: the caller uses to_light even when the sample carries no light (media whose transmittance is tracked
// along the shadow ray), so the all-black shortcut does not apply
__device__ inline void sample_env(const DeviceScene& sc, const RenderParams& P, Rng& r, V3& to_light, V4& out, bool need_direction = false) {
    if (sc.env_black && !need_direction) {
        // every texel and every pdf is exactly 0: the bilinear result is +0 for any direction, so only the
        // three draws (Sampler.slang:289) have an effect
        r.s = pcg_hash(pcg_hash(pcg_hash(r.s)));
        to_light = v3s(0.0f); out = v4(0.0f, 0.0f, 0.0f, 0.0f);
        return;
    }
    float x0 = r.uf(), x1 = r.uf(), x2 = r.uf();
    uint32_t w = sc.env_w, h = sc.env_h, size = w * h;
    uint32_t idx = (uint32_t)(x0 * (float)size);
    idx = idx < size - 1 ? idx : size - 1;
    AliasEntry e = sc.alias[idx];
    uint32_t ei;
    if (x1 < e.importance) { ei = idx; x1 /= e.importance; }
    else { ei = e.alias; x1 = (x1 - e.importance) / (1.0f - e.importance); }
    uint32_t px, py;
    if ((w & (w - 1u)) == 0u) { px = ei & (w - 1u); py = ei >> (31u - (uint32_t)__builtin_clz(w)); }   // power-of-two width: no integer division
    else { px = ei % w; py = ei / w; }
    float u = ((float)px + x1) / (float)w;
    float sp, cp; sincos_(u * (2.0f * VPT_PI) - VPT_PI, &sp, &cp);
    // (Measured and not kept, round 6: the two cosines below depend on the texel ROW alone; a per-row table of them, filled on the device by these very
    // expressions, made the shade stage 1.2 % SLOWER on the atrium and the bust — the stage waits on its gathers, and the table adds a dependent one:
    // profiles/r06_envrows_pairwise_ab.log.)
    float step = VPT_PI / (float)h;
    float theta0 = (float)py * step;
    float ct = cos_(theta0) * (1.0f - x2) + cos_(theta0 + step) * x2;
    float theta = acos_(clamp_(ct, -1.0f, 1.0f));
    float st = sin_(theta);
    float v = theta * VPT_1_OVER_PI;
    to_light = v3(sp * st, -ct, (-cp) * st);
    to_light = rotate_sc(to_light, v3(0.0f, 1.0f, 0.0f), P.sky_rot[0], P.sky_rot[1]);   // rotate(.., sky_azimuth / 180 * pi)
    to_light = rotate_sc(to_light, v3(1.0f, 0.0f, 0.0f), P.sky_rot[2], P.sky_rot[3]);   // rotate(.., sky_altitude / 180 * pi)
    out = env_sample(sc, u, v);
    out.x *= P.sky_intensity; out.y *= P.sky_intensity; out.z *= P.sky_intensity;
}
// World-space data of one light triangle (Sampler.slang:375-404), evaluated once per emissive triangle by
// k_precompute_emissive; `area` is also the area ClosestHit.slang:276-287 recomputes when a path hits the light.
__device__ inline void emissive_tri_compute(const DeviceScene& sc, const EmissiveDesc& em, uint32_t ti, EmissiveTri& t) {
    const MeshDesc me = sc.meshes[em.mesh];
    const uint32_t* idx = sc.indices + me.index_offset + ti * 3;
    const vpt_vertex* vb = sc.vertices + me.vertex_offset;
    const vpt_vertex &a = vb[idx[0]], &b = vb[idx[1]], &c = vb[idx[2]];
    V3 p0 = mat_point(em.xform, ld3(a.position)), p1 = mat_point(em.xform, ld3(b.position)), p2 = mat_point(em.xform, ld3(c.position));
    V3 nrm = normalize(cross(p2 - p0, p1 - p0));
    t.area = length(cross(p1 - p0, p2 - p0)) * 0.5f;
    t.p0[0] = p0.x; t.p0[1] = p0.y; t.p0[2] = p0.z; t.p1[0] = p1.x; t.p1[1] = p1.y; t.p1[2] = p1.z; t.p2[0] = p2.x; t.p2[1] = p2.y; t.p2[2] = p2.z;
    t.nrm[0] = nrm.x; t.nrm[1] = nrm.y; t.nrm[2] = nrm.z;
    t.u0 = a.texcoord[0]; t.v0 = a.texcoord[1]; t.u1 = b.texcoord[0]; t.v1 = b.texcoord[1]; t.u2 = c.texcoord[0]; t.v2 = c.texcoord[1]; t.pad = 0.0f;
}

// SampleEmissiveTriangle, :348-422 (1+1+2 draws; none if there is no emissive mesh)
// (results are built in locals and assigned once at the end: with stores to the callers' cpdf on two paths the optimiser merges them into one store through a
// selected ADDRESS, which keeps cpdf in scratch memory)
__device__ inline void sample_emissive(const DeviceScene& sc, Rng& r, V3 pos, V3& to_light_out, V4& cpdf_out, uint32_t& gid) {
    gid = 0xffffffffu;
    uint32_t n = sc.emissive_count;
    V3 to_light = v3s(0.0f); V4 cpdf = v4(0.0f, 0.0f, 0.0f, 0.0f);
    if (n == 0) { to_light_out = to_light; cpdf_out = cpdf; return; }
    uint32_t mi = (uint32_t)floor_(r.uf() * (float)n);
    mi = mi < n - 1 ? mi : n - 1;
    const LightSampler ls = sc.lights[mi];
    uint32_t tc = ls.tri_count;
    uint32_t ti = (uint32_t)floor_(r.uf() * (float)tc);
    ti = ti < tc - 1 ? ti : tc - 1;
    gid = ls.gid_base + ti;
    const float4* tq = reinterpret_cast<const float4*>(sc.emissive_tri + ls.tri_base + ti);
    float4 q0 = tq[0], q1 = tq[1], q2 = tq[2], q3 = tq[3], q4 = tq[4];
    V3 p0 = v3(q0.x, q0.y, q0.z), p1 = v3(q1.x, q1.y, q1.z), p2 = v3(q2.x, q2.y, q2.z), nrm = v3(q3.x, q3.y, q3.z);
    float area = q0.w;
    float x0 = r.uf(), x1 = r.uf();
    float su = sqrt_(x0);
    float b0 = 1.0f - su, b1 = x1 * su, b2 = 1.0f - b0 - b1;
    V3 tp = (b0 * p0 + b1 * p1) + b2 * p2;
    float uu = (b0 * q1.w + b1 * q3.w) + b2 * q4.y;
    float vv = (b0 * q2.w + b1 * q4.x) + b2 * q4.z;
    to_light = normalize(tp - pos);
    float d2 = dot(tp - pos, tp - pos);
    float ct = fabs_(dot(nrm, to_light));
    cpdf.w = d2 / ((float)n * (float)tc * area * ct);
    if (sc.all_plain || ls.uniform) {
        cpdf.x = ls.radiance[0]; cpdf.y = ls.radiance[1]; cpdf.z = ls.radiance[2];
    } else {
        TexTaps k;
        tex_issue(sc.texels, ls.tex, uu, vv, k);
        V4 te = tex_finish(k);
        cpdf.x = ls.emissive_color[0] * te.x; cpdf.y = ls.emissive_color[1] * te.y; cpdf.z = ls.emissive_color[2] * te.z;
    }
    to_light_out = to_light; cpdf_out = cpdf;
}

// Camera ray + AA jitter + DOF, RayGen.slang:35-50 (4 draws, the DOF pair always drawn).
__device__ inline void camera_ray(const RenderParams& P, Rng& r, uint32_t x, uint32_t y, V3& origin, V3& direction) {
    float j0 = r.uf(), j1 = r.uf();
    float cx = ((float)x + 0.5f) + (j0 * (0.5f - -0.5f) + -0.5f);
    float cy = ((float)y + 0.5f) + (j1 * (0.5f - -0.5f) + -0.5f);
    float dx = (cx / (float)P.width) * 2.0f - 1.0f, dy = (cy / (float)P.height) * 2.0f - 1.0f;
    V4 o4 = mat_v4(P.view_inv, v4(0.0f, 0.0f, 0.0f, 1.0f));
    origin = v3(o4.x, o4.y, o4.z);
    V4 tg = mat_v4(P.proj_inv, v4(dx, dy, 1.0f, 1.0f));
    V3 tn = normalize(v3(tg.x, tg.y, tg.z));
    V4 dd = mat_v4(P.view_inv, v4(tn.x, tn.y, tn.z, 0.0f));
    direction = v3(dd.x, dd.y, dd.z);
    V3 focus = origin + direction * max_(P.focus_distance, 0.001f);
    V2 rc = random_circle(r);
    float rox = rc.x * 0.5f * P.dof_strength, roy = rc.y * 0.5f * P.dof_strength;
    V3 right = v3(P.view_inv[0], P.view_inv[1], P.view_inv[2]);
    V3 upv = v3(P.view_inv[4], P.view_inv[5], P.view_inv[6]);
    origin = origin + (rox * right + roy * upv);
    direction = normalize(focus - origin);
}

}  // namespace vpt
