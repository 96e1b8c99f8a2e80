// atmosphere.hpp — planet atmosphere on the device.
// Reference semantics: Atmosphere.slang:7-201 (density profiles, ratio-tracked transmittance, delta-tracked scatter
// distance), Sampler.slang:195-215 (Rayleigh sampling), 431-476 (sun disk / ImportanceSampleSky), RTCommon.slang:174-211
// (sphere intersection, Rayleigh / approximate-Mie phase).  Expression order is part of the parity contract.
#pragma once
#include "volume.hpp"

namespace vpt {

__device__ inline float c_rayleigh(int ch) { return ch == 0 ? 5.802f * 1e-6f : (ch == 1 ? 13.558f * 1e-6f : 33.100f * 1e-6f); }  // Atmosphere.slang:7-11
__device__ inline float c_ozone(int ch) { return ch == 0 ? 0.650f * 1e-6f : (ch == 1 ? 1.881f * 1e-6f : 0.085f * 1e-6f); }
#define VPT_C_MIE_SCATTERING (3.996f * 1e-6f)
#define VPT_C_MIE_ABSORPTION (4.40f * 1e-6f)
#define VPT_C_MIE (VPT_C_MIE_SCATTERING + VPT_C_MIE_ABSORPTION)

__device__ inline V2 intersect_sphere(V3 org, V3 dir, V3 center, float radius) {  // RTCommon.slang:174-192
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
__device__ inline float rayleigh_phase(V3 V, V3 L) { float ct = dot(V, L); return (3.0f / (16.0f * VPT_PI)) * (1.0f + ct * ct); }
__device__ inline float phase_mie(V3 V, V3 L) {  // RTCommon.slang:204-211, g = 0.85
    float ct = dot(V, L);
    float g = min_(0.85f, 0.9381f);
    float k = 1.55f * g - 0.55f * g * g * g;
    float kc = k * ct;
    return (1.0f - k * k) / ((4.0f * VPT_PI) * (1.0f - kc) * (1.0f - kc));
}
__device__ inline V3 sample_rayleigh(Rng& r, V3 dir) {  // Sampler.slang:195-215
    float r1 = r.uf(), r2 = r.uf();
    float u = -pow_(2.0f * (2.0f * r1 - 1.0f) + sqrt_(4.0f * pow_(2.0f * r1 - 1.0f, 2.0f) + 1.0f), 1.0f / 3.0f);
    float ct = u - (1.0f / u);
    float sp, cp; sincos_(2.0f * VPT_PI * r2, &sp, &cp);
    float st = sqrt_(1.0f - ct * ct);
    V3 nd = v3(st * cp, st * sp, ct);
    V3 up = fabs_(dir.y) < 0.9999999f ? v3(0.0f, 1.0f, 0.0f) : v3(0.0f, 0.0f, 1.0f);
    V3 t = normalize(cross(up, dir));
    V3 b = cross(dir, t);
    return normalize((nd.x * t + nd.y * b) + nd.z * dir);
}
__device__ inline void sample_sun_disk(const DeviceScene& sc, const RenderParams& P, Rng& r, float sun_theta, V3& to_light, V4& cpdf) {  // Sampler.slang:431-462
    float az = P.sky_azimuth / 180.0f * VPT_PI, al = P.sky_altitude / 180.0f * VPT_PI;
    V3 sun = rotate(v3(0.0f, 0.0f, -1.0f), v3(1.0f, 0.0f, 0.0f), al);
    sun = rotate(sun, v3(0.0f, 1.0f, 0.0f), az);
    float ctm = cos_(sun_theta);
    float phi = 2.0f * VPT_PI * r.uf();
    float ct = lerp(ctm, 1.0f, r.uf());
    float st = sqrt_(1.0f - ct * ct);
    float sp, cp; sincos_(phi, &sp, &cp);
    V3 local = v3(cp * st, sp * st, ct);
    V3 w = normalize(sun);
    V3 up = fabs_(w.z) < 0.999f ? v3(0.0f, 0.0f, 1.0f) : v3(1.0f, 0.0f, 0.0f);
    V3 u = normalize(cross(up, w));
    V3 v = cross(w, u);
    to_light = (u * local.x + v * local.y) + w * local.z;
    float solid = 2.0f * VPT_PI * (1.0f - ctm);
    V3 c = (2e5f * ld3(sc.atm.sun_color)) * P.sky_intensity;
    cpdf = v4(c.x, c.y, c.z, 1.0f / solid);
}
// Sampler::ImportanceSampleSky, Sampler.slang:464-476
__device__ inline void sample_sky(const DeviceScene& sc, const RenderParams& P, Rng& r, V3& to_light, V4& out) {
    if (sc.atm_on) sample_sun_disk(sc, P, r, 0.004675f, to_light, out);
    else sample_env(sc, P, r, to_light, out, sc.hetero != 0u);
}
__device__ inline float atmosphere_height(const DeviceScene& sc, V3 p) { return length(p - ld3(sc.atm.planet_position)) - sc.atm.planet_radius; }
__device__ inline float rayleigh_density(const DeviceScene& sc, float h) { return exp_(-h / sc.atm.rayleigh_density_falloff); }
__device__ inline float mie_density(const DeviceScene& sc, float h) { return exp_(-h / sc.atm.mie_density_falloff); }
__device__ inline float ozone_density(const DeviceScene& sc, float h) { return exp_(-(fabs_(h - sc.atm.ozone_peak) / sc.atm.ozone_density_falloff)); }
struct AtmCoef { float ray, mie, ozo, majorant; };
__device__ inline AtmCoef atmosphere_coefficients(const DeviceScene& sc, int ch) {  // Atmosphere.slang:52-60 == 151-155
    AtmCoef k;
    k.ray = c_rayleigh(ch) * sc.atm.rayleigh_multiplier[ch];
    k.mie = VPT_C_MIE * sc.atm.mie_multiplier[ch];
    k.ozo = c_ozone(ch) * sc.atm.ozone_multiplier[ch];
    k.majorant = (rayleigh_density(sc, 0.0f) * k.ray + mie_density(sc, 0.0f) * k.mie) + ozone_density(sc, sc.atm.ozone_peak) * k.ozo;
    return k;
}
// CalculateTransmittanceThroughAtmosphere, Atmosphere.slang:33-107: only component `ch` of the result is set
__device__ inline V3 atmosphere_transmittance(const DeviceScene& sc, Rng& r, V3 org, V3 dir, int ch) {
    V2 planet = intersect_sphere(org, dir, ld3(sc.atm.planet_position), sc.atm.planet_radius);
    if (planet.y > 0.0f) return v3s(0.0f);
    V2 at = intersect_sphere(org, dir, ld3(sc.atm.planet_position), sc.atm.planet_radius + sc.atm.atmosphere_height);
    float tmin = max_(at.x, 0.0f), tmax = at.y;
    if (tmax < 0.0f) return v3s(1.0f);
    AtmCoef k = atmosphere_coefficients(sc, ch);
    if (k.majorant <= 0.0f) return v3s(1.0f);
    float t = 0.0f, tr = 1.0f;
    for (int i = 0; i < 1000; i++) {
        float dt = -log_(1.0f - r.uf()) / k.majorant;
        t += dt;
        if (t >= tmax - tmin) break;
        float h = atmosphere_height(sc, org + dir * (t + tmin));
        if (h < 0.0f) break;
        float dr = rayleigh_density(sc, h) * k.ray, dm = mie_density(sc, h) * k.mie, dz = ozone_density(sc, h) * k.ozo;
        tr *= 1.0f - (dr + dm + dz) / k.majorant;
        float p = tr;
        if (r.uf() > p) { tr = 0.0f; break; }
        tr /= p;
    }
    V3 out = v3s(0.0f);
    if (ch == 0) out.x = tr; else if (ch == 1) out.y = tr; else out.z = tr;
    return out;
}
// SampleAtmosphereScatterDistance, Atmosphere.slang:117-201; comp 0 Rayleigh, 1 Mie, 2 ozone, -1 none
__device__ inline float atmosphere_scatter_distance(const DeviceScene& sc, Rng& r, V3 org, V3 dir, int ch, int& comp) {
    V2 at = intersect_sphere(org, dir, ld3(sc.atm.planet_position), sc.atm.planet_radius + sc.atm.atmosphere_height);
    float tmin_a = max_(at.x, 0.0f), tmax_a = at.y;
    comp = -1;
    V2 planet = intersect_sphere(org, dir, ld3(sc.atm.planet_position), sc.atm.planet_radius);
    float tmin_p = planet.x;
    if (tmax_a < 0.0f) return -1.0f;
    AtmCoef k = atmosphere_coefficients(sc, ch);
    if (k.majorant <= 0.0f) return -1.0f;
    float t = tmin_a;
    for (int i = 0; i < 1000; i++) {
        float dt = -log_(1.0f - r.uf()) / k.majorant;
        t += dt;
        if (t >= tmax_a) break;
        if (tmin_p > 0.0f && t >= tmin_p) break;
        float h = atmosphere_height(sc, org + dir * t);
        float dr = rayleigh_density(sc, h) * k.ray, dm = mie_density(sc, h) * k.mie, dz = ozone_density(sc, h) * k.ozo;
        float dens = (dr + dm) + dz;
        if (dens / k.majorant < r.uf()) continue;
        float pr = dr / dens, pm = dm / dens;
        float x = r.uf();
        if (x <= pr) comp = 0; else if (x <= pr + pm) comp = 1; else comp = 2;
        return t;
    }
    return -1.0f;
}
// The per-channel product surface / volume NEE applies to a sky sample (ClosestHit.slang:335-349 == RayGen.slang:328-343)
__device__ inline V3 nee_atmosphere_transmittance(const DeviceScene& sc, Rng& r, V3 tr, V3 org, V3 dir, int color_channel) {
    if (color_channel == -1) {
        tr.x *= atmosphere_transmittance(sc, r, org, dir, 0).x;
        tr.y *= atmosphere_transmittance(sc, r, org, dir, 1).y;
        tr.z *= atmosphere_transmittance(sc, r, org, dir, 2).z;
        return tr;
    }
    return tr * atmosphere_transmittance(sc, r, org, dir, color_channel);
}

// ScatteredInVolume, RayGen.slang:162-263: the nearest scatter among the boxes and (with an atmosphere) the
// delta-tracked atmosphere collision — which first fixes the colour channel the collision is sampled for — wins if it
// lies before the geometry (`dgeo` = GetDistanceToGeometry, < 0: none).
// Returns -1: no scatter, -2: atmosphere (component `comp`, channel `cc`), >= 0: box index; `sd` = distance.
__device__ inline int scattered_in_media(const DeviceScene& sc, V3 org, V3 dir, Rng& r, float dgeo, float depth, int cc_in, float& sd, int& comp, int& cc) {
    int sv = nearest_box_scatter(sc, org, dir, r, depth, sd);
    cc = cc_in; comp = -1;
    if (sc.atm_on) {
        if (cc == -1) { float pick = r.uf(); cc = pick < 0.33333f ? 0 : (pick < 0.66666f ? 1 : 2); }
        float ad = atmosphere_scatter_distance(sc, r, org, dir, cc, comp);
        if (ad >= 0.0f && (ad < sd || sd < 0.0f)) { sd = ad; sv = -2; }
    }
    if (sd >= 0.0f && (dgeo < 0.0f || sd < dgeo)) return sv;
    return -1;
}

}  // namespace vpt
