// volume.hpp — homogeneous box volumes on the device.
// Reference semantics: Volume.slang:183-207 (ray/box), 141-147 (depth-dependent anisotropy), 261-297 (free-flight
// sampling, homogeneous branch), 350-406 (phase functions), 419-446 (Beer-Lambert transmittance through every box);
// Sampler.slang:168-284 (HG / Draine / HG+Draine sampling), RTCommon.slang:213-227 (phase pdfs).
// Expression order is part of the parity contract (compiled with -ffp-contract=off, vpt_fp32.h math).
#pragma once
#include "shading.hpp"

namespace vpt {

__device__ inline V3 sample_draine(Rng& r, V3 dir, float g, float a) {  // Sampler.slang:217-266
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
    float sp, cp; sincos_(2.0f * VPT_PI * r2, &sp, &cp);
    float st = sqrt_(1.0f - ct * ct);
    V3 nd = v3(st * cp, st * sp, ct);
    V3 up = fabs_(dir.y) < 0.9999999f ? v3(0.0f, 1.0f, 0.0f) : v3(0.0f, 0.0f, 1.0f);
    V3 t = normalize(cross(up, dir));
    V3 b = cross(dir, t);
    return normalize((nd.x * t + nd.y * b) + nd.z * dir);
}
struct HgDraineFit { float ghg, gd, alpha_d, w_d; };
__device__ inline HgDraineFit hg_draine_fit(float d) {  // Sampler.slang:271-274 == Volume.slang:397-400
    HgDraineFit f;
    f.ghg = exp_(-(0.0990567f / (d - 1.67154f)));
    f.gd = exp_(-(2.20679f / (d + 3.91029f)) - 0.428934f);
    f.alpha_d = exp_(3.62489f - (8.29288f / (d + 5.52825f)));
    f.w_d = exp_(-(0.599085f / (d - 0.641583f)) - 0.665888f);
    return f;
}
__device__ inline float phase_hg(V3 V, V3 L, float g) {  // RTCommon.slang:213-220
    if (g == 0.0f) return 1.0f / (4.0f * VPT_PI);
    float ct = dot(V, L);
    return (1.0f / (4.0f * VPT_PI)) * ((1.0f - g * g) / pow_(1.0f + g * g - 2.0f * g * ct, 1.5f));
}
__device__ inline float phase_draine(V3 V, V3 L, float g, float a) {  // RTCommon.slang:222-227
    float ct = dot(V, L);
    return ((1.0f - g * g) * (1.0f + a * ct * ct)) / (4.0f * (1.0f + (a * (1.0f + 2.0f * g * g)) / 3.0f) * VPT_PI * pow_(1.0f + g * g - 2.0f * g * ct, 1.5f));
}
struct VolIsect { float tn, tf; };
__device__ inline VolIsect ray_aabb(V3 org, V3 dir, const float* bmin, const float* bmax) {  // Volume.slang:183-207
    V3 inv = v3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
    V3 t0 = (ld3(bmin) - org) * inv, t1 = (ld3(bmax) - org) * inv;
    V3 ts = v3(min_(t0.x, t1.x), min_(t0.y, t1.y), min_(t0.z, t1.z));
    V3 tb = v3(max_(t0.x, t1.x), max_(t0.y, t1.y), max_(t0.z, t1.z));
    VolIsect r;
    r.tn = max_(max_(ts.x, ts.y), max_(ts.x, ts.z));
    r.tf = min_(min_(tb.x, tb.y), min_(tb.x, tb.z));
    if (r.tf < 0.0f || r.tn > r.tf) { r.tn = -1.0f; r.tf = -1.0f; }
    return r;
}
__device__ inline float effective_anisotropy(const vpt_volume& v, float depth) {  // Volume.slang:141-147
    if (v.approximated_scattering != 0) {
        float s = v.anisotropy > 0.0f ? 1.0f : (v.anisotropy < 0.0f ? -1.0f : 0.0f);
        return pow_(fabs_(v.anisotropy), 1.0f + depth) * s;
    }
    return v.anisotropy;
}
__device__ inline V3 volume_scatter_direction(uint32_t phase, const vpt_volume& v, V3 dir, Rng& r, uint32_t depth) {  // Volume.slang:350-368
    if (phase == VPT_PHASE_HENYEY_GREENSTEIN) return sample_hg(r, dir, effective_anisotropy(v, (float)depth));
    if (phase == VPT_PHASE_DRAINE) return sample_draine(r, dir, effective_anisotropy(v, (float)depth), v.alpha);
    HgDraineFit f = hg_draine_fit(v.droplet_size);  // Sampler.slang:268-284
    float ghg = pow_(max_(f.ghg, 0.0f), 1.0f + (float)depth);
    float gd = pow_(max_(f.gd, 0.0f), 1.0f + (float)depth);
    float u = r.uf();
    if (u < f.w_d) return sample_hg(r, dir, ghg);
    return sample_draine(r, dir, gd, f.alpha_d);
}
__device__ inline float volume_phase(uint32_t phase, const vpt_volume& v, V3 V, V3 L, uint32_t depth) {  // Volume.slang:370-406
    if (phase == VPT_PHASE_HENYEY_GREENSTEIN) return phase_hg(V, L, effective_anisotropy(v, (float)depth));
    if (phase == VPT_PHASE_DRAINE) return phase_draine(V, L, effective_anisotropy(v, (float)depth), v.alpha);
    HgDraineFit f = hg_draine_fit(v.droplet_size);
    return lerp(phase_hg(V, L, f.ghg), phase_draine(V, L, f.gd, f.alpha_d), f.w_d);
}
// ---- heterogeneous boxes: density from a dense grid (the reference's NanoVDB tree, densified) --------------------
__device__ inline float effective_density(const vpt_volume& v, float base, float depth) {  // Volume.slang:149-156
    if (v.approximated_scattering != 0) return base * pow_(v.approximated_scattering_falloff, depth);
    return base;
}
// SampleNanoVDBBuffer, Volume.slang:69-117, on a dense grid whose index box is [0, dim)
__device__ inline float sample_density_grid(const DeviceScene& sc, const vpt_volume& v, Rng& r, V3 x) {
    const DensityGrid& g = sc.grids[v.density_data_index];
    V3 n = (x - ld3(v.corner_min)) / (ld3(v.corner_max) - ld3(v.corner_min));
    n.y = 1.0f - n.y;
    V3 gp = n * v3((float)g.dim[0], (float)g.dim[1], (float)g.dim[2]);
    int cx = f2i_clamped(floor_(gp.x), -1, (int)g.dim[0]), cy = f2i_clamped(floor_(gp.y), -1, (int)g.dim[1]), cz = f2i_clamped(floor_(gp.z), -1, (int)g.dim[2]);
    r.s = pcg_hash(r.s); cx += (int)(r.s % 3u) - 1;
    r.s = pcg_hash(r.s); cy += (int)(r.s % 3u) - 1;
    r.s = pcg_hash(r.s); cz += (int)(r.s % 3u) - 1;
    cx = min(max(cx, 0), (int)g.dim[0] - 1); cy = min(max(cy, 0), (int)g.dim[1] - 1); cz = min(max(cz, 0), (int)g.dim[2] - 1);
    float value = g.values[(size_t)cx + (size_t)cy * g.dim[0] + (size_t)cz * g.dim[0] * g.dim[1]];
    return clamp_(value / g.max_density * v.grid_sharpness, 0.0f, 1.0f);
}
struct VolBlock { int index; V3 lo, hi; };
struct VolTrav { V3 block_size; float eps, t_enter, t_exit; };
__device__ inline VolTrav make_traversal(const vpt_volume& v, VolIsect is) {  // Volume.slang:119-127
    VolTrav c;
    V3 ext = ld3(v.corner_max) - ld3(v.corner_min);
    c.block_size = ext / v3s(32.0f);
    c.eps = 0.0001f * max_(ext.x, max_(ext.y, ext.z));
    c.t_enter = max_(is.tn, 0.0f);
    c.t_exit = is.tf;
    return c;
}
__device__ inline VolBlock block_info(const vpt_volume& v, V3 pos, const VolTrav& c) {  // Volume.slang:129-147
    V3 rel = (pos - ld3(v.corner_min)) / (ld3(v.corner_max) - ld3(v.corner_min));
    int ix = f2i_clamped(rel.x * 32.0f, 0, 31), iy = f2i_clamped(rel.y * 32.0f, 0, 31), iz = f2i_clamped(rel.z * 32.0f, 0, 31);
    VolBlock b;
    b.index = ix + iy * 32 + iz * 32 * 32;
    b.lo = ld3(v.corner_min) + c.block_size * v3((float)ix, (float)iy, (float)iz);
    b.hi = b.lo + c.block_size;
    return b;
}
__device__ inline VolIsect ray_aabb3(V3 org, V3 dir, V3 lo, V3 hi) { float a[3] = {lo.x, lo.y, lo.z}, b[3] = {hi.x, hi.y, hi.z}; return ray_aabb(org, dir, a, b); }
__device__ inline V3 blackbody(float kelvin) {  // RTCommon.slang:139-172
    float temp = kelvin / 100.0f;
    float r, g, b;
    if (temp <= 66.0f) r = 255.0f; else r = 329.698727446f * pow_(temp - 60.0f, -0.1332047592f);
    if (temp <= 66.0f) g = 99.4708025861f * log_(temp) - 161.1195681661f; else g = 288.1221695283f * pow_(temp - 60.0f, -0.0755148492f);
    if (temp >= 66.0f) b = 255.0f; else if (temp <= 19.0f) b = 0.0f; else b = 138.5177312231f * log_(temp - 10.0f) - 305.0447927307f;
    V3 c = v3(r, g, b) / 255.0f;
    return v3(clamp_(c.x, 0.0f, 1.0f), clamp_(c.y, 0.0f, 1.0f), clamp_(c.z, 0.0f, 1.0f));
}
// GetEmissionFromTemperatureAtPoint, Volume.slang:233-258 (reads the volume's density grid: see vpt.h)
__device__ inline V3 temperature_emission(const DeviceScene& sc, const vpt_volume& v, Rng& r, V3 x) {
    if (!v.has_temperature_data) return v3s(0.0f);
    float tn = sample_density_grid(sc, v, r, x);
    V3 color;
    if (v.use_blackbody) color = blackbody(tn * (float)(v.kelvin_max - v.kelvin_min) + (float)v.kelvin_min);
    else color = ld3(v.temperature_color);
    float intensity = pow_(tn, v.temperature_gamma) * v.temperature_scale;
    return intensity * v3(pow_(color.x, v.emissive_color_gamma), pow_(color.y, v.emissive_color_gamma), pow_(color.z, v.emissive_color_gamma));
}
// ProcessHeterogeneousVolumeScattering, Volume.slang:299-348: delta tracking block by block
__device__ inline float heterogeneous_scatter(const DeviceScene& sc, const vpt_volume& v, V3 org, V3 dir, Rng& r, float depth, VolIsect is) {
    const DensityGrid& g = sc.grids[v.density_data_index];
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
        float dens = effective_density(v, sample_density_grid(sc, v, r, pos) * v.density, depth);
        if (dens / maxd < r.uf()) continue;
        return c.t_enter + t;
    }
    return -1.0f;
}
// ProcessHeterogeneousVolumeTransmittance, Volume.slang:448-520: ratio tracking + roulette, block by block
__device__ inline float heterogeneous_transmittance(const DeviceScene& sc, const vpt_volume& v, Rng& r, V3 org, V3 dir, float depth, VolIsect is) {
    const DensityGrid& g = sc.grids[v.density_data_index];
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
        float dens = effective_density(v, sample_density_grid(sc, v, r, pos) * v.density, depth);
        tr *= 1.0f - (dens / maxd);
        float p = tr;
        if (r.uf() > p) return 0.0f;
        tr /= p;
    }
    return tr;
}
// Volume::CalculateVolumesTransmittance, Volume.slang:419-446: Beer-Lambert for homogeneous boxes, tracked (random
// draws) for heterogeneous ones — which is why callers evaluate it only after the shadow ray is known to be clear
__device__ inline float volumes_transmittance(const DeviceScene& sc, Rng& r, V3 org, V3 dir, float depth) {
    float tr = 1.0f;
    for (uint32_t i = 0; i < sc.volume_count; i++) {
        const vpt_volume& v = sc.volumes[i];
        VolIsect is = ray_aabb(org, dir, v.corner_min, v.corner_max);
        is.tn = max_(is.tn, 0.0f);
        if (v.density_data_index >= 0 && is.tf >= 0.0f) {
            tr *= heterogeneous_transmittance(sc, v, r, org, dir, depth, is);
            if (tr <= 0.0f) return 0.0f;
        } else {
            float len = is.tf - is.tn;
            if (len > 0.0f) tr *= exp_(-v.density * len);
        }
    }
    return clamp_(tr, 0.0f, 1.0f);
}
// Volume::DoesRayScatterInVolume, Volume.slang:261-297
__device__ inline float does_ray_scatter(const DeviceScene& sc, const vpt_volume& v, V3 org, V3 dir, Rng& r, float depth, float ignore_if_farther) {
    VolIsect is = ray_aabb(org, dir, v.corner_min, v.corner_max);
    if (is.tf < 0.0f) return -1.0f;
    if (ignore_if_farther >= 0.0f && is.tn > ignore_if_farther) return -1.0f;
    float inside = is.tf - max_(is.tn, 0.0f);
    if (inside <= 0.0f) return -1.0f;
    if (v.density_data_index >= 0) return heterogeneous_scatter(sc, v, org, dir, r, depth, is);
    float sd = -log_(r.uf()) / v.density;  // Sampler.slang:425-428
    if (sd < inside) return max_(is.tn, 0.0f) + sd;
    return -1.0f;
}
// The box part of ScatteredInVolume, RayGen.slang:162-210: boxes in order of entry distance (the reference's
// exchange sort, reproduced literally because ties are common and it is not stable), each crossed box draws a
// free-flight distance, the nearest scatter wins.  Returns the box index or -1; `sd` = its distance (-1: none).
__device__ inline int nearest_box_scatter(const DeviceScene& sc, V3 org, V3 dir, Rng& r, float depth, float& sd) {
    const int n = (int)sc.volume_count;
    float dist[VPT_MAX_VOLUMES]; int idx[VPT_MAX_VOLUMES];
    for (int i = 0; i < n; i++) {
        VolIsect is = ray_aabb(org, dir, sc.volumes[i].corner_min, sc.volumes[i].corner_max);
        dist[i] = max_(0.0f, is.tn); idx[i] = i;
    }
    for (int i = 0; i < n; i++)
        for (int j = i + 1; j < n; j++)
            if (dist[j] < dist[i]) { float td = dist[i]; int ti = idx[i]; dist[i] = dist[j]; idx[i] = idx[j]; dist[j] = td; idx[j] = ti; }
    sd = -1.0f; int sv = -1;
    for (int i = 0; i < n; i++) {
        float t = does_ray_scatter(sc, sc.volumes[idx[i]], org, dir, r, depth, sd);
        if (t >= 0.0f && (t < sd || sd < 0.0f)) { sd = t; sv = idx[i]; }
    }
    return sv;
}

}  // namespace vpt
