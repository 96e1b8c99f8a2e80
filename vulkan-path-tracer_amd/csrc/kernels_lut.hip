// kernels_lut.hip — energy-compensation lookup tables (Turquin 2019) computed on the GPU.
// Replaces LookupTableCalculator::CalculateTable (reference LookupTableCalculator.cpp:44-157) and its two compute
// shaders LookupReflect.slang:24-90 / LookupRefract.slang:23-103.
//
// The reference runs sampleCount/20 dispatches of 20 samples; dispatch i reseeds every cell with
// Sampler(y + x*x + Seed_i) and adds finalValue/20 to the cell, and the host divides by the dispatch count at the
// end.  A cell never talks to another cell, so here ONE thread owns a cell and walks its dispatches in order
// (same fp32 additions in the same order); long runs are cut into several launches that carry the partial sums
// in the table, like the reference's command-buffer splits.  Seed_i = PCG(i*2 + sampleCount + PCG(time_ms)): the
// reference feeds wall-clock milliseconds there, the C-ABI takes the value as an argument so a table is reproducible.
#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "shading.hpp"

namespace vpt {
namespace {

constexpr uint32_t kSamplesPerDispatch = 20u;  // LookupTableCalculator.cpp:80

// KIND 0: LookupReflect; 1: LookupRefract with ABOVE_SURFACE (Eta = 1/ior); 2: BELOW_SURFACE (Eta = ior)
template <int KIND>
__global__ __launch_bounds__(64) void k_lut(float* table, uint32_t sx, uint32_t sy, uint32_t sz, uint32_t sample_count, uint32_t time_hash,
                                            uint32_t first_dispatch, uint32_t n_dispatches) {
    const uint32_t index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= sx * sy * sz) return;
    const uint32_t x = index % sx, y = (index / sx) % sy, z = index / (sx * sy);
    Bsdf m;
    m.base = v3s(1.0f); m.spec = v3s(1.0f); m.emissive = v3s(0.0f);
    m.metallic = 0.0f; m.transmission = 0.0f; m.pm = m.pd = m.pg = 0.0f;
    m.lut_r = m.lut_o = m.lut_i = nullptr; m.ec = false;
    float vc;
    if (KIND == 0) {  // LookupReflect.slang:35-49
        vc = clamp_((float)x / (float)sx, 0.05f, 0.999f);
        const float rough = clamp_((float)y / (float)sy, 0.0001f, 1.0f);
        const float aniso = (float)z / (float)sz;
        const float aspect = sqrt_(1.0f - sqrt_(aniso) * 0.9f);
        m.ax = max_(0.0001f, rough / aspect); m.ay = max_(0.0001f, rough * aspect);
        m.roughness = rough; m.anisotropy = aniso; m.ior = 0.0f; m.eta = 1.0f;
    } else {          // LookupRefract.slang:35-51
        vc = clamp_(pow_((float)x / ((float)sx - 1.0f), 2.0f), 0.01f, 0.9999f);
        const float rough = clamp_((float)y / ((float)sy - 1.0f), 0.01f, 1.0f);
        const float ior = 1.0f + clamp_((float)z / ((float)sz - 1.0f), 0.0001f, 1.0f);
        m.ax = rough; m.ay = rough; m.roughness = rough; m.anisotropy = 0.0f; m.ior = ior;
        m.eta = KIND == 1 ? (1.0f / ior) : ior;
    }
    float cell = table[index];
    for (uint32_t i = first_dispatch; i < first_dispatch + n_dispatches; i++) {
        const uint32_t seed = pcg_hash(i * 2u + sample_count + time_hash);  // LookupTableCalculator.cpp:102
        Rng r; r.s = y + x * x + seed;
        float final_value = 0.0f;
        for (uint32_t k = 0; k < kSamplesPerDispatch; k++) {
            const float mag = sqrt_(1.0f - vc * vc);
            const float phi = r.uf() * VPT_2PI;
            float s, c; sincos_(phi, &s, &c);
            const V3 V = normalize(v3(mag * c, mag * s, vc));
            const V3 H = ggx_sample(r, V, m.ax, m.ay);
            const float gv = m.smith(V);
            m.set_view(V, 1.0f, 1.0f);   // reflection_f's 1 / (4 V.z)
            if (KIND == 0) {
                const V3 L = normalize(reflect(-V, H));
                if (L.z <= 0.0f) continue;
                const Bsdf::ReflCommon rc = m.reflection_common(V, L, normalize(V + L), gv);
                if (rc.pdf <= 0.0f) continue;
                const float f = m.reflection_f(rc, V, v3s(1.0f), gv).x;
                if (isnan_(f) || isinf_(f)) continue;
                final_value += f / rc.pdf;
            } else {
                const float F = m.fresnel(fabs_(dot(V, H)));
                float val = 0.0f;
                if (r.uf() < F) {
                    const V3 L = normalize(reflect(-V, H));
                    if (L.z > 0.0f) {
                        const Bsdf::ReflCommon rc = m.reflection_common(V, L, normalize(V + L), gv);
                        const float f = m.reflection_f(rc, V, v3s(1.0f), gv).x;
                        if (rc.pdf > 0.0f && !isnan_(f) && !isinf_(f)) val += f / rc.pdf;
                    }
                } else {
                    const V3 L = normalize(refract(-V, H, m.eta));
                    if (L.z < 0.0f) {
                        const Eval e = m.refraction(V, L, v3s(1.0f), gv);
                        if (e.pdf > 0.0f && !isnan_(e.f.x) && !isinf_(e.f.x)) val += e.f.x / e.pdf;
                    }
                }
                if (!isnan_(val) && !isinf_(val)) final_value += val;
            }
        }
        cell += final_value / (float)kSamplesPerDispatch;
    }
    table[index] = cell;
}

}  // namespace

void launch_lut(hipStream_t s, int kind, float* table, uint32_t sx, uint32_t sy, uint32_t sz, uint32_t sample_count, uint32_t time_hash,
                uint32_t first_dispatch, uint32_t n_dispatches) {
    const uint32_t cells = sx * sy * sz, blocks = (cells + 63u) / 64u;
    if (kind == 0) hipLaunchKernelGGL(k_lut<0>, dim3(blocks), dim3(64), 0, s, table, sx, sy, sz, sample_count, time_hash, first_dispatch, n_dispatches);
    else if (kind == 1) hipLaunchKernelGGL(k_lut<1>, dim3(blocks), dim3(64), 0, s, table, sx, sy, sz, sample_count, time_hash, first_dispatch, n_dispatches);
    else hipLaunchKernelGGL(k_lut<2>, dim3(blocks), dim3(64), 0, s, table, sx, sy, sz, sample_count, time_hash, first_dispatch, n_dispatches);
}

}  // namespace vpt
