// vpt_fp32.h — the fp32 arithmetic contract of the drop-in boundary.
//
// The reference leaves every elementary operation of its shaders to the Vulkan
// driver and the Slang->SPIR-V lowering (sin/cos/pow/log/acos/atan2, normalize,
// reflect/refract, lerp, the ray/triangle test inside VK_KHR_ray_tracing; see
// SURVEY.md §8c "Arithmetic living outside /root/reference"). None of it is
// pinned upstream.  To make "identical scene + identical seed => identical
// radiance" a testable statement, this header pins ONE definition of those
// primitives.  Both sides of the parity test include it:
//   * the HIP kernels under vulkan-path-tracer_amd/csrc (device code), and
//   * the CPU oracle under oracle/ (host code),
// and both are compiled with -ffp-contract=off and without fast-math, so the
// only fused multiply-adds are the explicit vptfp::fma() calls below
// (v_fma_f32 on gfx950, vfmadd on x86-64-v3): the two sides then agree bit for
// bit.  Everything algorithmic (integrator, BSDF, NEE/MIS, BVH, env tables,
// bloom, tonemap) is written separately on each side; only these leaf
// primitives are shared.
//
// Accuracy of the elementary functions vs. correctly rounded results is
// checked in tests/test_fp32_contract.py (<= 4 ulp on the ranges the path uses).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define VPT_HD __host__ __device__ inline
#else
#define VPT_HD static inline
#endif

namespace vptfp {

// ---------------------------------------------------------------- bit helpers
VPT_HD uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
VPT_HD float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
VPT_HD float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
VPT_HD float fabs_(float x) { return u2f(f2u(x) & 0x7fffffffu); }
VPT_HD bool isnan_(float x) { return x != x; }
VPT_HD bool isinf_(float x) { return (f2u(x) & 0x7fffffffu) == 0x7f800000u; }
VPT_HD float sqrt_(float x) { return __builtin_sqrtf(x); }    // IEEE correctly rounded on both sides
VPT_HD float floor_(float x) { return __builtin_floorf(x); }  // exact

// min/max/clamp with ONE defined NaN behaviour (second operand wins on NaN
// compare), so CPU and GPU agree; SPIR-V FMin/FMax leave it undefined.
VPT_HD float min_(float a, float b) { return (b < a) ? b : a; }
VPT_HD float max_(float a, float b) { return (a < b) ? b : a; }
VPT_HD float clamp_(float x, float lo, float hi) { return min_(max_(x, lo), hi); }
VPT_HD float saturate_(float x) { return clamp_(x, 0.0f, 1.0f); }
// float -> int with the out-of-range cases pinned (a bare cast differs between x86 and gfx950 there): NaN and x < lo give lo
VPT_HD int f2i_clamped(float x, int lo, int hi) { if (!(x >= (float)lo)) return lo; if (x >= (float)hi) return hi; return (int)x; }

// ---------------------------------------------------------------- RNG
// PCG hash, reference Sampler.slang:4-9 == PathTracer.cpp:130-134.
VPT_HD uint32_t pcg_hash(uint32_t seed) {
    uint32_t state = seed * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
// float(hash) / float(UINT_MAX): float(UINT_MAX) rounds to 2^32, so this is an
// exact scale of the RNE-converted hash; range [0,1] inclusive (Sampler.slang:38-43).
VPT_HD float u32_to_unit(uint32_t h) { return (float)h * 2.3283064365386963e-10f; }

// ---------------------------------------------------------------- elementary functions
// Cody-Waite reduction by pi/2 (three-term split) + Cephes single-precision
// minimax polynomials. Valid for |x| < ~1e4 (the path feeds angles <= 2*pi and
// degree->radian rotations).
VPT_HD void sincos_(float x, float* s_out, float* c_out) {
    float ax = fabs_(x);
    if (!(ax < 3.0e4f)) {  // out of contract range / NaN / inf
        *s_out = u2f(0x7fc00000u);
        *c_out = u2f(0x7fc00000u);
        return;
    }
    float fj = floor_(ax * 0.636619772367581343f + 0.5f);
    int j = (int)fj;
    float r = fma(fj, -1.5703125f, ax);
    r = fma(fj, -4.837512969970703125e-4f, r);
    r = fma(fj, -7.54978995489188216e-8f, r);
    float z = r * r;
    float ps = fma(fma(fma(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * r, r);
    float pc = fma(fma(fma(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z,
                   fma(-0.5f, z, 1.0f));
    float s, c;
    switch (j & 3) {
        case 0: s = ps; c = pc; break;
        case 1: s = pc; c = -ps; break;
        case 2: s = -ps; c = -pc; break;
        default: s = -pc; c = ps; break;
    }
    if (f2u(x) >> 31) s = -s;
    *s_out = s;
    *c_out = c;
}
VPT_HD float sin_(float x) { float s, c; sincos_(x, &s, &c); return s; }
VPT_HD float cos_(float x) { float s, c; sincos_(x, &s, &c); return c; }

VPT_HD float log_(float x) {
    if (isnan_(x) || x < 0.0f) return u2f(0x7fc00000u);
    if (x == 0.0f) return u2f(0xff800000u);
    if (isinf_(x)) return x;
    int e = 0;
    uint32_t u = f2u(x);
    if (u < 0x00800000u) { x = x * 8388608.0f; u = f2u(x); e = -23; }  // denormal
    e += (int)(u >> 23) - 126;
    float m = u2f((u & 0x007fffffu) | 0x3f000000u);  // [0.5,1)
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; } else { m = m - 1.0f; }
    float z = m * m;
    float y = 7.0376836292e-2f;
    y = fma(y, m, -1.1514610310e-1f);
    y = fma(y, m, 1.1676998740e-1f);
    y = fma(y, m, -1.2420140846e-1f);
    y = fma(y, m, 1.4249322787e-1f);
    y = fma(y, m, -1.6668057665e-1f);
    y = fma(y, m, 2.0000714765e-1f);
    y = fma(y, m, -2.4999993993e-1f);
    y = fma(y, m, 3.3333331174e-1f);
    y = y * m * z;
    float fe = (float)e;
    y = fma(fe, -2.12194440e-4f, y);
    y = fma(-0.5f, z, y);
    float r = m + y;
    return fma(fe, 0.693359375f, r);
}

// 2^n for n in [-126,127]
VPT_HD float pow2i_(int n) { return u2f((uint32_t)(n + 127) << 23); }

VPT_HD float exp_(float x) {
    if (isnan_(x)) return x;
    if (x > 88.7228317f) return u2f(0x7f800000u);
    if (x < -103.9f) return 0.0f;
    float fx = floor_(fma(x, 1.44269504088896341f, 0.5f));
    float r = fma(fx, -0.693359375f, x);
    r = fma(fx, 2.12194440e-4f, r);
    float z = r * r;
    float p = 1.9875691500e-4f;
    p = fma(p, r, 1.3981999507e-3f);
    p = fma(p, r, 8.3334519073e-3f);
    p = fma(p, r, 4.1665795894e-2f);
    p = fma(p, r, 1.6666665459e-1f);
    p = fma(p, r, 5.0000001201e-1f);
    float y = fma(p, z, r) + 1.0f;
    int n = (int)fx;
    int a = n / 2;
    int b = n - a;
    return y * pow2i_(a) * pow2i_(b);
}

// pow as the GPU drivers do it: exp(y*log(x)); x<0 -> NaN, pow(0,y>0)=0.  The two constant exponents the
// path uses all over (pow(pdf, 2.0f) in PowerHeuristics, pow(., 2.0f) in the GGX distribution, pow(V.z, 1/2)
// for the refraction table) fold to a multiply / a square root, as shader compilers fold them.
VPT_HD float pow_(float x, float y) {
    if (y == 2.0f) return x * x;
    if (y == 0.5f) return sqrt_(x);
    if (y == 0.0f) return 1.0f;
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : u2f(0x7f800000u);
    return exp_(y * log_(x));
}

VPT_HD float asin_(float x) {
    float a = fabs_(x);
    if (!(a <= 1.0f)) return u2f(0x7fc00000u);
    bool big = a > 0.5f;
    float z, s;
    if (big) { z = 0.5f * (1.0f - a); s = sqrt_(z); } else { z = a * a; s = a; }
    float p = 4.2163199048e-2f;
    p = fma(p, z, 2.4181311049e-2f);
    p = fma(p, z, 4.5470025998e-2f);
    p = fma(p, z, 7.4953002686e-2f);
    p = fma(p, z, 1.6666752422e-1f);
    float r = fma(p * z, s, s);
    if (big) r = 1.57079632679489661923f - (r + r);
    return (f2u(x) >> 31) ? -r : r;
}

// asin_ restricted to |x| <= 0.5: its first branch, operation for operation.
VPT_HD float asin_small_(float x) {
    float a = fabs_(x);
    float z = a * a, s = a;
    float p = 4.2163199048e-2f;
    p = fma(p, z, 2.4181311049e-2f);
    p = fma(p, z, 4.5470025998e-2f);
    p = fma(p, z, 7.4953002686e-2f);
    p = fma(p, z, 1.6666752422e-1f);
    float r = fma(p * z, s, s);
    return (f2u(x) >> 31) ? -r : r;
}

// acos(x) = pi - 2 asin(sqrt((1 + x) / 2)) for x < -0.5, 2 asin(sqrt((1 - x) / 2)) for x > 0.5, pi / 2 - asin(x) between.  In all three the
// argument of asin is at most 0.5 (sqrt of less than 0.25 rounds to at most 0.5), so it is always asin_'s first branch: written with ONE
// evaluation of it on the selected argument (1 - x is 1 + (-x) in IEEE arithmetic) — the same operations on every input as three
// separate asin_ calls, at a third of the instructions where lanes of a wave take different branches.
VPT_HD float acos_(float x) {
    if (!(fabs_(x) <= 1.0f)) return u2f(0x7fc00000u);
    const bool lo = x < -0.5f, hi = x > 0.5f;
    float t = x;
    if (lo || hi) t = sqrt_(0.5f * (1.0f + (lo ? x : -x)));
    const float a = asin_small_(t);
    if (lo) return 3.14159265358979323846f - 2.0f * a;
    if (hi) return 2.0f * a;
    return 1.57079632679489661923f - a;
}

VPT_HD float atan_(float x) {
    float a = fabs_(x);
    float y0;
    if (a > 2.414213562373095f) { y0 = 1.57079632679489661923f; a = -(1.0f / a); }
    else if (a > 0.4142135623730950f) { y0 = 0.785398163397448309616f; a = (a - 1.0f) / (a + 1.0f); }
    else { y0 = 0.0f; }
    float z = a * a;
    float p = 8.05374449538e-2f;
    p = fma(p, z, -1.38776856032e-1f);
    p = fma(p, z, 1.99777106478e-1f);
    p = fma(p, z, -3.33329491539e-1f);
    float r = y0 + fma(p * z, a, a);
    return (f2u(x) >> 31) ? -r : r;
}

VPT_HD float atan2_(float y, float x) {
    const float PI = 3.14159265358979323846f;
    if (isnan_(x) || isnan_(y)) return u2f(0x7fc00000u);
    if (x == 0.0f) {
        if (y == 0.0f) return 0.0f;
        return (y > 0.0f) ? 1.57079632679489661923f : -1.57079632679489661923f;
    }
    float r = atan_(y / x);
    if (x < 0.0f) r = (f2u(y) >> 31) ? r - PI : r + PI;
    return r;
}

// ---------------------------------------------------------------- vectors
struct V2 { float x, y; };
struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };

VPT_HD V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
VPT_HD V3 v3s(float s) { return v3(s, s, s); }
VPT_HD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
VPT_HD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
VPT_HD V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
VPT_HD V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
VPT_HD V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
VPT_HD V3 operator*(float s, V3 a) { return v3(a.x * s, a.y * s, a.z * s); }
VPT_HD V3 operator/(V3 a, V3 b) { return v3(a.x / b.x, a.y / b.y, a.z / b.z); }
// vector / scalar: one IEEE reciprocal, three multiplies (how the shader compilers lower a splat divide)
VPT_HD V3 operator/(V3 a, float s) { float r = 1.0f / s; return v3(a.x * r, a.y * r, a.z * r); }
VPT_HD float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
VPT_HD V3 cross(V3 a, V3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
VPT_HD float length(V3 a) { return sqrt_(dot(a, a)); }
// normalize(v) = v * (1/|v|): one IEEE divide, three multiplies.
VPT_HD V3 normalize(V3 a) { float inv = 1.0f / sqrt_(dot(a, a)); return a * inv; }
VPT_HD V3 reflect(V3 i, V3 n) { float d2 = 2.0f * dot(n, i); return i - n * d2; }
VPT_HD V3 refract(V3 i, V3 n, float eta) {
    float ni = dot(n, i);
    float k = 1.0f - eta * eta * (1.0f - ni * ni);
    if (k < 0.0f) return v3(0.0f, 0.0f, 0.0f);
    float c = eta * ni + sqrt_(k);
    return i * eta - n * c;
}
VPT_HD float lerp(float a, float b, float t) { return a + (b - a) * t; }
VPT_HD V3 lerp(V3 a, V3 b, float t) { return v3(lerp(a.x, b.x, t), lerp(a.y, b.y, t), lerp(a.z, b.z, t)); }
VPT_HD float smoothstep(float e0, float e1, float x) {
    float t = saturate_((x - e0) / (e1 - e0));
    return t * t * (3.0f - 2.0f * t);
}
VPT_HD float max3(V3 a) { return max_(a.x, max_(a.y, a.z)); }

// Rodrigues rotation, reference RTCommon.slang:37-45 (axis is already unit at
// every call site, but the reference normalises it; so do we).
VPT_HD V3 rotate(V3 v, V3 axis, float theta) {
    float s, c;
    sincos_(theta, &s, &c);
    V3 a = normalize(axis);
    return (v * c) + (cross(a, v) * s) + (a * dot(a, v)) * (1.0f - c);
}

// The same rotation with sincos_(theta) supplied by the caller (an angle that is constant for a whole render is evaluated once).
VPT_HD V3 rotate_sc(V3 v, V3 axis, float s, float c) {
    V3 a = normalize(axis);
    return (v * c) + (cross(a, v) * s) + (a * dot(a, v)) * (1.0f - c);
}

// ---------------------------------------------------------------- matrices
// float[16] column-major exactly as glm stores a mat4: m[col*4+row].
VPT_HD V3 mat_point(const float* m, V3 p) {  // mul(M, float4(p,1)).xyz
    return v3(((m[0] * p.x + m[4] * p.y) + m[8] * p.z) + m[12],
              ((m[1] * p.x + m[5] * p.y) + m[9] * p.z) + m[13],
              ((m[2] * p.x + m[6] * p.y) + m[10] * p.z) + m[14]);
}
VPT_HD V3 mat_vector(const float* m, V3 p) {  // mul(M, float4(p,0)).xyz
    return v3((m[0] * p.x + m[4] * p.y) + m[8] * p.z,
              (m[1] * p.x + m[5] * p.y) + m[9] * p.z,
              (m[2] * p.x + m[6] * p.y) + m[10] * p.z);
}
VPT_HD V4 mat_v4(const float* m, V4 p) {
    V4 r;
    r.x = ((m[0] * p.x + m[4] * p.y) + m[8] * p.z) + m[12] * p.w;
    r.y = ((m[1] * p.x + m[5] * p.y) + m[9] * p.z) + m[13] * p.w;
    r.z = ((m[2] * p.x + m[6] * p.y) + m[10] * p.z) + m[14] * p.w;
    r.w = ((m[3] * p.x + m[7] * p.y) + m[11] * p.z) + m[15] * p.w;
    return r;
}
// mul(n, WorldToObject()).xyz : row vector times the 3x3 of the inverse (Surface.slang:49,60).
// inv3 is the upper-left 3x3 of the inverse instance matrix, row-major inv3[row*3+col].
VPT_HD V3 rowvec_mat3(V3 n, const float* inv3) {
    return v3((n.x * inv3[0] + n.y * inv3[3]) + n.z * inv3[6],
              (n.x * inv3[1] + n.y * inv3[4]) + n.z * inv3[7],
              (n.x * inv3[2] + n.y * inv3[5]) + n.z * inv3[8]);
}
// Host-side: inverse of the upper-left 3x3 of a column-major mat4, evaluated in
// double and rounded once (the driver's WorldToObject is unpinned).
static inline void inverse3x3_from_mat4(const float* m, float* inv3) {
    double a = m[0], b = m[4], c = m[8];
    double d = m[1], e = m[5], f = m[9];
    double g = m[2], h = m[6], i = m[10];
    double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    double det = a * A + b * B + c * C;
    double id = 1.0 / det;
    inv3[0] = (float)(A * id);
    inv3[1] = (float)(-(b * i - c * h) * id);
    inv3[2] = (float)((b * f - c * e) * id);
    inv3[3] = (float)(B * id);
    inv3[4] = (float)((a * i - c * g) * id);
    inv3[5] = (float)(-(a * f - c * d) * id);
    inv3[6] = (float)(C * id);
    inv3[7] = (float)(-(a * h - b * g) * id);
    inv3[8] = (float)((a * e - b * d) * id);
}

// ---------------------------------------------------------------- sampler arithmetic
// Vulkan linear filtering at mip 0 (texel centres at +0.5): unnormalised coordinate
// x = u*size - 0.5, i0 = floor(x), weight = frac(x); REPEAT or CLAMP_TO_EDGE addressing.
// The reference leaves this to the texture unit (PathTracer.cpp:84-94); NaN/huge
// coordinates are mapped to 0 so both sides index identically.
VPT_HD float rint_(float x) { return __builtin_rintf(x); }  // RNE, exact
VPT_HD void texel_coords(float u, int size, bool repeat, int* i0, int* i1, float* w) {
    float x = u * (float)size - 0.5f;
    if (!(fabs_(x) < 1.0e9f)) x = 0.0f;
    float fl = floor_(x);
    *w = x - fl;
    int a = (int)fl;
    int b = a + 1;
    if (repeat) {
        if ((size & (size - 1)) == 0) {  // power of two: the two's-complement mask IS the non-negative remainder (no integer division)
            a &= size - 1; b &= size - 1;
        } else {
            a %= size; if (a < 0) a += size;
            b %= size; if (b < 0) b += size;
        }
    } else {
        a = a < 0 ? 0 : (a > size - 1 ? size - 1 : a);
        b = b < 0 ? 0 : (b > size - 1 ? size - 1 : b);
    }
    *i0 = a;
    *i1 = b;
}
// 2D-array layer selection: RNE(clamp(layer, 0, layers-1)).
VPT_HD int lut_layer(float layer, int layers) {
    float l = rint_(clamp_(layer, 0.0f, (float)(layers - 1)));
    if (!(l >= 0.0f)) l = 0.0f;
    return (int)l;
}
// UNORM8 texel -> float: b / 255, correctly rounded, without a division sequence (16 of them per bilinear RGBA tap):
// q = b * RN(1/255) is within an ulp, and one residual step q + (b - 255 q) * RN(1/255) lands on RN(b / 255) for every
// b in 0..255 (checked exhaustively against the division, tests/test_fp32_contract.py).
VPT_HD float unorm8_to_float(uint32_t b) {
    const float x = (float)b, r = 0.0039215688593685627f;
    const float q = x * r;
    return fma(fma(-255.0f, q, x), r, q);
}
// float -> UNORM8 store: RNE(saturate(c)*255); NaN -> 0.
VPT_HD uint8_t unorm8(float c) {
    float s = saturate_(c);
    if (!(s >= 0.0f)) s = 0.0f;
    return (uint8_t)(int)rint_(s * 255.0f);
}

// ---------------------------------------------------------------- ray / triangle
// World-space triangle as both sides intersect it: v0 and the two edges
// (e1 = v1 - v0, e2 = v2 - v0 computed in fp32 from mat_point()-transformed
// vertices). Moller-Trumbore, two-sided, no culling (FORCE_OPAQUE, no cull
// flags: RayGen.slang:90, RTCommon.slang:54). Returns true and (t,u,v) if the
// Sliver triangles (edges parallel to within 1e-5 rad, or a zero edge) are not part of the scene's geometry for
// intersection purposes, like the degenerate primitives a hardware BVH builder drops: their determinant is a rounding
// residue for EVERY ray, so u and v land in range at arbitrary t, far from the triangle — results would depend on the
// order an acceleration structure happens to visit its leaves in.  Both BVH builders and the brute-force loop skip them.
VPT_HD bool triangle_degenerate(V3 e1, V3 e2) {
    V3 n = cross(e1, e2);
    return !(dot(n, n) > 1.0e-10f * (dot(e1, e1) * dot(e2, e2)));
}

// ray hits with tmin < t < tmax; u,v are the barycentrics of v1,v2
// (ClosestHit.slang:45). Ties between triangles are resolved by the caller
// (smaller global triangle id wins) so the result is traversal-order free.
VPT_HD bool ray_triangle(V3 o, V3 d, V3 v0, V3 e1, V3 e2, float tmin, float tmax, float* t_out, float* u_out,
                         float* v_out) {
    V3 p = cross(d, e2);
    float det = dot(e1, p);
    if (det == 0.0f) return false;
    float inv = 1.0f / det;
    V3 s = o - v0;
    float u = dot(s, p) * inv;
    if (!(u >= 0.0f && u <= 1.0f)) return false;
    V3 q = cross(s, e1);
    float v = dot(d, q) * inv;
    if (!(v >= 0.0f && u + v <= 1.0f)) return false;
    float t = dot(e2, q) * inv;
    if (!(t > tmin && t < tmax)) return false;
    *t_out = t;
    *u_out = u;
    *v_out = v;
    return true;
}

// Locality of a hit.  For a ray numerically inside a triangle's plane `det` above is a rounding residue with a large
// relative error, and fp32 can put an accepted hit visibly in front of or behind the triangle (measured: 0.0095 at
// t = 12 for a ray grazing a 0.02-wide triangle at 2e-4 rad).  A box hierarchy only looks at a triangle where the ray is
// inside its boxes, brute force looks everywhere — so such a hit would exist or not depending on the acceleration
// structure.  Rule: a candidate accepted by ray_triangle() counts only if t lies where the ray is inside the triangle's
// own bounding box (padded by half of what the BVH builders add).  That depends on the ray and the triangle alone.
// The oracle applies it to every candidate; the HIP traversal validates the winning candidate after the search and
// searches again without it in the (about 1 per 1e9 rays) case that it fails.
VPT_HD bool hit_is_local(V3 o, V3 d, V3 v0, V3 e1, V3 e2, float t) {
    const float ox[3] = {o.x, o.y, o.z}, dx[3] = {d.x, d.y, d.z};
    const float ax[3] = {v0.x, v0.y, v0.z}, bx[3] = {v0.x + e1.x, v0.y + e1.y, v0.z + e1.z}, cx[3] = {v0.x + e2.x, v0.y + e2.y, v0.z + e2.z};
    float tn = -3.0e38f, tf = 3.0e38f;
    for (int k = 0; k < 3; k++) {
        float lo = min_(ax[k], min_(bx[k], cx[k])), hi = max_(ax[k], max_(bx[k], cx[k]));
        const float pad = 1.0e-5f * max_(fabs_(lo), fabs_(hi)) + 1.0e-6f;
        lo -= pad; hi += pad;
        if (dx[k] != 0.0f) {
            const float id = 1.0f / dx[k];
            const float t0 = (lo - ox[k]) * id, t1 = (hi - ox[k]) * id;
            tn = max_(tn, min_(t0, t1)); tf = min_(tf, max_(t0, t1));
        } else if (ox[k] < lo || ox[k] > hi) {
            return false;
        }
    }
    return t * 1.000002f >= tn && t * 0.999998f <= tf;
}

}  // namespace vptfp
