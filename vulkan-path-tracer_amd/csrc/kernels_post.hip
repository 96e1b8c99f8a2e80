// kernels_post.hip — bloom chain + ACES tonemap (reference Shaders/PostProcess/*.slang,
// schedule PostProcessor.cpp:193-246).  HBM-bound streaming kernels: one float4 (16 B) per lane per
// access, rows contiguous across the wave; the two 4x4 box blurs stage their input tile in LDS.  Tap
// summation order is the reference's (x offset outer, y offset inner) so results are bit-identical to
// the scalar restatement.
#include "kernels.hpp"

namespace vpt {
using namespace vptfp;

static inline uint32_t cdiv_(uint32_t a, uint32_t b) { return (a + b - 1) / b; }
__device__ inline int iclamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// BloomDownSample.slang:32-45 (FirstDispatch): soft threshold.
__global__ __launch_bounds__(256) void k_bloom_threshold(const float4* in, float4* out, uint32_t n, float threshold, float falloff) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 p = in[i];
    V3 c = v3(p.x, p.y, p.z);
    float br = dot(c, v3(0.2126f, 0.7152f, 0.0722f));
    float f = smoothstep(threshold - falloff, threshold + falloff, br);
    c = c * f;
    out[i] = make_float4(c.x, c.y, c.z, 1.0f);
}

// Tile geometry shared by the two blur kernels: a block of 256 threads produces 64 x 4 output texels from an
// input tile staged in LDS (coalesced float4 rows, edge texels clamped while loading, which is exactly the
// reference's clamp(samplePos)), then every thread sums its 16 taps from LDS in the reference's order
// (x offset outer, y offset inner) so the result stays bit-identical to the scalar restatement.
constexpr int kTileW = 64, kTileH = 4;
constexpr int kDownW = 2 * kTileW + 2, kDownH = 2 * kTileH + 2;   // 130 x 10 input texels per down-sample tile
constexpr int kUpW = kTileW / 2 + 3, kUpH = kTileH / 2 + 3;       // 35 x 5 input texels per up-sample tile

// BloomDownSample.slang:46-63: 16 taps around 2*xy, divided by 25 (pow(range*2+1, 2)), times strength.
__global__ __launch_bounds__(256) void k_bloom_down(const float4* in, int iw, int ih, float4* out, int ow, int oh, float strength) {
    __shared__ float4 tile[kDownH][kDownW];
    const int x0 = blockIdx.x * kTileW, y0 = blockIdx.y * kTileH;
    const int sx0 = 2 * x0 - 2, sy0 = 2 * y0 - 2;
    for (int i = threadIdx.x; i < kDownW * kDownH; i += 256) {
        int ty = i / kDownW, tx = i - ty * kDownW;
        tile[ty][tx] = in[(size_t)iclamp(sy0 + ty, 0, ih - 1) * iw + iclamp(sx0 + tx, 0, iw - 1)];
    }
    __syncthreads();
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const int x = x0 + lx, y = y0 + ly;
    if (x >= ow || y >= oh) return;
    V3 c = v3s(0.0f);
    for (int a = -2; a < 2; a++)
        for (int b = -2; b < 2; b++) {
            float4 p = tile[2 * ly + b + 2][2 * lx + a + 2];
            c = c + v3(p.x, p.y, p.z);
        }
    c = c / 25.0f;
    c = c * strength;
    out[(size_t)y * ow + x] = make_float4(c.x, c.y, c.z, 1.0f);
}

// BloomUpSample.slang:30-48: 16 taps around xy/2 + 1 of the coarser mip, /25, *strength, added to the finer mip.
__global__ __launch_bounds__(256) void k_bloom_up(const float4* in, int iw, int ih, float4* out, int ow, int oh, float strength) {
    __shared__ float4 tile[kUpH][kUpW];
    const int x0 = blockIdx.x * kTileW, y0 = blockIdx.y * kTileH;
    const int sx0 = x0 / 2 - 1, sy0 = y0 / 2 - 1;
    for (int i = threadIdx.x; i < kUpW * kUpH; i += 256) {
        int ty = i / kUpW, tx = i - ty * kUpW;
        tile[ty][tx] = in[(size_t)iclamp(sy0 + ty, 0, ih - 1) * iw + iclamp(sx0 + tx, 0, iw - 1)];
    }
    __syncthreads();
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const int x = x0 + lx, y = y0 + ly;
    if (x >= ow || y >= oh) return;
    V3 c = v3s(0.0f);
    for (int a = -2; a < 2; a++)
        for (int b = -2; b < 2; b++) {
            float4 p = tile[y / 2 + b + 1 - sy0][x / 2 + a + 1 - sx0];
            c = c + v3(p.x, p.y, p.z);
        }
    c = c / 25.0f;
    c = c * strength;
    float4 cur = out[(size_t)y * ow + x];
    c = c + v3(cur.x, cur.y, cur.z);
    out[(size_t)y * ow + x] = make_float4(c.x, c.y, c.z, 1.0f);
}

__device__ inline V3 aces_fitted(V3 c) {  // Tonemap.slang:20-55
    V3 a = v3((0.59719f * c.x + 0.35458f * c.y) + 0.04823f * c.z, (0.07600f * c.x + 0.90834f * c.y) + 0.01566f * c.z,
              (0.02840f * c.x + 0.13383f * c.y) + 0.83777f * c.z);
    V3 n = a * (a + v3s(0.0245786f)) - v3s(0.000090537f);
    V3 d = a * (0.983729f * a + v3s(0.4329510f)) + v3s(0.238081f);
    V3 r = n / d;
    V3 q = v3((1.60475f * r.x + -0.53108f * r.y) + -0.07367f * r.z, (-0.10208f * r.x + 1.10813f * r.y) + -0.00605f * r.z,
              (-0.00327f * r.x + -0.07276f * r.y) + 1.07602f * r.z);
    return v3(saturate_(q.x), saturate_(q.y), saturate_(q.z));
}

// Tonemap.slang:159-176: hdr + bloom tap at uv = xy/size (no half-texel offset), exposure, gamma, ACES, RGBA8.
__global__ __launch_bounds__(256) void k_tonemap(const float4* hdr, const float4* bloom, uchar4* out, int w, int h, float exposure,
                                                 float gamma, int linear_tap) {
    int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    float4 p = hdr[(size_t)y * w + x];
    V3 c = v3(p.x, p.y, p.z);
    float u = (float)x / (float)w, v = (float)y / (float)h;
    V3 bl;
    if (linear_tap) {
        int x0, x1, y0, y1; float fx, fy;
        texel_coords(u, w, false, &x0, &x1, &fx);
        texel_coords(v, h, false, &y0, &y1, &fy);
        float4 p00 = bloom[(size_t)y0 * w + x0], p10 = bloom[(size_t)y0 * w + x1], p01 = bloom[(size_t)y1 * w + x0], p11 = bloom[(size_t)y1 * w + x1];
        V3 a = lerp(v3(p00.x, p00.y, p00.z), v3(p10.x, p10.y, p10.z), fx);
        V3 b = lerp(v3(p01.x, p01.y, p01.z), v3(p11.x, p11.y, p11.z), fx);
        bl = lerp(a, b, fy);
    } else {
        int tx = iclamp((int)floor_(u * (float)w), 0, w - 1), ty = iclamp((int)floor_(v * (float)h), 0, h - 1);
        float4 q = bloom[(size_t)ty * w + tx];
        bl = v3(q.x, q.y, q.z);
    }
    c = c + bl;
    c = c * exposure;
    float ig = 1.0f / gamma;
    c = v3(pow_(c.x, ig), pow_(c.y, ig), pow_(c.z, ig));
    c = aces_fitted(c);
    out[(size_t)y * w + x] = make_uchar4(unorm8(c.x), unorm8(c.y), unorm8(c.z), 255);
}

void launch_bloom_threshold(hipStream_t s, const float* in, float* out, uint32_t w, uint32_t h, float threshold, float falloff) {
    uint32_t n = w * h;
    hipLaunchKernelGGL(k_bloom_threshold, dim3(cdiv_(n, 256)), dim3(256), 0, s, reinterpret_cast<const float4*>(in), reinterpret_cast<float4*>(out), n, threshold, falloff);
}
void launch_bloom_down(hipStream_t s, const float* in, uint32_t iw, uint32_t ih, float* out, uint32_t ow, uint32_t oh, float strength) {
    hipLaunchKernelGGL(k_bloom_down, dim3(cdiv_(ow, 64), cdiv_(oh, 4)), dim3(256), 0, s, reinterpret_cast<const float4*>(in), (int)iw, (int)ih,
                       reinterpret_cast<float4*>(out), (int)ow, (int)oh, strength);
}
void launch_bloom_up(hipStream_t s, const float* in, uint32_t iw, uint32_t ih, float* out, uint32_t ow, uint32_t oh, float strength) {
    hipLaunchKernelGGL(k_bloom_up, dim3(cdiv_(ow, 64), cdiv_(oh, 4)), dim3(256), 0, s, reinterpret_cast<const float4*>(in), (int)iw, (int)ih,
                       reinterpret_cast<float4*>(out), (int)ow, (int)oh, strength);
}
void launch_tonemap(hipStream_t s, const float* hdr, const float* bloom0, uint8_t* out, uint32_t w, uint32_t h, float exposure, float gamma,
                    bool linear_tap) {
    hipLaunchKernelGGL(k_tonemap, dim3(cdiv_(w, 64), cdiv_(h, 4)), dim3(256), 0, s, reinterpret_cast<const float4*>(hdr),
                       reinterpret_cast<const float4*>(bloom0), reinterpret_cast<uchar4*>(out), (int)w, (int)h, exposure, gamma, linear_tap ? 1 : 0);
}

}  // namespace vpt
