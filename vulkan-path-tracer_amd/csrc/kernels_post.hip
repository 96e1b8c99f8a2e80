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

// BloomDownSample.slang:32-45 for one texel: what k_bloom_threshold stores, as a value (the fused schedule never stores mip 0).
__device__ __forceinline__ float4 soft_threshold(float4 p, float threshold, float falloff) {
    V3 c = v3(p.x, p.y, p.z);
    const float br = dot(c, v3(0.2126f, 0.7152f, 0.0722f));
    const float f = smoothstep(threshold - falloff, threshold + falloff, br);
    c = c * f;
    return make_float4(c.x, c.y, c.z, 1.0f);
}

// BloomDownSample.slang:46-63: 16 taps around 2*xy, divided by 25 (pow(range*2+1, 2)), times strength.
// FIRST: the input is the HDR image itself and the soft threshold (the reference's FirstDispatch pass) is applied while the
// tile is staged — each texel once per tile — so the thresholded full-resolution mip is never written or read back.
template <bool FIRST>
__global__ __launch_bounds__(256) void k_bloom_down(const float4* in, int iw, int ih, float4* out, int ow, int oh, float strength, float threshold, float falloff) {
    __shared__ float4 tile[kDownH][kDownW];
    const int x0 = blockIdx.x * kTileW, y0 = blockIdx.y * kTileH;
    const int sx0 = 2 * x0 - 2, sy0 = 2 * y0 - 2;
    for (int i = threadIdx.x; i < kDownW * kDownH; i += 256) {
        int ty = i / kDownW, tx = i - ty * kDownW;
        const float4 p = in[(size_t)iclamp(sy0 + ty, 0, ih - 1) * iw + iclamp(sx0 + tx, 0, iw - 1)];
        tile[ty][tx] = FIRST ? soft_threshold(p, threshold, falloff) : p;
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

// ---- fused schedule (vpt_api.hip vpt_postprocess): the same values through fewer passes over memory.
//   first down-sample reads the HDR image and thresholds on the fly          (k_bloom_down<true>: mip 0 is never stored)
//   mips small enough for one CU's LDS go down AND up inside one launch       (k_bloom_tail: 10 launches of ~8 us become 1)
//   the last up-sample, the threshold of the texel it is added to and the tonemap are one kernel (k_post_final)
// Every value is produced by the expressions of the unfused kernels above in the same order, so the RGBA8 output and the
// optional bloom mip 0 stay bit-identical (tests/test_gpu_post.py runs both schedules).

constexpr int kTailThreads = 1024, kTailMaxLevels = 8, kTailMaxTexels = 2048 + 512 + 128 + 32 + 8 + 2 + 1 + 1;
struct BloomTail {
    float4* base;              // the mip above the tail (global memory): read by the first down-sample, updated by the last up-sample
    int bw, bh;
    int levels;                // mips held in LDS
    int w[kTailMaxLevels], h[kTailMaxLevels], off[kTailMaxLevels];   // size and first texel of each in the LDS array
    float strength;
};
__device__ __forceinline__ V3 down_taps(const float4* src, int iw, int ih, int x, int y) {   // BloomDownSample.slang:46-63, tap order kept
    V3 c = v3s(0.0f);
    for (int a = -2; a < 2; a++)
        for (int b = -2; b < 2; b++) {
            const float4 p = src[(size_t)iclamp(2 * y + b, 0, ih - 1) * iw + iclamp(2 * x + a, 0, iw - 1)];
            c = c + v3(p.x, p.y, p.z);
        }
    return c;
}
__device__ __forceinline__ V3 up_taps(const float4* src, int iw, int ih, int x, int y) {     // BloomUpSample.slang:30-48
    V3 c = v3s(0.0f);
    for (int a = -2; a < 2; a++)
        for (int b = -2; b < 2; b++) {
            const float4 p = src[(size_t)iclamp(y / 2 + b + 1, 0, ih - 1) * iw + iclamp(x / 2 + a + 1, 0, iw - 1)];
            c = c + v3(p.x, p.y, p.z);
        }
    return c;
}
__global__ __launch_bounds__(kTailThreads) void k_bloom_tail(BloomTail t) {
    __shared__ float4 m[kTailMaxTexels];
    for (int k = 0; k < t.levels; k++) {   // down: base -> level 0 -> level 1 ...
        const float4* src = k == 0 ? t.base : m + t.off[k - 1];
        const int iw = k == 0 ? t.bw : t.w[k - 1], ih = k == 0 ? t.bh : t.h[k - 1];
        for (int i = threadIdx.x; i < t.w[k] * t.h[k]; i += kTailThreads) {
            const int y = i / t.w[k], x = i - y * t.w[k];
            V3 c = down_taps(src, iw, ih, x, y);
            c = c / 25.0f;
            c = c * t.strength;
            m[t.off[k] + i] = make_float4(c.x, c.y, c.z, 1.0f);
        }
        __syncthreads();
    }
    for (int k = t.levels - 1; k >= 0; k--) {   // up: level k is added into level k - 1 (into the base for k == 0)
        float4* dst = k == 0 ? t.base : m + t.off[k - 1];
        const int ow = k == 0 ? t.bw : t.w[k - 1], oh = k == 0 ? t.bh : t.h[k - 1];
        // the 16 taps of an output texel depend on (x / 2, y / 2) only: one evaluation serves the 2 x 2 texels that share them
        const int qw = (ow + 1) / 2, qh = (oh + 1) / 2;
        for (int i = threadIdx.x; i < qw * qh; i += kTailThreads) {
            const int qy = i / qw, qx = i - qy * qw;
            V3 c = up_taps(m + t.off[k], t.w[k], t.h[k], 2 * qx, 2 * qy);
            c = c / 25.0f;
            c = c * t.strength;
            for (int dy = 0; dy < 2; dy++)
                for (int dx = 0; dx < 2; dx++) {
                    const int x = 2 * qx + dx, y = 2 * qy + dy;
                    if (x >= ow || y >= oh) continue;
                    const float4 cur = dst[(size_t)y * ow + x];
                    const V3 r = c + v3(cur.x, cur.y, cur.z);
                    dst[(size_t)y * ow + x] = make_float4(r.x, r.y, r.z, 1.0f);
                }
        }
        __syncthreads();
    }
}

// The last up-sample (mip 1 -> mip 0), the soft threshold of the mip-0 texel it lands on, and Tonemap.slang:159-176, for a
// 64 x 4 tile of pixels.  The tonemap taps bloom mip 0 at (x, y) or one texel up / left of it (uv = xy / size carries no half
// texel), so the block evaluates mip 0 on the 65 x 5 texels [x0 - 1, x0 + 63] x [y0 - 1, y0 + 3] into LDS:
//   bloom0[t] = soft_threshold(hdr[t]) + (16 taps of mip 1 around t / 2 + 1) / 25 * strength      (UP = false: no mip 1, first term only)
// from an LDS copy of the HDR tile and of the 36 x 6 mip-1 texels those taps can touch.
template <bool UP, bool LINEAR>
__global__ __launch_bounds__(256) void k_post_final(const float4* hdr, const float4* mip1, int mw, int mh, float4* bloom0_out, uchar4* out, int w, int h,
                                                    float threshold, float falloff, float strength, float exposure, float gamma) {
    constexpr int BW = kTileW + 1, BH = kTileH + 1;           // 65 x 5 texels of mip 0
    constexpr int MW = kTileW / 2 + 4, MH = kTileH / 2 + 4;   // 36 x 6 texels of mip 1
    __shared__ float4 s_hdr[BH][BW];
    __shared__ float4 s_bloom[BH][BW];
    __shared__ float4 s_m1[UP ? MH : 1][UP ? MW : 1];
    const int x0 = blockIdx.x * kTileW, y0 = blockIdx.y * kTileH;
    for (int i = threadIdx.x; i < BW * BH; i += 256) {
        const int ty = i / BW, tx = i - ty * BW;
        s_hdr[ty][tx] = hdr[(size_t)iclamp(y0 - 1 + ty, 0, h - 1) * w + iclamp(x0 - 1 + tx, 0, w - 1)];
    }
    const int mx0 = x0 / 2 - 2, my0 = y0 / 2 - 2;             // first mip-1 texel of the staged tile (x0, y0 are even)
    if (UP) {
        for (int i = threadIdx.x; i < MW * MH; i += 256) {
            const int ty = i / MW, tx = i - ty * MW;
            s_m1[ty][tx] = mip1[(size_t)iclamp(my0 + ty, 0, mh - 1) * mw + iclamp(mx0 + tx, 0, mw - 1)];
        }
    }
    __syncthreads();
    // the blurred mip-1 value a mip-0 texel receives depends on (gx / 2, gy / 2) only: evaluate it once per such position —
    // 33 x 3 of them under this tile ((x0 - 1) / 2 ... (x0 + 63) / 2) — instead of once per texel
    constexpr int UW = kTileW / 2 + 1, UH = kTileH / 2 + 1;
    __shared__ float4 s_up[UP ? UH : 1][UP ? UW : 1];
    const int ux0 = x0 / 2 - 1, uy0 = y0 / 2 - 1;             // first position (may be -1 at the image's edge: never used there)
    if (UP) {
        for (int i = threadIdx.x; i < UW * UH; i += 256) {
            const int ty = i / UW, tx = i - ty * UW;
            const int px = ux0 + tx, py = uy0 + ty;           // = gx / 2, gy / 2 of the texels it serves
            V3 u = v3s(0.0f);
            for (int a = -2; a < 2; a++)
                for (int b = -2; b < 2; b++) {
                    // the tile entry at the UNclamped position holds the clamped texel (the loader clamped while fetching): the
                    // reference's clamp(xy / 2 + (a, b) + 1)
                    const float4 p = s_m1[py + b + 1 - my0][px + a + 1 - mx0];
                    u = u + v3(p.x, p.y, p.z);
                }
            u = u / 25.0f;
            u = u * strength;
            s_up[ty][tx] = make_float4(u.x, u.y, u.z, 0.0f);
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < BW * BH; i += 256) {
        const int ty = i / BW, tx = i - ty * BW;
        const int gx = iclamp(x0 - 1 + tx, 0, w - 1), gy = iclamp(y0 - 1 + ty, 0, h - 1);   // the mip-0 texel this entry stands for
        const float4 th = soft_threshold(s_hdr[ty][tx], threshold, falloff);
        V3 c = v3(th.x, th.y, th.z);
        if (UP) {
            const float4 u = s_up[gy / 2 - uy0][gx / 2 - ux0];
            c = v3(u.x, u.y, u.z) + c;   // k_bloom_up: blurred + what mip 0 held
        }
        s_bloom[ty][tx] = make_float4(c.x, c.y, c.z, 1.0f);
    }
    __syncthreads();
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const int x = x0 + lx, y = y0 + ly;
    if (x >= w || y >= h) return;
    if (bloom0_out) bloom0_out[(size_t)y * w + x] = s_bloom[ly + 1][lx + 1];
    const float4 p = s_hdr[ly + 1][lx + 1];
    V3 c = v3(p.x, p.y, p.z);
    const float u = (float)x / (float)w, v = (float)y / (float)h;
    // mip-0 texel (tx, ty) sits at s_bloom[ty - (y0 - 1)][tx - (x0 - 1)]; the taps are x - 1 .. x, y - 1 .. y, clamped to the image
    auto B = [&](int tx, int ty) { const float4 q = s_bloom[ty - (y0 - 1)][tx - (x0 - 1)]; return v3(q.x, q.y, q.z); };
    V3 bl;
    if (LINEAR) {
        int xa, xb, ya, yb; float fx, fy;
        texel_coords(u, w, false, &xa, &xb, &fx);
        texel_coords(v, h, false, &ya, &yb, &fy);
        const V3 a = lerp(B(xa, ya), B(xb, ya), fx);
        const V3 b = lerp(B(xa, yb), B(xb, yb), fx);
        bl = lerp(a, b, fy);
    } else {
        const int tx = iclamp((int)floor_(u * (float)w), 0, w - 1), ty = iclamp((int)floor_(v * (float)h), 0, h - 1);
        bl = B(tx, ty);
    }
    c = c + bl;
    c = c * exposure;
    const float ig = 1.0f / gamma;
    c = v3(pow_(c.x, ig), pow_(c.y, ig), pow_(c.z, ig));
    c = aces_fitted(c);
    out[(size_t)y * w + x] = make_uchar4(unorm8(c.x), unorm8(c.y), unorm8(c.z), 255);
}

void launch_bloom_threshold(hipStream_t s, const float* in, float* out, uint32_t w, uint32_t h, float threshold, float falloff) {
    uint32_t n = w * h;
    hipLaunchKernelGGL(k_bloom_threshold, dim3(cdiv_(n, 256)), dim3(256), 0, s, reinterpret_cast<const float4*>(in), reinterpret_cast<float4*>(out), n, threshold, falloff);
}
void launch_bloom_down(hipStream_t s, const float* in, uint32_t iw, uint32_t ih, float* out, uint32_t ow, uint32_t oh, float strength) {
    hipLaunchKernelGGL(k_bloom_down<false>, dim3(cdiv_(ow, 64), cdiv_(oh, 4)), dim3(256), 0, s, reinterpret_cast<const float4*>(in), (int)iw, (int)ih,
                       reinterpret_cast<float4*>(out), (int)ow, (int)oh, strength, 0.0f, 0.0f);
}
void launch_bloom_down_first(hipStream_t s, const float* hdr, uint32_t iw, uint32_t ih, float* out, uint32_t ow, uint32_t oh, float strength, float threshold, float falloff) {
    hipLaunchKernelGGL(k_bloom_down<true>, dim3(cdiv_(ow, 64), cdiv_(oh, 4)), dim3(256), 0, s, reinterpret_cast<const float4*>(hdr), (int)iw, (int)ih,
                       reinterpret_cast<float4*>(out), (int)ow, (int)oh, strength, threshold, falloff);
}
void launch_bloom_tail(hipStream_t s, float* base, uint32_t bw, uint32_t bh, const uint32_t* w, const uint32_t* h, uint32_t levels, float strength) {
    BloomTail t{};
    t.base = reinterpret_cast<float4*>(base); t.bw = (int)bw; t.bh = (int)bh; t.levels = (int)levels; t.strength = strength;
    int off = 0;
    for (uint32_t k = 0; k < levels; k++) { t.w[k] = (int)w[k]; t.h[k] = (int)h[k]; t.off[k] = off; off += (int)(w[k] * h[k]); }
    hipLaunchKernelGGL(k_bloom_tail, dim3(1), dim3(kTailThreads), 0, s, t);
}
void launch_post_final(hipStream_t s, const float* hdr, const float* mip1, uint32_t mw, uint32_t mh, float* bloom0_out, uint8_t* out, uint32_t w, uint32_t h,
                       float threshold, float falloff, float strength, float exposure, float gamma, bool linear_tap) {
    const dim3 g(cdiv_(w, 64), cdiv_(h, 4)), b(256);
#define VPT_PF(UP, LIN) hipLaunchKernelGGL((k_post_final<UP, LIN>), g, b, 0, s, reinterpret_cast<const float4*>(hdr), reinterpret_cast<const float4*>(mip1), (int)mw, (int)mh, \
                                           reinterpret_cast<float4*>(bloom0_out), reinterpret_cast<uchar4*>(out), (int)w, (int)h, threshold, falloff, strength, exposure, gamma)
    if (mip1) { if (linear_tap) VPT_PF(true, true); else VPT_PF(true, false); }
    else { if (linear_tap) VPT_PF(false, true); else VPT_PF(false, false); }
#undef VPT_PF
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
