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
    // rgb only (alpha is not read): 15.6 KB per block instead of 20.8, so that the 2025 blocks of the 1080p first pass are resident together
    // (10 per CU) rather than in one full round and a 13 % tail
    __shared__ float tile[kDownH][kDownW][3];
    const int x0 = blockIdx.x * kTileW, y0 = blockIdx.y * kTileH;
    const int sx0 = 2 * x0 - 2, sy0 = 2 * y0 - 2;
    for (int i = threadIdx.x; i < kDownW * kDownH; i += 256) {
        int ty = i / kDownW, tx = i - ty * kDownW;
        float4 p = in[(size_t)iclamp(sy0 + ty, 0, ih - 1) * iw + iclamp(sx0 + tx, 0, iw - 1)];
        if (FIRST) p = soft_threshold(p, threshold, falloff);
        tile[ty][tx][0] = p.x; tile[ty][tx][1] = p.y; tile[ty][tx][2] = p.z;
    }
    __syncthreads();
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const int x = x0 + lx, y = y0 + ly;
    if (x >= ow || y >= oh) return;
    V3 c = v3s(0.0f);
    for (int a = -2; a < 2; a++)
        for (int b = -2; b < 2; b++) {
            const float* p = tile[2 * ly + b + 2][2 * lx + a + 2];
            c = c + v3(p[0], p[1], p[2]);
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
    float4* base;              // the mip above the tail (global memory): read by the first down-sample; updated by the last up-sample unless top_out
    float4* top_out;           // staged variant: the tail's first level, finished, goes here instead (the up-sample INTO the base — four times
                               // as many texels, on this one CU — is left to the up chain, which runs on all of them)
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
// One block walks ~14 dependent phases (six levels down, six up, the base in and out); what a phase costs is the latency of its slowest
// instruction chain, so the staged variant keeps every operand a phase needs where it is cheapest to reach: the level loop is unrolled
// (sizes and offsets of a level are scalars, not indexed loads from the argument block) and every source is an LDS array by type
// (a pointer that may be global or LDS compiles to flat loads, which wait for both memory pipelines).
// STAGED: the base mip (<= kTailMaxBase texels) is copied into LDS once, as rgb — its 16-tap reads by the first down-sample (32 k
// 16-byte loads through one CU's L1) and its read-modify-write by the last up-sample become one coalesced read and one coalesced write.
constexpr int kTailMaxBase = 8448;
template <bool STAGED>
__global__ __launch_bounds__(kTailThreads) void k_bloom_tail(BloomTail t) {
    __shared__ float4 m[kTailMaxTexels];
    __shared__ float base3[STAGED ? kTailMaxBase * 3 : 3];
    if (STAGED) {   // all of a thread's loads in flight together (one block: nothing else hides a round trip)
        constexpr int R = (kTailMaxBase + kTailThreads - 1) / kTailThreads;
        float4 v[R];
        const int n = t.bw * t.bh;
#pragma unroll
        for (int r = 0; r < R; r++) { const int i = (int)threadIdx.x + r * kTailThreads; if (i < n) v[r] = t.base[i]; }
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int i = (int)threadIdx.x + r * kTailThreads;
            if (i < n) { base3[3 * i] = v[r].x; base3[3 * i + 1] = v[r].y; base3[3 * i + 2] = v[r].z; }
        }
        __syncthreads();
    }
    if (STAGED) {
#pragma unroll
        for (int k = 0; k < kTailMaxLevels; k++) {   // down: base -> level 0 -> level 1 ...
            if (k >= t.levels) break;
            const int iw = k == 0 ? t.bw : t.w[k > 0 ? k - 1 : 0], ih = k == 0 ? t.bh : t.h[k > 0 ? k - 1 : 0];
            const int ow = t.w[k], n = ow * t.h[k];
            const float4* src = m + t.off[k > 0 ? k - 1 : 0];
            for (int i = threadIdx.x; i < n; i += kTailThreads) {
                const int y = i / ow, x = i - y * ow;
                V3 c = v3s(0.0f);
                for (int a = -2; a < 2; a++)
                    for (int b = -2; b < 2; b++) {
                        const int si = iclamp(2 * y + b, 0, ih - 1) * iw + iclamp(2 * x + a, 0, iw - 1);
                        if (k == 0) { const float* p = base3 + 3 * si; c = c + v3(p[0], p[1], p[2]); }
                        else { const float4 p = src[si]; c = c + v3(p.x, p.y, p.z); }
                    }
                c = c / 25.0f;
                c = c * t.strength;
                m[t.off[k] + i] = make_float4(c.x, c.y, c.z, 1.0f);
            }
            __syncthreads();
        }
#pragma unroll
        for (int k = kTailMaxLevels - 1; k >= 0; k--) {   // up: level k is added into level k - 1 (into the base for k == 0)
            if (k >= t.levels || k == 0) continue;
            const int ow = t.w[k - 1], oh = t.h[k - 1];
            const int iw = t.w[k], ih = t.h[k];
            const float4* src = m + t.off[k];
            float4* dstl = m + t.off[k - 1];
            // the 16 taps of an output texel depend on (x / 2, y / 2) only: one evaluation serves the 2 x 2 texels that share them
            const int qw = (ow + 1) / 2, qh = (oh + 1) / 2;
            for (int i = threadIdx.x; i < qw * qh; i += kTailThreads) {
                const int qy = i / qw, qx = i - qy * qw;
                V3 c = v3s(0.0f);
                for (int a = -2; a < 2; a++)
                    for (int b = -2; b < 2; b++) {
                        const float4 p = src[iclamp(qy + b + 1, 0, ih - 1) * iw + iclamp(qx + a + 1, 0, iw - 1)];
                        c = c + v3(p.x, p.y, p.z);
                    }
                c = c / 25.0f;
                c = c * t.strength;
                for (int dy = 0; dy < 2; dy++)
                    for (int dx = 0; dx < 2; dx++) {
                        const int x = 2 * qx + dx, y = 2 * qy + dy;
                        if (x >= ow || y >= oh) continue;
                        const float4 q = dstl[y * ow + x];
                        const V3 r = c + v3(q.x, q.y, q.z);
                        dstl[y * ow + x] = make_float4(r.x, r.y, r.z, 1.0f);
                    }
            }
            __syncthreads();
        }
        for (int i = threadIdx.x; i < t.w[0] * t.h[0]; i += kTailThreads) t.top_out[i] = m[t.off[0] + i];
        return;
    }
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

// Several down-samples in one launch (the small middle levels, where a launch costs more than its work): out[0] = down(in), out[1] =
// down(out[0]), out[2] = down(out[1]), every level written.  A block owns an 8 x 4 tile of the last level and the texels of the levels
// below that fold into it (a level with an odd size has one more row / column than twice the next one: the last tile owns it), and it
// evaluates, level by level in LDS, the texels of each level that the next one taps — its own and a halo that neighbouring blocks
// evaluate too (1.4-1.6 x redundant on levels of a few thousand texels).  Every texel by k_bloom_down's expression in its tap order; a
// tile entry at an out-of-range position holds the clamped texel's value, so the next level indexes the tile with unclamped taps.
constexpr int kDcMax = 3, kDcTW = 8, kDcTH = 4;
constexpr int kDcW1 = 2 * kDcTW + 3, kDcH1 = 2 * kDcTH + 3, kDcW0 = 2 * kDcW1 + 3, kDcH0 = 2 * kDcH1 + 3;   // 19 x 11, 41 x 25
struct DownChain {
    int n;                       // levels produced (2 or 3)
    const float4* in; int iw, ih;
    float4* out[kDcMax]; int w[kDcMax], h[kDcMax];
    float strength;
};
__global__ __launch_bounds__(256) void k_bloom_down_chain(DownChain c) {
    __shared__ float4 tile0[kDcW0 * kDcH0];
    __shared__ float4 tile1[kDcW1 * kDcH1];
    const int top = c.n - 1;
    int olx[kDcMax], ohx[kDcMax], oly[kDcMax], ohy[kDcMax];   // owned texels of each level
    int nlx[kDcMax], nhx[kDcMax], nly[kDcMax], nhy[kDcMax];   // evaluated texels (unclamped positions)
#pragma unroll
    for (int j = kDcMax - 1; j >= 0; j--) {
        if (j > top) { olx[j] = ohx[j] = oly[j] = ohy[j] = nlx[j] = nhx[j] = nly[j] = nhy[j] = 0; continue; }
        if (j == top) {
            olx[j] = blockIdx.x * kDcTW; ohx[j] = min(olx[j] + kDcTW - 1, c.w[j] - 1);
            oly[j] = blockIdx.y * kDcTH; ohy[j] = min(oly[j] + kDcTH - 1, c.h[j] - 1);
            nlx[j] = olx[j]; nhx[j] = ohx[j]; nly[j] = oly[j]; nhy[j] = ohy[j];
        } else {
            const int u = j + 1 < kDcMax ? j + 1 : j;
            olx[j] = 2 * olx[u]; ohx[j] = ohx[u] == c.w[u] - 1 ? c.w[j] - 1 : 2 * ohx[u] + 1;
            oly[j] = 2 * oly[u]; ohy[j] = ohy[u] == c.h[u] - 1 ? c.h[j] - 1 : 2 * ohy[u] + 1;
            nlx[j] = min(olx[j], 2 * iclamp(nlx[u], 0, c.w[u] - 1) - 2); nhx[j] = max(ohx[j], 2 * iclamp(nhx[u], 0, c.w[u] - 1) + 1);
            nly[j] = min(oly[j], 2 * iclamp(nly[u], 0, c.h[u] - 1) - 2); nhy[j] = max(ohy[j], 2 * iclamp(nhy[u], 0, c.h[u] - 1) + 1);
        }
    }
#pragma unroll
    for (int j = 0; j < kDcMax; j++) {
        if (j > top) break;
        const int tw = nhx[j] - nlx[j] + 1, th = nhy[j] - nly[j] + 1;
        const int pw = j > 0 ? nhx[j - 1] - nlx[j - 1] + 1 : 0;
        const float4* prev = j == 1 ? tile0 : tile1;
        float4* mine = j == 0 ? tile0 : tile1;
        for (int i = threadIdx.x; i < tw * th; i += 256) {
            const int ty = i / tw, tx = i - ty * tw;
            const int px = nlx[j] + tx, py = nly[j] + ty;
            const int x = iclamp(px, 0, c.w[j] - 1), y = iclamp(py, 0, c.h[j] - 1);
            V3 v = v3s(0.0f);
#pragma unroll
            for (int a = -2; a < 2; a++)
#pragma unroll
                for (int b = -2; b < 2; b++) {
                    float4 p;
                    if (j == 0) p = c.in[(size_t)iclamp(2 * y + b, 0, c.ih - 1) * c.iw + iclamp(2 * x + a, 0, c.iw - 1)];
                    else p = prev[(2 * y + b - nly[j > 0 ? j - 1 : 0]) * pw + (2 * x + a - nlx[j > 0 ? j - 1 : 0])];
                    v = v + v3(p.x, p.y, p.z);
                }
            v = v / 25.0f;
            v = v * c.strength;
            const float4 r = make_float4(v.x, v.y, v.z, 1.0f);
            if (j < top) mine[i] = r;
            if (px == x && py == y && x >= olx[j] && x <= ohx[j] && y >= oly[j] && y <= ohy[j]) c.out[j][(size_t)y * c.w[j] + x] = r;
        }
        __syncthreads();
    }
}

// Several up-samples in one launch: out[j] = blur(out[j + 1]) / 25 * strength + mip[j] for j = n - 1 .. 0, where out[n] is the level above
// as it stands in memory (the tail's result).  Only out[0] is written: the levels between are consumed by the next level down alone,
// so a block evaluates the part of each that its 64 x 8 tile of level 0 depends on — 35 x 7, 21 x 7, 14 x 7 texels — in LDS, top level
// first, every texel by k_bloom_up's expression in its tap order.  A tile entry at an out-of-range position holds the clamped texel's
// value (the reference clamps tap coordinates), so the next level indexes the tile with unclamped taps.
constexpr int kChainMax = 4, kChainTileH = 8, kChainW1 = kTileW / 2 + 3, kChainRows = kChainTileH / 2 + 3;   // level 1 of the chain: 35 x 7
struct UpChain {
    int n;                          // levels computed (1 .. kChainMax)
    float4* mip[kChainMax + 1];     // mip[0]: updated in place; mip[1 .. n - 1]: the down-sampled levels (read only); mip[n]: final, read only
    int w[kChainMax + 1], h[kChainMax + 1];
    float strength;
};
__global__ __launch_bounds__(256) void k_bloom_up_chain(UpChain c) {
    __shared__ float4 tile[kChainMax - 1][kChainRows * (kChainW1 + 1)];
    int lox[kChainMax], hix[kChainMax], loy[kChainMax], hiy[kChainMax];
    lox[0] = blockIdx.x * kTileW; hix[0] = lox[0] + kTileW - 1; loy[0] = blockIdx.y * kChainTileH; hiy[0] = loy[0] + kChainTileH - 1;
#pragma unroll
    for (int j = 1; j < kChainMax; j++) {   // the positions of level j the tile of level j - 1 taps: x / 2 - 1 .. x / 2 + 2 of its (clamped) texels
        lox[j] = iclamp(lox[j - 1], 0, c.w[j - 1] - 1) / 2 - 1; hix[j] = iclamp(hix[j - 1], 0, c.w[j - 1] - 1) / 2 + 2;
        loy[j] = iclamp(loy[j - 1], 0, c.h[j - 1] - 1) / 2 - 1; hiy[j] = iclamp(hiy[j - 1], 0, c.h[j - 1] - 1) / 2 + 2;
    }
    // Everything a thread reads from memory is requested before the first level is evaluated (the levels follow each other through LDS and
    // barriers; a fetch inside each would put a memory round trip into every one of them): the texel of each level it will add its blur
    // to, and, for the top level of the chain, the 16 taps of the level above.
    // Levels 1 .. n - 1: one tile entry per thread (<= 252 entries).  Level 0: a thread owns two neighbouring texels of one row; their
    // taps depend on (x / 2, y / 2) only, which they share.
    float4 cur[kChainMax];          // levels >= 1
    int ex[kChainMax], ey[kChainMax]; bool eon[kChainMax];
#pragma unroll
    for (int j = 1; j < kChainMax; j++) {
        const int tw = hix[j] - lox[j] + 1, th = hiy[j] - loy[j] + 1;
        eon[j] = j < c.n && (int)threadIdx.x < tw * th;
        const int ty = (int)threadIdx.x / tw, tx = (int)threadIdx.x - ty * tw;
        ex[j] = iclamp(lox[j] + tx, 0, c.w[j] - 1); ey[j] = iclamp(loy[j] + ty, 0, c.h[j] - 1);
        if (eon[j]) cur[j] = c.mip[j][(size_t)ey[j] * c.w[j] + ex[j]];
    }
    const int qx = (int)(threadIdx.x & 31u), qy = (int)(threadIdx.x >> 6), dy = (int)((threadIdx.x >> 5) & 1u);
    const int x0 = lox[0] + 2 * qx, y0 = loy[0] + 2 * qy + dy;         // this thread's texels of level 0: (x0, y0), (x0 + 1, y0)
    const bool on0 = x0 < c.w[0] && y0 < c.h[0], on1 = on0 && x0 + 1 < c.w[0];
    float4 c0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), c1 = c0;
    if (on0) c0 = c.mip[0][(size_t)y0 * c.w[0] + x0];
    if (on1) c1 = c.mip[0][(size_t)y0 * c.w[0] + x0 + 1];
    float4 tap[16];
    const int jt = c.n - 1;                                           // the chain's top level: its taps come from memory
#pragma unroll
    for (int j = 0; j < kChainMax; j++) {
        if (j != jt) continue;
        const int x = j == 0 ? x0 : ex[j], y = j == 0 ? y0 : ey[j];
        if (j == 0 ? on0 : eon[j])
#pragma unroll
            for (int a = -2; a < 2; a++)
#pragma unroll
                for (int b = -2; b < 2; b++)
                    tap[(a + 2) * 4 + (b + 2)] = c.mip[j + 1][(size_t)iclamp(y / 2 + b + 1, 0, c.h[j + 1] - 1) * c.w[j + 1] + iclamp(x / 2 + a + 1, 0, c.w[j + 1] - 1)];
    }
#pragma unroll
    for (int j = kChainMax - 1; j >= 0; j--) {
        if (j >= c.n) continue;
        const bool act = j == 0 ? on0 : eon[j];
        if (act) {
            const int x = j == 0 ? x0 : ex[j], y = j == 0 ? y0 : ey[j];
            V3 u = v3s(0.0f);
            if (j == jt) {
#pragma unroll
                for (int k = 0; k < 16; k++) u = u + v3(tap[k].x, tap[k].y, tap[k].z);   // (a outer, b inner): the order they were fetched in
            } else {
                const int sj = j + 1 < kChainMax ? j + 1 : 0;         // (never read with j == kChainMax - 1: that level is always the top)
                const int sw = hix[sj] - lox[sj] + 1;
                for (int a = -2; a < 2; a++)
                    for (int b = -2; b < 2; b++) {
                        const float4 p = tile[sj - 1 >= 0 ? sj - 1 : 0][(y / 2 + b + 1 - loy[sj]) * sw + (x / 2 + a + 1 - lox[sj])];
                        u = u + v3(p.x, p.y, p.z);
                    }
            }
            u = u / 25.0f;
            u = u * c.strength;
            if (j == 0) {
                const V3 r0 = u + v3(c0.x, c0.y, c0.z);
                c.mip[0][(size_t)y0 * c.w[0] + x0] = make_float4(r0.x, r0.y, r0.z, 1.0f);
                if (on1) { const V3 r1 = u + v3(c1.x, c1.y, c1.z); c.mip[0][(size_t)y0 * c.w[0] + x0 + 1] = make_float4(r1.x, r1.y, r1.z, 1.0f); }
            } else {
                const V3 r = u + v3(cur[j].x, cur[j].y, cur[j].z);
                tile[j - 1][threadIdx.x] = make_float4(r.x, r.y, r.z, 1.0f);   // entry index = ty * tw + tx = threadIdx.x
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
bool bloom_tail_is_staged(uint32_t bw, uint32_t bh) { return (uint64_t)bw * bh <= (uint64_t)kTailMaxBase; }
void launch_bloom_tail(hipStream_t s, float* base, float* top_out, uint32_t bw, uint32_t bh, const uint32_t* w, const uint32_t* h, uint32_t levels, float strength) {
    BloomTail t{};
    t.base = reinterpret_cast<float4*>(base); t.top_out = reinterpret_cast<float4*>(top_out); t.bw = (int)bw; t.bh = (int)bh; t.levels = (int)levels; t.strength = strength;
    int off = 0;
    for (uint32_t k = 0; k < levels; k++) { t.w[k] = (int)w[k]; t.h[k] = (int)h[k]; t.off[k] = off; off += (int)(w[k] * h[k]); }
    if (bloom_tail_is_staged(bw, bh)) hipLaunchKernelGGL(k_bloom_tail<true>, dim3(1), dim3(kTailThreads), 0, s, t);
    else hipLaunchKernelGGL(k_bloom_tail<false>, dim3(1), dim3(kTailThreads), 0, s, t);
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
void launch_bloom_down_chain(hipStream_t s, const float* in, uint32_t iw, uint32_t ih, float* const* outs, const uint32_t* w, const uint32_t* h, uint32_t n, float strength) {
    DownChain c{};
    c.n = (int)n; c.in = reinterpret_cast<const float4*>(in); c.iw = (int)iw; c.ih = (int)ih; c.strength = strength;
    for (uint32_t j = 0; j < kDcMax; j++) { const uint32_t k = j < n ? j : n - 1; c.out[j] = reinterpret_cast<float4*>(outs[k]); c.w[j] = (int)w[k]; c.h[j] = (int)h[k]; }
    hipLaunchKernelGGL(k_bloom_down_chain, dim3(cdiv_(w[n - 1], kDcTW), cdiv_(h[n - 1], kDcTH)), dim3(256), 0, s, c);
}
void launch_bloom_up_chain(hipStream_t s, float* const* mips, const uint32_t* w, const uint32_t* h, uint32_t n, float strength) {
    UpChain c{};
    c.n = (int)n; c.strength = strength;
    for (uint32_t j = 0; j <= kChainMax; j++) {   // unused levels repeat the last one: the range arithmetic stays defined
        const uint32_t k = j <= n ? j : n;
        c.mip[j] = reinterpret_cast<float4*>(mips[k]); c.w[j] = (int)w[k]; c.h[j] = (int)h[k];
    }
    hipLaunchKernelGGL(k_bloom_up_chain, dim3(cdiv_(w[0], kTileW), cdiv_(h[0], kChainTileH)), dim3(256), 0, s, c);
}
void launch_tonemap(hipStream_t s, const float* hdr, const float* bloom0, uint8_t* out, uint32_t w, uint32_t h, float exposure, float gamma,
                    bool linear_tap) {
    hipLaunchKernelGGL(k_tonemap, dim3(cdiv_(w, 64), cdiv_(h, 4)), dim3(256), 0, s, reinterpret_cast<const float4*>(hdr),
                       reinterpret_cast<const float4*>(bloom0), reinterpret_cast<uchar4*>(out), (int)w, (int)h, exposure, gamma, linear_tap ? 1 : 0);
}

}  // namespace vpt
