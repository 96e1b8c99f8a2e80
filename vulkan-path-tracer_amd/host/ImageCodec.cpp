// ImageCodec.cpp — PNG and JPEG texture decoding for the scene importer (stands in for stb_image behind
// VulkanHelper::AssetImporter::ImportTexture; reference call sites PathTracer.cpp:812-836 LoadTexture and :239,259,279,299,319).
//
// The reference's importer decodes LDR textures with stb_image (VulkanHelper's dependency; the submodule is empty in this
// snapshot, so neither it nor stb is in /root/reference).  Decoder choice changes JPEG texels, so this file restates stb_image's
// published algorithm (v2.2x series) rather than libjpeg's: its fixed-point "islow"-derived IDCT with 12-bit constants and
// its 10 / 17-bit descales, its triangle-filter chroma upsampling for the 2x1, 1x2 and 2x2 cases (nearest otherwise) and its
// 20-bit fixed-point YCbCr conversion; for PNG its conventions for what the format leaves open (16-bit samples keep the high
// byte, sub-byte grey is scaled by 255 / (2^depth - 1), a tRNS colour key is compared before the depth conversion).  Parity
// with stb_image itself is unpinned here (it cannot be run); tests/test_image_codecs.py holds this decoder against PIL
// (libpng: byte-exact; libjpeg-turbo: within the 2 code values by which two correct JPEG decoders may differ), and
// vulkan-path-tracer_amd/imagefiles.py is the same decoder in numpy, byte-identical to this one by test.
// Everything is decoded to RGBA8, as stbi_load(..., 4) delivers it to LoadTexture.
#include <zlib.h>

#include <algorithm>
#include <cstdlib>
#include <memory>
#include <cstring>

#include "SceneLoader.h"

namespace vpthost {
namespace {

// No image this importer accepts has more texels than this (the texel pool vpt_set_scene accepts is < 4 GiB; ADVICE r2: a
// crafted header must not force a multi-GiB allocation before any pixel data has been validated).
constexpr uint64_t kMaxTexels = 1ull << 28;   // 16384 x 16384
constexpr int kMaxJpegScans = 64;             // a progressive file of real encoders has about ten

// ------------------------------------------------------------------------------------------------ PNG
inline int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return pa <= pb && pa <= pc ? a : (pb <= pc ? b : c);
}

struct PngInfo {
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0, channels = 0;
    uint8_t palette[256][4];
    int palette_n = 0;
    bool has_key = false;
    uint16_t key[3] = {0, 0, 0};   // tRNS colour key at the file's own depth
};

// One (sub)image of `w` x `h` pixels, filtered scanlines at `src` -> one 16-bit sample per channel in `dst` (w*h*channels)
bool png_unfilter(const PngInfo& I, const uint8_t* src, size_t avail, uint32_t w, uint32_t h, std::vector<uint16_t>& dst, size_t& used) {
    const size_t bits = (size_t)I.channels * I.depth, row = (w * bits + 7) / 8, bpp = bits >= 8 ? bits / 8 : 1;
    if (avail < (row + 1) * (size_t)h) return false;
    std::vector<uint8_t> cur(row), prev(row, 0);
    dst.assign((size_t)w * h * I.channels, 0);
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t* s = src + (row + 1) * y;
        const int ft = s[0];
        if (ft > 4) return false;
        s++;
        for (size_t x = 0; x < row; x++) {
            const int a = x >= bpp ? cur[x - bpp] : 0, b = prev[x], c = x >= bpp ? prev[x - bpp] : 0;
            int v = s[x];
            switch (ft) { case 1: v += a; break; case 2: v += b; break; case 3: v += (a + b) >> 1; break; case 4: v += paeth(a, b, c); break; default: break; }
            cur[x] = (uint8_t)v;
        }
        uint16_t* d = &dst[(size_t)y * w * I.channels];
        const size_t n = (size_t)w * I.channels;
        if (I.depth == 8) for (size_t i = 0; i < n; i++) d[i] = cur[i];
        else if (I.depth == 16) for (size_t i = 0; i < n; i++) d[i] = (uint16_t)((cur[2 * i] << 8) | cur[2 * i + 1]);
        else for (size_t i = 0; i < n; i++) d[i] = (uint16_t)((cur[(i * I.depth) >> 3] >> (8 - I.depth - ((i * I.depth) & 7))) & ((1 << I.depth) - 1));
        prev.swap(cur);
    }
    used = (row + 1) * (size_t)h;
    return true;
}

}  // namespace

bool DecodePNG(const std::string& f, const std::string& path, TextureAsset& out, std::string& error) {
    static const unsigned char sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (f.size() < 33 || std::memcmp(f.data(), sig, 8) != 0) { error = "not a PNG: " + path; return false; }
    auto be32 = [&](size_t o) { return ((uint32_t)(uint8_t)f[o] << 24) | ((uint32_t)(uint8_t)f[o + 1] << 16) | ((uint32_t)(uint8_t)f[o + 2] << 8) | (uint8_t)f[o + 3]; };
    PngInfo I;
    std::string idat, trns;
    for (size_t p = 8; p + 12 <= f.size();) {
        const uint32_t len = be32(p);
        const std::string type = f.substr(p + 4, 4);
        if ((uint64_t)p + 12 + len > f.size()) break;
        if (type == "IHDR") {
            if (len < 13) { error = "bad PNG header: " + path; return false; }
            I.w = be32(p + 8); I.h = be32(p + 12); I.depth = (uint8_t)f[p + 16]; I.ctype = (uint8_t)f[p + 17]; I.interlace = (uint8_t)f[p + 20];
        } else if (type == "PLTE") {
            if (len > 768 || len % 3) { error = "bad PNG palette: " + path; return false; }
            I.palette_n = (int)(len / 3);
            for (int i = 0; i < I.palette_n; i++) { for (int c = 0; c < 3; c++) I.palette[i][c] = (uint8_t)f[p + 8 + 3 * i + c]; I.palette[i][3] = 255; }
        } else if (type == "tRNS") trns = f.substr(p + 8, len);
        else if (type == "IDAT") idat.append(f, p + 8, len);
        else if (type == "IEND") break;
        p += 12 + (size_t)len;
    }
    I.channels = I.ctype == 0 ? 1 : I.ctype == 2 ? 3 : I.ctype == 3 ? 1 : I.ctype == 4 ? 2 : I.ctype == 6 ? 4 : 0;
    const bool depth_ok = I.ctype == 0 ? (I.depth == 1 || I.depth == 2 || I.depth == 4 || I.depth == 8 || I.depth == 16)
                        : I.ctype == 3 ? (I.depth == 1 || I.depth == 2 || I.depth == 4 || I.depth == 8) : (I.depth == 8 || I.depth == 16);
    if (I.channels == 0 || !depth_ok || I.interlace > 1 || I.w == 0 || I.h == 0) { error = "unsupported PNG format: " + path; return false; }
    if ((uint64_t)I.w * I.h > kMaxTexels) { error = "PNG larger than 2^28 texels: " + path; return false; }
    if (I.ctype == 3) {
        if (I.palette_n == 0) { error = "PNG palette missing: " + path; return false; }
        for (size_t i = 0; i < trns.size() && i < (size_t)I.palette_n; i++) I.palette[i][3] = (uint8_t)trns[i];
    } else if (!trns.empty() && (I.ctype == 0 || I.ctype == 2)) {
        if (trns.size() < (size_t)I.channels * 2) { error = "bad PNG tRNS: " + path; return false; }
        I.has_key = true;
        for (int c = 0; c < I.channels; c++) I.key[c] = (uint16_t)(((uint8_t)trns[2 * c] << 8) | (uint8_t)trns[2 * c + 1]);
    }
    // the inflated size is known from the header: allocate exactly that much
    static const int xo[7] = {0, 4, 0, 2, 0, 1, 0}, yo[7] = {0, 0, 4, 0, 2, 0, 1}, xs[7] = {8, 8, 4, 4, 2, 2, 1}, ys[7] = {8, 8, 8, 4, 4, 2, 2};
    const size_t bits = (size_t)I.channels * I.depth;
    size_t total = 0;
    if (!I.interlace) total = (((size_t)I.w * bits + 7) / 8 + 1) * I.h;
    else for (int k = 0; k < 7; k++) {
        const uint32_t pw = (I.w - xo[k] + xs[k] - 1) / xs[k], ph = (I.h - yo[k] + ys[k] - 1) / ys[k];
        if (I.w > (uint32_t)xo[k] && I.h > (uint32_t)yo[k] && pw && ph) total += (((size_t)pw * bits + 7) / 8 + 1) * ph;
    }
    // deflate expands at most ~1032 : 1: a stream too short for the size the header declares is rejected BEFORE the (up to 2 GiB) buffers
    // below are allocated and zero-filled — a crafted header with a few KB of IDAT costs nothing
    if ((uint64_t)idat.size() * 1032ull + 1024ull < (uint64_t)total) { error = "PNG inflate failed: " + path; return false; }
    std::vector<uint8_t> raw(total);
    uLongf dl = (uLongf)raw.size();
    if (uncompress(raw.data(), &dl, (const Bytef*)idat.data(), (uLong)idat.size()) != Z_OK || dl != raw.size()) { error = "PNG inflate failed: " + path; return false; }
    std::vector<uint16_t> img((size_t)I.w * I.h * I.channels), sub;
    size_t used = 0;
    if (!I.interlace) {
        if (!png_unfilter(I, raw.data(), raw.size(), I.w, I.h, img, used)) { error = "bad PNG scanline: " + path; return false; }
    } else {   // Adam7: seven sub-images, each filtered on its own
        size_t off = 0;
        for (int k = 0; k < 7; k++) {
            if (I.w <= (uint32_t)xo[k] || I.h <= (uint32_t)yo[k]) continue;
            const uint32_t pw = (I.w - xo[k] + xs[k] - 1) / xs[k], ph = (I.h - yo[k] + ys[k] - 1) / ys[k];
            if (!pw || !ph) continue;
            if (!png_unfilter(I, raw.data() + off, raw.size() - off, pw, ph, sub, used)) { error = "bad PNG scanline: " + path; return false; }
            off += used;
            for (uint32_t y = 0; y < ph; y++)
                for (uint32_t x = 0; x < pw; x++)
                    for (int c = 0; c < I.channels; c++)
                        img[((size_t)(yo[k] + y * ys[k]) * I.w + xo[k] + (size_t)x * xs[k]) * I.channels + c] = sub[((size_t)y * pw + x) * I.channels + c];
        }
    }
    // samples -> RGBA8
    out.Width = I.w; out.Height = I.h; out.Channels = 4; out.Data.resize((size_t)I.w * I.h * 4);
    const int scale = I.depth == 1 ? 255 : I.depth == 2 ? 85 : I.depth == 4 ? 17 : 1;   // sub-byte grey: 255 / (2^depth - 1)
    auto to8 = [&](uint16_t v) -> uint8_t { return I.depth == 16 ? (uint8_t)(v >> 8) : I.depth == 8 ? (uint8_t)v : (uint8_t)(v * scale); };
    for (size_t i = 0; i < (size_t)I.w * I.h; i++) {
        const uint16_t* s = &img[i * I.channels];
        uint8_t* d = &out.Data[i * 4];
        switch (I.ctype) {
            case 0: d[0] = d[1] = d[2] = to8(s[0]); d[3] = (I.has_key && s[0] == I.key[0]) ? 0 : 255; break;
            case 2: d[0] = to8(s[0]); d[1] = to8(s[1]); d[2] = to8(s[2]); d[3] = (I.has_key && s[0] == I.key[0] && s[1] == I.key[1] && s[2] == I.key[2]) ? 0 : 255; break;
            case 3: {
                if (s[0] >= I.palette_n) { error = "PNG palette index out of range: " + path; return false; }
                std::memcpy(d, I.palette[s[0]], 4);
                break;
            }
            case 4: d[0] = d[1] = d[2] = to8(s[0]); d[3] = to8(s[1]); break;
            default: d[0] = to8(s[0]); d[1] = to8(s[1]); d[2] = to8(s[2]); d[3] = to8(s[3]); break;
        }
    }
    return true;
}

// ------------------------------------------------------------------------------------------------ JPEG
namespace {

const uint8_t kZigzag[64 + 15] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                  35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
                                  63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};   // the tail absorbs runs past the block's end

struct Huff {
    bool present = false;
    uint8_t values[256];
    int mincode[17], maxcode[18], valptr[17];   // canonical decoding per code length (ITU T.81 F.2.2.3)
    bool build(const uint8_t counts[16], const uint8_t* vals, int nvals) {
        std::memcpy(values, vals, (size_t)nvals);
        int code = 0, k = 0;
        for (int len = 1; len <= 16; len++) {
            valptr[len] = k; mincode[len] = code;
            code += counts[len - 1]; k += counts[len - 1];
            maxcode[len] = counts[len - 1] ? code - 1 : -1;
            if (code > (1 << len)) return false;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        present = true;
        return k == nvals;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int x = 0, y = 0, w2 = 0, h2 = 0;   // size in samples; padded to whole MCUs
    int dc_pred = 0;
    int coeff_w = 0, coeff_h = 0;        // blocks (progressive)
    std::vector<uint8_t> data;
    std::vector<int16_t> coeff;
};

struct Jpeg {
    const uint8_t* p = nullptr; const uint8_t* end = nullptr;
    uint32_t bitbuf = 0; int bitcnt = 0;
    int marker = -1;                     // marker met inside entropy-coded data (0xff00 stuffing removed)
    bool nomore = false;
    int img_x = 0, img_y = 0, ncomp = 0;
    int h_max = 1, v_max = 1, mcu_x = 0, mcu_y = 0, mcu_w = 0, mcu_h = 0;
    bool progressive = false, rgb = false;
    int adobe_transform = -1; bool jfif = false;
    int restart_interval = 0, todo = 0;
    int spec_start = 0, spec_end = 63, succ_high = 0, succ_low = 0, eob_run = 0;
    int scan_n = 0, order[4] = {0, 0, 0, 0};
    uint16_t dequant[4][64];
    Huff hdc[4], hac[4];
    Component comp[4];
    std::string err;

    bool fail(const char* m) { if (err.empty()) err = m; return false; }
    int get8() { return p < end ? *p++ : 0; }
    int get16() { const int a = get8(); return (a << 8) | get8(); }

    void grow() {   // refill the bit buffer to >= 25 bits; after a marker only zero bits follow
        while (bitcnt <= 24) {
            int b = nomore ? 0 : get8();
            if (b == 0xff && !nomore) {
                int c = get8();
                while (c == 0xff) c = get8();   // fill bytes
                if (c != 0) { marker = c; nomore = true; b = 0; }
            }
            bitbuf |= (uint32_t)b << (24 - bitcnt);
            bitcnt += 8;
        }
    }
    int bits(int n) {   // n in 0..16
        if (n == 0) return 0;
        if (bitcnt < n) grow();
        const int v = (int)(bitbuf >> (32 - n));
        bitbuf <<= n; bitcnt -= n;
        return v;
    }
    int bit() { return bits(1); }
    int decode(const Huff& h) {
        if (bitcnt < 16) grow();
        int code = 0;
        for (int len = 1; len <= 16; len++) {
            code = (int)(bitbuf >> (32 - len));
            if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len]) {
                bitbuf <<= len; bitcnt -= len;
                return h.values[h.valptr[len] + code - h.mincode[len]];
            }
        }
        return -1;
    }
    int extend_receive(int n) {   // T.81 F.2.2.1 EXTEND(RECEIVE(n), n)
        if (n == 0) return 0;
        const int v = bits(n);
        return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
    }
    void reset_entropy() {
        bitbuf = 0; bitcnt = 0; nomore = false; marker = -1;
        for (int i = 0; i < 4; i++) comp[i].dc_pred = 0;
        todo = restart_interval ? restart_interval : 0x7fffffff;
        eob_run = 0;
    }

    bool block_baseline(int16_t data[64], const Huff& dc, const Huff& ac, int ci) {
        const uint16_t* dq = dequant[comp[ci].tq];
        std::memset(data, 0, 64 * sizeof(int16_t));
        const int t = decode(dc);
        if (t < 0 || t > 15) return fail("bad huffman code");
        const int diff = extend_receive(t);
        comp[ci].dc_pred += diff;
        data[0] = (int16_t)(comp[ci].dc_pred * dq[0]);
        for (int k = 1; k < 64;) {
            const int rs = decode(ac);
            if (rs < 0) return fail("bad huffman code");
            const int s = rs & 15, r = rs >> 4;
            if (s == 0) { if (rs != 0xf0) break; k += 16; }
            else { k += r; const int z = kZigzag[k++]; data[z] = (int16_t)(extend_receive(s) * dq[z]); }
        }
        return true;
    }
    bool block_prog_dc(int16_t data[64], const Huff& dc, int ci) {
        if (spec_end != 0) return fail("can't merge dc and ac");
        if (succ_high == 0) {
            std::memset(data, 0, 64 * sizeof(int16_t));
            const int t = decode(dc);
            if (t < 0 || t > 15) return fail("bad huffman code");
            comp[ci].dc_pred += extend_receive(t);
            data[0] = (int16_t)(comp[ci].dc_pred * (1 << succ_low));
        } else if (bit()) data[0] = (int16_t)(data[0] + (1 << succ_low));
        return true;
    }
    bool block_prog_ac(int16_t data[64], const Huff& ac) {
        if (spec_start == 0) return fail("can't merge dc and ac");
        if (succ_high == 0) {
            const int shift = succ_low;
            if (eob_run) { --eob_run; return true; }
            int k = spec_start;
            do {
                const int rs = decode(ac);
                if (rs < 0) return fail("bad huffman code");
                const int s = rs & 15, r = rs >> 4;
                if (s == 0) {
                    if (r < 15) { eob_run = 1 << r; if (r) eob_run += bits(r); --eob_run; break; }
                    k += 16;
                } else { k += r; const int z = kZigzag[k++]; data[z] = (int16_t)(extend_receive(s) * (1 << shift)); }
            } while (k <= spec_end);
        } else {   // refinement of already-seen coefficients
            const int16_t bitv = (int16_t)(1 << succ_low);
            auto refine = [&](int16_t* q) { if (bit() && (*q & bitv) == 0) { if (*q > 0) *q = (int16_t)(*q + bitv); else *q = (int16_t)(*q - bitv); } };
            if (eob_run) {
                --eob_run;
                for (int k = spec_start; k <= spec_end; k++) { int16_t* q = &data[kZigzag[k]]; if (*q != 0) refine(q); }
            } else {
                int k = spec_start;
                do {
                    const int rs = decode(ac);
                    if (rs < 0) return fail("bad huffman code");
                    int s = rs & 15, r = rs >> 4;
                    if (s == 0) {
                        if (r < 15) { eob_run = (1 << r) - 1; if (r) eob_run += bits(r); r = 64; }   // run to the end of the band
                    } else {
                        if (s != 1) return fail("bad huffman code");
                        s = bit() ? bitv : -bitv;
                    }
                    while (k <= spec_end) {
                        int16_t* q = &data[kZigzag[k++]];
                        if (*q != 0) refine(q);
                        else { if (r == 0) { *q = (int16_t)s; break; } --r; }
                    }
                } while (k <= spec_end);
            }
        }
        return true;
    }

    // stb_image's integer IDCT: 12-bit constants, column pass descaled by 10 bits, row pass by 17 with the +128 level shift folded in
    static inline int f2f(double x) { return (int)(x * 4096 + 0.5); }
    static inline uint8_t clamp8(int x) { return (uint8_t)(x < 0 ? 0 : x > 255 ? 255 : x); }
    static void idct(uint8_t* out, int stride, const int16_t d[64]) {
        int val[64];
#define VPT_IDCT_1D(s0, s1, s2, s3, s4, s5, s6, s7)                                                                     \
        int t0, t1, t2, t3, p1, p2, p3, p4, p5, x0, x1, x2, x3;                                                          \
        p2 = s2; p3 = s6; p1 = (p2 + p3) * 2217; t2 = p1 + p3 * -7567; t3 = p1 + p2 * 3135;                              \
        p2 = s0; p3 = s4; t0 = (p2 + p3) * 4096; t1 = (p2 - p3) * 4096;                                                  \
        x0 = t0 + t3; x3 = t0 - t3; x1 = t1 + t2; x2 = t1 - t2;                                                          \
        t0 = s7; t1 = s5; t2 = s3; t3 = s1;                                                                              \
        p3 = t0 + t2; p4 = t1 + t3; p1 = t0 + t3; p2 = t1 + t2; p5 = (p3 + p4) * 4816;                                   \
        t0 = t0 * 1223; t1 = t1 * 8410; t2 = t2 * 12586; t3 = t3 * 6149;                                                 \
        p1 = p5 + p1 * -3685; p2 = p5 + p2 * -10497; p3 = p3 * -8034; p4 = p4 * -1597;                                   \
        t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3;
        for (int i = 0; i < 8; i++) {
            const int16_t* c = d + i; int* v = val + i;
            VPT_IDCT_1D(c[0], c[8], c[16], c[24], c[32], c[40], c[48], c[56])
            x0 += 512; x1 += 512; x2 += 512; x3 += 512;
            v[0] = (x0 + t3) >> 10; v[56] = (x0 - t3) >> 10; v[8] = (x1 + t2) >> 10; v[48] = (x1 - t2) >> 10;
            v[16] = (x2 + t1) >> 10; v[40] = (x2 - t1) >> 10; v[24] = (x3 + t0) >> 10; v[32] = (x3 - t0) >> 10;
        }
        for (int i = 0; i < 8; i++) {
            const int* v = val + 8 * i; uint8_t* o = out + (size_t)stride * i;
            VPT_IDCT_1D(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7])
            x0 += 65536 + (128 << 17); x1 += 65536 + (128 << 17); x2 += 65536 + (128 << 17); x3 += 65536 + (128 << 17);
            o[0] = clamp8((x0 + t3) >> 17); o[7] = clamp8((x0 - t3) >> 17); o[1] = clamp8((x1 + t2) >> 17); o[6] = clamp8((x1 - t2) >> 17);
            o[2] = clamp8((x2 + t1) >> 17); o[5] = clamp8((x2 - t1) >> 17); o[3] = clamp8((x3 + t0) >> 17); o[4] = clamp8((x3 - t0) >> 17);
        }
#undef VPT_IDCT_1D
    }

    bool read_dqt(int len) {
        while (len > 0) {
            const int q = get8(), prec = q >> 4, t = q & 15;
            if ((prec != 0 && prec != 1) || t > 3) return fail("bad DQT");
            for (int i = 0; i < 64; i++) dequant[t][kZigzag[i]] = (uint16_t)(prec ? get16() : get8());
            len -= prec ? 129 : 65;
        }
        return len == 0 || fail("bad DQT length");
    }
    bool read_dht(int len) {
        while (len > 0) {
            const int q = get8(), tc = q >> 4, th = q & 15;
            if (tc > 1 || th > 3) return fail("bad DHT");
            uint8_t counts[16]; int n = 0;
            for (int i = 0; i < 16; i++) { counts[i] = (uint8_t)get8(); n += counts[i]; }
            if (n > 256) return fail("bad DHT");
            uint8_t vals[256];
            for (int i = 0; i < n; i++) vals[i] = (uint8_t)get8();
            if (!(tc == 0 ? hdc[th] : hac[th]).build(counts, vals, n)) return fail("bad code lengths");
            len -= 17 + n;
        }
        return len == 0 || fail("bad DHT length");
    }
    bool read_sof(int len, bool prog) {
        progressive = prog;
        if (len < 11) return fail("bad SOF length");
        if (get8() != 8) return fail("only 8-bit JPEG is supported");
        img_y = get16(); img_x = get16(); ncomp = get8();
        if (img_x <= 0 || img_y <= 0) return fail("empty JPEG");
        if ((uint64_t)img_x * (uint64_t)img_y > kMaxTexels) return fail("JPEG larger than 2^28 texels");
        if (ncomp != 1 && ncomp != 3) return fail("only 1- and 3-component JPEG is supported");   // CMYK / YCCK: not a texture format
        if (len != 8 + 3 * ncomp) return fail("bad SOF length");
        rgb = false; int rgbn = 0;
        static const char ids[3] = {'R', 'G', 'B'};
        for (int i = 0; i < ncomp; i++) {
            Component& c = comp[i];
            c.id = get8(); if (ncomp == 3 && c.id == ids[i]) rgbn++;
            const int q = get8(); c.h = q >> 4; c.v = q & 15; c.tq = get8();
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return fail("bad SOF component");
        }
        rgb = rgbn == 3;
        h_max = v_max = 1;
        for (int i = 0; i < ncomp; i++) { h_max = std::max(h_max, comp[i].h); v_max = std::max(v_max, comp[i].v); }
        for (int i = 0; i < ncomp; i++) if (h_max % comp[i].h || v_max % comp[i].v) return fail("bad sampling factors");
        mcu_w = h_max * 8; mcu_h = v_max * 8;
        mcu_x = (img_x + mcu_w - 1) / mcu_w; mcu_y = (img_y + mcu_h - 1) / mcu_h;
        for (int i = 0; i < ncomp; i++) {
            Component& c = comp[i];
            c.x = (img_x * c.h + h_max - 1) / h_max; c.y = (img_y * c.v + v_max - 1) / v_max;
            c.w2 = mcu_x * c.h * 8; c.h2 = mcu_y * c.v * 8;
            c.data.assign((size_t)c.w2 * c.h2, 0);
            if (progressive) { c.coeff_w = c.w2 / 8; c.coeff_h = c.h2 / 8; c.coeff.assign((size_t)c.w2 * c.h2, 0); }
        }
        return true;
    }
    bool read_sos(int len) {
        scan_n = get8();
        if (scan_n < 1 || scan_n > ncomp || len != 6 + 2 * scan_n) return fail("bad SOS");
        for (int i = 0; i < scan_n; i++) {
            const int id = get8(), q = get8();
            int which = -1;
            for (int k = 0; k < ncomp; k++) if (comp[k].id == id) which = k;
            if (which < 0) return fail("bad SOS component");
            comp[which].td = q >> 4; comp[which].ta = q & 15;
            if (comp[which].td > 3 || comp[which].ta > 3) return fail("bad SOS tables");
            order[i] = which;
        }
        spec_start = get8(); spec_end = get8();
        const int q = get8(); succ_high = q >> 4; succ_low = q & 15;
        if (progressive) { if (spec_start > 63 || spec_end > 63 || spec_start > spec_end || succ_high > 13 || succ_low > 13) return fail("bad SOS"); }
        else { if (spec_start != 0 || succ_high != 0 || succ_low != 0) return fail("bad SOS"); spec_end = 63; }
        return true;
    }
    bool restart_if_due() {   // after every `restart_interval` MCUs: RSTn, predictors and bit buffer start over
        if (--todo > 0) return true;
        if (bitcnt < 24) grow();
        if (marker < 0xd0 || marker > 0xd7) return true;   // no RSTn here: the scan is over (or the file truncated) - what is decoded stays
        reset_entropy();
        return true;
    }
    bool decode_scan() {
        reset_entropy();
        int16_t blk[64];
        if (scan_n == 1) {   // non-interleaved: the component's own blocks, ceil(size / 8) per axis
            const int n = order[0]; Component& c = comp[n];
            const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
            for (int j = 0; j < h; j++)
                for (int i = 0; i < w; i++) {
                    if (!progressive) {
                        if (!hdc[c.td].present || !hac[c.ta].present) return fail("missing huffman table");
                        if (!block_baseline(blk, hdc[c.td], hac[c.ta], n)) return false;
                        idct(&c.data[(size_t)c.w2 * j * 8 + i * 8], c.w2, blk);
                    } else {
                        int16_t* d = &c.coeff[64 * ((size_t)i + (size_t)j * c.coeff_w)];
                        if (spec_start == 0) { if (!hdc[c.td].present) return fail("missing huffman table"); if (!block_prog_dc(d, hdc[c.td], n)) return false; }
                        else { if (!hac[c.ta].present) return fail("missing huffman table"); if (!block_prog_ac(d, hac[c.ta])) return false; }
                    }
                    if (!restart_if_due()) return false;
                    if (nomore && marker >= 0 && (marker < 0xd0 || marker > 0xd7) && bitcnt <= 0) return true;
                }
            return true;
        }
        for (int j = 0; j < mcu_y; j++)
            for (int i = 0; i < mcu_x; i++) {
                for (int k = 0; k < scan_n; k++) {
                    const int n = order[k]; Component& c = comp[n];
                    for (int y = 0; y < c.v; y++)
                        for (int x = 0; x < c.h; x++) {
                            const int x2 = (i * c.h + x) * 8, y2 = (j * c.v + y) * 8;
                            if (!progressive) {
                                if (!hdc[c.td].present || !hac[c.ta].present) return fail("missing huffman table");
                                if (!block_baseline(blk, hdc[c.td], hac[c.ta], n)) return false;
                                idct(&c.data[(size_t)c.w2 * y2 + x2], c.w2, blk);
                            } else {   // an interleaved progressive scan can only be a DC scan
                                if (!hdc[c.td].present) return fail("missing huffman table");
                                if (!block_prog_dc(&c.coeff[64 * ((size_t)(x2 >> 3) + (size_t)(y2 >> 3) * c.coeff_w)], hdc[c.td], n)) return false;
                            }
                        }
                }
                if (!restart_if_due()) return false;
            }
        return true;
    }
    void finish_progressive() {   // dequantise and transform the blocks the image actually covers
        for (int n = 0; n < ncomp; n++) {
            Component& c = comp[n];
            const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
            for (int j = 0; j < h; j++)
                for (int i = 0; i < w; i++) {
                    int16_t* d = &c.coeff[64 * ((size_t)i + (size_t)j * c.coeff_w)];
                    for (int k = 0; k < 64; k++) d[k] = (int16_t)(d[k] * dequant[c.tq][k]);
                    idct(&c.data[(size_t)c.w2 * j * 8 + i * 8], c.w2, d);
                }
        }
    }
    int next_marker() {
        if (marker >= 0) { const int m = marker; marker = -1; return m; }
        int x = get8();
        if (x != 0xff) return -1;
        while (x == 0xff) x = get8();
        return x;
    }
    bool decode_file() {
        if (get8() != 0xff || get8() != 0xd8) return fail("not a JPEG");
        bool have_sof = false, have_scan = false;
        int scans = 0;
        while (true) {
            int m = next_marker();
            while (m < 0 && p < end) m = next_marker();   // garbage between segments
            if (m < 0) break;
            if (m == 0xd9) break;
            if (m == 0xda) {
                if (!have_sof) return fail("SOS before SOF");
                if (++scans > kMaxJpegScans) return fail("too many scans");   // every scan walks every MCU of the image: bounded work for a crafted file
                const int len = get16();
                if (!read_sos(len)) return false;
                if (!decode_scan()) return false;
                have_scan = true;
                if (marker < 0) {   // the scan's data ends at the next marker: skip to it
                    while (p < end) { if (*p++ == 0xff) { while (p < end && *p == 0xff) p++; if (p < end && *p != 0) { marker = *p++; break; } } }
                }
                nomore = false;
                continue;
            }
            if (m >= 0xd0 && m <= 0xd7) continue;   // a restart marker outside a scan carries nothing
            const int len = get16() - 2;
            if (len < 0 || p + len > end) return fail("bad segment length");
            const uint8_t* seg_end = p + len;
            if (m == 0xdb) { if (!read_dqt(len)) return false; }
            else if (m == 0xc4) { if (!read_dht(len)) return false; }
            else if (m == 0xc0 || m == 0xc1 || m == 0xc2) { if (have_sof) return fail("two SOF segments"); if (!read_sof(len + 2, m == 0xc2)) return false; have_sof = true; }
            else if (m == 0xdd) { if (len != 2) return fail("bad DRI"); restart_interval = get16(); }
            else if (m == 0xee && len >= 12 && std::memcmp(p, "Adobe", 5) == 0) adobe_transform = p[11];
            else if (m == 0xe0 && len >= 5 && std::memcmp(p, "JFIF", 5) == 0) jfif = true;
            else if ((m >= 0xc3 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc)) return fail("unsupported JPEG coding process (lossless / hierarchical / arithmetic)");
            p = seg_end;
        }
        if (!have_sof || !have_scan) return fail("no image data");
        if (progressive) finish_progressive();
        return true;
    }
};

// stb_image's row resamplers: `out` gets w * hs samples from the rows `near` (the one the output row lies in) and `far`
void resample_row(uint8_t* out, const uint8_t* near_, const uint8_t* far_, int w, int hs, int vs) {
    if (hs == 1 && vs == 1) { std::memcpy(out, near_, (size_t)w); return; }
    if (hs == 1 && vs == 2) { for (int i = 0; i < w; i++) out[i] = (uint8_t)((3 * near_[i] + far_[i] + 2) >> 2); return; }
    if (hs == 2 && vs == 1) {
        const uint8_t* in = near_;
        if (w == 1) { out[0] = out[1] = in[0]; return; }
        out[0] = in[0]; out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
        int i;
        for (i = 1; i < w - 1; i++) { const int n = 3 * in[i] + 2; out[i * 2] = (uint8_t)((n + in[i - 1]) >> 2); out[i * 2 + 1] = (uint8_t)((n + in[i + 1]) >> 2); }
        out[i * 2] = (uint8_t)((in[w - 2] * 3 + in[w - 1] + 2) >> 2); out[i * 2 + 1] = in[w - 1];
        return;
    }
    if (hs == 2 && vs == 2) {
        if (w == 1) { out[0] = out[1] = (uint8_t)((3 * near_[0] + far_[0] + 2) >> 2); return; }
        int t1 = 3 * near_[0] + far_[0];
        out[0] = (uint8_t)((t1 + 2) >> 2);
        for (int i = 1; i < w; i++) {
            const int t0 = t1; t1 = 3 * near_[i] + far_[i];
            out[i * 2 - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4); out[i * 2] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
        }
        out[w * 2 - 1] = (uint8_t)((t1 + 2) >> 2);
        return;
    }
    for (int i = 0; i < w; i++) for (int j = 0; j < hs; j++) out[i * hs + j] = near_[i];   // other factors: nearest
}

}  // namespace

bool DecodeJPEG(const std::string& f, const std::string& path, TextureAsset& out, std::string& error) {
    std::unique_ptr<Jpeg> jp(new Jpeg());
    Jpeg& J = *jp;
    J.p = (const uint8_t*)f.data(); J.end = J.p + f.size();
    std::memset(J.dequant, 0, sizeof(J.dequant));
    if (!J.decode_file()) { error = "JPEG: " + (J.err.empty() ? std::string("decode failed") : J.err) + ": " + path; return false; }
    const int W = J.img_x, H = J.img_y;
    out.Width = (uint32_t)W; out.Height = (uint32_t)H; out.Channels = 4; out.Data.assign((size_t)W * H * 4, 255);
    struct Res { int hs, vs, ystep, ypos, w_lores; const uint8_t* line0; const uint8_t* line1; std::vector<uint8_t> buf; } rs[3];
    for (int k = 0; k < J.ncomp; k++) {
        Res& r = rs[k]; const Component& c = J.comp[k];
        r.hs = J.h_max / c.h; r.vs = J.v_max / c.v; r.ystep = r.vs >> 1; r.ypos = 0;
        r.w_lores = (W + r.hs - 1) / r.hs; r.line0 = r.line1 = c.data.data();
        r.buf.assign((size_t)r.w_lores * r.hs + 8, 0);
    }
    // 3 components are YCbCr unless the file says RGB (component ids 'R','G','B', or an Adobe segment with transform 0)
    const bool is_rgb = J.ncomp == 3 && (J.rgb || (J.adobe_transform == 0 && !J.jfif));
    constexpr int kCrR = 5743 << 8, kCrG = 2925 << 8, kCbG = 1410 << 8, kCbB = 7258 << 8;   // (int)(c * 4096 + 0.5) << 8 for 1.402, 0.71414, 0.34414, 1.772
    for (int j = 0; j < H; j++) {
        const uint8_t* row[3] = {nullptr, nullptr, nullptr};
        for (int k = 0; k < J.ncomp; k++) {
            Res& r = rs[k]; const Component& c = J.comp[k];
            const bool bot = r.ystep >= (r.vs >> 1);
            resample_row(r.buf.data(), bot ? r.line1 : r.line0, bot ? r.line0 : r.line1, r.w_lores, r.hs, r.vs);
            row[k] = r.buf.data();
            if (++r.ystep >= r.vs) { r.ystep = 0; r.line0 = r.line1; if (++r.ypos < c.y) r.line1 += c.w2; }
        }
        uint8_t* d = &out.Data[(size_t)j * W * 4];
        if (J.ncomp == 1) for (int i = 0; i < W; i++) { d[4 * i] = d[4 * i + 1] = d[4 * i + 2] = row[0][i]; }
        else if (is_rgb) for (int i = 0; i < W; i++) { d[4 * i] = row[0][i]; d[4 * i + 1] = row[1][i]; d[4 * i + 2] = row[2][i]; }
        else for (int i = 0; i < W; i++) {
            const int yf = (row[0][i] << 20) + (1 << 19), cb = row[1][i] - 128, cr = row[2][i] - 128;
            int r = yf + cr * kCrR, g = yf + cr * -kCrG + (int)((uint32_t)(cb * -kCbG) & 0xffff0000u), b = yf + cb * kCbB;
            r >>= 20; g >>= 20; b >>= 20;
            d[4 * i] = Jpeg::clamp8(r); d[4 * i + 1] = Jpeg::clamp8(g); d[4 * i + 2] = Jpeg::clamp8(b);
        }
    }
    return true;
}

bool DecodeImage(const std::string& bytes, const std::string& name, TextureAsset& out, std::string& error) {
    if (bytes.size() >= 8 && (uint8_t)bytes[0] == 137 && bytes[1] == 'P' && bytes[2] == 'N' && bytes[3] == 'G') return DecodePNG(bytes, name, out, error);
    if (bytes.size() >= 3 && (uint8_t)bytes[0] == 0xff && (uint8_t)bytes[1] == 0xd8) return DecodeJPEG(bytes, name, out, error);
    error = "unsupported image format (need PNG or JPEG): " + name;
    return false;
}

}  // namespace vpthost
