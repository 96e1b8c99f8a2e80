// SceneLoader.cpp — minimal JSON + glTF 2.0 + PNG readers (see SceneLoader.h).
#include "SceneLoader.h"
#include <cmath>

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>

namespace vpthost {
namespace {

// ---------------------------------------------------------------- JSON
struct JVal;
using JPtr = std::shared_ptr<JVal>;
// Files are untrusted: every array index and every required key goes through accessors that never touch memory they should not.
// A bad access yields a shared Null value and raises g_malformed, which ImportScene turns into an error.
thread_local bool g_malformed = false;
struct JVal {
    enum Type { Null, Bool, Num, Str, Arr, Obj } type = Null;
    double num = 0;
    bool b = false;
    std::string str;
    std::vector<JPtr> arr;
    std::map<std::string, JPtr> obj;
    static const JVal& null_value() { static const JVal v; return v; }
    const JVal* get(const std::string& k) const { auto it = obj.find(k); return it == obj.end() ? nullptr : it->second.get(); }
    bool has(const std::string& k) const { return obj.count(k) != 0; }
    double number(const std::string& k, double def) const { const JVal* v = get(k); return v && v->type == Num ? v->num : def; }
    size_t size() const { return arr.size(); }
    const JVal& operator[](size_t i) const { if (i >= arr.size() || !arr[i]) { g_malformed = true; return null_value(); } return *arr[i]; }
    // required member: the shared Null (and the malformed flag) when it is missing
    const JVal& req(const std::string& k) const { const JVal* v = get(k); if (!v) { g_malformed = true; return null_value(); } return *v; }
    // array index stored as a JSON number: SIZE_MAX (always out of range) for anything that is not a non-negative integer
    size_t index(const std::string& k, double def) const { const double d = number(k, def); return (d >= 0.0 && d < 9.0e15) ? (size_t)d : (size_t)-1; }
    size_t as_index() const { return (type == Num && num >= 0.0 && num < 9.0e15) ? (size_t)num : (size_t)-1; }
};
constexpr int kMaxJsonDepth = 256;
struct JParser {
    const std::string& s; size_t p = 0; bool ok = true;
    explicit JParser(const std::string& t) : s(t) {}
    void ws() { while (p < s.size() && (s[p] == ' ' || s[p] == '\n' || s[p] == '\t' || s[p] == '\r')) p++; }
    JPtr parse(int depth = 0) {
        ws();
        auto v = std::make_shared<JVal>();
        if (p >= s.size() || depth > kMaxJsonDepth) { ok = false; return v; }   // nesting is bounded: the parser recurses
        char c = s[p];
        if (c == '{') {
            v->type = JVal::Obj; p++; ws();
            if (p < s.size() && s[p] == '}') { p++; return v; }
            while (ok) {
                ws(); JPtr k = parse(depth + 1); ws();
                if (!ok || k->type != JVal::Str || p >= s.size() || s[p] != ':') { ok = false; break; }
                p++;
                v->obj[k->str] = parse(depth + 1); ws();
                if (p < s.size() && s[p] == ',') { p++; continue; }
                if (p < s.size() && s[p] == '}') { p++; break; }
                ok = false;
            }
        } else if (c == '[') {
            v->type = JVal::Arr; p++; ws();
            if (p < s.size() && s[p] == ']') { p++; return v; }
            while (ok) {
                v->arr.push_back(parse(depth + 1)); ws();
                if (p < s.size() && s[p] == ',') { p++; continue; }
                if (p < s.size() && s[p] == ']') { p++; break; }
                ok = false;
            }
        } else if (c == '"') {
            v->type = JVal::Str; p++;
            while (p < s.size() && s[p] != '"') {
                if (s[p] == '\\' && p + 1 < s.size()) {
                    char e = s[p + 1];
                    v->str += e == 'n' ? '\n' : e == 't' ? '\t' : e;  // \uXXXX is not needed for the asset paths we read
                    p += 2;
                } else v->str += s[p++];
            }
            p++;
        } else if (s.compare(p, 4, "true") == 0) { v->type = JVal::Bool; v->b = true; p += 4; }
        else if (s.compare(p, 5, "false") == 0) { v->type = JVal::Bool; p += 5; }
        else if (s.compare(p, 4, "null") == 0) { p += 4; }
        else {
            char* end = nullptr;
            v->type = JVal::Num; v->num = std::strtod(s.c_str() + p, &end);
            if (end == s.c_str() + p) ok = false;
            p = (size_t)(end - s.c_str());
        }
        return v;
    }
};

bool read_file(const std::string& path, std::string& out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::ostringstream ss; ss << f.rdbuf(); out = ss.str();
    return true;
}
std::string dir_of(const std::string& path) { size_t k = path.find_last_of("/\\"); return k == std::string::npos ? std::string(".") : path.substr(0, k); }

Mat4 from_rows(const double r[4][4]) { Mat4 m; for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) m.at(i, j) = (float)r[i][j]; return m; }

// node local matrix in double (TRS or explicit matrix)
void node_matrix(const JVal& n, double M[4][4]) {
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) M[i][j] = i == j;
    if (const JVal* m = n.get("matrix")) { for (int c = 0; c < 4; c++) for (int r = 0; r < 4; r++) M[r][c] = (*m)[c * 4 + r].num; return; }
    double T[4][4], R[4][4], S[4][4];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) T[i][j] = R[i][j] = S[i][j] = i == j;
    if (const JVal* t = n.get("translation")) for (int i = 0; i < 3; i++) T[i][3] = (*t)[i].num;
    if (const JVal* q = n.get("rotation")) {
        double x = (*q)[0].num, y = (*q)[1].num, z = (*q)[2].num, w = (*q)[3].num;
        double r[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)},
                          {2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)},
                          {2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)}};
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i][j] = r[i][j];
    }
    if (const JVal* sc = n.get("scale")) for (int i = 0; i < 3; i++) S[i][i] = (*sc)[i].num;
    double TR[4][4];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { TR[i][j] = 0; for (int k = 0; k < 4; k++) TR[i][j] += T[i][k] * R[k][j]; }
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { M[i][j] = 0; for (int k = 0; k < 4; k++) M[i][j] += TR[i][k] * S[k][j]; }
}
void matmul(const double A[4][4], const double B[4][4], double C[4][4]) {
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { double s = 0; for (int k = 0; k < 4; k++) s += A[i][k] * B[k][j]; C[i][j] = s; }
}
// F * M * F with F = diag(1,-1,1,1): glTF is Y-up, the reference's world is Y-down
void flip_y(double M[4][4]) { for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) if ((i == 1) != (j == 1)) M[i][j] = -M[i][j]; }

void default_material(vpt_material& m) {  // PathTracer.h:14-33
    std::memset(&m, 0, sizeof(m));
    for (int i = 0; i < 3; i++) { m.base_color[i] = 1; m.specular_color[i] = 1; m.medium_color[i] = 1; }
    m.roughness = 1.0f; m.ior = 1.5f;
    m.base_color_texture = 0; m.normal_texture = 1; m.roughness_texture = 2; m.metallic_texture = 3; m.emissive_texture = 4;
}

}  // namespace

bool LoadPNG(const std::string& path, TextureAsset& out, std::string& error) {
    std::string f;
    if (!read_file(path, f)) { error = "cannot open " + path; return false; }
    return DecodePNG(f, path, out, error);
}
bool LoadImage(const std::string& path, TextureAsset& out, std::string& error) {   // PNG or JPEG, by content
    std::string f;
    if (!read_file(path, f)) { error = "cannot open " + path; return false; }
    return DecodeImage(f, path, out, error);
}
// base64 payload of a "data:<mime>;base64,<payload>" URI (glTF 2.0 section 2.6); false if `uri` is not one
static bool decode_data_uri(const std::string& uri, std::string& out) {
    if (uri.compare(0, 5, "data:") != 0) return false;
    size_t comma = uri.find(',');
    if (comma == std::string::npos || uri.find(";base64") == std::string::npos || uri.find(";base64") > comma) return false;
    out.clear();
    uint32_t acc = 0; int bits = 0;
    for (size_t i = comma + 1; i < uri.size(); i++) {
        const char ch = uri[i];
        int v = (ch >= 'A' && ch <= 'Z') ? ch - 'A' : (ch >= 'a' && ch <= 'z') ? ch - 'a' + 26 : (ch >= '0' && ch <= '9') ? ch - '0' + 52 : ch == '+' ? 62 : ch == '/' ? 63 : -1;
        if (v < 0) continue;  // '=' padding, whitespace
        acc = (acc << 6) | (uint32_t)v; bits += 6;
        if (bits >= 8) { bits -= 8; out.push_back((char)((acc >> bits) & 0xffu)); }
    }
    return true;
}
bool LoadHDR(const std::string& path, std::vector<float>& rgba, uint32_t& width, uint32_t& height, std::string& error) {
    std::string f;
    if (!read_file(path, f)) { error = "cannot open " + path; return false; }
    size_t p = 0;
    auto line = [&](std::string& out) { out.clear(); while (p < f.size() && f[p] != '\n') out.push_back(f[p++]); if (p < f.size()) p++; return true; };
    std::string ln;
    line(ln);
    if (ln.rfind("#?RADIANCE", 0) != 0 && ln.rfind("#?RGBE", 0) != 0) { error = "not a Radiance HDR file: " + path; return false; }
    bool fmt = false;
    while (p < f.size()) { line(ln); if (ln.empty()) break; if (ln == "FORMAT=32-bit_rle_rgbe") fmt = true; }
    if (!fmt) { error = "unsupported HDR format (need 32-bit_rle_rgbe): " + path; return false; }
    line(ln);
    int h = 0, w = 0;
    if (sscanf(ln.c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0) { error = "unsupported HDR orientation (need -Y h +X w): " + path; return false; }
    width = (uint32_t)w; height = (uint32_t)h;
    rgba.assign((size_t)w * h * 4, 1.0f);
    std::vector<uint8_t> scan((size_t)w * 4);
    auto put = [&](int y) {
        for (int x = 0; x < w; x++) {
            const uint8_t* s = &scan[(size_t)x * 4]; float* d = &rgba[((size_t)y * w + x) * 4];
            if (s[3] == 0) { d[0] = d[1] = d[2] = 0.0f; }
            else { float sc = std::ldexp(1.0f, (int)s[3] - 136); d[0] = (float)s[0] * sc; d[1] = (float)s[1] * sc; d[2] = (float)s[2] * sc; }
        }
    };
    for (int y = 0; y < h; y++) {
        if (p + 4 > f.size()) { error = "truncated HDR: " + path; return false; }
        const uint8_t b0 = (uint8_t)f[p], b1 = (uint8_t)f[p + 1], b2 = (uint8_t)f[p + 2], b3 = (uint8_t)f[p + 3];
        if (w < 8 || w >= 32768 || b0 != 2 || b1 != 2 || (b2 & 0x80)) {  // flat scanline
            if (p + (size_t)w * 4 > f.size()) { error = "truncated HDR: " + path; return false; }
            std::memcpy(scan.data(), f.data() + p, (size_t)w * 4); p += (size_t)w * 4;
        } else {  // new RLE: the four components of the scanline stored one after another
            if ((((int)b2 << 8) | b3) != w) { error = "bad HDR scanline width: " + path; return false; }
            p += 4;
            for (int c = 0; c < 4; c++)
                for (int x = 0; x < w;) {
                    if (p >= f.size()) { error = "truncated HDR: " + path; return false; }
                    int n = (uint8_t)f[p++];
                    if (n > 128) {
                        n -= 128;
                        if (n == 0 || x + n > w || p >= f.size()) { error = "bad HDR run: " + path; return false; }
                        uint8_t v = (uint8_t)f[p++];
                        for (int k = 0; k < n; k++) scan[(size_t)(x++) * 4 + c] = v;
                    } else {
                        if (n == 0 || x + n > w || p + n > f.size()) { error = "bad HDR run: " + path; return false; }
                        for (int k = 0; k < n; k++) scan[(size_t)(x++) * 4 + c] = (uint8_t)f[p++];
                    }
                }
        }
        put(y);
    }
    return true;
}

bool SavePNG(const std::string& path, const uint8_t* rgba, uint32_t w, uint32_t h, std::string& error) {
    if (!rgba || w == 0 || h == 0) { error = "SavePNG: empty image"; return false; }
    std::vector<uint8_t> raw(((size_t)w * 4 + 1) * h);
    for (uint32_t y = 0; y < h; y++) { raw[((size_t)w * 4 + 1) * y] = 0; std::memcpy(&raw[((size_t)w * 4 + 1) * y + 1], rgba + (size_t)y * w * 4, (size_t)w * 4); }
    uLongf cl = compressBound((uLong)raw.size());
    std::vector<uint8_t> comp(cl);
    if (compress2(comp.data(), &cl, raw.data(), (uLong)raw.size(), 6) != Z_OK) { error = "SavePNG: deflate failed"; return false; }
    std::string out("\x89PNG\r\n\x1a\n", 8);
    auto chunk = [&](const char* type, const uint8_t* d, uint32_t n) {
        uint8_t len[4] = {(uint8_t)(n >> 24), (uint8_t)(n >> 16), (uint8_t)(n >> 8), (uint8_t)n};
        out.append((const char*)len, 4);
        std::string body(type, 4); body.append((const char*)d, n);
        out += body;
        uint32_t c = (uint32_t)crc32(0L, (const Bytef*)body.data(), (uInt)body.size());
        uint8_t cb[4] = {(uint8_t)(c >> 24), (uint8_t)(c >> 16), (uint8_t)(c >> 8), (uint8_t)c};
        out.append((const char*)cb, 4);
    };
    uint8_t ihdr[13] = {(uint8_t)(w >> 24), (uint8_t)(w >> 16), (uint8_t)(w >> 8), (uint8_t)w, (uint8_t)(h >> 24), (uint8_t)(h >> 16), (uint8_t)(h >> 8), (uint8_t)h, 8, 6, 0, 0, 0};
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", comp.data(), (uint32_t)cl);
    chunk("IEND", nullptr, 0);
    std::ofstream o(path, std::ios::binary);
    if (!o) { error = "cannot write " + path; return false; }
    o.write(out.data(), (std::streamsize)out.size());
    return (bool)o;
}

bool LoadLookupTables(const std::string& path, std::vector<float>& r, std::vector<float>& o, std::vector<float>& i, std::string& error) {
    std::string f;
    const size_t n0 = 64 * 64 * 32, n1 = 128 * 128 * 32;
    if (!read_file(path, f) || f.size() != (n0 + 2 * n1) * 4) { error = "bad lookup table file " + path; return false; }
    const float* p = reinterpret_cast<const float*>(f.data());
    r.assign(p, p + n0); o.assign(p + n0, p + n0 + n1); i.assign(p + n0 + n1, p + n0 + 2 * n1);
    return true;
}

bool ImportScene(const std::string& gltfPath, SceneAsset& sc, std::string& error) {
    std::string text, glbBin;
    if (!read_file(gltfPath, text)) { error = "cannot open " + gltfPath; return false; }
    if (text.size() >= 20 && text.compare(0, 4, "glTF") == 0) {  // binary container (.glb): 12-byte header, JSON chunk, optional BIN chunk
        auto le32 = [&](size_t o) { return (uint32_t)(uint8_t)text[o] | ((uint32_t)(uint8_t)text[o + 1] << 8) | ((uint32_t)(uint8_t)text[o + 2] << 16) | ((uint32_t)(uint8_t)text[o + 3] << 24); };
        if (le32(4) != 2u) { error = "unsupported .glb version in " + gltfPath; return false; }
        std::string json;
        for (size_t p = 12; p + 8 <= text.size();) {
            const uint32_t len = le32(p), type = le32(p + 4);
            if (p + 8 + len > text.size()) { error = "truncated .glb: " + gltfPath; return false; }
            if (type == 0x4E4F534Au) json = text.substr(p + 8, len);          // "JSON"
            else if (type == 0x004E4942u && glbBin.empty()) glbBin = text.substr(p + 8, len);  // "BIN\0"
            p += 8 + ((len + 3u) & ~3u);
        }
        if (json.empty()) { error = "no JSON chunk in " + gltfPath; return false; }
        text.swap(json);
    }
    g_malformed = false;
    JParser jp(text);
    JPtr root = jp.parse();
    if (!jp.ok || root->type != JVal::Obj) { error = "JSON parse error in " + gltfPath; return false; }
    const JVal& g = *root;
    const std::string base = dir_of(gltfPath);
    std::vector<std::string> bufs;
    if (const JVal* bs = g.get("buffers"))
        for (size_t i = 0; i < bs->size(); i++) {
            std::string d;
            const JVal* uri = (*bs)[i].get("uri");
            if (!uri) { if (i != 0 || glbBin.empty()) { error = "glTF buffer without uri"; return false; } d = glbBin; }  // the .glb BIN chunk
            else if (!decode_data_uri(uri->str, d) && !read_file(base + "/" + uri->str, d)) { error = "cannot read glTF buffer " + uri->str; return false; }
            bufs.push_back(std::move(d));
        }
    auto accessor = [&](size_t idx, std::vector<double>& out, int& ncomp) -> bool {
        const JVal& a = g.req("accessors")[idx];
        const JVal& bv = g.req("bufferViews")[a.index("bufferView", 0)];
        int ct = (int)a.number("componentType", 5126);
        const std::string& ty = a.req("type").str;
        ncomp = ty == "SCALAR" ? 1 : ty == "VEC2" ? 2 : ty == "VEC3" ? 3 : ty == "VEC4" ? 4 : ty == "MAT4" ? 16 : 0;
        const size_t bi = bv.index("buffer", 0);
        if (g_malformed || ncomp == 0 || bi >= bufs.size()) return false;
        const size_t count = a.index("count", 0);
        const size_t csz = ct == 5126 || ct == 5125 ? 4 : (ct == 5123 || ct == 5122 ? 2 : 1);
        const size_t off0 = bv.index("byteOffset", 0), off1 = a.index("byteOffset", 0);
        size_t stride = bv.index("byteStride", 0);
        if (!stride) stride = csz * ncomp;
        if (stride < csz * ncomp) return false;   // (glTF: byteStride >= the element size) — with it, count * element size <= buffer size bounds the allocation below
        const std::string& buf = bufs[bi];
        // the last element must end inside the buffer; 128-bit arithmetic, so no crafted count / stride can wrap the check
        const unsigned __int128 end = (unsigned __int128)off0 + off1 + (count ? (unsigned __int128)(count - 1) * stride + csz * ncomp : 0);
        if (count == (size_t)-1 || off0 == (size_t)-1 || off1 == (size_t)-1 || stride == (size_t)-1 || end > buf.size()) return false;
        const size_t off = off0 + off1;
        out.resize(count * ncomp);
        for (size_t k = 0; k < count; k++)
            for (int c = 0; c < ncomp; c++) {
                const char* p = buf.data() + off + k * stride + c * csz;
                double v;
                if (ct == 5126) { float x; std::memcpy(&x, p, 4); v = x; }
                else if (ct == 5125) { uint32_t x; std::memcpy(&x, p, 4); v = x; }
                else if (ct == 5123) { uint16_t x; std::memcpy(&x, p, 2); v = x; }
                else if (ct == 5122) { int16_t x; std::memcpy(&x, p, 2); v = x; }
                else if (ct == 5121) { v = (uint8_t)*p; }
                else { v = (int8_t)*p; }
                out[k * ncomp + c] = v;
            }
        return true;
    };

    // default textures in LoadDefaultTexture order (PathTracer.cpp:1557-1582): base, normal, roughness R8, metallic R8, emissive
    sc = SceneAsset();
    auto deftex = [&](std::vector<uint8_t> d, uint32_t ch) { TextureAsset t; t.Width = t.Height = 1; t.Channels = ch; t.Data = std::move(d); sc.Textures.push_back(t); };
    deftex({255, 255, 255, 255}, 4); deftex({128, 128, 255, 255}, 4); deftex({255}, 1); deftex({255}, 1); deftex({255, 255, 255, 255}, 4);
    std::map<std::string, uint32_t> texCache;
    auto textureIndex = [&](const JVal* ref, bool single, uint32_t& outIdx) -> bool {
        if (!ref) return true;
        const JVal& tex = g.req("textures")[ref->index("index", 0)];
        const JVal& img = g.req("images")[tex.index("source", 0)];
        if (g_malformed) { error = "glTF texture / image index out of range"; return false; }
        // image source: external file, data: URI, or a bufferView (the usual .glb form); PNG or JPEG, glTF's two core formats
        const JVal* iuri = img.get("uri");
        std::string key = (iuri ? iuri->str : "bufferView:" + std::to_string((long long)img.number("bufferView", -1))) + (single ? "#r" : "");
        auto it = texCache.find(key);
        if (it == texCache.end()) {
            TextureAsset t;
            std::string bytes;
            if (!iuri) {
                const JVal* bvs = g.get("bufferViews");
                const long long bi = (long long)img.number("bufferView", -1);
                if (!bvs || bi < 0 || (size_t)bi >= bvs->size()) { error = "glTF image without uri or bufferView"; return false; }
                const JVal& bv = (*bvs)[(size_t)bi];
                const size_t bidx = bv.index("buffer", 0), off = bv.index("byteOffset", 0), len = bv.index("byteLength", 0);
                if (bidx >= bufs.size() || off > bufs[bidx].size() || len > bufs[bidx].size() - off) { error = "glTF image bufferView out of range"; return false; }
                bytes = bufs[bidx].substr(off, len);
                if (!DecodeImage(bytes, key, t, error)) return false;
            } else if (decode_data_uri(iuri->str, bytes)) {
                if (!DecodeImage(bytes, "data URI", t, error)) return false;
            } else if (!LoadImage(base + "/" + iuri->str, t, error)) return false;
            if (single) {  // LoadTexture(..., onlySingleChannel=true) keeps R (PathTracer.cpp:826-836)
                std::vector<uint8_t> r((size_t)t.Width * t.Height);
                for (size_t i = 0; i < r.size(); i++) r[i] = t.Data[i * 4];
                t.Data = std::move(r); t.Channels = 1;
            }
            sc.Textures.push_back(std::move(t));
            it = texCache.emplace(key, (uint32_t)sc.Textures.size() - 1).first;
        }
        outIdx = it->second;
        return true;
    };

    if (const JVal* mats = g.get("materials"))
        for (size_t i = 0; i < mats->size(); i++) {
            const JVal& m = (*mats)[i];
            vpt_material pm; default_material(pm);
            const JVal* pbr = m.get("pbrMetallicRoughness");
            const JVal* ext = m.get("extensions");
            auto extObj = [&](const char* name) -> const JVal* { return ext ? ext->get(name) : nullptr; };
            if (pbr) {
                if (const JVal* bc = pbr->get("baseColorFactor")) for (int k = 0; k < 3; k++) pm.base_color[k] = (float)(*bc)[k].num;
                pm.metallic = (float)pbr->number("metallicFactor", 1.0);
                pm.roughness = (float)pbr->number("roughnessFactor", 1.0);
            } else { pm.metallic = 1.0f; }
            double strength = 1.0;
            if (const JVal* es = extObj("KHR_materials_emissive_strength")) strength = es->number("emissiveStrength", 1.0);
            if (const JVal* ef = m.get("emissiveFactor")) for (int k = 0; k < 3; k++) pm.emissive_color[k] = (float)((*ef)[k].num * strength);
            if (const JVal* e = extObj("KHR_materials_ior")) pm.ior = (float)e->number("ior", 1.5);
            if (const JVal* e = extObj("KHR_materials_transmission")) pm.transmission = (float)e->number("transmissionFactor", 0.0);
            if (const JVal* e = extObj("KHR_materials_specular")) if (const JVal* c = e->get("specularColorFactor")) for (int k = 0; k < 3; k++) pm.specular_color[k] = (float)(*c)[k].num;
            if (const JVal* e = extObj("KHR_materials_anisotropy")) {   // rotation: radians (glTF) -> degrees in [0, 360) (Editor.cpp:325)
                pm.anisotropy = (float)e->number("anisotropyStrength", 0.0);
                double deg = std::fmod(e->number("anisotropyRotation", 0.0) * (180.0 / 3.14159265358979323846), 360.0);
                if (deg < 0.0) deg += 360.0;
                if (deg == 0.0) deg = 0.0;   // no negative zero
                pm.anisotropy_rotation = (float)deg;
            }
            if (!textureIndex(pbr ? pbr->get("baseColorTexture") : nullptr, false, pm.base_color_texture)) return false;
            if (!textureIndex(m.get("normalTexture"), false, pm.normal_texture)) return false;
            if (!textureIndex(pbr ? pbr->get("metallicRoughnessTexture") : nullptr, true, pm.roughness_texture)) return false;
            if (!textureIndex(pbr ? pbr->get("metallicRoughnessTexture") : nullptr, true, pm.metallic_texture)) return false;
            if (!textureIndex(m.get("emissiveTexture"), false, pm.emissive_texture)) return false;
            sc.Materials.push_back(pm);
            const JVal* nm = m.get("name");
            sc.MaterialNames.push_back(nm ? nm->str : std::string());
        }
    if (sc.Materials.empty()) { vpt_material pm; default_material(pm); sc.Materials.push_back(pm); sc.MaterialNames.push_back("default"); }

    // meshes: one MeshAsset per primitive
    std::map<std::pair<int, int>, std::pair<uint32_t, uint32_t>> primMesh;
    if (const JVal* meshes = g.get("meshes"))
        for (size_t mi = 0; mi < meshes->size(); mi++) {
            const JVal& prims = (*meshes)[mi].req("primitives");
            for (size_t pi = 0; pi < prims.size(); pi++) {
                const JVal& p = prims[pi];
                const JVal& attr = p.req("attributes");
                std::vector<double> pos, nrm, uv, idx; int nc;
                if (!attr.has("POSITION") || !accessor(attr.index("POSITION", 0), pos, nc) || nc != 3) { error = "primitive without a readable POSITION accessor"; return false; }
                size_t nv = pos.size() / 3;
                if (attr.has("NORMAL") && !accessor(attr.index("NORMAL", 0), nrm, nc)) { error = "unreadable NORMAL accessor"; return false; }
                if (attr.has("TEXCOORD_0") && !accessor(attr.index("TEXCOORD_0", 0), uv, nc)) { error = "unreadable TEXCOORD_0 accessor"; return false; }
                MeshAsset me; me.Vertices.resize(nv);
                for (size_t v = 0; v < nv; v++) {
                    vpt_vertex& o = me.Vertices[v];
                    o.position[0] = (float)pos[v * 3]; o.position[1] = -(float)pos[v * 3 + 1]; o.position[2] = (float)pos[v * 3 + 2];
                    if (nrm.size() == nv * 3) { o.normal[0] = (float)nrm[v * 3]; o.normal[1] = -(float)nrm[v * 3 + 1]; o.normal[2] = (float)nrm[v * 3 + 2]; }
                    else { o.normal[0] = o.normal[1] = o.normal[2] = 0; }
                    if (uv.size() == nv * 2) { o.texcoord[0] = (float)uv[v * 2]; o.texcoord[1] = (float)uv[v * 2 + 1]; } else { o.texcoord[0] = o.texcoord[1] = 0; }
                }
                if (p.has("indices")) { if (!accessor(p.index("indices", 0), idx, nc)) { error = "unreadable index accessor"; return false; } } else { idx.resize(nv); for (size_t k = 0; k < nv; k++) idx[k] = (double)k; }
                me.Indices.resize(idx.size() / 3 * 3);
                for (size_t t = 0; t < idx.size(); t++) if (!(idx[t] >= 0.0 && idx[t] < (double)nv)) { error = "vertex index out of range"; return false; }
                for (size_t t = 0; t + 2 < idx.size(); t += 3) {  // mirror => swap the winding so geometric and vertex normals agree
                    me.Indices[t] = (uint32_t)idx[t]; me.Indices[t + 1] = (uint32_t)idx[t + 2]; me.Indices[t + 2] = (uint32_t)idx[t + 1];
                }
                sc.Meshes.push_back(std::move(me));
                primMesh[{(int)mi, (int)pi}] = {(uint32_t)sc.Meshes.size() - 1, (uint32_t)std::min<size_t>(p.index("material", 0), 0xffffffffu)};
            }
        }

    // node hierarchy
    struct Walk { size_t node; double M[4][4]; };
    std::vector<Walk> stack;
    const JVal* nodes = g.get("nodes");
    const JVal* scenes = g.get("scenes");
    std::vector<char> visited(nodes ? nodes->size() : 0, 0);   // a node has one parent in valid glTF: seeing one twice means a cycle or a DAG
    if (nodes && scenes && scenes->size()) {
        const JVal& roots = (*scenes)[g.index("scene", 0)].req("nodes");
        for (size_t i = roots.size(); i-- > 0;) { Walk w; w.node = roots[i].as_index(); for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) w.M[a][b] = a == b; stack.push_back(w); }
    }
    while (!stack.empty()) {
        Walk w = stack.back(); stack.pop_back();
        if (w.node >= visited.size() || visited[w.node]) { error = "glTF node index out of range or node reachable twice"; return false; }
        visited[w.node] = 1;
        const JVal& n = (*nodes)[w.node];
        double L[4][4], M[4][4];
        node_matrix(n, L); matmul(w.M, L, M);
        double F[4][4]; std::memcpy(F, M, sizeof(F)); flip_y(F);
        if (n.has("mesh")) {
            const size_t mi_ = n.index("mesh", 0);
            const JVal& prims = g.req("meshes")[mi_].req("primitives");
            if (g_malformed) { error = "glTF node references a missing mesh"; return false; }
            int mi = (int)mi_;
            for (size_t pi = 0; pi < prims.size(); pi++) {
                auto pm = primMesh[{mi, (int)pi}];
                MeshInstance inst; inst.MeshIndex = pm.first; inst.MaterialIndex = pm.second; inst.Transform = from_rows(F);
                sc.MeshInstances.push_back(inst);
            }
        }
        if (n.has("camera") && sc.Cameras.empty()) {
            const JVal& c = g.req("cameras")[n.index("camera", 0)];
            CameraAsset cam;
            if (const JVal* p = c.get("perspective")) { cam.AspectRatio = (float)p->number("aspectRatio", 16.0 / 9.0); cam.FOV = degrees((float)p->number("yfov", 0.7853981633974483)); }
            cam.ViewMatrix = inverse(from_rows(F));
            sc.Cameras.push_back(cam);
        }
        if (const JVal* ch = n.get("children"))
            for (size_t i = ch->size(); i-- > 0;) { Walk c; c.node = (*ch)[i].as_index(); std::memcpy(c.M, M, sizeof(M)); stack.push_back(c); }
    }
    if (g_malformed) { error = "malformed glTF: an index or a required member is missing or out of range"; return false; }
    if (sc.Meshes.empty()) { error = "No meshes found in scene"; return false; }  // PathTracer.cpp:180
    return true;
}

}  // namespace vpthost
