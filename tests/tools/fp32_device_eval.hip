// fp32_device_eval — the DEVICE compile of include/vpt_fp32.h, one leaf function at a time (test utility, not part of the product).
// HIP-vs-oracle parity cannot see a difference between the two compiles of the shared header except through whole renders; this tool runs the
// header's functions on the GPU on arrays of inputs so that tests/test_gpu_fp32_device.py can compare them BIT FOR BIT with the host compile
// (oracle/oracle.cpp orc_fp32_eval / orc_leaf_eval, same function ids), which tests/test_fp32_contract.py in turn holds against float64.
// Built by the test with the product's flags (vulkan-path-tracer_amd/_build.py FLAGS: -ffp-contract=off, no fast-math, correctly rounded div / sqrt).
//   fp32_device_eval elem <fn> <n> x.bin y.bin out.bin        fn: 0 sin 1 cos 2 log 3 exp 4 asin 5 acos 6 atan2 7 pow 8 sqrt 9 division 10 pcg_hash->unit float
//   fp32_device_eval leaf <fn> <n> <nin> <nout> in.bin out.bin  fn as orc_leaf_eval
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/vpt_fp32.h"
using namespace vptfp;

__global__ void k_elem(int fn, const float* x, const float* y, float* out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    switch (fn) {
        case 0: out[i] = sin_(x[i]); break;
        case 1: out[i] = cos_(x[i]); break;
        case 2: out[i] = log_(x[i]); break;
        case 3: out[i] = exp_(x[i]); break;
        case 4: out[i] = asin_(x[i]); break;
        case 5: out[i] = acos_(x[i]); break;
        case 6: out[i] = atan2_(x[i], y[i]); break;
        case 7: out[i] = pow_(x[i], y[i]); break;
        case 8: out[i] = sqrt_(x[i]); break;
        case 9: out[i] = x[i] / y[i]; break;
        default: out[i] = u32_to_unit(pcg_hash(f2u(x[i]))); break;
    }
}
__global__ void k_leaf(int fn, const float* in, float* out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    switch (fn) {
        case 0: {
            const float* a = in + (size_t)i * 17; float t = 0, u = 0, v = 0;
            bool h = ray_triangle(v3(a[0], a[1], a[2]), v3(a[3], a[4], a[5]), v3(a[6], a[7], a[8]), v3(a[9], a[10], a[11]), v3(a[12], a[13], a[14]), a[15], a[16], &t, &u, &v);
            out[i * 4] = h ? 1.0f : 0.0f; out[i * 4 + 1] = t; out[i * 4 + 2] = u; out[i * 4 + 3] = v; break;
        }
        case 1: { const float* a = in + (size_t)i * 3; int i0, i1; float w; texel_coords(a[0], (int)a[1], a[2] != 0.0f, &i0, &i1, &w); out[i * 3] = (float)i0; out[i * 3 + 1] = (float)i1; out[i * 3 + 2] = w; break; }
        case 2: out[i] = (float)lut_layer(in[i * 2], (int)in[i * 2 + 1]); break;
        case 3: { const float* a = in + (size_t)i * 7; V3 r = refract(v3(a[0], a[1], a[2]), v3(a[3], a[4], a[5]), a[6]); out[i * 3] = r.x; out[i * 3 + 1] = r.y; out[i * 3 + 2] = r.z; break; }
        case 4: out[i] = smoothstep(in[i * 3], in[i * 3 + 1], in[i * 3 + 2]); break;
        case 5: { const float* a = in + (size_t)i * 6; V3 r = reflect(v3(a[0], a[1], a[2]), v3(a[3], a[4], a[5])); out[i * 3] = r.x; out[i * 3 + 1] = r.y; out[i * 3 + 2] = r.z; break; }
        case 6: { const float* a = in + (size_t)i * 3; V3 r = normalize(v3(a[0], a[1], a[2])); out[i * 3] = r.x; out[i * 3 + 1] = r.y; out[i * 3 + 2] = r.z; break; }
        case 7: out[i] = (float)unorm8(in[i]); break;
        case 8: { const float* a = in + (size_t)i * 16; out[i] = hit_is_local(v3(a[0], a[1], a[2]), v3(a[3], a[4], a[5]), v3(a[6], a[7], a[8]), v3(a[9], a[10], a[11]), v3(a[12], a[13], a[14]), a[15]) ? 1.0f : 0.0f; break; }
        case 10: out[i] = unorm8_to_float((uint32_t)in[i]); break;
        default: { const float* a = in + (size_t)i * 6; out[i] = triangle_degenerate(v3(a[0], a[1], a[2]), v3(a[3], a[4], a[5])) ? 1.0f : 0.0f; break; }
    }
}

static std::vector<float> read_floats(const char* path, size_t n) {
    std::vector<float> v(n);
    FILE* f = fopen(path, "rb");
    if (!f || fread(v.data(), 4, n, f) != n) { fprintf(stderr, "cannot read %zu floats from %s\n", n, path); exit(3); }
    fclose(f);
    return v;
}
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 4; } } while (0)

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const bool elem = !strcmp(argv[1], "elem");
    if ((elem && argc != 7) || (!elem && argc != 8)) return 2;
    const int fn = atoi(argv[2]);
    const uint32_t n = (uint32_t)atol(argv[3]);
    const size_t nin = elem ? 1 : (size_t)atoi(argv[4]), nout = elem ? 1 : (size_t)atoi(argv[5]);
    const char* out_path = elem ? argv[6] : argv[7];
    std::vector<float> x = read_floats(elem ? argv[4] : argv[6], n * nin), y;
    if (elem) y = read_floats(argv[5], n);
    float *dx = nullptr, *dy = nullptr, *dout = nullptr;
    CHECK(hipMalloc((void**)&dx, x.size() * 4)); CHECK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    if (elem) { CHECK(hipMalloc((void**)&dy, y.size() * 4)); CHECK(hipMemcpy(dy, y.data(), y.size() * 4, hipMemcpyHostToDevice)); }
    CHECK(hipMalloc((void**)&dout, (size_t)n * nout * 4));
    if (elem) hipLaunchKernelGGL(k_elem, dim3((n + 255) / 256), dim3(256), 0, 0, fn, dx, dy, dout, n);
    else hipLaunchKernelGGL(k_leaf, dim3((n + 255) / 256), dim3(256), 0, 0, fn, dx, dout, n);
    CHECK(hipDeviceSynchronize());
    std::vector<float> out((size_t)n * nout);
    CHECK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
    FILE* f = fopen(out_path, "wb");
    if (!f || fwrite(out.data(), 4, out.size(), f) != out.size()) return 5;
    fclose(f);
    return 0;
}
