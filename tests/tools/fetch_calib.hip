// Calibration probe (not part of the product): what does rocprofv3's FETCH_SIZE read for the access patterns of the path kernels?
// MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide coalesced streaming read and is uncalibrated
// for other patterns.  Kernels over a 1 GiB float4 array (64 Mi entries, far beyond L2 + Infinity Cache):
//   k_stream   : lane i reads entry i                              (the record streams of extend / shade / fused phase 1)
//   k_subset80 : lane i reads entry idx[i], idx = every entry kept with probability 0.8, in order   (regrouped hits: fused phase 2)
//   k_subset50 : the same with probability 0.5
//   k_random   : idx = a random permutation                         (slot-addressed records: pathLight in join, frame sums)
// Run under  rocprofv3 --pmc FETCH_SIZE  and divide FETCH_SIZE x 1024 by the bytes each kernel asks for (printed).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <random>
__global__ __launch_bounds__(256) void k_stream(const float4* a, float* out, uint32_t n) {
    float s = 0.0f;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) { float4 v = a[i]; s += v.x + v.w; }
    if (s == 1234.5f) out[0] = s;
}
__global__ __launch_bounds__(256) void k_gather(const float4* a, const uint32_t* idx, float* out, uint32_t m) {
    float s = 0.0f;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < m; i += gridDim.x * 256u) { float4 v = a[idx[i]]; s += v.x + v.w; }
    if (s == 1234.5f) out[0] = s;
}
int main() {
    const uint32_t n = 64u << 20;
    float4* a; float* out; uint32_t* d_idx;
    (void)hipMalloc(&a, (size_t)n * 16); (void)hipMemset(a, 0, (size_t)n * 16); (void)hipMalloc(&out, 4); (void)hipMalloc(&d_idx, (size_t)n * 4);
    std::mt19937 rng(7);
    auto run = [&](const char* name, const std::vector<uint32_t>& idx) {
        (void)hipMemcpy(d_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice);
        k_gather<<<4096, 256>>>(a, d_idx, out, (uint32_t)idx.size());
        (void)hipDeviceSynchronize();
        printf("%s: %zu entries, %zu bytes of records + %zu bytes of indices\n", name, idx.size(), idx.size() * 16, idx.size() * 4);
    };
    k_stream<<<4096, 256>>>(a, out, n); (void)hipDeviceSynchronize();
    printf("k_stream: %u entries, %zu bytes\n", n, (size_t)n * 16);
    for (double keep : {0.8, 0.5}) {
        std::vector<uint32_t> idx; idx.reserve(n);
        std::uniform_real_distribution<double> u(0, 1);
        for (uint32_t i = 0; i < n; i++) if (u(rng) < keep) idx.push_back(i);
        run(keep > 0.6 ? "k_gather subset80" : "k_gather subset50", idx);
    }
    { std::vector<uint32_t> idx(n / 4); for (uint32_t i = 0; i < n / 4; i++) idx[i] = (uint32_t)(rng() % n); run("k_gather random", idx); }
    return 0;
}
