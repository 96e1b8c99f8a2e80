// gather_calib — how fast can a CU fetch 64-byte BVH nodes that every lane picks for itself?  (Test utility, not linked by the product.)
//
// The vote-scheduled traversal kernels fetch one 64-byte node per lane and step with four loads (3 x dwordx4 + 1 x dwordx2), every
// lane at an address of its own.  This measures what that access pattern costs in the vector memory pipeline, detached from
// the traversal arithmetic, for four ways of getting the same 64 bytes per lane out of an L2-resident array:
//   A  per-lane: 4 x global_load_dwordx4 from the lane's own node                     (what the kernels do)
//   B  per-lane, one load: 1 x global_load_dwordx4 from the lane's own node           (cost of ONE scattered load instruction)
//   C  quad-cooperative: in load j the four lanes of a quad read the four 16-byte pieces of the node of the quad's lane j — 64
//      contiguous bytes per quad per instruction — straight into LDS (global_load_lds_dwordx4, lane-linear), then every lane
//      reads its own node back with 4 x ds_read_b128
//   D  like C but into registers (no transposition: only the memory side is measured)
//   E  like A, but 3 of 8 fetches go to one of 96 "top of the tree" nodes that the block keeps in LDS, through ONE generic pointer
//      per lane (flat_load_dwordx4: the lane's address decides between LDS and the vector cache)
//   F  like E with every fetch from global memory (the same index distribution: what E is to be compared with)
// Every variant chases: the next node index comes out of the loaded data, as in a traversal.
//   hipcc --offload-arch=gfx950 -O3 -o gather_calib gather_calib.hip && ./gather_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int kBlock = 256, kIters = 256;

__device__ __forceinline__ uint32_t mix(uint32_t a, uint4 v) { return (a * 747796405u + 2891336453u) ^ v.x ^ v.y ^ v.z ^ v.w; }

template <int MODE>
__global__ __launch_bounds__(kBlock, 8) void k_gather(const uint4* nodes, uint32_t mask, uint32_t* out) {
    __shared__ uint4 stage[MODE == 2 ? kBlock * 4 : MODE == 4 ? 96 * 4 : 1];
    if (MODE == 4) { for (uint32_t i = threadIdx.x; i < 96u * 4u; i += kBlock) stage[i] = nodes[i]; __syncthreads(); }
    uint32_t idx = (blockIdx.x * kBlock + threadIdx.x) * 2654435761u;
    uint32_t acc = 0;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (int it = 0; it < kIters; it++) {
        const uint32_t node = idx & mask;
        if (MODE == 0) {
            const uint4* p = nodes + (size_t)node * 4;
            const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc = mix(mix(mix(mix(acc, a), b), c), d);
        } else if (MODE == 1) {
            const uint4 a = nodes[(size_t)node * 4];
            acc = mix(acc, a);
        } else if (MODE == 4 || MODE == 5) {
            const bool top = ((idx >> 28) & 7u) < 3u;
            const uint32_t nd = top ? (idx >> 8) % 96u : node;
            const uint4* p = (MODE == 4 && top) ? (const uint4*)&stage[nd * 4] : nodes + (size_t)nd * 4;
            const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc = mix(mix(mix(mix(acc, a), b), c), d);
        } else {
            uint4 r[4];
#define COOP(J)                                                                                                                          \
            {                                                                                                                            \
                const uint32_t nj = (uint32_t)__builtin_amdgcn_mov_dpp((int)node, (J) * 0x55, 0xf, 0xf, true); /* quad_perm:[J,J,J,J] */  \
                const uint4* p = nodes + (size_t)nj * 4 + (lane & 3u);                                                                   \
                if (MODE == 2) /* lane-linear destination: the wave's slab for load J, 16 bytes per lane */                             \
                    __builtin_amdgcn_global_load_lds(p, (__attribute__((address_space(3))) void*)&stage[(wave * 4 + (J)) * 64], 16, 0, 0); \
                else r[J] = *p;                                                                                                          \
            }
            COOP(0) COOP(1) COOP(2) COOP(3)
#undef COOP
            if (MODE == 2) {
                __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0): the LDS-direct loads have landed
                __builtin_amdgcn_wave_barrier();
                // the lane's own node: load j = (lane & 3) of its quad, pieces 0..3 sit in the quad's four lanes
                const uint4* s = &stage[(wave * 4 + (lane & 3u)) * 64 + (lane & ~3u)];
                const uint4 a = s[0], b = s[1], c = s[2], d = s[3];
                acc = mix(mix(mix(mix(acc, a), b), c), d);
                __builtin_amdgcn_wave_barrier();
            } else {
                acc = mix(mix(mix(mix(acc, r[0]), r[1]), r[2]), r[3]);
            }
        }
        idx = acc;
    }
    out[blockIdx.x * kBlock + threadIdx.x] = acc;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate / 1e6;
    printf("device %s, %d CUs, %.2f GHz\n", prop.name, cus, ghz);
    for (uint32_t log2_nodes : {14u, 18u}) {   // 1 MB (L2 of one XCD holds it many times over), 16 MB, 256 MB of 64-byte nodes
        const uint32_t n = 1u << log2_nodes;
        std::vector<uint32_t> h((size_t)n * 16);
        uint32_t s = 12345u;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s; }
        uint4* d = nullptr; uint32_t* out = nullptr;
        CK(hipMalloc(&d, (size_t)n * 64)); CK(hipMemcpy(d, h.data(), (size_t)n * 64, hipMemcpyHostToDevice));
        const int blocks = cus * 8;
        CK(hipMalloc(&out, (size_t)blocks * kBlock * 4));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const char* names[6] = {"A per-lane 4 x dwordx4", "B per-lane 1 x dwordx4", "C quad-coop -> LDS -> ds_read", "D quad-coop -> registers", "E 3/8 from LDS via flat", "F 3/8 to 96 hot nodes, global"};
        for (int mode = 0; mode < 6; mode++) {
            float best = 1e30f;
            for (int rep = 0; rep < 4; rep++) {
                CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k_gather<0>, dim3(blocks), dim3(kBlock), 0, 0, d, n - 1, out);
                else if (mode == 1) hipLaunchKernelGGL(k_gather<1>, dim3(blocks), dim3(kBlock), 0, 0, d, n - 1, out);
                else if (mode == 2) hipLaunchKernelGGL(k_gather<2>, dim3(blocks), dim3(kBlock), 0, 0, d, n - 1, out);
                else if (mode == 3) hipLaunchKernelGGL(k_gather<3>, dim3(blocks), dim3(kBlock), 0, 0, d, n - 1, out);
                else if (mode == 4) hipLaunchKernelGGL(k_gather<4>, dim3(blocks), dim3(kBlock), 0, 0, d, n - 1, out);
                else hipLaunchKernelGGL(k_gather<5>, dim3(blocks), dim3(kBlock), 0, 0, d, n - 1, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            const double fetches = (double)blocks * kBlock * kIters;          // lane-level node fetches
            const double per_cu_clk = fetches / cus / (best * 1e-3 * ghz * 1e9);
            printf("nodes %8u (%4u MB)  %-32s %8.3f ms  %6.2f Gnode/s  %.3f nodes/clk/CU  (%.1f clk per 64-lane fetch per CU)\n", n, n / 16384, names[mode], best,
                   fetches / best / 1e6, per_cu_clk, 64.0 / per_cu_clk);
        }
        CK(hipFree(d)); CK(hipFree(out));
    }
    return 0;
}
