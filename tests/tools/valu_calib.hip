// Calibration probe (not part of the product): what do rocprofv3's SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU / GRBM_GUI_ACTIVE read on a
// kernel whose VALU issue rate is known?  Three kernels of independent chains, 8 waves per SIMD, 2048 blocks x 256 threads:
//   fma: 16 independent v_fma_f32 chains;  rcp: 16 independent v_rcp_f32 chains (transcendental rate);  mix: fma + int ops.
// Prints instructions per SIMD-cycle from HIP-event timing at the device's reported clock; run it under
//   rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -- ./valu_calib
// and compare VALUBusy = SQ_ACTIVE_INST_VALU / CU_NUM / GRBM_GUI_ACTIVE(per XCD) with the known rate.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int kIters = 4096;
__global__ __launch_bounds__(256) void k_fma(float* out, float a, float b) {
    float x[16];
    for (int i = 0; i < 16; i++) x[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < kIters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) x[i] = __builtin_fmaf(x[i], a, b);
    }
    float s = 0; for (int i = 0; i < 16; i++) s += x[i];
    if (s == 12345.678f) out[0] = s;
}
__global__ __launch_bounds__(256) void k_rcp(float* out, float a) {
    float x[16];
    for (int i = 0; i < 16; i++) x[i] = threadIdx.x * 1e-3f + i + a;
    for (int it = 0; it < kIters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) x[i] = __builtin_amdgcn_rcpf(x[i]);
    }
    float s = 0; for (int i = 0; i < 16; i++) s += x[i];
    if (s == 12345.678f) out[0] = s;
}
__global__ __launch_bounds__(256) void k_mix(float* out, float a, float b, unsigned m) {
    float x[8]; unsigned u[8];
    for (int i = 0; i < 8; i++) { x[i] = threadIdx.x * 1e-3f + i; u[i] = threadIdx.x * 7u + i; }
    for (int it = 0; it < kIters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) { x[i] = __builtin_fmaf(x[i], a, b); u[i] = (u[i] ^ m) + (u[i] >> 3); }
    }
    float s = 0; unsigned t = 0; for (int i = 0; i < 8; i++) { s += x[i]; t += u[i]; }
    if (s == 12345.678f || t == 0x12345u) out[0] = s + t;
}
int main() {
    float* d; (void)hipMalloc(&d, 4);
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const double clock_hz = p.clockRate * 1e3, simds = p.multiProcessorCount * 4.0;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = p.multiProcessorCount * 8;   // 8 blocks of 4 waves per CU = 8 waves per SIMD
    const double waves = blocks * 4.0;
    for (int which = 0; which < 3; which++) {
        for (int rep = 0; rep < 3; rep++) {
            (void)hipEventRecord(e0);
            if (which == 0) k_fma<<<blocks, 256>>>(d, 1.0001f, 0.5f);
            if (which == 1) k_rcp<<<blocks, 256>>>(d, 1.5f);
            if (which == 2) k_mix<<<blocks, 256>>>(d, 1.0001f, 0.5f, 0x9e3779b9u);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double per_wave = which == 2 ? kIters * 8.0 * 4.0 : kIters * 16.0;   // mix: fma + xor + shift + add per element
            if (rep == 2)
                printf("%s: %.3f ms, %.3g wave-instructions, %.3f cycles per wave-instruction per SIMD at %.0f MHz (%d CUs)\n",
                       which == 0 ? "fma" : which == 1 ? "rcp" : "mix", ms, per_wave * waves, ms * 1e-3 * clock_hz * simds / (per_wave * waves), clock_hz / 1e6, p.multiProcessorCount);
        }
    }
    return 0;
}
