// wave.hpp — wave64 building blocks shared by the stage kernels: lane id, ballot prefix, one-atomic-per-wave append,
// and the persistent work-fetch chunk size.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Region markers for the ISA budget (tests/tools/isa_budget.py attributes every instruction between two markers to a step of the
// vote loop).  A marker is a comment in the assembly — no instruction — and it is part of the product build on purpose: a build
// with markers only for measuring gets a different register allocation than the one that ships.
#define VPT_MARK(name) asm volatile("; VPT_MARK " name)

namespace vpt {

__device__ inline uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ inline uint32_t lanes_below(unsigned long long mask) {  // popcount of mask bits below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
// One atomic per wave: returns this lane's slot in the output stream (valid where pred).
__device__ inline uint32_t wave_append(bool pred, uint32_t* counter) {
    unsigned long long mask = __ballot(pred);
    uint32_t total = (uint32_t)__popcll(mask);
    uint32_t base = 0;
    if (total) {
        uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
        if (lane_id() == leader) base = atomicAdd(counter, total);
        base = __shfl(base, (int)leader);
    }
    return base + lanes_below(mask);
}

// One atomic on a single word costs ~11 ns under contention (MI355X_MICROARCH.md 'dequeue': a head word
// saturates at ~88 dequeues/us), so the number of fetches per launch is kept near 8k: a wave takes
// n/8192 queue entries per fetch, rounded up to whole waves, between 64 and 1024.
__device__ inline uint32_t fetch_chunk(uint32_t n) {
    uint32_t c = ((n >> 13) + 63u) & ~63u;
    return c < 64u ? 64u : (c > 1024u ? 1024u : c);
}

// Stream records are written once and read once, a bounce later: stores and loads that bypass cache retention leave L2 to the
// scene data (BVH nodes, triangles, textures) the kernels gather.
__device__ __forceinline__ void st_stream(float4* p, float4 v) {
    __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y); __builtin_nontemporal_store(v.z, &p->z); __builtin_nontemporal_store(v.w, &p->w);
}
__device__ __forceinline__ float4 ld_stream(const float4* p) {
    return make_float4(__builtin_nontemporal_load(&p->x), __builtin_nontemporal_load(&p->y), __builtin_nontemporal_load(&p->z), __builtin_nontemporal_load(&p->w));
}

}  // namespace vpt
