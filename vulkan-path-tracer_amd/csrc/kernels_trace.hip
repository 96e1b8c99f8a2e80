// kernels_trace.hip — ray-stream traversal kernels (replace the driver-side TraceRay / RayQuery of RayGen.slang:90 and
// RTCommon.slang:54-63 for ray streams that live in HBM) and the trace lab that times them on identical rays.
//
//   k_trace_base   one ray per lane, a wave works through 64 rays at a time until the slowest lane is done
//                  (the structure of round 1's extend kernel, kept as the measured baseline)
//   k_trace_vote   persistent lanes with a wave-level vote: every iteration the wave executes ONE kind of step for the
//                  lanes that want it — an inner-node step (fetch 64 B, four slab tests, order, push), a triangle step
//                  (ONE triangle of the lane's current leaf) or a fetch step (retire finished rays, load new ones into
//                  the idle lanes) — chosen by ballot/popcount as the kind most lanes are waiting for.  A lane that
//                  reached a leaf no longer drags the whole wave through the triangle code while its neighbours sit at
//                  inner nodes, a leaf of four triangles no longer makes one-triangle lanes wait, and a finished lane is
//                  refilled instead of idling until the slowest ray of its 64 is done.  Results are per ray, so the
//                  order rays are worked on cannot change a bit of them.
//
// Closest-hit search: ties in t go to the smaller global triangle id (traverse.hpp), so every kernel here returns the
// same (t, u, v, primitive, instance) for a ray whatever it visits first.
#include "kernels.hpp"
#include "traverse.hpp"
#include "wave.hpp"
#include "vote.hpp"

namespace vpt {

namespace {

__device__ inline V3 xyz4(float4 v) { return vptfp::v3(v.x, v.y, v.z); }

__device__ inline void load_ray(const TraceArgs& a, uint32_t i, uint32_t& rid, V3& o, V3& d) {
    rid = a.order ? a.order[i] : i;
    o = xyz4(a.ro[rid]);
    d = xyz4(a.rd[rid]);
    if (a.normalize_dir) d = vptfp::normalize(d);  // RayGen.slang:70
}

// hit record of a finished closest-hit search: prim / inst come from the winning triangle's record; returns the instance
__device__ inline uint32_t store_closest(const TraceArgs& a, const BvhTri* tris, uint32_t rid, bool found, float t, float u, float v, uint32_t slot) {
    uint32_t prim = 0xffffffffu, inst = 0xffffffffu;
    if (found) { prim = a.store_gid ? tris[slot].gid : tris[slot].prim; inst = tris[slot].inst; }
    st_stream(&a.hit[rid], make_float4(found ? t : -1.0f, found ? u : 0.0f, found ? v : 0.0f, __uint_as_float(prim)));
    a.hinst[rid] = inst;
    return inst;
}
}  // namespace

// ------------------------------------------------------------------ baseline: 64 rays per wave at a time
template <bool ANY, bool COUNT>
__global__ __launch_bounds__(kTraverseBlock, 8) void k_trace_base(DeviceScene sc, TraceArgs a, Counters* ctr) {
    extern __shared__ __align__(16) unsigned char smem[];
    const TravStack stack = make_stack(smem, sc.stack_overflow);
    GlobalSceneSrc src{sc.nodes, sc.tris, false};
    const uint32_t n = a.n_dev ? *a.n_dev : a.n;
    const uint32_t chunk = fetch_chunk(n);
    TravStats st; st.nodes = 0; st.tris = 0;
    while (true) {
        uint32_t base = 0;
        if (lane_id() == 0) base = atomicAdd(a.head, chunk);
        base = __shfl(base, 0);
        if (base >= n) break;
        for (uint32_t k = 0; k < chunk; k += 64) {
            uint32_t i = base + k + lane_id();
            if (i >= n) break;
            uint32_t rid; V3 o, d;
            load_ray(a, i, rid, o, d);
            if (ANY) {
                int slot = 0; float t = 0.0f; uint32_t gid = 0;
                bool occ = trace_occluded_pass<COUNT, false, false>(src, o, d, a.tmin, a.tmax, 0.0f, 0u, stack, st, 0xffffffffu, 0xffffffffu, slot, t, gid);
                a.hit[rid] = make_float4(occ ? 1.0f : -1.0f, 0.0f, 0.0f, 0.0f);
            } else {
                HitRec h;
                bool found = trace_closest_pass<COUNT, false>(src, o, d, a.tmin, a.tmax, stack, h, st, 0xffffffffu, 0xffffffffu);
                a.hit[rid] = make_float4(found ? h.t : -1.0f, found ? h.u : 0.0f, found ? h.v : 0.0f, __uint_as_float(a.store_gid ? h.gid : h.prim));
                a.hinst[rid] = h.inst;
            }
        }
    }
    if (COUNT) {
        atomicAdd(&ctr->stat_nodes, (unsigned long long)st.nodes);
        atomicAdd(&ctr->stat_tris, (unsigned long long)st.tris);
    }
}

// ------------------------------------------------------------------ vote-scheduled persistent lanes
// Lane state: `cur` >= 0 inner node to visit; < 0 leaf code ~(first << 3 | count - 1) with `first` advancing as the
// triangles are consumed; kLaneDone / kLaneIdle.  The wave owns a chunk [w_next, w_end) of the stream (one atomic per
// chunk) and deals its entries to idle lanes in fetch steps.
template <bool ANY, bool COUNT, bool WIDE8>
__global__ __launch_bounds__(kTraverseBlock, 8) void k_trace_vote(DeviceScene sc, TraceArgs a, Counters* ctr) {
    extern __shared__ __align__(16) unsigned char smem[];
    const LaneStack S = make_lane_stack(smem, sc.stack_overflow);
    const BvhNode* const nodes = sc.nodes;
    const BvhTri* const tris = sc.tris;
    const uint32_t n = a.n_dev ? *a.n_dev : a.n;
    const uint32_t chunk = fetch_chunk(n);
    const uint32_t fetch_at = (a.param & 0xffu) ? (a.param & 0xffu) : 16u;  // idle lanes that trigger a fetch step (64: only when all are idle)
    const bool weighted = ((a.param >> 8) & 1u) != 0u;  // vote by lanes served per instruction issued: a triangle step costs about half a node step
    // every wave starts on its own 64 entries without an atomic (8192 waves fetching at once would queue ~90 us on the cursor);
    // entries beyond the grid's static part are fetched chunk-wise through the cursor
    const uint32_t n_static = gridDim.x * (kTraverseBlock / 64u) * 64u;
    uint32_t w_next = (blockIdx.x * (kTraverseBlock / 64u) + (threadIdx.x >> 6)) * 64u, w_end = w_next + 64u < n ? w_next + 64u : n;
    if (w_next >= n) { w_next = 0u; w_end = 0u; }
    bool exhausted = false;
    int cur = kLaneIdle, sp = 0;
    uint32_t rid = 0, bslot = 0xffffffffu, bgid = 0xffffffffu;
    V3 o = vptfp::v3(0.0f, 0.0f, 0.0f), d = o, inv = o;
    float best_t = 0.0f, bu = 0.0f, bv = 0.0f;
    uint32_t st_nodes = 0, st_tris = 0;
    uint32_t qi = 0u;  // the ray's position in the queue: where its shade class goes (a.cls)
    // Vote loop: one kind of step per iteration.  (Giving each kind its own inner loop, which lets the lane state stay in fixed
    // registers across the back edge, was measured 5-7 % slower on closest-hit rays: the vote then sticks to a kind for too long.)
    while (true) {
        const bool busy = cur < kLaneDone;
        const bool at_node = busy && cur >= 0;
        const bool at_leaf = busy && cur < 0;
        const uint32_t nn = (uint32_t)__popcll(__ballot(at_node)), nl = (uint32_t)__popcll(__ballot(at_leaf));
        if (!exhausted && (64u - nn - nl >= fetch_at || nn + nl == 0u)) {
            // ---- fetch step: retire finished rays, deal new ones to the idle lanes
            if (cur == kLaneDone) {
                if (ANY) a.hit[rid] = make_float4(bslot != 0xffffffffu ? 1.0f : -1.0f, 0.0f, 0.0f, 0.0f);
                else {
                    const uint32_t inst = store_closest(a, tris, rid, bslot != 0xffffffffu, best_t, bu, bv, bslot);
                    if (a.cls) a.cls[qi] = bslot != 0xffffffffu ? sc.inst_class[inst] : (unsigned char)kShadeMiss;  // the shade-queue sort key
                }
                cur = kLaneIdle;
            }
            if (w_next >= w_end) {
                if (n_static >= n) exhausted = true;
                else {
                    uint32_t base = 0;
                    if (lane_id() == 0) base = atomicAdd(a.head, chunk);
                    base = n_static + __builtin_amdgcn_readfirstlane(base);
                    if (base >= n) exhausted = true;
                    else { w_next = base; w_end = base + chunk < n ? base + chunk : n; }
                }
            }
            if (!exhausted) {
                const unsigned long long m_idle = __ballot(cur == kLaneIdle);
                const uint32_t i = w_next + lanes_below(m_idle);
                if (cur == kLaneIdle && i < w_end) {
                    rid = a.order ? a.order[i] : (a.valid && a.valid[i] == kHole) ? kHole : i;
                    qi = i;
                    if (rid == kHole) { if (a.cls) a.cls[i] = 0xffu; }   // a hole has no class: the classify step drops it
                    if (rid != kHole) {  // a hole: the tail of some wave's last chunk of the queue (vote.hpp WaveAppender)
                        o = xyz4(ld_stream(&a.ro[rid]));
                        d = xyz4(ld_stream(&a.rd[rid]));
                        if (a.normalize_dir) d = vptfp::normalize(d);  // RayGen.slang:70
                        inv = safe_inverse(d);
                        best_t = a.tmax; bslot = 0xffffffffu; bgid = 0xffffffffu;
                        sp = 0; cur = 0;  // root
                    }
                }
                const uint32_t want = (uint32_t)__popcll(m_idle), left = w_end - w_next;
                w_next += want < left ? want : left;
            }
        } else if (nn + nl == 0u) {
            break;
        } else if (weighted ? nn > 2u * nl : nn >= nl) {
            if (at_node) {  // ---- inner-node step
                if (COUNT) st_nodes++;
                if (WIDE8) vote_node8_step(sc.nodes8, S, cur, sp, o, inv, a.tmin, best_t);
                else vote_node_step<ANY>(nodes, S, cur, sp, o, inv, a.tmin, best_t);
            }
        } else {
            if (at_leaf) {  // ---- triangle step: ONE triangle of the lane's leaf
                if (COUNT) st_tris++;
                if (ANY) { if (vote_tri_step_any(tris, S, cur, sp, o, d, a.tmin, a.tmax, a.tmax, 0xffffffffu)) bslot = 0u; }
                else vote_tri_step_closest(tris, S, cur, sp, o, d, a.tmin, a.tmax, best_t, bu, bv, bslot, bgid);
            }
        }
    }
    if (cur == kLaneDone) {  // rays that finished after the stream ran dry
        if (ANY) a.hit[rid] = make_float4(bslot != 0xffffffffu ? 1.0f : -1.0f, 0.0f, 0.0f, 0.0f);
        else {
            const uint32_t inst = store_closest(a, tris, rid, bslot != 0xffffffffu, best_t, bu, bv, bslot);
            if (a.cls) a.cls[qi] = bslot != 0xffffffffu ? sc.inst_class[inst] : (unsigned char)kShadeMiss;
        }
    }
    if (COUNT) {
        atomicAdd(&ctr->stat_nodes, (unsigned long long)st_nodes);
        atomicAdd(&ctr->stat_tris, (unsigned long long)st_tris);
    }
}

// ------------------------------------------------------------------ launch
int trace_blocks_per_cu(uint32_t variant, bool any) {
    int nb = 0;
    const size_t lds = kVoteStackBytes;
    if (variant == VPT_TRACE_BASE) {
        if (any) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace_base<true, false>, kTraverseBlock, lds);
        else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace_base<false, false>, kTraverseBlock, lds);
    } else if (variant == VPT_TRACE_VOTE8) {
        if (any) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace_vote<true, false, true>, kTraverseBlock, lds);
        else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace_vote<false, false, true>, kTraverseBlock, lds);
    } else {
        if (any) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace_vote<true, false, false>, kTraverseBlock, lds);
        else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace_vote<false, false, false>, kTraverseBlock, lds);
    }
    return nb > 0 ? nb : 1;
}

void launch_trace(hipStream_t s, uint32_t blocks, uint32_t variant, bool any, bool count, const DeviceScene& sc, const TraceArgs& a, Counters* ctr) {
    const size_t lds = kVoteStackBytes;
    const dim3 g(blocks), b(kTraverseBlock);
#define VPT_LT(K) do { if (any) { if (count) hipLaunchKernelGGL((K<true, true>), g, b, lds, s, sc, a, ctr); else hipLaunchKernelGGL((K<true, false>), g, b, lds, s, sc, a, ctr); } \
                       else { if (count) hipLaunchKernelGGL((K<false, true>), g, b, lds, s, sc, a, ctr); else hipLaunchKernelGGL((K<false, false>), g, b, lds, s, sc, a, ctr); } } while (0)
#define VPT_LV(W) do { if (any) { if (count) hipLaunchKernelGGL((k_trace_vote<true, true, W>), g, b, lds, s, sc, a, ctr); else hipLaunchKernelGGL((k_trace_vote<true, false, W>), g, b, lds, s, sc, a, ctr); } \
                       else { if (count) hipLaunchKernelGGL((k_trace_vote<false, true, W>), g, b, lds, s, sc, a, ctr); else hipLaunchKernelGGL((k_trace_vote<false, false, W>), g, b, lds, s, sc, a, ctr); } } while (0)
    if (variant == VPT_TRACE_BASE) VPT_LT(k_trace_base); else if (variant == VPT_TRACE_VOTE8) VPT_LV(true); else VPT_LV(false);
#undef VPT_LV
#undef VPT_LT
}

}  // namespace vpt
