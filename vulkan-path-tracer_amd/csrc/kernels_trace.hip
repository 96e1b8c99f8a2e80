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

#if VPT_LAB
// ------------------------------------------------------------------ baseline: 64 rays per wave at a time (laboratory build only)
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

#endif  // VPT_LAB

// ------------------------------------------------------------------ vote-scheduled persistent lanes
// Lane state: `cur` >= 0 inner node to visit; < 0 leaf code ~(first << 3 | count - 1) with `first` advancing as the
// triangles are consumed; kLaneDone / kLaneIdle.  The wave owns a chunk [w_next, w_end) of the stream (one atomic per
// chunk) and deals its entries to idle lanes in fetch steps.
// TUNED: the vote parameters are the compile-time defaults (fetch step at 16 idle lanes, weighted vote: kVoteParamDefault), which
// the pipeline always uses; the lab's other settings go through the instantiation that reads them from a.param.
// STRICT (VPT_FLAG_LOCAL_HITS, traverse.hpp trace_closest_strict / trace_occluded_strict): a closest-hit winner is validated when its
// ray retires and the ray is traced again without that triangle if the hit was not local to it; an any-hit stop is validated on the spot.
// CULL (closest-hit, four-wide tree): stale stack entries are dropped at the pop (vote.hpp LaneStack::pop_or_done_cull).
template <bool ANY, bool COUNT, bool WIDE8, bool TUNED, bool STRICT = false, bool CULL = false, bool PK = false, bool TRI2 = false, bool SPLIT4 = false>
__global__ __launch_bounds__(kTraverseBlock, 8) void k_trace_vote(DeviceScene sc, TraceArgs a, Counters* ctr) {
    static_assert(!SPLIT4 || (!WIDE8 && !CULL && !STRICT && !PK), "the split-order tree is an experiment on the four-wide node (any-hit searches ignore its order tables)");
    static_assert(!(TRI2 && STRICT), "the two-triangle steps have no validating form: VPT_FLAG_LOCAL_HITS keeps the one-triangle step");
    extern __shared__ __align__(16) unsigned char smem[];
    const LaneStack S = make_lane_stack(smem, sc.stack_overflow);
    const BvhNode* const nodes = SPLIT4 ? sc.nodes4s : sc.nodes;
    const BvhTri* const tris = sc.tris;
    const TreeTop top = stage_tree_top(smem, nodes, sc.node_count, ANY && !WIDE8 && (TUNED || ((a.param >> 16) & 1u) == 0u));   // any-hit only (vote.hpp vote_node_step); lab: bit 16 switches it off
    uint32_t oct = 0u;   // SPLIT4: the ray's direction octant
    const uint32_t n = a.n_dev ? *a.n_dev : a.n;
    const uint32_t chunk = fetch_chunk(n);
    const uint32_t fetch_at = TUNED ? kVoteFetchAt : (a.param & 0xffu) ? (a.param & 0xffu) : 16u;  // idle lanes that trigger a fetch step (64: only when all are idle)
    const bool weighted = TUNED || ((a.param >> 8) & 1u) != 0u;  // vote by lanes served per instruction issued: a triangle step costs about half a node step
    const uint32_t w4 = TUNED ? kVoteWeight4 : ((a.param >> 12) & 15u) ? ((a.param >> 12) & 15u) : kVoteWeight4;   // the weight in quarters (lab: bits 12-15)
    // every wave starts on its own 64 entries without an atomic (8192 waves fetching at once would queue ~90 us on the cursor);
    // entries beyond the grid's static part are fetched chunk-wise through the cursor
    const uint32_t n_static = gridDim.x * (kTraverseBlock / 64u) * 64u;
    // wave-uniform on purpose (readfirstlane): with the wave index taken from threadIdx the compiler must treat the chunk bounds, and
    // through them `exhausted` and every branch of the vote loop, as divergent, which turns the loop into exec-masked regions with
    // ~48 register copies per iteration at their joins (profiles/r03_trace_isa_budget.md)
    uint32_t w_next = __builtin_amdgcn_readfirstlane((blockIdx.x * (kTraverseBlock / 64u) + (threadIdx.x >> 6)) * 64u), w_end = w_next + 64u < n ? w_next + 64u : n;
    if (w_next >= n) { w_next = 0u; w_end = 0u; }
    bool exhausted = false;
    int cur = kLaneIdle, sp = 0;
    uint32_t rid = 0, bslot = 0xffffffffu, bgid = 0xffffffffu;
    V3 o = vptfp::v3(0.0f, 0.0f, 0.0f), d = o, inv = o;
    float best_t = 0.0f, bu = 0.0f, bv = 0.0f;
    uint32_t st_nodes = 0, st_tris = 0;
    uint32_t qi = 0u;  // the ray's position in the queue: where its shade class goes (a.cls)
    uint32_t ex0 = 0xffffffffu, ex1 = 0xffffffffu;   // STRICT: triangles excluded from this ray's search
    bool validated = false;                           // STRICT: the lane's finished ray holds a validated winner (or none)
    // Vote loop: one kind of step per iteration.  (Giving each kind its own inner loop, which lets the lane state stay in fixed
    // registers across the back edge, was measured 5-7 % slower on closest-hit rays: the vote then sticks to a kind for too long.)
    // The fetch step sits in an OUTER loop and the node / triangle steps in an inner one that re-votes every iteration: the ray
    // (o, d, inv, rid, qi) is then loop-invariant where the steps run, and the compiler no longer copies the whole lane state
    // between two register sets on every iteration (29 v_mov per step before, profiles/r03_trace_isa_budget.md).
    while (true) {
        uint32_t nn, nl;
        while (true) {
            VPT_MARK("vote");
            const bool busy = cur < kLaneDone;
            const bool at_node = busy && cur >= 0;
            const bool at_leaf = busy && cur < 0;
            nn = (uint32_t)__popcll(__ballot(at_node)); nl = (uint32_t)__popcll(__ballot(at_leaf));
            if ((!exhausted && 64u - nn - nl >= fetch_at) || nn + nl == 0u) break;
            const bool node_wins = weighted ? 4u * nn > w4 * nl : nn >= nl;
            // two predicated regions in sequence rather than if / else on the (uniform) vote: the if / else form is compiled into a
            // flag-linked pair of regions that hands the lane state from one register set to another and back (14 v_mov per step)
            VPT_MARK("node");
            if (node_wins & at_node) {  // ---- inner-node step
                if (COUNT) st_nodes++;
                if (WIDE8) vote_node8_step(sc.nodes8, S, cur, sp, o, inv, a.tmin, best_t);
                else if (SPLIT4 && ANY) vote_node_step<true, true, false, LaneStack, false, true>(nodes, top, S, cur, sp, o, inv, a.tmin, best_t);
                else if (SPLIT4) vote_node4s_step(nodes, S, cur, sp, o, inv, oct, a.tmin, best_t);
                else vote_node_step<ANY, ANY, CULL, LaneStack, PK>(nodes, top, S, cur, sp, o, inv, a.tmin, best_t);
            }
            VPT_MARK("tri");
            if (!node_wins & at_leaf) {  // ---- triangle step: ONE triangle of the lane's leaf
                if (COUNT) st_tris += TRI2 ? (((uint32_t)(~cur)) & 7u ? 2u : 1u) : 1u;
                if (TRI2) {   // the product instantiation: up to two triangles of the leaf per step (-3 ... -6 %, profiles/r04_trace_lab_tri2_*.json)
                    if (ANY) { if (vote_tri2_step_any(tris, S, cur, sp, o, d, a.tmin, a.tmax, a.tmax, 0xffffffffu)) bslot = 0u; }
                    else vote_tri2_step_closest(tris, S, cur, sp, o, d, a.tmin, a.tmax, best_t, bu, bv, bslot, bgid);
                }
                else if (ANY) { if (vote_tri_step_any<STRICT>(tris, S, cur, sp, o, d, a.tmin, a.tmax, a.tmax, 0xffffffffu)) bslot = 0u; }
                else vote_tri_step_closest<STRICT, CULL>(tris, S, cur, sp, o, d, a.tmin, a.tmax, best_t, bu, bv, bslot, bgid, ex0, ex1);
            }
        }
        VPT_MARK("exit");
        if (STRICT && !ANY) {   // finished rays: is the winner's hit local to the winner?  If not: once more without that triangle
            bool again = false;
            if (cur == kLaneDone && !validated) {
                validated = true;
                if (bslot != 0xffffffffu) {
                    const float4* q = reinterpret_cast<const float4*>(tris + bslot);
                    const float4 ta = q[0], tb = q[1], tc = q[2];
                    if (!vptfp::hit_is_local(o, d, vptfp::v3(ta.x, ta.y, ta.z), vptfp::v3(ta.w, tb.x, tb.y), vptfp::v3(tb.z, tb.w, tc.x), best_t)) {
                        ex1 = ex0; ex0 = bgid;
                        best_t = a.tmax; bslot = 0xffffffffu; bgid = 0xffffffffu; sp = 0; cur = 0; validated = false;
                        again = true;
                    }
                }
            }
            if (exhausted && __ballot(again) != 0ull) continue;   // nothing left to fetch, but a lane is busy again
        }
        if (exhausted) break;   // nothing left to fetch and (inner loop's exit) no lane busy
        VPT_MARK("fetch");
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
                    if (SPLIT4) oct = (inv.x < 0.0f ? 1u : 0u) | (inv.y < 0.0f ? 2u : 0u) | (inv.z < 0.0f ? 4u : 0u);
                    best_t = a.tmax; bslot = 0xffffffffu; bgid = 0xffffffffu;
                    ex0 = 0xffffffffu; ex1 = 0xffffffffu; validated = false;
                    sp = 0; cur = 0;  // root
                }
            }
            const uint32_t want = (uint32_t)__popcll(m_idle), left = w_end - w_next;
            w_next += want < left ? want : left;
        }
    }
    VPT_MARK("done");
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

#if VPT_LAB
// ------------------------------------------------------------------ ray slots in LDS (trace lab variant VPT_TRACE_POOL, closest hit; laboratory build only)
// In k_trace_vote a ray lives in a lane's registers, so a step of one kind runs on the lanes whose OWN ray wants it: 42 of 64 lanes in
// a node step, 21 in a triangle step (profiles/r04_vote_sim_*.txt; the counters say the same).  Here a wave owns SLOTS ray slots in LDS —
// ray, best hit, traversal state and stack of every ray — and a step runs on up to 64 of the slots that want it, whichever they are:
// lanes are workers, not owners.  The host model puts that at -21 / -31 % VALU wave-instructions per ray at 96 / 128 slots (+ 20 per step
// for the selection and the state traffic).  What it costs on the device: 10-13 KB of LDS per wave, i.e. 4 or 3 waves per SIMD instead of
// 8 behind every 64-byte node fetch, and the same number of L1 accesses per ray as before.  DUAL: an iteration runs a node step AND (when
// enough slots wait at leaves) a triangle step on disjoint slots, with the loads of both in flight together.
// Results are per ray: bit-identical hits by the lab's check.
constexpr int kPoolRows = 16;     // state dwords per slot: o 0-2, d 3-5, 1/d 6-8, best_t 9, u 10, v 11, best slot 12, cur 13, sp 14, ray id 15
constexpr size_t pool_wave_bytes(int slots, int stack) { return (size_t)(kPoolRows + stack) * slots * 4 + 128; }   // + the step's two slot lists (one byte per entry)
constexpr size_t pool_lds_bytes(int slots, int stack) { return pool_wave_bytes(slots, stack) * (kTraverseBlock / 64); }
template <int SLOTS, int STACK>
struct PoolStack {
    uint32_t* stk;  // LDS: entry k of this slot at stk[k * SLOTS]
    uint32_t* ovf;  // global: entries beyond STACK
    __device__ __forceinline__ void push(int& sp, int v) const {
        if (sp < STACK) stk[sp * SLOTS] = (uint32_t)v;
        else if (sp < STACK + kStackOverflow) ovf[sp - STACK] = (uint32_t)v;
        sp++;
    }
    __device__ __forceinline__ void pop_or_done(int& sp, int& cur) const {
        if (sp == 0) cur = kLaneDone;
        else {
            sp--;
            int v = (int)stk[(sp < STACK ? sp : STACK - 1) * SLOTS];
            asm volatile("" : "+v"(v));   // (vote.hpp LaneStack::pop_or_done)
            if (sp >= STACK) v = (int)ovf[sp - STACK];
            cur = v;
        }
    }
};
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// vote.hpp vote_node_step<closest hit> on node words that are already in registers
template <class STK>
__device__ __forceinline__ void pool_node_step(const uint4& w0, const uint4& w1, const uint4& w2, const uint4& w3, const STK& S, int& cur, int& sp, V3 o, V3 inv, float tmin, float tlimit) {
    NodeData n;
    unpack_node(w0, w1, w2, w3, n);
    RaySlab slab; slab.o = o; slab.inv = inv;
    slab.negx = inv.x < 0.0f; slab.negy = inv.y < 0.0f; slab.negz = inv.z < 0.0f;
    float t0, t1, t2, t3;
    node_entries(n, slab, tmin, tlimit, t0, t1, t2, t3);
    int c0 = n.c0, c1 = n.c1, c2 = n.c2, c3 = n.c3;
    cswap(t0, c0, t1, c1); cswap(t2, c2, t3, c3); cswap(t0, c0, t2, c2); cswap(t1, c1, t3, c3); cswap(t1, c1, t2, c2);
    if (t0 < kMissT) {  // nearest child next, the others pushed far -> near
        if (t3 < kMissT) S.push(sp, c3);
        if (t2 < kMissT) S.push(sp, c2);
        if (t1 < kMissT) S.push(sp, c1);
        cur = c0;
    } else S.pop_or_done(sp, cur);
}
template <bool COUNT, int SLOTS, int STACK, bool DUAL>
__global__ __launch_bounds__(kTraverseBlock, 3) void k_trace_pool(DeviceScene sc, TraceArgs a, Counters* ctr) {
    constexpr size_t kWaveBytes = pool_wave_bytes(SLOTS, STACK);
    constexpr uint32_t kHalf2 = (uint32_t)SLOTS - 64u;   // slots of the second half (lanes below this look after two slots)
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    uint32_t* const W = reinterpret_cast<uint32_t*>(smem + wave * kWaveBytes);
    float* const F = reinterpret_cast<float*>(W);
    uint32_t* const stack0 = W + kPoolRows * SLOTS;
    unsigned char* const list_n = reinterpret_cast<unsigned char*>(stack0 + STACK * SLOTS);
    unsigned char* const list_l = list_n + 64;
    const uint32_t gw = __builtin_amdgcn_readfirstlane(blockIdx.x * (kTraverseBlock / 64u) + wave);
    uint32_t* const spill = sc.stack_overflow + (size_t)gw * SLOTS * kStackOverflow;
#define ST(row, slot) W[(row) * SLOTS + (slot)]
#define SF(row, slot) F[(row) * SLOTS + (slot)]
    const BvhNode* const nodes = sc.nodes;
    const BvhTri* const tris = sc.tris;
    const uint32_t n = a.n_dev ? *a.n_dev : a.n;
    const uint32_t chunk = fetch_chunk(n);
    const uint32_t fetch_at = (a.param & 0xffu) ? (a.param & 0xffu) : (uint32_t)SLOTS * 5u / 16u;   // idle SLOTS that trigger a fetch step
    const uint32_t w4 = ((a.param >> 12) & 15u) ? ((a.param >> 12) & 15u) : kVoteWeight4;
    const uint32_t tri_at = ((a.param >> 16) & 63u) ? ((a.param >> 16) & 63u) : 32u;                  // DUAL: slots at leaves that make an iteration carry a triangle step
    const uint32_t n_static = gridDim.x * (kTraverseBlock / 64u) * (uint32_t)SLOTS;   // every wave starts on its own SLOTS entries, no atomic
    uint32_t w_next = gw * (uint32_t)SLOTS, w_end = w_next + (uint32_t)SLOTS < n ? w_next + (uint32_t)SLOTS : n;
    if (w_next >= n) { w_next = 0u; w_end = 0u; }
    bool exhausted = false;
    uint32_t st_nodes = 0u, st_tris = 0u;
    const bool two = lane < kHalf2;   // this lane looks after slot lane + 64 as well
    ST(13, lane) = (uint32_t)kLaneIdle; if (two) ST(13, lane + 64u) = (uint32_t)kLaneIdle;
    wave_lds_sync();
    while (true) {
        // ---- vote over the pool: every lane looks at (up to) two slots
        const int c0 = (int)ST(13, lane), c1 = two ? (int)ST(13, lane + 64u) : kLaneIdle;
        const bool n0 = c0 >= 0 && c0 < kLaneDone, n1 = c1 >= 0 && c1 < kLaneDone, l0 = c0 < 0, l1 = c1 < 0;
        const unsigned long long bn0 = __ballot(n0), bn1 = __ballot(n1), bl0 = __ballot(l0), bl1 = __ballot(l1);
        const uint32_t nn = (uint32_t)(__popcll(bn0) + __popcll(bn1)), nl = (uint32_t)(__popcll(bl0) + __popcll(bl1));
        if ((!exhausted && (uint32_t)SLOTS - nn - nl >= fetch_at) || nn + nl == 0u) {
            // ---- fetch step: retire finished slots, deal new rays to the idle ones (two passes of 64 slots)
#pragma unroll 1
            for (uint32_t h = 0u; h < (SLOTS > 64 ? 2u : 1u); h++) {
                const uint32_t s = lane + 64u * h;
                const bool mine = h == 0u || two;
                int c = mine ? (int)ST(13, s) : kLaneDone - 1;   // (a slot that does not exist is neither done nor idle)
                if (c == kLaneDone) {
                    const uint32_t bslot = ST(12, s);
                    (void)store_closest(a, tris, ST(15, s), bslot != 0xffffffffu, SF(9, s), SF(10, s), SF(11, s), bslot);
                    c = kLaneIdle; ST(13, s) = (uint32_t)kLaneIdle;
                }
                if (exhausted) continue;
                if (w_next >= w_end) {
                    if (n_static >= n) exhausted = true;
                    else {
                        uint32_t base = 0u;
                        if (lane == 0u) base = atomicAdd(a.head, chunk);
                        base = n_static + __builtin_amdgcn_readfirstlane(base);
                        if (base >= n) exhausted = true;
                        else { w_next = base; w_end = base + chunk < n ? base + chunk : n; }
                    }
                }
                if (exhausted) continue;
                const unsigned long long m_idle = __ballot(c == kLaneIdle);
                const uint32_t i = w_next + lanes_below(m_idle);
                if (c == kLaneIdle && i < w_end) {
                    const uint32_t rid = a.order ? a.order[i] : i;
                    V3 o = xyz4(ld_stream(&a.ro[rid])), d = xyz4(ld_stream(&a.rd[rid]));
                    if (a.normalize_dir) d = vptfp::normalize(d);
                    const V3 inv = safe_inverse(d);
                    SF(0, s) = o.x; SF(1, s) = o.y; SF(2, s) = o.z; SF(3, s) = d.x; SF(4, s) = d.y; SF(5, s) = d.z; SF(6, s) = inv.x; SF(7, s) = inv.y; SF(8, s) = inv.z;
                    SF(9, s) = a.tmax; ST(12, s) = 0xffffffffu; ST(13, s) = 0u; ST(14, s) = 0u; ST(15, s) = rid;
                }
                const uint32_t want = (uint32_t)__popcll(m_idle), left = w_end - w_next;
                w_next += want < left ? want : left;
            }
            wave_lds_sync();
            if (exhausted) {   // nothing came in: done when no slot is busy
                const int d0 = (int)ST(13, lane), d1 = two ? (int)ST(13, lane + 64u) : kLaneIdle;
                if (__ballot(d0 < kLaneDone) == 0ull && __ballot(d1 < kLaneDone) == 0ull) break;
            }
            continue;
        }
        // ---- which kind(s) of step, and the (up to) 64 slots that take each
        const bool node_wins = (nn >= 64u && nl < 64u) ? true : (nl >= 64u && nn < 64u) ? false : 4u * nn > w4 * nl;
        const bool do_node = DUAL ? nn > 0u : node_wins, do_tri = DUAL ? (nl >= tri_at || nn == 0u) : !node_wins;
        if (do_node) {
            const uint32_t r0 = lanes_below(bn0), r1 = (uint32_t)__popcll(bn0) + lanes_below(bn1);
            if (n0 && r0 < 64u) list_n[r0] = (unsigned char)lane;
            if (n1 && r1 < 64u) list_n[r1] = (unsigned char)(lane + 64u);
        }
        if (do_tri) {
            const uint32_t r0 = lanes_below(bl0), r1 = (uint32_t)__popcll(bl0) + lanes_below(bl1);
            if (l0 && r0 < 64u) list_l[r0] = (unsigned char)lane;
            if (l1 && r1 < 64u) list_l[r1] = (unsigned char)(lane + 64u);
        }
        wave_lds_sync();
        const bool an = do_node && lane < (nn < 64u ? nn : 64u), al = do_tri && lane < (nl < 64u ? nl : 64u);
        // the loads of both steps first
        uint32_t sn = 0u, sl = 0u;
        int cur_n = 0, cur_l = 0;
        uint4 w0 = make_uint4(0u, 0u, 0u, 0u), w1 = w0, w2 = w0, w3 = w0;
        float4 ta = make_float4(0.0f, 0.0f, 0.0f, 0.0f), tb = ta, tc = ta;
        if (an) {
            sn = list_n[lane]; cur_n = (int)ST(13, sn);
            const uint4* p = reinterpret_cast<const uint4*>(nodes + cur_n);
            w0 = p[0]; w1 = p[1]; w2 = p[2]; w3 = p[3];
        }
        if (al) {
            sl = list_l[lane]; cur_l = (int)ST(13, sl);
            const float4* q = reinterpret_cast<const float4*>(tris + (int)(((uint32_t)(~cur_l)) >> 3));
            ta = q[0]; tb = q[1]; tc = q[2];
        }
        if (an) {   // ---- inner-node step
            if (COUNT) st_nodes++;
            int sp = (int)ST(14, sn);
            PoolStack<SLOTS, STACK> S; S.stk = stack0 + sn; S.ovf = spill + (size_t)sn * kStackOverflow;
            pool_node_step(w0, w1, w2, w3, S, cur_n, sp, vptfp::v3(SF(0, sn), SF(1, sn), SF(2, sn)), vptfp::v3(SF(6, sn), SF(7, sn), SF(8, sn)), a.tmin, SF(9, sn));
            ST(13, sn) = (uint32_t)cur_n; ST(14, sn) = (uint32_t)sp;
        }
        if (al) {   // ---- ONE triangle of the slot's leaf (vote.hpp vote_tri_step_closest; the best triangle's global id is read only on a tie in t)
            if (COUNT) st_tris++;
            int sp = (int)ST(14, sl);
            PoolStack<SLOTS, STACK> S; S.stk = stack0 + sl; S.ovf = spill + (size_t)sl * kStackOverflow;
            const uint32_t enc = (uint32_t)(~cur_l);
            const int first = (int)(enc >> 3);
            const uint32_t more = enc & 7u;
            float t, u, v;
            const bool hit = ray_triangle_flat(vptfp::v3(SF(0, sl), SF(1, sl), SF(2, sl)), vptfp::v3(SF(3, sl), SF(4, sl), SF(5, sl)), vptfp::v3(ta.x, ta.y, ta.z), vptfp::v3(ta.w, tb.x, tb.y),
                                               vptfp::v3(tb.z, tb.w, tc.x), a.tmin, a.tmax, t, u, v);
            const uint32_t bslot = ST(12, sl);
            const float best_t = SF(9, sl);
            bool better = hit & ((bslot == 0xffffffffu) | (t < best_t));
            if (hit & (bslot != 0xffffffffu) & (t == best_t)) better = __float_as_uint(tc.w) < tris[bslot].gid;   // ties in t go to the smaller global id
            if (better) { SF(9, sl) = t; SF(10, sl) = u; SF(11, sl) = v; ST(12, sl) = (uint32_t)first; }
            if (more) cur_l = ~(int)((((uint32_t)first + 1u) << 3) | (more - 1u));
            else S.pop_or_done(sp, cur_l);
            ST(13, sl) = (uint32_t)cur_l; ST(14, sl) = (uint32_t)sp;
        }
        wave_lds_sync();
    }
#undef ST
#undef SF
    if (COUNT) {
        atomicAdd(&ctr->stat_nodes, (unsigned long long)st_nodes);
        atomicAdd(&ctr->stat_tris, (unsigned long long)st_tris);
    }
}

// ------------------------------------------------------------------ two rays per lane (trace lab variant VPT_TRACE_PAIR, closest hit)
// The pool's idea without its LDS traffic: a lane keeps TWO rays in its registers and, in a step of the voted kind, serves whichever of them wants it
// (the first if both do).  A lane then takes part in a node step when either ray stands at a node (0.66 -> 0.88 of the lanes by the model,
// tests/tools/vote_sim "two rays per lane": -8 ... -12 % VALU per ray on top of the two-triangle step).  The price: the select of ~14 state registers into
// the step and the write-back behind it, 17 more live registers, a second stack per lane in LDS.  Results are per ray: bit-identical hits by the lab's check.
constexpr size_t kPairLdsBytes = 2 * kVoteStackBytes;
struct RayRegs {
    int cur, sp;
    uint32_t rid, bslot, bgid;
    V3 o, d, inv;
    float best_t, bu, bv;
};
template <bool COUNT>
__global__ __launch_bounds__(kTraverseBlock, 5) void k_trace_pair(DeviceScene sc, TraceArgs a, Counters* ctr) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t* const stk0 = reinterpret_cast<uint32_t*>(smem) + threadIdx.x;
    uint32_t* const stk1 = stk0 + kVoteStackRows * kTraverseBlock;
    uint32_t* const ovf0 = sc.stack_overflow + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 2u * kStackOverflow;
    const BvhNode* const nodes = sc.nodes;
    const BvhTri* const tris = sc.tris;
    TreeTop top; top.lds = nullptr; top.count = 0;
    const uint32_t n = a.n_dev ? *a.n_dev : a.n;
    const uint32_t chunk = fetch_chunk(n);
    const uint32_t fetch_at = (a.param & 0xffu) ? (a.param & 0xffu) : 48u;   // idle RAYS (of 128) that trigger a fetch step
    const uint32_t n_static = gridDim.x * (kTraverseBlock / 64u) * 128u;     // every wave starts on its own 128 entries, no atomic
    uint32_t w_next = __builtin_amdgcn_readfirstlane((blockIdx.x * (kTraverseBlock / 64u) + (threadIdx.x >> 6)) * 128u), w_end = w_next + 128u < n ? w_next + 128u : n;
    if (w_next >= n) { w_next = 0u; w_end = 0u; }
    bool exhausted = false;
    RayRegs r0, r1;
    r0.cur = kLaneIdle; r0.sp = 0; r0.rid = 0u; r0.bslot = 0xffffffffu; r0.bgid = 0xffffffffu; r0.o = vptfp::v3(0.0f, 0.0f, 0.0f); r0.d = r0.o; r0.inv = r0.o; r0.best_t = 0.0f; r0.bu = 0.0f; r0.bv = 0.0f;
    r1 = r0;
    uint32_t st_nodes = 0u, st_tris = 0u;
    auto retire_and_refill = [&](RayRegs& r) {   // every lane of the wave calls
        if (r.cur == kLaneDone) { (void)store_closest(a, tris, r.rid, r.bslot != 0xffffffffu, r.best_t, r.bu, r.bv, r.bslot); r.cur = kLaneIdle; }
        if (exhausted) return;
        if (w_next >= w_end) {
            if (n_static >= n) exhausted = true;
            else {
                uint32_t base = 0u;
                if (lane_id() == 0u) base = atomicAdd(a.head, chunk);
                base = n_static + __builtin_amdgcn_readfirstlane(base);
                if (base >= n) exhausted = true;
                else { w_next = base; w_end = base + chunk < n ? base + chunk : n; }
            }
        }
        if (exhausted) return;
        const unsigned long long m_idle = __ballot(r.cur == kLaneIdle);
        const uint32_t i = w_next + lanes_below(m_idle);
        if (r.cur == kLaneIdle && i < w_end) {
            r.rid = a.order ? a.order[i] : i;
            r.o = xyz4(ld_stream(&a.ro[r.rid]));
            r.d = xyz4(ld_stream(&a.rd[r.rid]));
            if (a.normalize_dir) r.d = vptfp::normalize(r.d);
            r.inv = safe_inverse(r.d);
            r.best_t = a.tmax; r.bslot = 0xffffffffu; r.bgid = 0xffffffffu; r.sp = 0; r.cur = 0;
        }
        const uint32_t want = (uint32_t)__popcll(m_idle), left = w_end - w_next;
        w_next += want < left ? want : left;
    };
    while (true) {
        const bool n0 = r0.cur >= 0 && r0.cur < kLaneDone, n1 = r1.cur >= 0 && r1.cur < kLaneDone, l0 = r0.cur < 0, l1 = r1.cur < 0;
        const uint32_t nn = (uint32_t)__popcll(__ballot(n0 | n1)), nl = (uint32_t)__popcll(__ballot(l0 | l1));
        const uint32_t busy = (uint32_t)(__popcll(__ballot(n0 | l0)) + __popcll(__ballot(n1 | l1)));
        if ((!exhausted && 128u - busy >= fetch_at) || busy == 0u) {
            retire_and_refill(r0);
            retire_and_refill(r1);
            if (exhausted && __ballot(r0.cur < kLaneDone) == 0ull && __ballot(r1.cur < kLaneDone) == 0ull) break;
            continue;
        }
        const bool node_wins = 4u * nn > kVoteWeight4 * nl;
        const bool take0 = node_wins ? n0 : l0, take1 = !take0 && (node_wins ? n1 : l1);
        if (take0 | take1) {
            // the served ray into the step's registers
            int cur = take0 ? r0.cur : r1.cur, sp = take0 ? r0.sp : r1.sp;
            const V3 o = vptfp::v3(take0 ? r0.o.x : r1.o.x, take0 ? r0.o.y : r1.o.y, take0 ? r0.o.z : r1.o.z);
            float best_t = take0 ? r0.best_t : r1.best_t;
            LaneStack S; S.stk = take0 ? stk0 : stk1; S.ovf = ovf0 + (take0 ? 0 : kStackOverflow); S.tq = nullptr;
            if (node_wins) {
                if (COUNT) st_nodes++;
                const V3 inv = vptfp::v3(take0 ? r0.inv.x : r1.inv.x, take0 ? r0.inv.y : r1.inv.y, take0 ? r0.inv.z : r1.inv.z);
                vote_node_step<false, false, false>(nodes, top, S, cur, sp, o, inv, a.tmin, best_t);
            } else {
                if (COUNT) st_tris += (((uint32_t)(~cur)) & 7u) ? 2u : 1u;
                const V3 d = vptfp::v3(take0 ? r0.d.x : r1.d.x, take0 ? r0.d.y : r1.d.y, take0 ? r0.d.z : r1.d.z);
                float bu = take0 ? r0.bu : r1.bu, bv = take0 ? r0.bv : r1.bv;
                uint32_t bslot = take0 ? r0.bslot : r1.bslot, bgid = take0 ? r0.bgid : r1.bgid;
                vote_tri2_step_closest(tris, S, cur, sp, o, d, a.tmin, a.tmax, best_t, bu, bv, bslot, bgid);
                if (take0) { r0.best_t = best_t; r0.bu = bu; r0.bv = bv; r0.bslot = bslot; r0.bgid = bgid; }
                else { r1.best_t = best_t; r1.bu = bu; r1.bv = bv; r1.bslot = bslot; r1.bgid = bgid; }
            }
            if (take0) { r0.cur = cur; r0.sp = sp; } else { r1.cur = cur; r1.sp = sp; }
        }
    }
    // (the loop leaves with every ray retired: the fetch step that found nothing busy stored the last results)
    if (COUNT) {
        atomicAdd(&ctr->stat_nodes, (unsigned long long)st_nodes);
        atomicAdd(&ctr->stat_tris, (unsigned long long)st_tris);
    }
}

#endif  // VPT_LAB

// ------------------------------------------------------------------ shadow rays
// LIGHT = false: visible <=> nothing is hit (ClosestHit.slang:139, 344-353).  LIGHT = true: visible <=> the closest hit is
// the sampled triangle (ClosestHit.slang:171-176, 358-370): that triangle is tested first by its own record, then the search
// looks for anything that beats it (traverse.hpp closest_is).
// TRI2: up to two triangles of a leaf per triangle step (vote.hpp vote_tri2_step_any: -3 ... -6 % on atrium and bust shadow rays, profiles/r04_trace_lab_tri2_*.json); the
// product instantiation only — the counting and the validating ones keep the one-triangle step, so the visit statistics stay what a ray needs.
template <bool LIGHT, bool COUNT, bool TUNED, bool STRICT = false, bool TRI2 = false>
__global__ __launch_bounds__(kTraverseBlock, 8) void k_trace_shadow(DeviceScene sc, const float4* RO, const float4* RD, unsigned char* vis, const uint32_t* n_dev,
                                                                  uint32_t* head, Counters* ctr, uint32_t param, uint32_t ray_queries) {
    static_assert(!(TRI2 && STRICT), "the two-triangle steps have no validating form: VPT_FLAG_LOCAL_HITS keeps the one-triangle step");
    extern __shared__ __align__(16) unsigned char smem[];
    const LaneStack S = make_lane_stack(smem, sc.stack_overflow);
    const BvhNode* const nodes = sc.nodes;
    const BvhTri* const tris = sc.tris;
    const TreeTop top = stage_tree_top(smem, nodes, sc.node_count, TUNED || ((param >> 16) & 1u) == 0u);
    const uint32_t n = *n_dev;
    const uint32_t chunk = fetch_chunk(n);
    const uint32_t fetch_at = TUNED ? kVoteFetchAt : (param & 0xffu) ? (param & 0xffu) : 16u;
    const bool weighted = TUNED || ((param >> 8) & 1u) != 0u;
    const uint32_t w4 = TUNED ? kVoteWeight4 : ((param >> 12) & 15u) ? ((param >> 12) & 15u) : kVoteWeight4;
    // RTCommon.slang:52-63 (USE_RAY_QUERIES: the direction as it is, [1e-4, 1e6]) or :64-84 (normalised direction, [1e-5, 1000]; sky rays only — light rays are not queued then, shade_core.hpp)
    const float tmin = ray_queries ? 0.0001f : 0.00001f, tmax = ray_queries ? 1000000.0f : 1000.0f;
    const uint32_t n_static = gridDim.x * (kTraverseBlock / 64u) * 64u;  // every wave starts on its own 64 entries, no atomic
    // wave-uniform by construction (see k_trace_vote)
    uint32_t w_next = __builtin_amdgcn_readfirstlane((blockIdx.x * (kTraverseBlock / 64u) + (threadIdx.x >> 6)) * 64u), w_end = w_next + 64u < n ? w_next + 64u : n;
    if (w_next >= n) { w_next = 0u; w_end = 0u; }
    bool exhausted = false;
    int cur = kLaneIdle, sp = 0;
    uint32_t rid = 0u, expect = 0xffffffffu;
    bool visible = false;
    V3 o = vptfp::v3(0.0f, 0.0f, 0.0f), d = o, inv = o;
    float tlim = tmax;
    uint32_t st_nodes = 0u, st_tris = 0u;
    while (true) {   // fetch steps outside, node / triangle steps in the inner loop, one kind per iteration (kernels_trace.hip k_trace_vote)
        while (true) {
            VPT_MARK("vote");
            const bool busy = cur < kLaneDone;
            const bool at_node = busy && cur >= 0;
            const bool at_leaf = busy && cur < 0;
            const uint32_t nn = (uint32_t)__popcll(__ballot(at_node)), nl = (uint32_t)__popcll(__ballot(at_leaf));
            if ((!exhausted && 64u - nn - nl >= fetch_at) || nn + nl == 0u) break;
            const bool node_wins = weighted ? 4u * nn > w4 * nl : nn >= nl;
            VPT_MARK("node");   // two predicated regions in sequence, not if / else (see k_trace_vote)
            if (node_wins & at_node) {
                if (COUNT) st_nodes++;
                vote_node_step<true>(nodes, top, S, cur, sp, o, inv, tmin, tlim);
            }
            VPT_MARK("tri");
            if (!node_wins & at_leaf) {
                if (COUNT) st_tris++;
                if (TRI2) { if (vote_tri2_step_any(tris, S, cur, sp, o, d, tmin, tmax, tlim, expect)) visible = false; }
                else if (vote_tri_step_any<STRICT>(tris, S, cur, sp, o, d, tmin, tmax, tlim, expect)) visible = false;
            }
        }
        VPT_MARK("exit");
        if (exhausted) break;
        VPT_MARK("fetch");
        if (cur == kLaneDone) { vis[rid] = visible ? 1 : 0; cur = kLaneIdle; }
        if (w_next >= w_end) {
            if (n_static >= n) exhausted = true;
            else {
                uint32_t base = 0u;
                if (lane_id() == 0u) base = atomicAdd(head, chunk);
                base = n_static + __builtin_amdgcn_readfirstlane(base);
                if (base >= n) exhausted = true;
                else { w_next = base; w_end = base + chunk < n ? base + chunk : n; }
            }
        }
        if (!exhausted) {
            const unsigned long long m_idle = __ballot(cur == kLaneIdle);
            const uint32_t i = w_next + lanes_below(m_idle);
            if (cur == kLaneIdle && i < w_end) {
                const float4 rd = ld_stream(&RD[i]);
                expect = __float_as_uint(rd.z);
                if (expect != kRayHole) {
                    const float4 ro = ld_stream(&RO[i]);
                    rid = i;
                    o = vptfp::v3(ro.x, ro.y, ro.z); d = vptfp::v3(ro.w, rd.x, rd.y);
                    if (!ray_queries) d = vptfp::normalize(d);
                    inv = safe_inverse(d);
                    tlim = tmax; visible = true;  // until an occluder / a closer triangle is found
                    sp = 0; cur = 0;
                    if (LIGHT) {
                        const uint32_t slot = sc.tri_slot_of_gid[expect];
                        bool hit_it = false;
                        if (slot != 0xffffffffu) {  // 0xffffffff: the sampled light triangle is a sliver, nothing can hit it
                            const float4* q = reinterpret_cast<const float4*>(tris + slot);
                            const float4 ta = q[0], tb = q[1], tc = q[2];
                            if (COUNT) st_tris++;
                            float u, v;
                            hit_it = vptfp::ray_triangle(o, d, vptfp::v3(ta.x, ta.y, ta.z), vptfp::v3(ta.w, tb.x, tb.y), vptfp::v3(tb.z, tb.w, tc.x), tmin, tmax, &tlim, &u, &v);
                            if (STRICT) { if (hit_it) hit_it = vptfp::hit_is_local(o, d, vptfp::v3(ta.x, ta.y, ta.z), vptfp::v3(ta.w, tb.x, tb.y), vptfp::v3(tb.z, tb.w, tc.x), tlim); }   // traverse.hpp closest_is
                        }
                        if (!hit_it) {
                            // media NEE (RayGen.slang:296-299 compares a miss as "hit (0, 0)"): a ray flagged in RD.w is also visible when
                            // it hits nothing at all — then this becomes a plain occlusion query; otherwise the sample is not visible
                            if (rd.w != 0.0f) { tlim = tmax; expect = 0xffffffffu; }
                            else { visible = false; cur = kLaneDone; }
                        }
                    }
                }
            }
            const uint32_t want = (uint32_t)__popcll(m_idle), left = w_end - w_next;
            w_next += want < left ? want : left;
        }
    }
    VPT_MARK("done");
    if (cur == kLaneDone) vis[rid] = visible ? 1 : 0;
    if (COUNT) {
        atomicAdd(&ctr->stat_shadow_nodes, (unsigned long long)st_nodes);
        atomicAdd(&ctr->stat_shadow_tris, (unsigned long long)st_tris);
    }
}

// ------------------------------------------------------------------ launch
// The PRODUCT library holds four instantiations of k_trace_vote (closest hit: default, counting, validating, validating + counting — the
// pipeline always runs the compile-time vote parameters) and eight of k_trace_shadow.  Everything else — the baseline loop, the eight-wide
// tree, culling, packed arithmetic, ray pools, ray pairs, the one-triangle step for the A/B, run-time vote parameters, any-hit k_trace_vote —
// was measured slower (profiles/REJECTED.md) and lives in the LABORATORY build only (-DVPT_LAB=1: libvpt_hip_lab.so, include/vpt_lab.h, tests/tools/trace_lab.py).
int trace_blocks_per_cu(uint32_t variant, bool any) {
    int nb = 0;
#if VPT_LAB
    if (variant == VPT_TRACE_PAIR) {
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace_pair<false>, kTraverseBlock, kPairLdsBytes);
        return nb > 0 ? nb : 1;
    }
    if (variant == VPT_TRACE_POOL) {   // (the 128-slot form: 3 blocks per CU; the launch scales the grid for the smaller pools)
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace_pool<false, 128, 10, false>, kTraverseBlock, pool_lds_bytes(128, 10));
        return nb > 0 ? nb : 1;
    }
    const size_t lds_lab = variant == VPT_TRACE_BASE ? kVoteStackBytes : kVoteLdsBytes;
    if (variant == VPT_TRACE_BASE) {
        if (any) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace_base<true, false>, kTraverseBlock, lds_lab);
        else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace_base<false, false>, kTraverseBlock, lds_lab);
        return nb > 0 ? nb : 1;
    }
    if (variant == VPT_TRACE_VOTE4S) {
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace_vote<false, false, false, true, false, false, false, true, true>, kTraverseBlock, lds_lab);
        return nb > 0 ? nb : 1;
    }
    if (variant == VPT_TRACE_VOTE8) {
        if (any) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace_vote<true, false, true, false>, kTraverseBlock, lds_lab);
        else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace_vote<false, false, true, false>, kTraverseBlock, lds_lab);
        return nb > 0 ? nb : 1;
    }
    if (any) {
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace_vote<true, false, false, true, false, false, false, true>, kTraverseBlock, lds_lab);
        return nb > 0 ? nb : 1;
    }
#endif
    (void)variant; (void)any;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace_vote<false, false, false, true, false, false, false, true>, kTraverseBlock, kVoteLdsBytes);
    return nb > 0 ? nb : 1;
}

void launch_trace(hipStream_t s, uint32_t blocks, uint32_t variant, bool any, bool count, const DeviceScene& sc, const TraceArgs& a, Counters* ctr) {
#if VPT_LAB
    if (variant == VPT_TRACE_PAIR) {   // closest hit only
        if (count) hipLaunchKernelGGL((k_trace_pair<true>), dim3(blocks), dim3(kTraverseBlock), kPairLdsBytes, s, sc, a, ctr);
        else hipLaunchKernelGGL((k_trace_pair<false>), dim3(blocks), dim3(kTraverseBlock), kPairLdsBytes, s, sc, a, ctr);
        return;
    }
    if (variant == VPT_TRACE_POOL) {   // closest hit only (the lab refuses the any-hit form).  param bits 8-9: slots per wave / LDS stack entries 128/10, 96/10, 80/8, 64/8
        // (3, 4, 5, 6 blocks per CU: the grid — and with it the slot-indexed spill region — keeps blocks x slots constant); bit 10: DUAL
        const uint32_t cfg = (a.param >> 8) & 3u;
        const bool dual = ((a.param >> 10) & 1u) != 0u;
        const dim3 b(kTraverseBlock);
#define VPT_LP(SL, STK, G) do { const dim3 g(G); const size_t lds = pool_lds_bytes(SL, STK);                                                   \
            if (dual) { if (count) hipLaunchKernelGGL((k_trace_pool<true, SL, STK, true>), g, b, lds, s, sc, a, ctr);                               \
                        else hipLaunchKernelGGL((k_trace_pool<false, SL, STK, true>), g, b, lds, s, sc, a, ctr); }                                 \
            else { if (count) hipLaunchKernelGGL((k_trace_pool<true, SL, STK, false>), g, b, lds, s, sc, a, ctr);                                  \
                   else hipLaunchKernelGGL((k_trace_pool<false, SL, STK, false>), g, b, lds, s, sc, a, ctr); } } while (0)
        if (cfg == 0u) VPT_LP(128, 10, blocks);
        else if (cfg == 1u) VPT_LP(96, 10, blocks + blocks / 3u);
        else if (cfg == 2u) VPT_LP(80, 8, blocks + (blocks * 2u) / 3u);
        else VPT_LP(64, 8, blocks * 2u);
#undef VPT_LP
        return;
    }
    {
        const size_t lds = variant == VPT_TRACE_BASE ? kVoteStackBytes : kVoteLdsBytes;
        const dim3 g(blocks), b(kTraverseBlock);
#define VPT_LT(K) do { if (any) { if (count) hipLaunchKernelGGL((K<true, true>), g, b, lds, s, sc, a, ctr); else hipLaunchKernelGGL((K<true, false>), g, b, lds, s, sc, a, ctr); } \
                       else { if (count) hipLaunchKernelGGL((K<false, true>), g, b, lds, s, sc, a, ctr); else hipLaunchKernelGGL((K<false, false>), g, b, lds, s, sc, a, ctr); } } while (0)
#define VPT_LV(W, T) do { if (any) { if (count) hipLaunchKernelGGL((k_trace_vote<true, true, W, T>), g, b, lds, s, sc, a, ctr); else hipLaunchKernelGGL((k_trace_vote<true, false, W, T>), g, b, lds, s, sc, a, ctr); } \
                          else { if (count) hipLaunchKernelGGL((k_trace_vote<false, true, W, T>), g, b, lds, s, sc, a, ctr); else hipLaunchKernelGGL((k_trace_vote<false, false, W, T>), g, b, lds, s, sc, a, ctr); } } while (0)
        if (variant == VPT_TRACE_VOTE4S) {   // product vote parameters, two triangles per step; the counting form one.  Any-hit: the plain node step on the split-order tree (its order tables masked off)
            if (any) {
                if (count) hipLaunchKernelGGL((k_trace_vote<true, true, false, true, false, false, false, false, true>), g, b, lds, s, sc, a, ctr);
                else hipLaunchKernelGGL((k_trace_vote<true, false, false, true, false, false, false, true, true>), g, b, lds, s, sc, a, ctr);
            } else if (count) hipLaunchKernelGGL((k_trace_vote<false, true, false, true, false, false, false, false, true>), g, b, lds, s, sc, a, ctr);
            else hipLaunchKernelGGL((k_trace_vote<false, false, false, true, false, false, false, true, true>), g, b, lds, s, sc, a, ctr);
            return;
        }
        if (variant == VPT_TRACE_BASE) { VPT_LT(k_trace_base); return; }
        if (variant == VPT_TRACE_VOTE8) { VPT_LV(true, false); return; }
        if (!sc.strict_hits) {
            if (a.one_tri && !count) {   // trace lab bit 19: ONE triangle per triangle step (round 3's step) with the product vote parameters, for the A/B against the product's two
                if (any) hipLaunchKernelGGL((k_trace_vote<true, false, false, true, false, false, false, false>), g, b, lds, s, sc, a, ctr);
                else hipLaunchKernelGGL((k_trace_vote<false, false, false, true, false, false, false, false>), g, b, lds, s, sc, a, ctr);
                return;
            }
            if (a.packed) {   // trace lab bit 18: packed plane arithmetic in the node step, product vote parameters
                if (any) { if (count) hipLaunchKernelGGL((k_trace_vote<true, true, false, true, false, false, true>), g, b, lds, s, sc, a, ctr); else hipLaunchKernelGGL((k_trace_vote<true, false, false, true, false, false, true>), g, b, lds, s, sc, a, ctr); }
                else { if (count) hipLaunchKernelGGL((k_trace_vote<false, true, false, true, false, false, true>), g, b, lds, s, sc, a, ctr); else hipLaunchKernelGGL((k_trace_vote<false, false, false, true, false, false, true>), g, b, lds, s, sc, a, ctr); }
                return;
            }
            if (a.cull && !any) {   // stale-entry culling (closest-hit only)
                if (count) hipLaunchKernelGGL((k_trace_vote<false, true, false, false, false, true>), g, b, lds, s, sc, a, ctr);
                else if (a.param == kVoteParamDefault) hipLaunchKernelGGL((k_trace_vote<false, false, false, true, false, true>), g, b, lds, s, sc, a, ctr);
                else hipLaunchKernelGGL((k_trace_vote<false, false, false, false, false, true>), g, b, lds, s, sc, a, ctr);
                return;
            }
            if (a.param != kVoteParamDefault) { VPT_LV(false, false); return; }   // run-time vote parameters
            if (any) {   // (the pipeline's shadow rays go through k_trace_shadow: the any-hit form of k_trace_vote exists for the lab's ray sets)
                if (count) hipLaunchKernelGGL((k_trace_vote<true, true, false, true>), g, b, lds, s, sc, a, ctr);
                else hipLaunchKernelGGL((k_trace_vote<true, false, false, true, false, false, false, true>), g, b, lds, s, sc, a, ctr);
                return;
            }
        } else if (any) {
            if (count) hipLaunchKernelGGL((k_trace_vote<true, true, false, true, true>), g, b, lds, s, sc, a, ctr);
            else hipLaunchKernelGGL((k_trace_vote<true, false, false, true, true>), g, b, lds, s, sc, a, ctr);
            return;
        }
#undef VPT_LV
#undef VPT_LT
    }
#endif
    // ---- the product's closest-hit search: compile-time vote parameters; two triangles per triangle step unless visits are counted or hits validated
    (void)variant; (void)any;
    const dim3 g(blocks), b(kTraverseBlock);
    if (sc.strict_hits) {   // VPT_FLAG_LOCAL_HITS: the validating instantiations
        if (count) hipLaunchKernelGGL((k_trace_vote<false, true, false, true, true>), g, b, kVoteLdsBytes, s, sc, a, ctr);
        else hipLaunchKernelGGL((k_trace_vote<false, false, false, true, true>), g, b, kVoteLdsBytes, s, sc, a, ctr);
    } else if (count) hipLaunchKernelGGL((k_trace_vote<false, true, false, true>), g, b, kVoteLdsBytes, s, sc, a, ctr);
    else hipLaunchKernelGGL((k_trace_vote<false, false, false, true, false, false, false, true>), g, b, kVoteLdsBytes, s, sc, a, ctr);
}

void launch_trace_shadow(hipStream_t s, uint32_t blocks, bool light, bool count, const DeviceScene& sc, const StreamState& ss, Counters* ctr,
                         StreamCounters* sctr, uint32_t param, uint32_t ray_queries) {
    const size_t lds = kVoteLdsBytes;
    const dim3 g(blocks), b(kTraverseBlock);
    const float4 *RO = light ? ss.LTO : ss.SKO, *RD = light ? ss.LTD : ss.SKD;
    unsigned char* vis = light ? ss.vis_light : ss.vis_sky;
    uint32_t *len = light ? &sctr->light_len.v : &sctr->sky_len.v, *head = light ? &sctr->light_head.v : &sctr->sky_head.v;
#define VPT_LS(L, C, T, ST, T2) hipLaunchKernelGGL((k_trace_shadow<L, C, T, ST, T2>), g, b, lds, s, sc, RO, RD, vis, len, head, ctr, param, ray_queries)
#if VPT_LAB
    if (param != kVoteParamDefault && !sc.strict_hits) {   // run-time vote parameters (trace-lab sweeps through the pipeline)
        if (light) { if (count) VPT_LS(true, true, false, false, false); else VPT_LS(true, false, false, false, false); }
        else { if (count) VPT_LS(false, true, false, false, false); else VPT_LS(false, false, false, false, false); }
        return;
    }
#endif
    // the product instantiations: compile-time vote parameters; two triangles per triangle step unless visits are counted or hits validated
    // (VPT_FLAG_LOCAL_HITS with vpt_config.count_traversal runs the counting + validating one, so the visit statistics are not silently zero)
    if (sc.strict_hits) {
        if (light) { if (count) VPT_LS(true, true, true, true, false); else VPT_LS(true, false, true, true, false); }
        else { if (count) VPT_LS(false, true, true, true, false); else VPT_LS(false, false, true, true, false); }
    } else if (count) {
        if (light) VPT_LS(true, true, true, false, false); else VPT_LS(false, true, true, false, false);
    } else {
        if (light) VPT_LS(true, false, true, false, true); else VPT_LS(false, false, true, false, true);
    }
#undef VPT_LS
}
int trace_shadow_blocks_per_cu() {
    int a = 0, b = 0;
    const size_t lds = kVoteLdsBytes;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, k_trace_shadow<true, false, true, false, true>, kTraverseBlock, lds);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, k_trace_shadow<false, false, true, false, true>, kTraverseBlock, lds);
    int nb = a < b ? a : b;
    return nb > 0 ? nb : 1;
}

}  // namespace vpt
