// vote.hpp — building blocks of the vote-scheduled persistent traversal kernels (kernels_trace.hip, kernels_stream.hip):
// the per-lane stack and ONE inner-node visit / ONE triangle test of a lane's ray, written so that a wave can execute a
// single kind of step per iteration for the lanes that want it.
#pragma once
#include "traverse.hpp"
#include "wave.hpp"

namespace vpt {

constexpr int kLaneIdle = 0x7fffffff;  // the lane holds no ray
constexpr int kLaneDone = 0x7ffffffe;  // the lane's ray is finished, its result not yet written
constexpr uint32_t kVoteParamDefault = 256u + 16u;  // weighted vote, fetch step at 16 idle lanes (profiles/r02_trace_lab_*.json)
constexpr uint32_t kVoteFetchAt = 24u;  // idle lanes that trigger a fetch step in the product instantiation (Msamples/s at 16 / 24 / 32: atrium 1331 / 1351 / 1343, glass bust 2907 / 2938 / 2938)
constexpr uint32_t kVoteWeight4 = 8u;   // a node step wins the vote when 4 x (lanes at nodes) > kVoteWeight4 x (lanes in leaves)
constexpr uint32_t kRayHole = 0xfffffffeu;  // SKD.z / LTD.z of a shadow-ray stream entry nobody wrote
constexpr uint32_t kHole = 0xffffffffu;  // a stream entry nobody wrote (tail of a wave's last chunk, see WaveAppender)

constexpr int kVoteStackRows = kStackDepth;   // LDS rows of the vote kernels' stacks
constexpr size_t kVoteStackBytes = (size_t)kVoteStackRows * kTraverseBlock * 4;
// Stale-entry culling (closest-hit search, CULL instantiations): next to every LDS stack entry sits one byte, the entry distance of
// the pushed child quantised DOWN (sign-free float bits >> 20: 3 mantissa bits, 12.5 % steps; clamped into a byte).  When the ray has
// found a closer hit since the push, a pop sees `byte > quantised best_t` — which implies entry distance > best_t, strictly — and drops
// the entry instead of spending a whole node step (four slab tests) on finding out that nothing in it can win.  A pruned node holds no
// candidate with t <= best_t, so hits (ties included) cannot change.  Spilled entries carry no byte and are never culled.
constexpr uint32_t kCullBias = 900u;   // float bits >> 20 of 2^-8 .. 2^23 map to 902 .. 1150; the byte clamps below and above (conservatively)
__device__ __forceinline__ uint32_t cull_quant(float t) {
    const uint32_t b = __float_as_uint(t) >> 20;
    const uint32_t q = b > kCullBias ? b - kCullBias : 0u;
    return q < 255u ? q : 255u;
}
struct LaneStack {
    uint32_t* stk;  // LDS: entry k of this lane at stk[k * kTraverseBlock]
    uint32_t* ovf;  // global: entries beyond kStackDepth
    unsigned char* tq;  // LDS (CULL kernels only): byte k of this lane at tq[k * kTraverseBlock]
    __device__ __forceinline__ void push_t(int& sp, int v, float t) const {
        if (sp < kStackDepth) { stk[sp * kTraverseBlock] = (uint32_t)v; tq[sp * kTraverseBlock] = (unsigned char)cull_quant(t); }
        else if (sp < kStackDepth + kStackOverflow) ovf[sp - kStackDepth] = (uint32_t)v;
        sp++;
    }
    __device__ __forceinline__ void pop_or_done_cull(int& sp, int& cur, float best_t) const {
        const uint32_t qb = cull_quant(best_t);
        while (true) {
            if (sp == 0) { cur = kLaneDone; return; }
            sp--;
            if (sp < kStackDepth) {
                if ((uint32_t)tq[sp * kTraverseBlock] > qb) continue;   // entered beyond the best hit found since: nothing in it can win
                cur = (int)stk[sp * kTraverseBlock];
                return;
            }
            cur = (int)ovf[sp - kStackDepth];
            return;
        }
    }
    __device__ __forceinline__ void push(int& sp, int v) const {
        if (sp < kStackDepth) stk[sp * kTraverseBlock] = (uint32_t)v;
        else if (sp < kStackDepth + kStackOverflow) ovf[sp - kStackDepth] = (uint32_t)v;
        sp++;
    }
    // The LDS read is unconditional (row clamped) and the spill read sits behind its own rare branch: written as one
    // `sp < kStackDepth ? stk[..] : ovf[..]` the two loads are merged into a flat_load of a selected address, which is
    // slower than ds_read_b32 and waits on both memory counters.
    __device__ __forceinline__ void pop_or_done(int& sp, int& cur) const {
        if (sp == 0) cur = kLaneDone;
        else {
            sp--;
            int v = (int)stk[(sp < kStackDepth ? sp : kStackDepth - 1) * kTraverseBlock];
            asm volatile("" : "+v"(v));   // keeps the two loads apart (the optimiser would otherwise select between the addresses again)
            if (sp >= kStackDepth) v = (int)ovf[sp - kStackDepth];
            cur = v;
        }
    }
};
__device__ __forceinline__ LaneStack make_lane_stack(unsigned char* smem, uint32_t* overflow) {
    LaneStack S;
    S.stk = reinterpret_cast<uint32_t*>(smem) + threadIdx.x;
    S.ovf = overflow + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * kStackOverflow;
    S.tq = smem + kVoteStackBytes + threadIdx.x;   // the bytes live where the any-hit kernels keep their tree top (kVoteTopBytes >= kStackDepth * kTraverseBlock)
    return S;
}
// The block's LDS copy of the tree top (device_types.hpp kBvhTopNodes) sits behind the stacks.  A node index below `top` is fetched
// from it through a GENERIC pointer — one flat_load per 16-byte piece, the lane's address decides between LDS and the vector L1 —
// so the lanes at the top of the tree, which every ray passes, cost the L1 nothing.  Why: the traversal kernels sit at ~0.9 L1
// accesses per clock per CU with the texture-address unit 90-96 % busy (profiles/r03_atrium_p2_f129_summary.md): the L1 takes one
// per-lane access per clock, four per node visit (profiles/r03_trace_isa_budget.md, tests/tools/gather_calib.hip mode E).
constexpr size_t kVoteTopBytes = (size_t)kBvhTopNodes * sizeof(BvhNode);
constexpr size_t kVoteLdsBytes = kVoteStackBytes + kVoteTopBytes;
static_assert(kVoteTopBytes >= (size_t)kStackDepth * kTraverseBlock, "the culling bytes of the closest-hit kernels share the tree-top region");
struct TreeTop {
    const uint4* lds;   // generic pointer into LDS
    int count;          // nodes held there (0: none — the lab's other variants)
};
__device__ __forceinline__ TreeTop stage_tree_top(unsigned char* smem, const BvhNode* nodes, uint32_t node_count, bool enable) {
    TreeTop t;
    uint4* dst = reinterpret_cast<uint4*>(smem + kVoteStackBytes);
    t.lds = dst;
    t.count = enable ? (int)(node_count < (uint32_t)kBvhTopNodes ? node_count : (uint32_t)kBvhTopNodes) : 0;
    const uint4* src = reinterpret_cast<const uint4*>(nodes);
    for (int i = (int)threadIdx.x; i < t.count * 4; i += (int)blockDim.x) dst[i] = src[i];
    __syncthreads();
    return t;
}

// One inner-node visit: 64 B fetch, four slab tests against [tmin, tlimit], then either the nearest hit child with the
// others pushed far -> near (closest-hit search) or the hit children in slot order (any-hit search).  (Measured and not kept: branch-free pushes through
// a trash row — closest-hit -8 %, any-hit +2 %; farthest-child-first for the light-identity queries — shadow stage -25 %.)
// USE_TOP: only the any-hit kernels use the LDS tree top.  Measured (atrium / glass bust, bench kernel means): shadow stage 4.46 -> 3.94 ms
// and 0.318 -> 0.311 ms with it — those kernels are L1-bound with VALU issue to spare (VALUBusy 82-85 %); the closest-hit extend kernel,
// VALU-saturated, pays for the six instructions of the address select: 6.37 -> 6.71 ms and 1.12 -> 1.19 ms, so it keeps the plain load.
// MASKED: the node comes from the split-order tree, whose step_x / step_y carry order tables in their mantissas (vote_node4s_step below): two v_and.
template <bool ANY, bool USE_TOP = ANY, bool CULL = false, class STK = LaneStack, bool PK = false, bool MASKED = false>
__device__ __forceinline__ void vote_node_step(const BvhNode* nodes, const TreeTop& top, const STK& S, int& cur, int& sp, V3 o, V3 inv, float tmin, float tlimit) {
    const uint4* p = (USE_TOP && cur < top.count) ? top.lds + cur * 4 : reinterpret_cast<const uint4*>(nodes + cur);
    NodeData n;
    if (MASKED) { uint4 a = p[0], c = p[2]; a.w &= 0x7f800000u; c.z &= 0x7f800000u; unpack_node(a, p[1], c, p[3], n); }
    else unpack_node(p[0], p[1], p[2], p[3], n);
    RaySlab slab; slab.o = o; slab.inv = inv;
    slab.negx = inv.x < 0.0f; slab.negy = inv.y < 0.0f; slab.negz = inv.z < 0.0f;
    float t0, t1, t2, t3;
    if (PK) node_entries_pk(n, slab, tmin, tlimit, t0, t1, t2, t3);   // (trace lab, bit 18: the plane arithmetic in packed instructions)
    else node_entries(n, slab, tmin, tlimit, t0, t1, t2, t3);
    int c0 = n.c0, c1 = n.c1, c2 = n.c2, c3 = n.c3;
    if (ANY) {   // order is irrelevant for an any-hit search: hit children in slot order
        int next = kLaneIdle;
        if (t3 < kMissT) next = c3;
        if (t2 < kMissT) { if (next != kLaneIdle) S.push(sp, next); next = c2; }
        if (t1 < kMissT) { if (next != kLaneIdle) S.push(sp, next); next = c1; }
        if (t0 < kMissT) { if (next != kLaneIdle) S.push(sp, next); next = c0; }
        if (next != kLaneIdle) cur = next; else S.pop_or_done(sp, cur);
    } else {
        cswap(t0, c0, t1, c1); cswap(t2, c2, t3, c3); cswap(t0, c0, t2, c2); cswap(t1, c1, t3, c3); cswap(t1, c1, t2, c2);
        if (t0 < kMissT) {  // nearest child next, the others pushed far -> near
            if (CULL) {
                if (t3 < kMissT) S.push_t(sp, c3, t3);
                if (t2 < kMissT) S.push_t(sp, c2, t2);
                if (t1 < kMissT) S.push_t(sp, c1, t1);
            } else {
                if (t3 < kMissT) S.push(sp, c3);
                if (t2 < kMissT) S.push(sp, c2);
                if (t1 < kMissT) S.push(sp, c1);
            }
            cur = c0;
        } else {
            if (CULL) S.pop_or_done_cull(sp, cur, tlimit); else S.pop_or_done(sp, cur);
        }
    }
}
// ---- split-order experiment (trace lab VPT_TRACE_VOTE4S, closest hit): one visit of a four-wide node of the split-order tree
// (bvh_build.hpp BvhBuildOptions::nodes4s).  The four slab tests as above, but instead of sorting the entry distances (five compare-exchanges
// on distance + child code, 25 VALU) the hit children are visited in the order the node's binary splits give for the ray's direction octant:
// three table bits (bit `oct` of two bytes in step_x's mantissa, of one in step_y's) drive a three-exchange butterfly on the child codes and
// the hit flags.  A closest-hit result does not depend on the order (ties in t go to the smaller global id); the order only decides how soon
// best_t shrinks.  oct = negx | negy << 1 | negz << 2, kept per ray by the caller.
__device__ __forceinline__ void vote_node4s_step(const BvhNode* nodes, const LaneStack& S, int& cur, int& sp, V3 o, V3 inv, uint32_t oct, float tmin, float tlimit) {
    const uint4* p = reinterpret_cast<const uint4*>(nodes + cur);
    const uint4 w0 = p[0], w1 = p[1], w2 = p[2], w3 = p[3];
    NodeData n;
    unpack_node(make_uint4(w0.x, w0.y, w0.z, w0.w & 0x7f800000u), w1, make_uint4(w2.x, w2.y, w2.z & 0x7f800000u, w2.w), w3, n);
    RaySlab slab; slab.o = o; slab.inv = inv;
    slab.negx = inv.x < 0.0f; slab.negy = inv.y < 0.0f; slab.negz = inv.z < 0.0f;
    float t0, t1, t2, t3;
    node_entries(n, slab, tmin, tlimit, t0, t1, t2, t3);
    bool h0 = t0 < kMissT, h1 = t1 < kMissT, h2 = t2 < kMissT, h3 = t3 < kMissT;
    int c0 = n.c0, c1 = n.c1, c2 = n.c2, c3 = n.c3;
    const bool swl = ((w0.w >> oct) & 1u) != 0u, swr = ((w0.w >> (oct + 8u)) & 1u) != 0u, swt = ((w2.z >> oct) & 1u) != 0u;
#define VPT_SWAPI(C, A, B) { const int t_ = (C) ? B : A; B = (C) ? A : B; A = t_; }
#define VPT_SWAPB(C, A, B) { const bool t_ = (C) ? B : A; B = (C) ? A : B; A = t_; }
    VPT_SWAPI(swl, c0, c1) VPT_SWAPB(swl, h0, h1)
    VPT_SWAPI(swr, c2, c3) VPT_SWAPB(swr, h2, h3)
    VPT_SWAPI(swt, c0, c2) VPT_SWAPB(swt, h0, h2)
    VPT_SWAPI(swt, c1, c3) VPT_SWAPB(swt, h1, h3)
#undef VPT_SWAPI
#undef VPT_SWAPB
    if (!(h0 | h1 | h2 | h3)) { S.pop_or_done(sp, cur); return; }
    // every hit child except the first is pushed, last first
    if (h3 & (h0 | h1 | h2)) S.push(sp, c3);
    if (h2 & (h0 | h1)) S.push(sp, c2);
    if (h1 & h0) S.push(sp, c1);
    cur = h0 ? c0 : h1 ? c1 : h2 ? c2 : c3;
}
// ---- BVH8 experiment: one visit of an eight-wide node (device_types.hpp BvhNode8).  Eight slab tests on the shared grid, then
// the hit children in OCTANT order — slot XOR (sign bits of the ray direction), ascending — instead of a distance sort: the
// children and the hit mask are permuted into that order with three conditional butterfly stages, the first hit child is
// visited next, the others are pushed so that they pop in the same order.
__device__ __forceinline__ void vote_node8_step(const BvhNode8* nodes, const LaneStack& S, int& cur, int& sp, V3 o, V3 inv, float tmin, float tlimit) {
    const uint4* p = reinterpret_cast<const uint4*>(nodes + cur);
    const uint4 w0 = p[0], w1 = p[1], w2 = p[2], w3 = p[3], w4 = p[4], w5 = p[5];
    const bool negx = inv.x < 0.0f, negy = inv.y < 0.0f, negz = inv.z < 0.0f;
    const float ax = __uint_as_float((w0.w & 0xffu) << 23) * inv.x;
    const float ay = __uint_as_float(((w0.w >> 8) & 0xffu) << 23) * inv.y;
    const float az = __uint_as_float(((w0.w >> 16) & 0xffu) << 23) * inv.z;
    const float bx = (__uint_as_float(w0.x) - o.x) * inv.x, by = (__uint_as_float(w0.y) - o.y) * inv.y, bz = (__uint_as_float(w0.z) - o.z) * inv.z;
    // lo: x = (w1.x, w1.y), y = (w1.z, w1.w), z = (w2.x, w2.y); hi: x = (w2.z, w2.w), y = (w3.x, w3.y), z = (w3.z, w3.w)
    uint32_t hits = 0u;
#define VPT_HALF(NX, FX, NY, FY, NZ, FZ, BASE)                                                                                        \
    {                                                                                                                                 \
        const uint32_t nx = negx ? FX : NX, fx = negx ? NX : FX, ny = negy ? FY : NY, fy = negy ? NY : FY, nz = negz ? FZ : NZ, fz = negz ? NZ : FZ; \
        _Pragma("unroll") for (int k = 0; k < 4; k++) {                                                                               \
            const float tn = fmax_(fmax_(__builtin_fmaf((float)((nx >> (8 * k)) & 0xffu), ax, bx), __builtin_fmaf((float)((ny >> (8 * k)) & 0xffu), ay, by)), \
                                   fmax_(__builtin_fmaf((float)((nz >> (8 * k)) & 0xffu), az, bz), tmin));                             \
            const float tf = fmin_(fmin_(__builtin_fmaf((float)((fx >> (8 * k)) & 0xffu), ax, bx), __builtin_fmaf((float)((fy >> (8 * k)) & 0xffu), ay, by)), \
                                   fmin_(__builtin_fmaf((float)((fz >> (8 * k)) & 0xffu), az, bz), tlimit));                           \
            hits |= (tn <= tf * 1.0000005f ? 1u : 0u) << (BASE + k);                                                                  \
        }                                                                                                                             \
    }
    VPT_HALF(w1.x, w2.z, w1.z, w3.x, w2.x, w3.z, 0)
    VPT_HALF(w1.y, w2.w, w1.w, w3.y, w2.y, w3.w, 4)
#undef VPT_HALF
    // children and hit bits into key order: key = slot ^ (negx | negy << 1 | negz << 2)
    int c0 = (int)w4.x, c1 = (int)w4.y, c2 = (int)w4.z, c3 = (int)w4.w, c4 = (int)w5.x, c5 = (int)w5.y, c6 = (int)w5.z, c7 = (int)w5.w;
#define VPT_SWAP(C, A, B) { const int t_ = (C) ? B : A; B = (C) ? A : B; A = t_; }
    VPT_SWAP(negx, c0, c1) VPT_SWAP(negx, c2, c3) VPT_SWAP(negx, c4, c5) VPT_SWAP(negx, c6, c7)
    VPT_SWAP(negy, c0, c2) VPT_SWAP(negy, c1, c3) VPT_SWAP(negy, c4, c6) VPT_SWAP(negy, c5, c7)
    VPT_SWAP(negz, c0, c4) VPT_SWAP(negz, c1, c5) VPT_SWAP(negz, c2, c6) VPT_SWAP(negz, c3, c7)
#undef VPT_SWAP
    if (negx) hits = ((hits & 0x55u) << 1) | ((hits & 0xaau) >> 1);
    if (negy) hits = ((hits & 0x33u) << 2) | ((hits & 0xccu) >> 2);
    if (negz) hits = ((hits & 0x0fu) << 4) | ((hits & 0xf0u) >> 4);
    if (hits == 0u) { S.pop_or_done(sp, cur); return; }
    // every hit child except the first (lowest key) is pushed, highest key first
    if ((hits & 0x80u) && (hits & 0x7fu)) S.push(sp, c7);
    if ((hits & 0x40u) && (hits & 0x3fu)) S.push(sp, c6);
    if ((hits & 0x20u) && (hits & 0x1fu)) S.push(sp, c5);
    if ((hits & 0x10u) && (hits & 0x0fu)) S.push(sp, c4);
    if ((hits & 0x08u) && (hits & 0x07u)) S.push(sp, c3);
    if ((hits & 0x04u) && (hits & 0x03u)) S.push(sp, c2);
    if ((hits & 0x02u) && (hits & 0x01u)) S.push(sp, c1);
    cur = (hits & 1u) ? c0 : (hits & 2u) ? c1 : (hits & 4u) ? c2 : (hits & 8u) ? c3 : (hits & 16u) ? c4 : (hits & 32u) ? c5 : (hits & 64u) ? c6 : c7;
}

// vptfp::ray_triangle() without its early returns: the same operations in the same order (the build has no contraction and no
// fast-math, so u, v, t are the contract's values bit for bit wherever the contract accepts the hit), and ONE predicate at the
// end.  With det == 0 the reciprocal is an infinity, u, v, t are infinities or NaNs and the explicit det test rejects the hit as
// the contract does.  Why: in a vote step a wave runs the whole test anyway as soon as one lane passes each early-out, and the
// nested exits made the compiler carry the best-hit registers through five nested exec regions (50 register copies per triangle
// step, profiles/r03_trace_isa_budget.md); straight-line code has none.
__device__ __forceinline__ bool ray_triangle_flat(V3 o, V3 d, V3 v0, V3 e1, V3 e2, float tmin, float tmax, float& t, float& u, float& v) {
    const V3 p = vptfp::cross(d, e2);
    const float det = vptfp::dot(e1, p);
    const float inv = 1.0f / det;
    const V3 s = o - v0;
    u = vptfp::dot(s, p) * inv;
    const V3 q = vptfp::cross(s, e1);
    v = vptfp::dot(d, q) * inv;
    t = vptfp::dot(e2, q) * inv;
    return (det != 0.0f) & (u >= 0.0f) & (u <= 1.0f) & (v >= 0.0f) & (u + v <= 1.0f) & (t > tmin) & (t < tmax);
}

// One triangle of the lane's leaf, closest-hit search (ties in t -> smaller global id).  STRICT (VPT_FLAG_LOCAL_HITS): triangles
// ex0 / ex1 — winners of earlier passes over this ray whose hit was not local to them — are not candidates (traverse.hpp
// trace_closest_strict).
template <bool STRICT = false, bool CULL = false>
__device__ __forceinline__ void vote_tri_step_closest(const BvhTri* tris, const LaneStack& S, int& cur, int& sp, V3 o, V3 d, float tmin, float tmax,
                                                      float& best_t, float& bu, float& bv, uint32_t& bslot, uint32_t& bgid, uint32_t ex0 = 0xffffffffu,
                                                      uint32_t ex1 = 0xffffffffu) {
    const uint32_t enc = (uint32_t)(~cur);
    const int first = (int)(enc >> 3);
    const uint32_t more = enc & 7u;  // triangles left after this one
    const float4* q = reinterpret_cast<const float4*>(tris + first);
    const float4 ta = q[0], tb = q[1], tc = q[2];
    float t, u, v;
    const bool hit = ray_triangle_flat(o, d, vptfp::v3(ta.x, ta.y, ta.z), vptfp::v3(ta.w, tb.x, tb.y), vptfp::v3(tb.z, tb.w, tc.x), tmin, tmax, t, u, v);
    const uint32_t gid = __float_as_uint(tc.w);
    const bool better = hit & (!STRICT | ((gid != ex0) & (gid != ex1))) & ((bslot == 0xffffffffu) | (t < best_t) | ((t == best_t) & (gid < bgid)));
    best_t = better ? t : best_t; bu = better ? u : bu; bv = better ? v : bv; bslot = better ? (uint32_t)first : bslot; bgid = better ? gid : bgid;
    if (more) cur = ~(int)((((uint32_t)first + 1u) << 3) | (more - 1u));
    else if (CULL) S.pop_or_done_cull(sp, cur, best_t);
    else S.pop_or_done(sp, cur);
}
// UP TO TWO triangles of the lane's leaf in one step (the product instantiations of the closest-hit and the shadow kernels; trace-lab bit 19 selects the one-triangle step for the A/B): the second one's record is fetched with the first (a leaf's triangles are
// consecutive) and tested behind it with the best hit the first left, i.e. exactly what two single steps would have done.
__device__ __forceinline__ void vote_tri2_step_closest(const BvhTri* tris, const LaneStack& S, int& cur, int& sp, V3 o, V3 d, float tmin, float tmax,
                                                       float& best_t, float& bu, float& bv, uint32_t& bslot, uint32_t& bgid) {
    const uint32_t enc = (uint32_t)(~cur);
    const int first = (int)(enc >> 3);
    const uint32_t more = enc & 7u;  // triangles left after the first
    const float4* q = reinterpret_cast<const float4*>(tris + first);
    const float4* q2 = reinterpret_cast<const float4*>(tris + first + (more ? 1 : 0));
    const float4 ta = q[0], tb = q[1], tc = q[2], ua = q2[0], ub = q2[1], uc = q2[2];
    {
        float t, u, v;
        const bool hit = ray_triangle_flat(o, d, vptfp::v3(ta.x, ta.y, ta.z), vptfp::v3(ta.w, tb.x, tb.y), vptfp::v3(tb.z, tb.w, tc.x), tmin, tmax, t, u, v);
        const uint32_t gid = __float_as_uint(tc.w);
        const bool better = hit & ((bslot == 0xffffffffu) | (t < best_t) | ((t == best_t) & (gid < bgid)));
        best_t = better ? t : best_t; bu = better ? u : bu; bv = better ? v : bv; bslot = better ? (uint32_t)first : bslot; bgid = better ? gid : bgid;
    }
    if (more) {
        float t, u, v;
        const bool hit = ray_triangle_flat(o, d, vptfp::v3(ua.x, ua.y, ua.z), vptfp::v3(ua.w, ub.x, ub.y), vptfp::v3(ub.z, ub.w, uc.x), tmin, tmax, t, u, v);
        const uint32_t gid = __float_as_uint(uc.w);
        const bool better = hit & ((bslot == 0xffffffffu) | (t < best_t) | ((t == best_t) & (gid < bgid)));
        best_t = better ? t : best_t; bu = better ? u : bu; bv = better ? v : bv; bslot = better ? (uint32_t)first + 1u : bslot; bgid = better ? gid : bgid;
    }
    if (more >= 2u) cur = ~(int)((((uint32_t)first + 2u) << 3) | (more - 2u));
    else S.pop_or_done(sp, cur);
}
__device__ __forceinline__ bool vote_tri2_step_any(const BvhTri* tris, const LaneStack& S, int& cur, int& sp, V3 o, V3 d, float tmin, float tmax, float tlim, uint32_t expect) {
    const uint32_t enc = (uint32_t)(~cur);
    const int first = (int)(enc >> 3);
    const uint32_t more = enc & 7u;
    const float4* q = reinterpret_cast<const float4*>(tris + first);
    const float4* q2 = reinterpret_cast<const float4*>(tris + first + (more ? 1 : 0));
    const float4 ta = q[0], tb = q[1], tc = q[2], ua = q2[0], ub = q2[1], uc = q2[2];
    float t, u, v;
    bool hit = ray_triangle_flat(o, d, vptfp::v3(ta.x, ta.y, ta.z), vptfp::v3(ta.w, tb.x, tb.y), vptfp::v3(tb.z, tb.w, tc.x), tmin, tmax, t, u, v);
    bool stop = hit & ((t < tlim) | ((t == tlim) & (__float_as_uint(tc.w) < expect)));
    if (more) {
        hit = ray_triangle_flat(o, d, vptfp::v3(ua.x, ua.y, ua.z), vptfp::v3(ua.w, ub.x, ub.y), vptfp::v3(ub.z, ub.w, uc.x), tmin, tmax, t, u, v);
        stop = stop | (hit & ((t < tlim) | ((t == tlim) & (__float_as_uint(uc.w) < expect))));
    }
    if (stop) { cur = kLaneDone; return true; }
    if (more >= 2u) cur = ~(int)((((uint32_t)first + 2u) << 3) | (more - 2u));
    else S.pop_or_done(sp, cur);
    return false;
}

// One triangle of the lane's leaf, any-hit search: stops at the first triangle hit with t < tlim, or t == tlim and a
// smaller global id than `expect` (traverse.hpp: with tlim = tmax this is plain occlusion; with tlim = t_e of the sampled
// light triangle it decides "is the closest hit that triangle").  Returns true when the search is over.  STRICT: a triangle stops
// the search only if its hit is local to it (vpt_fp32.h hit_is_local; validated on the spot, stops are rare).
template <bool STRICT = false>
__device__ __forceinline__ bool vote_tri_step_any(const BvhTri* tris, const LaneStack& S, int& cur, int& sp, V3 o, V3 d, float tmin, float tmax,
                                                  float tlim, uint32_t expect) {
    const uint32_t enc = (uint32_t)(~cur);
    const int first = (int)(enc >> 3);
    const uint32_t more = enc & 7u;
    const float4* q = reinterpret_cast<const float4*>(tris + first);
    const float4 ta = q[0], tb = q[1], tc = q[2];
    float t, u, v;
    const bool hit = ray_triangle_flat(o, d, vptfp::v3(ta.x, ta.y, ta.z), vptfp::v3(ta.w, tb.x, tb.y), vptfp::v3(tb.z, tb.w, tc.x), tmin, tmax, t, u, v);
    const uint32_t gid = __float_as_uint(tc.w);
    bool stop = hit & ((t < tlim) | ((t == tlim) & (gid < expect)));
    if (STRICT) { if (stop) stop = vptfp::hit_is_local(o, d, vptfp::v3(ta.x, ta.y, ta.z), vptfp::v3(ta.w, tb.x, tb.y), vptfp::v3(tb.z, tb.w, tc.x), t); }
    if (stop) { cur = kLaneDone; return true; }
    if (more) cur = ~(int)((((uint32_t)first + 1u) << 3) | (more - 1u));
    else S.pop_or_done(sp, cur);
    return false;
}


// Wave-private chunked append to a stream in global memory.  A wave owns a STATIC first chunk (its index in the grid picks it:
// no atomic, so thousands of waves starting together do not queue on the stream counter) and reserves further chunks of
// kAppendChunk entries with ONE atomic each; lanes get their entries by ballot prefix, and an append may straddle two chunks.
// The only unwritten entries of a stream are therefore the tail of each participating wave's LAST chunk, which the wave marks
// as holes when it is done (consumers skip them).  The stream counter, preset to waves x kAppendChunk, holds the stream's
// LENGTH (holes included); every write lands in a 256-entry run owned by one wave.
// (kAppendChunk = 256 and kAppendExactBelow = 1 << 21 live in device_types.hpp: the host sizes the streams' slack by them)
// Short streams are appended to EXACTLY instead (one atomic per wave per append, no holes at all): with fewer than
// kAppendExactBelow entries to process a launch issues at most ~30 k such atomics per stream, while chunk tails of several
// thousand waves would outnumber the entries themselves.
struct WaveAppender {
    uint32_t base, used;  // wave-uniform
    uint32_t dyn_base;    // where the counter's reservations start in the stream (0 when the counter was preset to the static part)
    bool exact;
    __device__ __forceinline__ void init(uint32_t wave_in_grid, bool exact_mode, uint32_t dynamic_base = 0u) {
        exact = exact_mode; dyn_base = dynamic_base;
        base = exact_mode ? 0u : wave_in_grid * kAppendChunk;
        used = exact_mode ? kAppendChunk : 0u;   // exact mode owns no chunk: tail_count() == 0
    }
    // every lane of the wave calls; returns the entry index for lanes with pred
    __device__ __forceinline__ uint32_t append(bool pred, uint32_t* counter) {
        const unsigned long long m = __ballot(pred);
        const uint32_t cnt = (uint32_t)__popcll(m), off = lanes_below(m);
        if (exact) {
            uint32_t nb = 0u;
            if (cnt) {
                if (lane_id() == 0u) nb = atomicAdd(counter, cnt);
                nb = __builtin_amdgcn_readfirstlane(nb);
            }
            return nb + off;
        }
        const uint32_t room = kAppendChunk - used;
        uint32_t pos;
        if (cnt <= room) { pos = base + used + off; used += cnt; }
        else {
            uint32_t nb = 0u;
            if (lane_id() == 0u) nb = atomicAdd(counter, kAppendChunk);
            nb = dyn_base + __builtin_amdgcn_readfirstlane(nb);
            pos = off < room ? base + used + off : nb + (off - room);
            base = nb; used = cnt - room;
        }
        return pos;
    }
    // entries [tail_first(), tail_first() + tail_count()) of the wave's last chunk were never written
    __device__ __forceinline__ uint32_t tail_first() const { return base + used; }
    __device__ __forceinline__ uint32_t tail_count() const { return kAppendChunk - used; }
};

}  // namespace vpt
