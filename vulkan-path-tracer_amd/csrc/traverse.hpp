// traverse.hpp — BVH4 traversal for one ray per lane (replaces the driver-side TraceRay / RayQuery of
// RayGen.slang:90 and RTCommon.slang:54-63).
//
// * nodes in global memory are 64 B (four child boxes as 8-bit offsets on a per-node power-of-two grid + four
//   child codes: 4 x dwordx4, two nodes per cache line); a scene small enough to be staged into LDS uses the
//   same tree with fp32 boxes (128 B nodes: no decode, fewer VALU ops); triangles 48 B; closest-hit search visits the hit children nearest first (4-element sorting network on the
//   entry distances), any-hit search takes them in slot order;
// * the per-lane stack lives in LDS as stack[depth][lane] (bank = lane, conflict-free) for the first
//   kStackDepth entries and spills to a per-thread global region beyond that (rare: 3 pushes per level);
// * the triangle test is the shared fp32 contract vptfp::ray_triangle(), so (t,u,v) are bit-identical to the
//   oracle's; ties in t go to the smaller global triangle id, so the result does not depend on traversal
//   order or tree shape;
// * box tests are conservative (boxes are padded at build time and quantised outward; the interval test
//   carries a 4-ulp slack) — a box test only ever prunes, it never decides a hit.
#pragma once
#include "device_types.hpp"

namespace vpt {

constexpr int kStackDepth = 14;      // LDS entries per lane
constexpr int kStackOverflow = 82;   // global entries per lane: 3 pushes per level x binary depth bound 30, minus the LDS part
constexpr int kTraverseBlock = 256;

struct HitRec {
    float t, u, v;
    uint32_t prim, inst, gid;
    int slot;  // position of the triangle in the leaf-ordered array (to validate the winner, see trace_closest)
};

struct TravStats {
    uint32_t nodes, tris;
};

struct TravStack {
    uint32_t* lds;    // this lane's column: entry k at lds[k * kTraverseBlock]
    uint32_t* glob;   // this thread's overflow region
    int sp;
    __device__ inline void push(uint32_t v) {
        if (sp < kStackDepth) lds[sp * kTraverseBlock] = v;
        else if (sp < kStackDepth + kStackOverflow) glob[sp - kStackDepth] = v;
        sp++;
    }
    __device__ inline uint32_t pop() {
        sp--;
        return sp < kStackDepth ? lds[sp * kTraverseBlock] : glob[sp - kStackDepth];
    }
};
__device__ inline TravStack make_stack(unsigned char* smem, uint32_t* overflow) {
    TravStack s;
    s.lds = reinterpret_cast<uint32_t*>(smem) + threadIdx.x;
    s.glob = overflow + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * kStackOverflow;
    s.sp = 0;
    return s;
}

struct NodeData {
    float ox, oy, oz;
    float sx, sy, sz;   // grid steps (powers of two)
    uint32_t lox, loy, loz, hix, hiy, hiz;
    int c0, c1, c2, c3;
};
__device__ inline void unpack_node(const uint4& w0, const uint4& w1, const uint4& w2, const uint4& w3, NodeData& n) {
    n.ox = __uint_as_float(w0.x); n.oy = __uint_as_float(w0.y); n.oz = __uint_as_float(w0.z); n.sx = __uint_as_float(w0.w);
    n.lox = w1.x; n.loy = w1.y; n.loz = w1.z; n.hix = w1.w; n.hiy = w2.x; n.hiz = w2.y; n.sy = __uint_as_float(w2.z); n.sz = __uint_as_float(w2.w);
    n.c0 = (int)w3.x; n.c1 = (int)w3.y; n.c2 = (int)w3.z; n.c3 = (int)w3.w;
}

struct NodeDataWide {
    float4 minx, miny, minz, maxx, maxy, maxz;
    int c0, c1, c2, c3;
};

// Scene access either from global memory (quantised nodes) or from an LDS copy (small scenes, fp32 nodes).
struct GlobalSceneSrc {
    using Node = NodeData;
    const BvhNode* nodes;
    const BvhTri* tris;
    bool strict;  // DeviceScene::strict_hits
    __device__ inline void node(int i, NodeData& n) const {
        const uint4* p = reinterpret_cast<const uint4*>(nodes + i);
        unpack_node(p[0], p[1], p[2], p[3], n);
    }
    __device__ inline void tri(int i, float4& a, float4& b, float4& c) const {
        const float4* p = reinterpret_cast<const float4*>(tris + i);
        a = p[0]; b = p[1]; c = p[2];
    }
};
struct LdsSceneSrc {
    using Node = NodeDataWide;
    const float4* nodes;  // LDS, BvhNodeWide
    const float4* tris;   // LDS
    bool strict;          // DeviceScene::strict_hits
    __device__ inline void node(int i, NodeDataWide& n) const {
        const float4* p = nodes + i * 8;
        n.minx = p[0]; n.miny = p[1]; n.minz = p[2]; n.maxx = p[3]; n.maxy = p[4]; n.maxz = p[5];
        float4 c = p[6];
        n.c0 = __float_as_int(c.x); n.c1 = __float_as_int(c.y); n.c2 = __float_as_int(c.z); n.c3 = __float_as_int(c.w);
    }
    __device__ inline void tri(int i, float4& a, float4& b, float4& c) const {
        const float4* p = tris + i * 3;
        a = p[0]; b = p[1]; c = p[2];
    }
};

__device__ inline float fmin_(float a, float b) { return __builtin_fminf(a, b); }
__device__ inline float fmax_(float a, float b) { return __builtin_fmaxf(a, b); }
constexpr float kMissT = 3.0e38f;

__device__ inline V3 safe_inverse(V3 d) {
    // a zero component would give 0*inf = NaN in the slab test: clamp its reciprocal to +-1e30
    V3 inv;
    inv.x = (vptfp::fabs_(d.x) > 1e-30f) ? 1.0f / d.x : (vptfp::f2u(d.x) >> 31 ? -1e30f : 1e30f);
    inv.y = (vptfp::fabs_(d.y) > 1e-30f) ? 1.0f / d.y : (vptfp::f2u(d.y) >> 31 ? -1e30f : 1e30f);
    inv.z = (vptfp::fabs_(d.z) > 1e-30f) ? 1.0f / d.z : (vptfp::f2u(d.z) >> 31 ? -1e30f : 1e30f);
    return inv;
}

// Per-ray constants of the slab test: reciprocal direction and which plane byte is the near one per axis.
struct RaySlab {
    V3 o, inv;
    bool negx, negy, negz;
};
__device__ inline RaySlab make_slab(V3 o, V3 d) {
    RaySlab r;
    r.o = o; r.inv = safe_inverse(d);
    r.negx = r.inv.x < 0.0f; r.negy = r.inv.y < 0.0f; r.negz = r.inv.z < 0.0f;
    return r;
}
template <int K> __device__ inline float byte_f(uint32_t w) { return (float)((w >> (8 * K)) & 0xffu); }  // v_cvt_f32_ubyteK

// Entry distances of the ray into the four child boxes ([tmin, tlimit] clipped), kMissT for a miss.
// plane distance = (origin + q*step - o) * inv = q * (step*inv) + (origin - o)*inv: one fma per plane (step is a
// power of two, so step*inv is exact); the rounding of the second term moves a plane by a few ulp of the
// ray-to-node distance, which the build-time padding of every box covers.  The near/far byte is picked by the
// sign of the direction, so an inverted (unused) slot gives near > far on every axis and is never entered.
__device__ inline void node_entries(const NodeData& n, const RaySlab& r, float tmin, float tlimit, float& t0, float& t1, float& t2, float& t3) {
    const float ax = n.sx * r.inv.x, ay = n.sy * r.inv.y, az = n.sz * r.inv.z;
    const float bx = (n.ox - r.o.x) * r.inv.x, by = (n.oy - r.o.y) * r.inv.y, bz = (n.oz - r.o.z) * r.inv.z;
    const uint32_t nx = r.negx ? n.hix : n.lox, fx = r.negx ? n.lox : n.hix;
    const uint32_t ny = r.negy ? n.hiy : n.loy, fy = r.negy ? n.loy : n.hiy;
    const uint32_t nz = r.negz ? n.hiz : n.loz, fz = r.negz ? n.loz : n.hiz;
#define VPT_CHILD(K, T)                                                                                                     \
    {                                                                                                                       \
        float tn = fmax_(fmax_(__builtin_fmaf(byte_f<K>(nx), ax, bx), __builtin_fmaf(byte_f<K>(ny), ay, by)),               \
                         fmax_(__builtin_fmaf(byte_f<K>(nz), az, bz), tmin));                                               \
        float tf = fmin_(fmin_(__builtin_fmaf(byte_f<K>(fx), ax, bx), __builtin_fmaf(byte_f<K>(fy), ay, by)),               \
                         fmin_(__builtin_fmaf(byte_f<K>(fz), az, bz), tlimit));                                             \
        T = (tn <= tf * 1.0000005f) ? tn : kMissT;                                                                          \
    }
    VPT_CHILD(0, t0) VPT_CHILD(1, t1) VPT_CHILD(2, t2) VPT_CHILD(3, t3)
#undef VPT_CHILD
}
// The same with the 24 plane fmas and the four slack multiplies issued as 12 + 2 packed instructions (v_pk_fma_f32 / v_pk_mul_f32: two
// IEEE fp32 operations per issue slot, each rounded exactly as its scalar form, so every entry distance is the one node_entries() returns).
typedef float vpt_f2 __attribute__((ext_vector_type(2)));
__device__ inline void node_entries_pk(const NodeData& n, const RaySlab& r, float tmin, float tlimit, float& t0, float& t1, float& t2, float& t3) {
    const float ax = n.sx * r.inv.x, ay = n.sy * r.inv.y, az = n.sz * r.inv.z;
    const float bx = (n.ox - r.o.x) * r.inv.x, by = (n.oy - r.o.y) * r.inv.y, bz = (n.oz - r.o.z) * r.inv.z;
    const uint32_t nx = r.negx ? n.hix : n.lox, fx = r.negx ? n.lox : n.hix;
    const uint32_t ny = r.negy ? n.hiy : n.loy, fy = r.negy ? n.loy : n.hiy;
    const uint32_t nz = r.negz ? n.hiz : n.loz, fz = r.negz ? n.loz : n.hiz;
    const vpt_f2 ax2 = {ax, ax}, ay2 = {ay, ay}, az2 = {az, az}, bx2 = {bx, bx}, by2 = {by, by}, bz2 = {bz, bz};
#define VPT_PAIR(W, A, B, K0, K1) __builtin_elementwise_fma((vpt_f2){byte_f<K0>(W), byte_f<K1>(W)}, A, B)
    const vpt_f2 nx01 = VPT_PAIR(nx, ax2, bx2, 0, 1), nx23 = VPT_PAIR(nx, ax2, bx2, 2, 3), ny01 = VPT_PAIR(ny, ay2, by2, 0, 1), ny23 = VPT_PAIR(ny, ay2, by2, 2, 3);
    const vpt_f2 nz01 = VPT_PAIR(nz, az2, bz2, 0, 1), nz23 = VPT_PAIR(nz, az2, bz2, 2, 3);
    const vpt_f2 fx01 = VPT_PAIR(fx, ax2, bx2, 0, 1), fx23 = VPT_PAIR(fx, ax2, bx2, 2, 3), fy01 = VPT_PAIR(fy, ay2, by2, 0, 1), fy23 = VPT_PAIR(fy, ay2, by2, 2, 3);
    const vpt_f2 fz01 = VPT_PAIR(fz, az2, bz2, 0, 1), fz23 = VPT_PAIR(fz, az2, bz2, 2, 3);
#undef VPT_PAIR
    const float tn0 = fmax_(fmax_(nx01.x, ny01.x), fmax_(nz01.x, tmin)), tn1 = fmax_(fmax_(nx01.y, ny01.y), fmax_(nz01.y, tmin));
    const float tn2 = fmax_(fmax_(nx23.x, ny23.x), fmax_(nz23.x, tmin)), tn3 = fmax_(fmax_(nx23.y, ny23.y), fmax_(nz23.y, tmin));
    const vpt_f2 tf01 = {fmin_(fmin_(fx01.x, fy01.x), fmin_(fz01.x, tlimit)), fmin_(fmin_(fx01.y, fy01.y), fmin_(fz01.y, tlimit))};
    const vpt_f2 tf23 = {fmin_(fmin_(fx23.x, fy23.x), fmin_(fz23.x, tlimit)), fmin_(fmin_(fx23.y, fy23.y), fmin_(fz23.y, tlimit))};
    const vpt_f2 slack = {1.0000005f, 1.0000005f};
    const vpt_f2 s01 = tf01 * slack, s23 = tf23 * slack;
    t0 = (tn0 <= s01.x) ? tn0 : kMissT; t1 = (tn1 <= s01.y) ? tn1 : kMissT; t2 = (tn2 <= s23.x) ? tn2 : kMissT; t3 = (tn3 <= s23.y) ? tn3 : kMissT;
}
// fp32 nodes: the plain slab test.
__device__ inline float box_entry(float bx0, float by0, float bz0, float bx1, float by1, float bz1, V3 o, V3 inv, float tmin, float tlimit) {
    float t0x = (bx0 - o.x) * inv.x, t1x = (bx1 - o.x) * inv.x;
    float t0y = (by0 - o.y) * inv.y, t1y = (by1 - o.y) * inv.y;
    float t0z = (bz0 - o.z) * inv.z, t1z = (bz1 - o.z) * inv.z;
    float tn = fmax_(fmax_(fmin_(t0x, t1x), fmin_(t0y, t1y)), fmax_(fmin_(t0z, t1z), tmin));
    float tf = fmin_(fmin_(fmax_(t0x, t1x), fmax_(t0y, t1y)), fmin_(fmax_(t0z, t1z), tlimit));
    return (tn <= tf * 1.0000005f) ? tn : kMissT;
}
__device__ inline void node_entries(const NodeDataWide& n, const RaySlab& r, float tmin, float tlimit, float& t0, float& t1, float& t2, float& t3) {
    t0 = box_entry(n.minx.x, n.miny.x, n.minz.x, n.maxx.x, n.maxy.x, n.maxz.x, r.o, r.inv, tmin, tlimit);
    t1 = box_entry(n.minx.y, n.miny.y, n.minz.y, n.maxx.y, n.maxy.y, n.maxz.y, r.o, r.inv, tmin, tlimit);
    t2 = box_entry(n.minx.z, n.miny.z, n.minz.z, n.maxx.z, n.maxy.z, n.maxz.z, r.o, r.inv, tmin, tlimit);
    t3 = box_entry(n.minx.w, n.miny.w, n.minz.w, n.maxx.w, n.maxy.w, n.maxz.w, r.o, r.inv, tmin, tlimit);
}
__device__ inline void cswap(float& ta, int& ca, float& tb, int& cb) {
    bool sw = tb < ta;
    float tt = sw ? tb : ta; int ct = sw ? cb : ca;
    tb = sw ? ta : tb; cb = sw ? ca : cb;
    ta = tt; ca = ct;
}

// One closest-hit search (tmin < t < tmax; ties -> smaller global triangle id) that ignores triangles ex0 / ex1.
template <bool COUNT, bool STRICT, class Src>
__device__ inline bool trace_closest_pass(const Src& src, V3 o, V3 d, float tmin, float tmax, TravStack stack, HitRec& best, TravStats& st,
                                          uint32_t ex0, uint32_t ex1) {
    best.t = tmax; best.u = 0.0f; best.v = 0.0f; best.prim = 0xffffffffu; best.inst = 0xffffffffu; best.gid = 0xffffffffu; best.slot = 0;
    bool found = false;
    const RaySlab slab = make_slab(o, d);
    stack.sp = 0;
    int cur = 0;  // root is inner node 0
    while (true) {
        if (cur >= 0) {
            typename Src::Node n;
            src.node(cur, n);
            if (COUNT) st.nodes++;
            float t0, t1, t2, t3;
            node_entries(n, slab, tmin, best.t, t0, t1, t2, t3);
            int c0 = n.c0, c1 = n.c1, c2 = n.c2, c3 = n.c3;
            cswap(t0, c0, t1, c1); cswap(t2, c2, t3, c3); cswap(t0, c0, t2, c2); cswap(t1, c1, t3, c3); cswap(t1, c1, t2, c2);
            if (t0 < kMissT) {  // nearest child next, the others pushed far -> near
                if (t3 < kMissT) stack.push((uint32_t)c3);
                if (t2 < kMissT) stack.push((uint32_t)c2);
                if (t1 < kMissT) stack.push((uint32_t)c1);
                cur = c0;
                continue;
            }
        } else {
            uint32_t enc = (uint32_t)(~cur);
            int first = (int)(enc >> 3), cnt = (int)(enc & 7u) + 1;
            for (int k = 0; k < cnt; k++) {
                float4 a, b, c;
                src.tri(first + k, a, b, c);
                if (COUNT) st.tris++;
                float t, u, v;
                if (vptfp::ray_triangle(o, d, vptfp::v3(a.x, a.y, a.z), vptfp::v3(a.w, b.x, b.y), vptfp::v3(b.z, b.w, c.x),
                                        tmin, tmax, &t, &u, &v)) {
                    uint32_t gid = __float_as_uint(c.w);
                    if ((!STRICT || (gid != ex0 && gid != ex1)) && (!found || t < best.t || (t == best.t && gid < best.gid))) {
                        best.t = t; best.u = u; best.v = v;
                        best.prim = __float_as_uint(c.y); best.inst = __float_as_uint(c.z); best.gid = gid;
                        if (STRICT) best.slot = first + k;
                        found = true;
                    }
                }
            }
        }
        if (stack.sp == 0) break;
        cur = (int)stack.pop();
    }
    return found;
}

template <class Src>
__device__ inline bool slot_hit_is_local(const Src& src, int slot, V3 o, V3 d, float t) {
    float4 a, b, c;
    src.tri(slot, a, b, c);
    return vptfp::hit_is_local(o, d, vptfp::v3(a.x, a.y, a.z), vptfp::v3(a.w, b.x, b.y), vptfp::v3(b.z, b.w, c.x), t);
}
// Closest hit among the hits that are LOCAL to their triangle (vpt_fp32.h hit_is_local: for a ray numerically inside a
// triangle's plane fp32 can place the hit outside the triangle's own box, where a box hierarchy would or would not see it
// depending on its shape).  The guard is too expensive per candidate in SIMT form (it would run whenever any lane of the
// wave has a candidate), so the search runs unguarded, the WINNER is validated once, and in the ~1e-9 case that it fails
// the search is repeated without that triangle.  Even so it costs ~11 % of the Cornell throughput (about four validations
// of ~70 VALU per bounce), so it is a run-time option (VPT_FLAG_LOCAL_HITS); off, the unguarded winner is returned.  Candidates that are not the winner cannot matter: best.t only shrinks,
// and a local hit closer than the final winner is never pruned (boxes contain the triangle boxes).
// The path kernels are instantiated twice (STRICT template flag): the default instantiation sees strict == false as a
// compile-time constant and carries none of this code (the fused bounce kernel is instruction-cache sensitive).
template <bool COUNT, class Src>
__device__ inline bool trace_closest_strict(const Src& src, V3 o, V3 d, float tmin, float tmax, TravStack stack, HitRec& best, TravStats& st) {
    uint32_t ex0 = 0xffffffffu, ex1 = 0xffffffffu;
    while (true) {
        if (!trace_closest_pass<COUNT, true>(src, o, d, tmin, tmax, stack, best, st, ex0, ex1)) return false;
        if (slot_hit_is_local(src, best.slot, o, d, best.t)) return true;
        ex1 = ex0; ex0 = best.gid;
    }
}
template <bool COUNT, class Src>
__device__ inline bool trace_closest(const Src& src, V3 o, V3 d, float tmin, float tmax, TravStack stack, HitRec& best, TravStats& st) {
    if (!src.strict) return trace_closest_pass<COUNT, false>(src, o, d, tmin, tmax, stack, best, st, 0xffffffffu, 0xffffffffu);  // default: the unguarded winner
    return trace_closest_strict<COUNT>(src, o, d, tmin, tmax, stack, best, st);
}

// Shadow queries.  The reference asks for the CLOSEST committed hit and then only looks at (a) whether there
// is one (sky visibility, ClosestHit.slang:139) or (b) whether it is the sampled light triangle
// (ClosestHit.slang:171-176).  Both are decided exactly by an any-hit search:
//   (a) occluded  <=>  some triangle is hit with tmin < t < tmax;
//   (b) closest == expected  <=>  the expected triangle is hit at t_e (tested first, by its own record) and no
//       other triangle beats it under the closest-hit order used everywhere here (smaller t, ties to the
//       smaller global id), i.e. no hit with t < t_e or (t == t_e and gid < expected).
// The search stops at the first such triangle and never looks beyond t_e, so it visits far fewer nodes than a
// closest-hit traversal.  LIGHT = false: (a); LIGHT = true: (b) with `t_e`, `expect`.
// One any-hit search ignoring triangles ex0 / ex1; reports the triangle that stopped it (cand_slot, cand_t, cand_gid).
template <bool COUNT, bool LIGHT, bool STRICT, class Src>
__device__ inline bool trace_occluded_pass(const Src& src, V3 o, V3 d, float tmin, float tmax, float t_e, uint32_t expect, TravStack stack,
                                           TravStats& st, uint32_t ex0, uint32_t ex1, int& cand_slot, float& cand_t, uint32_t& cand_gid) {
    const float tlimit = LIGHT ? t_e : tmax;
    const RaySlab slab = make_slab(o, d);
    stack.sp = 0;
    int cur = 0;
    while (true) {
        if (cur >= 0) {
            typename Src::Node n;
            src.node(cur, n);
            if (COUNT) st.nodes++;
            float t0, t1, t2, t3;
            node_entries(n, slab, tmin, tlimit, t0, t1, t2, t3);
            int next = 0x7fffffff;  // order is irrelevant for an any-hit search: take hit children in slot order
            if (t3 < kMissT) next = n.c3;
            if (t2 < kMissT) { if (next != 0x7fffffff) stack.push((uint32_t)next); next = n.c2; }
            if (t1 < kMissT) { if (next != 0x7fffffff) stack.push((uint32_t)next); next = n.c1; }
            if (t0 < kMissT) { if (next != 0x7fffffff) stack.push((uint32_t)next); next = n.c0; }
            if (next != 0x7fffffff) { cur = next; continue; }
        } else {
            uint32_t enc = (uint32_t)(~cur);
            int first = (int)(enc >> 3), cnt = (int)(enc & 7u) + 1;
            for (int k = 0; k < cnt; k++) {
                float4 a, b, c;
                src.tri(first + k, a, b, c);
                if (COUNT) st.tris++;
                float t, u, v;
                if (vptfp::ray_triangle(o, d, vptfp::v3(a.x, a.y, a.z), vptfp::v3(a.w, b.x, b.y), vptfp::v3(b.z, b.w, c.x), tmin, tmax, &t, &u, &v)) {
                    const uint32_t gid = __float_as_uint(c.w);
                    if ((!STRICT || (gid != ex0 && gid != ex1)) && (!LIGHT || t < t_e || (t == t_e && gid < expect))) {
                        if (STRICT) { cand_slot = first + k; cand_t = t; cand_gid = gid; }
                        return true;
                    }
                }
            }
        }
        if (stack.sp == 0) break;
        cur = (int)stack.pop();
    }
    return false;
}

// Any-hit search over LOCAL hits only: the triangle that stops a pass is validated afterwards (see trace_closest).
template <bool COUNT, bool LIGHT, class Src>
__device__ inline bool trace_occluded_strict(const Src& src, V3 o, V3 d, float tmin, float tmax, float t_e, uint32_t expect, TravStack stack,
                                                   TravStats& st) {
    int slot = 0; float t = 0.0f; uint32_t gid = 0xffffffffu;
    uint32_t ex0 = 0xffffffffu, ex1 = 0xffffffffu;
    while (true) {
        if (!trace_occluded_pass<COUNT, LIGHT, true>(src, o, d, tmin, tmax, t_e, expect, stack, st, ex0, ex1, slot, t, gid)) return false;
        if (slot_hit_is_local(src, slot, o, d, t)) return true;
        ex1 = ex0; ex0 = gid;
    }
}
template <bool COUNT, bool LIGHT, class Src>
__device__ inline bool trace_occluded(const Src& src, V3 o, V3 d, float tmin, float tmax, float t_e, uint32_t expect, TravStack stack,
                                      TravStats& st) {
    int slot = 0; float t = 0.0f; uint32_t gid = 0xffffffffu;
    if (!src.strict) return trace_occluded_pass<COUNT, LIGHT, false>(src, o, d, tmin, tmax, t_e, expect, stack, st, 0xffffffffu, 0xffffffffu, slot, t, gid);
    return trace_occluded_strict<COUNT, LIGHT>(src, o, d, tmin, tmax, t_e, expect, stack, st);
}

// (b) in full: is the closest hit of the ray the triangle with global id `expect`?  `slot` is that triangle's
// position in the leaf-ordered triangle array.
template <bool COUNT, class Src>
__device__ inline bool closest_is(const Src& src, V3 o, V3 d, float tmin, float tmax, uint32_t expect, uint32_t slot, TravStack stack,
                                  TravStats& st) {
    float4 a, b, c;
    src.tri((int)slot, a, b, c);
    if (COUNT) st.tris++;
    float t_e, u, v;
    if (!vptfp::ray_triangle(o, d, vptfp::v3(a.x, a.y, a.z), vptfp::v3(a.w, b.x, b.y), vptfp::v3(b.z, b.w, c.x), tmin, tmax, &t_e, &u, &v)) return false;
    if (src.strict && !vptfp::hit_is_local(o, d, vptfp::v3(a.x, a.y, a.z), vptfp::v3(a.w, b.x, b.y), vptfp::v3(b.z, b.w, c.x), t_e)) return false;
    return !trace_occluded<COUNT, true>(src, o, d, tmin, tmax, t_e, expect, stack, st);
}

}  // namespace vpt
