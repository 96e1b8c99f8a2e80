// traverse.hpp — closest-hit BVH2 traversal for one ray per lane (replaces the driver-side
// TraceRay / RayQuery of RayGen.slang:90 and RTCommon.slang:54-63).
//
// * nodes are 64 B (both child boxes in the parent), triangles 48 B; near child first, far child
//   pushed on a per-lane stack that lives in LDS as stack[depth][lane] (bank = lane, conflict-free);
// * the triangle test is the shared fp32 contract vptfp::ray_triangle(), so (t,u,v) are bit-identical
//   to the oracle's; ties in t go to the smaller global triangle id, so the result does not depend
//   on traversal order or tree shape;
// * box tests are conservative (boxes are padded at build time; the interval test carries a 4-ulp
//   slack) — a box test only ever prunes, it never decides a hit.
#pragma once
#include "device_types.hpp"

namespace vpt {

constexpr int kStackDepth = 32;  // the builder bounds the tree depth to this
constexpr int kTraverseBlock = 256;

struct HitRec {
    float t, u, v;
    uint32_t prim, inst, gid;
};

struct TravStats {
    uint32_t nodes, tris;
};

// Scene access either from global memory or from an LDS copy (small scenes).
struct GlobalSceneSrc {
    const BvhNode* nodes;
    const BvhTri* tris;
    __device__ inline void node(int i, float4& a, float4& b, float4& c, int& l, int& r) const {
        const float4* p = reinterpret_cast<const float4*>(nodes + i);
        a = p[0]; b = p[1]; c = p[2];
        float4 d = p[3];
        l = __float_as_int(d.x); r = __float_as_int(d.y);
    }
    __device__ inline void tri(int i, float4& a, float4& b, float4& c) const {
        const float4* p = reinterpret_cast<const float4*>(tris + i);
        a = p[0]; b = p[1]; c = p[2];
    }
};
struct LdsSceneSrc {
    const float4* nodes;  // LDS
    const float4* tris;   // LDS
    __device__ inline void node(int i, float4& a, float4& b, float4& c, int& l, int& r) const {
        const float4* p = nodes + i * 4;
        a = p[0]; b = p[1]; c = p[2];
        float4 d = p[3];
        l = __float_as_int(d.x); r = __float_as_int(d.y);
    }
    __device__ inline void tri(int i, float4& a, float4& b, float4& c) const {
        const float4* p = tris + i * 3;
        a = p[0]; b = p[1]; c = p[2];
    }
};

__device__ inline float fmin_(float a, float b) { return __builtin_fminf(a, b); }
__device__ inline float fmax_(float a, float b) { return __builtin_fmaxf(a, b); }

// Entry distance of the ray into a box, or a negative value if it misses [tmin, tlimit].
__device__ inline float box_entry(float bx0, float by0, float bz0, float bx1, float by1, float bz1, V3 o, V3 inv,
                                  float tmin, float tlimit) {
    float t0x = (bx0 - o.x) * inv.x, t1x = (bx1 - o.x) * inv.x;
    float t0y = (by0 - o.y) * inv.y, t1y = (by1 - o.y) * inv.y;
    float t0z = (bz0 - o.z) * inv.z, t1z = (bz1 - o.z) * inv.z;
    float tn = fmax_(fmax_(fmin_(t0x, t1x), fmin_(t0y, t1y)), fmax_(fmin_(t0z, t1z), tmin));
    float tf = fmin_(fmin_(fmax_(t0x, t1x), fmax_(t0y, t1y)), fmin_(fmax_(t0z, t1z), tlimit));
    return (tn <= tf * 1.0000005f) ? tn : -1.0f;
}

// stack: this lane's column, entries at stack[k * stride].
template <bool COUNT, class Src>
__device__ inline bool trace_closest(const Src& src, V3 o, V3 d, float tmin, float tmax, uint32_t* stack, int stride,
                                     HitRec& best, TravStats& st) {
    best.t = tmax; best.u = 0.0f; best.v = 0.0f; best.prim = 0xffffffffu; best.inst = 0xffffffffu; best.gid = 0xffffffffu;
    bool found = false;
    V3 inv;
    // a zero component would give 0*inf = NaN in the slab test: clamp its reciprocal to +-1e30
    inv.x = (vptfp::fabs_(d.x) > 1e-30f) ? 1.0f / d.x : (vptfp::f2u(d.x) >> 31 ? -1e30f : 1e30f);
    inv.y = (vptfp::fabs_(d.y) > 1e-30f) ? 1.0f / d.y : (vptfp::f2u(d.y) >> 31 ? -1e30f : 1e30f);
    inv.z = (vptfp::fabs_(d.z) > 1e-30f) ? 1.0f / d.z : (vptfp::f2u(d.z) >> 31 ? -1e30f : 1e30f);
    int sp = 0;
    int cur = 0;  // root is inner node 0
    while (true) {
        if (cur >= 0) {
            float4 a, b, c; int l, r;
            src.node(cur, a, b, c, l, r);
            if (COUNT) st.nodes++;
            float tl = box_entry(a.x, a.y, a.z, a.w, b.x, b.y, o, inv, tmin, best.t);
            float tr = box_entry(b.z, b.w, c.x, c.y, c.z, c.w, o, inv, tmin, best.t);
            bool hl = tl >= 0.0f, hr = tr >= 0.0f;
            if (hl && hr) {
                bool lfirst = tl <= tr;
                int nearc = lfirst ? l : r, farc = lfirst ? r : l;
                if (sp < kStackDepth) { stack[sp * stride] = (uint32_t)farc; sp++; }
                cur = nearc;
                continue;
            } else if (hl) { cur = l; continue; }
            else if (hr) { cur = r; continue; }
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
                    if (!found || t < best.t || (t == best.t && gid < best.gid)) {
                        best.t = t; best.u = u; best.v = v;
                        best.prim = __float_as_uint(c.y); best.inst = __float_as_uint(c.z); best.gid = gid;
                        found = true;
                    }
                }
            }
        }
        if (sp == 0) break;
        sp--;
        cur = (int)stack[sp * stride];
    }
    return found;
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
template <bool COUNT, bool LIGHT, class Src>
__device__ inline bool trace_occluded(const Src& src, V3 o, V3 d, float tmin, float tmax, float t_e, uint32_t expect, uint32_t* stack,
                                      int stride, TravStats& st) {
    const float tlimit = LIGHT ? t_e : tmax;
    V3 inv;
    inv.x = (vptfp::fabs_(d.x) > 1e-30f) ? 1.0f / d.x : (vptfp::f2u(d.x) >> 31 ? -1e30f : 1e30f);
    inv.y = (vptfp::fabs_(d.y) > 1e-30f) ? 1.0f / d.y : (vptfp::f2u(d.y) >> 31 ? -1e30f : 1e30f);
    inv.z = (vptfp::fabs_(d.z) > 1e-30f) ? 1.0f / d.z : (vptfp::f2u(d.z) >> 31 ? -1e30f : 1e30f);
    int sp = 0;
    int cur = 0;
    while (true) {
        if (cur >= 0) {
            float4 a, b, c; int l, r;
            src.node(cur, a, b, c, l, r);
            if (COUNT) st.nodes++;
            float tl = box_entry(a.x, a.y, a.z, a.w, b.x, b.y, o, inv, tmin, tlimit);
            float tr = box_entry(b.z, b.w, c.x, c.y, c.z, c.w, o, inv, tmin, tlimit);
            bool hl = tl >= 0.0f, hr = tr >= 0.0f;
            if (hl && hr) {
                bool lfirst = tl <= tr;
                int nearc = lfirst ? l : r, farc = lfirst ? r : l;
                if (sp < kStackDepth) { stack[sp * stride] = (uint32_t)farc; sp++; }
                cur = nearc;
                continue;
            } else if (hl) { cur = l; continue; }
            else if (hr) { cur = r; continue; }
        } else {
            uint32_t enc = (uint32_t)(~cur);
            int first = (int)(enc >> 3), cnt = (int)(enc & 7u) + 1;
            for (int k = 0; k < cnt; k++) {
                float4 a, b, c;
                src.tri(first + k, a, b, c);
                if (COUNT) st.tris++;
                float t, u, v;
                if (vptfp::ray_triangle(o, d, vptfp::v3(a.x, a.y, a.z), vptfp::v3(a.w, b.x, b.y), vptfp::v3(b.z, b.w, c.x), tmin, tmax, &t, &u, &v)) {
                    if (!LIGHT) return true;
                    if (t < t_e || (t == t_e && __float_as_uint(c.w) < expect)) return true;
                }
            }
        }
        if (sp == 0) break;
        sp--;
        cur = (int)stack[sp * stride];
    }
    return false;
}

// (b) in full: is the closest hit of the ray the triangle with global id `expect`?  `slot` is that triangle's
// position in the leaf-ordered triangle array.
template <bool COUNT, class Src>
__device__ inline bool closest_is(const Src& src, V3 o, V3 d, float tmin, float tmax, uint32_t expect, uint32_t slot, uint32_t* stack,
                                  int stride, TravStats& st) {
    float4 a, b, c;
    src.tri((int)slot, a, b, c);
    if (COUNT) st.tris++;
    float t_e, u, v;
    if (!vptfp::ray_triangle(o, d, vptfp::v3(a.x, a.y, a.z), vptfp::v3(a.w, b.x, b.y), vptfp::v3(b.z, b.w, c.x), tmin, tmax, &t_e, &u, &v)) return false;
    return !trace_occluded<COUNT, true>(src, o, d, tmin, tmax, t_e, expect, stack, stride, st);
}

}  // namespace vpt
