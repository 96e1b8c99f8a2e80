// bvh_build.cpp — deterministic binned-SAH BVH2 over the flattened world-space triangles.
// Replaces BLASBuilder::Build + TLAS::New (reference PathTracer.cpp:488-505; the reference's BVH lives
// in the Vulkan driver).  Every reference instance gets its own BLAS (PathTracer.cpp:471-479) and is
// placed once, so instances are flattened into one world-space tree.
//
// Output layout (device_types.hpp): 64 B four-wide quantised nodes (collapsed from the binary SAH tree), 48 B triangles
// in leaf order, leaves of <= 4 triangles, binary depth bounded by kMaxDepth (bounds the traversal stack).
#include "bvh_build.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <future>

namespace vpt {

namespace {
constexpr int kLeafSize = 4;
constexpr int kMaxBins = 32;   // (the product builds with 16 bins, BvhBuildOptions::bins; 32 is the builder study's other setting: profiles/r06_builder_study.md)
constexpr int kMaxDepth = 30;  // < kStackDepth (32)
constexpr float kNodeCost = 0.7f;  // one two-box node test relative to one triangle test

struct Box {
    float lo[3], hi[3];
    void reset() { for (int a = 0; a < 3; a++) { lo[a] = 3.0e38f; hi[a] = -3.0e38f; } }
    void grow(const Box& b) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
    void grow(const float* p) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx < 0 || dy < 0 || dz < 0) return 0.0f;
        return dx * dy + dy * dz + dz * dx;
    }
};
struct Ref { Box b; float c[3]; uint32_t tri; };
struct TmpNode { Box b; int left, right; int first, count; };

// Subtrees of at least kParallelMin references, down to kParallelDepth levels below the root, are built by their own threads (the
// reference builds its acceleration structures on the device, PathTracer.cpp:484-505; here the host builder was 0.3 s of a 0.44 s
// vpt_set_scene on the 511 k-triangle bust).  The two halves of a split own disjoint ranges of `refs`, each thread numbers its nodes
// from zero in a vector of its own, and the parent appends left then right with the indices shifted — exactly the array the
// single-threaded depth-first recursion produces, whatever the thread timing.
constexpr int kParallelMin = 16384;
constexpr int kParallelDepth = 5;

struct Builder {
    std::vector<Ref> refs;
    std::vector<TmpNode> nodes;
    bool parallel = true;
    int bins = 16;

    int build(int first, int count, int depth) { return build(first, count, depth, nodes); }
    static void append_shifted(std::vector<TmpNode>& out, const std::vector<TmpNode>& sub) {
        const int off = (int)out.size();
        for (TmpNode t : sub) { if (t.left >= 0) { t.left += off; t.right += off; } out.push_back(t); }
    }
    int build(int first, int count, int depth, std::vector<TmpNode>& nodes) {
        TmpNode n; n.b.reset(); n.left = n.right = -1; n.first = first; n.count = count;
        Box cb; cb.reset();
        for (int i = first; i < first + count; i++) { n.b.grow(refs[i].b); cb.grow(refs[i].c); }
        int id = (int)nodes.size();
        nodes.push_back(n);
        if (count == 1) return id;
        const bool small = count <= kLeafSize;  // may stay a leaf; split only if SAH says the split is cheaper
        int need = 0; { int c = (count + kLeafSize - 1) / kLeafSize; while ((1 << need) < c) need++; }
        bool force_median = depth + need + 1 >= kMaxDepth;
        int best_axis = -1, best_bin = -1; float best_cost = 3.0e38f;
        if (!force_median) {
            for (int a = 0; a < 3; a++) {
                float ext = cb.hi[a] - cb.lo[a];
                if (!(ext > 0.0f)) continue;
                Box bb[kMaxBins]; int bc[kMaxBins];
                for (int k = 0; k < bins; k++) { bb[k].reset(); bc[k] = 0; }
                float scale = (float)bins / ext;
                for (int i = first; i < first + count; i++) {
                    int k = (int)((refs[i].c[a] - cb.lo[a]) * scale);
                    k = k < 0 ? 0 : (k >= bins ? bins - 1 : k);
                    bb[k].grow(refs[i].b); bc[k]++;
                }
                float la[kMaxBins]; int lc[kMaxBins];
                Box acc; acc.reset(); int cnt = 0;
                for (int k = 0; k < bins - 1; k++) { acc.grow(bb[k]); cnt += bc[k]; la[k] = acc.half_area(); lc[k] = cnt; }
                acc.reset(); cnt = 0;
                for (int k = bins - 1; k > 0; k--) {
                    acc.grow(bb[k]); cnt += bc[k];
                    if (lc[k - 1] == 0 || cnt == 0) continue;
                    float cost = la[k - 1] * (float)lc[k - 1] + acc.half_area() * (float)cnt;
                    if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = k; }
                }
            }
        }
        if (small) {
            // leaf cost = count triangle tests; split cost = one node visit + area-weighted child tests
            float pa = n.b.half_area();
            if (best_axis < 0 || !(pa > 0.0f) || kNodeCost + best_cost / pa >= (float)count) return id;
        }
        int mid;
        if (best_axis >= 0) {
            int a = best_axis;
            float ext = cb.hi[a] - cb.lo[a], scale = (float)bins / ext, lo = cb.lo[a];
            int bin = best_bin;
            auto it = std::stable_partition(refs.begin() + first, refs.begin() + first + count, [&](const Ref& r) {
                int k = (int)((r.c[a] - lo) * scale);
                k = k < 0 ? 0 : (k >= bins ? bins - 1 : k);
                return k < bin;
            });
            mid = (int)(it - refs.begin());
        } else {
            // object-median split on the widest centroid axis (also the depth-bounding fallback)
            int a = 0; float e = -1.0f;
            for (int k = 0; k < 3; k++) { float x = cb.hi[k] - cb.lo[k]; if (x > e) { e = x; a = k; } }
            mid = first + count / 2;
            std::nth_element(refs.begin() + first, refs.begin() + mid, refs.begin() + first + count,
                             [a](const Ref& x, const Ref& y) { return x.c[a] < y.c[a] || (x.c[a] == y.c[a] && x.tri < y.tri); });
        }
        if (mid == first || mid == first + count) mid = first + count / 2;
        int l, r;
        if (parallel && count >= kParallelMin && depth < kParallelDepth) {
            std::vector<TmpNode> lsub, rsub;
            std::future<int> fl = std::async(std::launch::async, [&]() { return build(first, mid - first, depth + 1, lsub); });
            build(mid, first + count - mid, depth + 1, rsub);
            fl.get();
            l = (int)nodes.size(); append_shifted(nodes, lsub);
            r = (int)nodes.size(); append_shifted(nodes, rsub);
        } else {
            l = build(first, mid - first, depth + 1, nodes);
            r = build(mid, first + count - mid, depth + 1, nodes);
        }
        nodes[id].left = l; nodes[id].right = r; nodes[id].count = 0;
        return id;
    }

    // ---- Spatial splits (Stich, Friedrich, Dietrich 2009): besides the object split of build(), a node may be cut by an axis-aligned
    // plane; triangles crossing it are referenced from BOTH children, each with the bounds of its clipped part, so the children do not
    // overlap.  References are lists per node here (they multiply); leaves append their triangle ids to leaf_tris in creation order.
    const std::vector<BvhTri>* src = nullptr;
    std::vector<uint32_t> leaf_tris;
    float root_area = 0.0f;
    size_t ref_budget = 0, ref_total = 0;

    // bounds of (triangle t) ∩ (slab lo <= x[a] <= hi) ∩ box `within`; false if empty
    bool clip_bounds(uint32_t t, int a, double lo, double hi, const Box& within, Box& out) const {
        const BvhTri& tr = (*src)[t];
        double poly[2][9][3]; int n = 3, cur = 0;
        for (int k = 0; k < 3; k++) { poly[0][0][k] = tr.v0[k]; poly[0][1][k] = (double)tr.v0[k] + tr.e1[k]; poly[0][2][k] = (double)tr.v0[k] + tr.e2[k]; }
        for (int side = 0; side < 2; side++) {   // keep x >= lo, then x <= hi
            const double plane = side == 0 ? lo : hi, sgn = side == 0 ? 1.0 : -1.0;
            int m = 0;
            for (int i = 0; i < n; i++) {
                const double* p = poly[cur][i]; const double* q = poly[cur][(i + 1) % n];
                const double dp = sgn * (p[a] - plane), dq = sgn * (q[a] - plane);
                if (dp >= 0.0) { for (int k = 0; k < 3; k++) poly[cur ^ 1][m][k] = p[k]; m++; }
                if ((dp > 0.0 && dq < 0.0) || (dp < 0.0 && dq > 0.0)) {
                    const double w = dp / (dp - dq);
                    for (int k = 0; k < 3; k++) poly[cur ^ 1][m][k] = p[k] + (q[k] - p[k]) * w;
                    poly[cur ^ 1][m][a] = plane;
                    m++;
                }
            }
            n = m; cur ^= 1;
            if (n == 0) return false;
        }
        out.reset();
        for (int i = 0; i < n; i++) {
            for (int k = 0; k < 3; k++) {   // outward-rounded float bounds of the double vertices
                float f = (float)poly[cur][i][k];
                float fl = (double)f > poly[cur][i][k] ? std::nextafter(f, -3.0e38f) : f;
                float fh = (double)f < poly[cur][i][k] ? std::nextafter(f, 3.0e38f) : f;
                out.lo[k] = std::min(out.lo[k], fl); out.hi[k] = std::max(out.hi[k], fh);
            }
        }
        for (int k = 0; k < 3; k++) { out.lo[k] = std::max(out.lo[k], within.lo[k]); out.hi[k] = std::min(out.hi[k], within.hi[k]); if (out.lo[k] > out.hi[k]) return false; }
        return true;
    }

    int make_leaf(const std::vector<Ref>& r, const Box& nb) {
        TmpNode n; n.b = nb; n.left = n.right = -1; n.first = (int)leaf_tris.size(); n.count = (int)r.size();
        for (const Ref& x : r) leaf_tris.push_back(x.tri);
        nodes.push_back(n);
        return (int)nodes.size() - 1;
    }

    int build_spatial(std::vector<Ref>& r, int depth) {
        const int count = (int)r.size();
        Box nb; nb.reset(); Box cb; cb.reset();
        for (const Ref& x : r) { nb.grow(x.b); cb.grow(x.c); }
        if (count == 1) return make_leaf(r, nb);
        const bool small = count <= kLeafSize;
        int need = 0; { int c = (count + kLeafSize - 1) / kLeafSize; while ((1 << need) < c) need++; }
        const bool force_median = depth + need + 1 >= kMaxDepth;
        // object split (as build())
        int o_axis = -1, o_bin = -1; float o_cost = 3.0e38f; float o_overlap = 0.0f;
        if (!force_median) {
            for (int a = 0; a < 3; a++) {
                float ext = cb.hi[a] - cb.lo[a];
                if (!(ext > 0.0f)) continue;
                Box bb[kMaxBins]; int bc[kMaxBins];
                for (int k = 0; k < bins; k++) { bb[k].reset(); bc[k] = 0; }
                float scale = (float)bins / ext;
                for (const Ref& x : r) {
                    int k = (int)((x.c[a] - cb.lo[a]) * scale);
                    k = k < 0 ? 0 : (k >= bins ? bins - 1 : k);
                    bb[k].grow(x.b); bc[k]++;
                }
                Box lb[kMaxBins]; int lc[kMaxBins];
                Box acc; acc.reset(); int cnt = 0;
                for (int k = 0; k < bins - 1; k++) { acc.grow(bb[k]); cnt += bc[k]; lb[k] = acc; lc[k] = cnt; }
                acc.reset(); cnt = 0;
                for (int k = bins - 1; k > 0; k--) {
                    acc.grow(bb[k]); cnt += bc[k];
                    if (lc[k - 1] == 0 || cnt == 0) continue;
                    float cost = lb[k - 1].half_area() * (float)lc[k - 1] + acc.half_area() * (float)cnt;
                    if (cost < o_cost) {
                        o_cost = cost; o_axis = a; o_bin = k;
                        Box in; for (int q = 0; q < 3; q++) { in.lo[q] = std::max(lb[k - 1].lo[q], acc.lo[q]); in.hi[q] = std::min(lb[k - 1].hi[q], acc.hi[q]); }
                        o_overlap = in.half_area();
                    }
                }
            }
        }
        // spatial split: only where the object split leaves the children overlapping noticeably, and while references may still multiply
        int s_axis = -1; float s_cost = 3.0e38f; double s_plane = 0.0;
        if (!force_median && o_axis >= 0 && o_overlap > 1.0e-5f * root_area && ref_total + (size_t)count <= ref_budget) {
            for (int a = 0; a < 3; a++) {
                const double lo = nb.lo[a], ext = (double)nb.hi[a] - lo;
                if (!(ext > 0.0)) continue;
                Box bb[kMaxBins]; int enter[kMaxBins], leave[kMaxBins];
                for (int k = 0; k < bins; k++) { bb[k].reset(); enter[k] = leave[k] = 0; }
                const double scale = bins / ext;
                for (const Ref& x : r) {
                    int k0 = (int)(((double)x.b.lo[a] - lo) * scale), k1 = (int)(((double)x.b.hi[a] - lo) * scale);
                    k0 = k0 < 0 ? 0 : (k0 >= bins ? bins - 1 : k0); k1 = k1 < k0 ? k0 : (k1 >= bins ? bins - 1 : k1);
                    enter[k0]++; leave[k1]++;
                    if (k0 == k1) { bb[k0].grow(x.b); continue; }
                    for (int k = k0; k <= k1; k++) {
                        Box part;
                        if (clip_bounds(x.tri, a, lo + k / scale, lo + (k + 1) / scale, x.b, part)) bb[k].grow(part);
                    }
                }
                float la[kMaxBins]; int lc[kMaxBins];
                Box acc; acc.reset(); int cnt = 0;
                for (int k = 0; k < bins - 1; k++) { acc.grow(bb[k]); cnt += enter[k]; la[k] = acc.half_area(); lc[k] = cnt; }
                acc.reset(); cnt = 0;
                for (int k = bins - 1; k > 0; k--) {
                    acc.grow(bb[k]); cnt += leave[k];
                    if (lc[k - 1] == 0 || cnt == 0) continue;
                    float cost = la[k - 1] * (float)lc[k - 1] + acc.half_area() * (float)cnt;
                    if (cost < s_cost) { s_cost = cost; s_axis = a; s_plane = lo + k / scale; }
                }
            }
        }
        const bool use_spatial = s_axis >= 0 && s_cost < o_cost;
        const float best_cost = use_spatial ? s_cost : o_cost;
        if (small) {
            float pa = nb.half_area();
            if ((o_axis < 0 && !use_spatial) || !(pa > 0.0f) || kNodeCost + best_cost / pa >= (float)count) return make_leaf(r, nb);
        }
        std::vector<Ref> L, R;
        if (use_spatial) {
            const int a = s_axis;
            for (const Ref& x : r) {
                if ((double)x.b.hi[a] <= s_plane) L.push_back(x);
                else if ((double)x.b.lo[a] >= s_plane) R.push_back(x);
                else {
                    Ref l = x, rr = x; bool hl = clip_bounds(x.tri, a, -1.0e300, s_plane, x.b, l.b), hr = clip_bounds(x.tri, a, s_plane, 1.0e300, x.b, rr.b);
                    if (hl) { for (int k = 0; k < 3; k++) l.c[k] = 0.5f * (l.b.lo[k] + l.b.hi[k]); L.push_back(l); }
                    if (hr) { for (int k = 0; k < 3; k++) rr.c[k] = 0.5f * (rr.b.lo[k] + rr.b.hi[k]); R.push_back(rr); }
                    if (!hl && !hr) L.push_back(x);   // cannot happen for a triangle inside its own bounds; keep it somewhere
                }
            }
            if (L.empty() || R.empty() || ((int)L.size() == count && (int)R.size() == count)) { L.clear(); R.clear(); }   // no progress: object split instead
            else ref_total += L.size() + R.size() - (size_t)count;
        }
        if (L.empty() && R.empty()) {
            if (o_axis >= 0) {
                const int a = o_axis;
                const float ext = cb.hi[a] - cb.lo[a], scale = (float)bins / ext, lo = cb.lo[a];
                for (const Ref& x : r) {
                    int k = (int)((x.c[a] - lo) * scale);
                    k = k < 0 ? 0 : (k >= bins ? bins - 1 : k);
                    (k < o_bin ? L : R).push_back(x);
                }
            }
            if (L.empty() || R.empty()) {   // object median on the widest centroid axis (also the depth-bounding fallback)
                int a = 0; float e = -1.0f;
                for (int k = 0; k < 3; k++) { float x = cb.hi[k] - cb.lo[k]; if (x > e) { e = x; a = k; } }
                std::vector<Ref> all = r;
                std::stable_sort(all.begin(), all.end(), [a](const Ref& x, const Ref& y) { return x.c[a] < y.c[a] || (x.c[a] == y.c[a] && x.tri < y.tri); });
                L.assign(all.begin(), all.begin() + count / 2); R.assign(all.begin() + count / 2, all.end());
            }
        }
        std::vector<Ref>().swap(r);   // the parent's list is not needed below
        TmpNode n; n.b = nb; n.left = n.right = -1; n.first = 0; n.count = 0;
        const int id = (int)nodes.size();
        nodes.push_back(n);
        const int l = build_spatial(L, depth + 1);
        const int rr = build_spatial(R, depth + 1);
        nodes[id].left = l; nodes[id].right = rr;
        return id;
    }
};

inline int32_t leaf_code(int first, int count) { return ~(int32_t)(((uint32_t)first << 3) | (uint32_t)(count - 1)); }
}  // namespace

void build_bvh(const std::vector<BvhTri>& tris_in, std::vector<BvhNode>& nodes_out, std::vector<BvhNodeWide>& wide_out, std::vector<BvhTri>& tris_out, int* depth_out,
               std::vector<BvhNode8>* nodes8_out, bool spatial_splits, bool parallel) {
    BvhBuildOptions opt; opt.spatial_splits = spatial_splits; opt.parallel = parallel; opt.nodes8 = nodes8_out;
    build_bvh_ex(tris_in, nodes_out, wide_out, tris_out, depth_out, opt);
}
void build_bvh_ex(const std::vector<BvhTri>& tris_in, std::vector<BvhNode>& nodes_out, std::vector<BvhNodeWide>& wide_out, std::vector<BvhTri>& tris_out, int* depth_out,
                  const BvhBuildOptions& opt) {
    std::vector<BvhNode8>* const nodes8_out = opt.nodes8;
    const bool spatial_splits = opt.spatial_splits;
    nodes_out.clear(); wide_out.clear(); tris_out.clear();
    if (opt.nodes4s) opt.nodes4s->clear();
    if (opt.sah_cost) *opt.sah_cost = 0.0;
    Builder b;
    b.parallel = opt.parallel;
    b.bins = opt.bins < 2 ? 2 : (opt.bins > kMaxBins ? kMaxBins : opt.bins);
    b.refs.resize(tris_in.size());
    float maxabs = 0.0f;
    for (size_t i = 0; i < tris_in.size(); i++) {
        const BvhTri& t = tris_in[i];
        float p[3][3];
        for (int a = 0; a < 3; a++) { p[0][a] = t.v0[a]; p[1][a] = t.v0[a] + t.e1[a]; p[2][a] = t.v0[a] + t.e2[a]; }
        Ref& r = b.refs[i];
        r.b.reset(); r.b.grow(p[0]); r.b.grow(p[1]); r.b.grow(p[2]);
        for (int a = 0; a < 3; a++) { r.c[a] = 0.5f * (r.b.lo[a] + r.b.hi[a]); maxabs = std::max(maxabs, std::max(std::fabs(r.b.lo[a]), std::fabs(r.b.hi[a]))); }
        r.tri = (uint32_t)i;
    }
    // Conservative padding so a box test can never cull a triangle the shared ray_triangle() accepts.
    const float pad = 2.0e-5f * maxabs + 1.0e-6f;
    auto empty_node = [&]() {
        BvhNode n; std::memset(&n, 0, sizeof(n));
        for (int a = 0; a < 3; a++) { n.set_step(a, 1u); n.lo[a] = 0xffffffffu; n.hi[a] = 0u; }  // inverted: never entered
        for (int k = 0; k < 4; k++) n.child[k] = leaf_code(0, 1);
        return n;
    };
    auto empty_wide = [&]() {
        BvhNodeWide n; std::memset(&n, 0, sizeof(n));
        for (int k = 0; k < 4; k++) {
            n.minx[k] = n.miny[k] = n.minz[k] = n.maxx[k] = n.maxy[k] = n.maxz[k] = 1.0e30f;  // unreachable point box
            n.child[k] = leaf_code(0, 1);
        }
        return n;
    };
    if (tris_in.empty()) {
        nodes_out.push_back(empty_node()); wide_out.push_back(empty_wide());
        if (opt.nodes4s) opt.nodes4s->push_back(empty_node());
        if (depth_out) *depth_out = 0;
        return;
    }
    if (spatial_splits) {
        b.src = &tris_in;
        Box rb; rb.reset();
        for (const Ref& x : b.refs) rb.grow(x.b);
        b.root_area = rb.half_area();
        b.ref_total = b.refs.size(); b.ref_budget = b.refs.size() + b.refs.size() / 2;   // at most 1.5 references per triangle
        std::vector<Ref> all; all.swap(b.refs);
        b.build_spatial(all, 0);
        tris_out.resize(b.leaf_tris.size());
        for (size_t i = 0; i < b.leaf_tris.size(); i++) tris_out[i] = tris_in[b.leaf_tris[i]];
    } else {
        b.build(0, (int)b.refs.size(), 0);
        tris_out.resize(tris_in.size());
        for (size_t i = 0; i < b.refs.size(); i++) tris_out[i] = tris_in[b.refs[i].tri];
    }

    // emit: collapse the binary tree into 4-wide nodes (a node adopts its grandchildren, largest box first),
    // depth-first order; a leaf root gets a wrapper node.
    int max_depth = 0;
    // Quantise the (padded) child boxes of one node: origin = their common lower corner, step = the smallest
    // power of two whose 255 steps span them; lower planes round down, upper planes round up, checked in double
    // (origin + q * step is exact there) so the decoded box is a superset of the fp32 one.
    auto put_boxes = [&](BvhNode& n, BvhNodeWide& w, const Box* bx, int nk) {
        for (int k = 0; k < nk; k++) {
            w.minx[k] = bx[k].lo[0] - pad; w.miny[k] = bx[k].lo[1] - pad; w.minz[k] = bx[k].lo[2] - pad;
            w.maxx[k] = bx[k].hi[0] + pad; w.maxy[k] = bx[k].hi[1] + pad; w.maxz[k] = bx[k].hi[2] + pad;
        }
        for (int a = 0; a < 3; a++) {
            float lo = bx[0].lo[a] - pad, hi = bx[0].hi[a] + pad;
            for (int k = 1; k < nk; k++) { lo = std::min(lo, bx[k].lo[a] - pad); hi = std::max(hi, bx[k].hi[a] + pad); }
            const double org = lo, ext = (double)hi - (double)lo;
            int e = 1;  // biased exponent, step = 2^(e-127)
            if (ext > 0.0) { int ex; std::frexp(ext / 255.0, &ex); e = std::min(std::max(ex + 127, 1), 254); }
            while (e < 254 && org + 255.0 * std::ldexp(1.0, e - 127) < (double)hi) e++;
            const double step = std::ldexp(1.0, e - 127);
            n.origin[a] = lo;
            n.set_step(a, (uint32_t)e);
            uint32_t wl = 0xffffffffu, wh = 0u;
            for (int k = 0; k < nk; k++) {
                const double cl = (double)(bx[k].lo[a] - pad), ch = (double)(bx[k].hi[a] + pad);
                int ql = (int)std::floor((cl - org) / step), qh = (int)std::ceil((ch - org) / step);
                ql = std::min(std::max(ql, 0), 255); qh = std::min(std::max(qh, 0), 255);
                while (ql > 0 && org + ql * step > cl) ql--;
                while (qh < 255 && org + qh * step < ch) qh++;
                wl = (wl & ~(0xffu << (8 * k))) | (uint32_t)ql << (8 * k);
                wh = (wh & ~(0xffu << (8 * k))) | (uint32_t)qh << (8 * k);
            }
            n.lo[a] = wl; n.hi[a] = wh;
        }
    };
    struct Item { int tmp; int out; int depth; };
    std::vector<Item> work;
    nodes_out.push_back(empty_node()); wide_out.push_back(empty_wide());
    if (b.nodes[0].left < 0) {
        put_boxes(nodes_out[0], wide_out[0], &b.nodes[0].b, 1);
        nodes_out[0].child[0] = wide_out[0].child[0] = leaf_code(b.nodes[0].first, b.nodes[0].count);
    } else {
        work.push_back({0, 0, 0});
    }
    while (!work.empty()) {
        Item it = work.back(); work.pop_back();
        max_depth = std::max(max_depth, it.depth);
        int kids[4]; int nk = 0;
        if (opt.pairwise) {   // the two children's children (a leaf child stays as it is): two binary levels per four-wide node, always
            const int lr[2] = {b.nodes[it.tmp].left, b.nodes[it.tmp].right};
            for (int side = 0; side < 2; side++) {
                if (b.nodes[lr[side]].left >= 0) { kids[nk++] = b.nodes[lr[side]].left; kids[nk++] = b.nodes[lr[side]].right; }
                else kids[nk++] = lr[side];
            }
        } else {
            kids[nk++] = b.nodes[it.tmp].left; kids[nk++] = b.nodes[it.tmp].right;
            while (nk < 4) {
                int best = -1; float ba = -1.0f;
                for (int k = 0; k < nk; k++)
                    if (b.nodes[kids[k]].left >= 0) { float a = b.nodes[kids[k]].b.half_area(); if (a > ba) { ba = a; best = k; } }
                if (best < 0) break;
                int t = kids[best];
                kids[best] = b.nodes[t].left; kids[nk++] = b.nodes[t].right;
            }
        }
        Box boxes[4];
        for (int k = 0; k < nk; k++) boxes[k] = b.nodes[kids[k]].b;
        put_boxes(nodes_out[it.out], wide_out[it.out], boxes, nk);
        for (int k = 0; k < nk; k++) {
            const TmpNode& c = b.nodes[kids[k]];
            if (c.left < 0) {
                nodes_out[it.out].child[k] = wide_out[it.out].child[k] = leaf_code(c.first, c.count);
            } else {
                int idx = (int)nodes_out.size();
                nodes_out.push_back(empty_node()); wide_out.push_back(empty_wide());
                nodes_out[it.out].child[k] = wide_out[it.out].child[k] = idx;
                work.push_back({kids[k], idx, it.depth + 1});
            }
        }
    }
    if (depth_out) *depth_out = max_depth;
    // The top of the tree first, in breadth-first order: nodes 0 .. kBvhTopNodes-1 are the ones the traversal kernels keep in LDS
    // (vote.hpp LaneStack / kernels_trace.hip: every ray visits them, and a node fetched from LDS costs the vector L1 nothing);
    // the rest keep their depth-first order (subtrees contiguous).  A pure renumbering: boxes, children and leaves are unchanged.
    auto top_first = [&](std::vector<BvhNode>& nv, std::vector<BvhNodeWide>* wv) {
        const size_t n = nv.size();
        std::vector<int> order; order.reserve(n);
        std::vector<char> taken(n, 0);
        order.push_back(0); taken[0] = 1;
        for (size_t head = 0; head < order.size() && order.size() < (size_t)kBvhTopNodes; head++)
            for (int k = 0; k < 4 && order.size() < (size_t)kBvhTopNodes; k++) {
                const int c = nv[order[head]].child[k];
                if (c >= 0 && !taken[c]) { taken[c] = 1; order.push_back(c); }
            }
        for (size_t i = 0; i < n; i++) if (!taken[i]) order.push_back((int)i);
        std::vector<int> new_index(n);
        for (size_t i = 0; i < n; i++) new_index[order[i]] = (int)i;
        std::vector<BvhNode> nn(n); std::vector<BvhNodeWide> nw(wv ? n : 0);
        for (size_t i = 0; i < n; i++) {
            nn[i] = nv[order[i]]; if (wv) nw[i] = (*wv)[order[i]];
            for (int k = 0; k < 4; k++) if (nn[i].child[k] >= 0) { nn[i].child[k] = new_index[nn[i].child[k]]; if (wv) nw[i].child[k] = nn[i].child[k]; }
        }
        nv.swap(nn); if (wv) wv->swap(nw);
    };
    top_first(nodes_out, &wide_out);
    if (opt.sah_cost) {
        double cost = 0.0;
        for (const TmpNode& t : b.nodes) cost += (t.left >= 0 ? (double)kNodeCost : (double)t.count) * (double)t.b.half_area();
        const double ra = (double)b.nodes[0].b.half_area();
        *opt.sah_cost = ra > 0.0 ? cost / ra : 0.0;
    }
    if (opt.nodes4s) {
        // ---- the split-order tree: slots 0,1 = the left child's children (or the left child itself in slot 0 when it is a leaf), slots 2,3 = the
        // right child's.  For every pair the axis along which its two centres differ most decides the order: a ray whose direction is negative
        // along it meets the pair in reverse.  Tables: bit `oct` (= negx | negy << 1 | negz << 2) of byte 0 / byte 1 of step_x's mantissa for the
        // left / right pair, of byte 0 of step_y's mantissa for the two pairs themselves.
        std::vector<BvhNode>& out = *opt.nodes4s;
        std::vector<BvhNodeWide> dummy_w;
        auto centre = [&](int t, int a) { return 0.5f * (b.nodes[t].b.lo[a] + b.nodes[t].b.hi[a]); };
        auto order_table = [&](int ta, int tb) -> uint32_t {   // bit oct = 1: tb comes first
            if (ta < 0 || tb < 0) return 0u;
            int axis = 0; float best = -1.0f;
            for (int a = 0; a < 3; a++) { const float d = std::fabs(centre(ta, a) - centre(tb, a)); if (d > best) { best = d; axis = a; } }
            const bool a_low = centre(ta, axis) <= centre(tb, axis);
            uint32_t tab = 0u;
            for (uint32_t oct = 0; oct < 8; oct++) { const bool neg = ((oct >> axis) & 1u) != 0u; if (neg == a_low) tab |= 1u << oct; }
            return tab;
        };
        struct Item4 { int tmp; int out; };
        std::vector<Item4> work4;
        out.push_back(empty_node()); dummy_w.push_back(empty_wide());
        if (b.nodes[0].left < 0) {
            put_boxes(out[0], dummy_w[0], &b.nodes[0].b, 1);
            out[0].child[0] = leaf_code(b.nodes[0].first, b.nodes[0].count);
        } else work4.push_back({0, 0});
        while (!work4.empty()) {
            const Item4 it = work4.back(); work4.pop_back();
            const int L = b.nodes[it.tmp].left, R = b.nodes[it.tmp].right;
            int kid[4] = {-1, -1, -1, -1};
            if (b.nodes[L].left >= 0) { kid[0] = b.nodes[L].left; kid[1] = b.nodes[L].right; } else kid[0] = L;
            if (b.nodes[R].left >= 0) { kid[2] = b.nodes[R].left; kid[3] = b.nodes[R].right; } else kid[2] = R;
            // put_boxes takes a dense list: unused slots are written as inverted boxes afterwards
            Box boxes[4]; int map[4]; int nk = 0;
            for (int k = 0; k < 4; k++) if (kid[k] >= 0) { boxes[nk] = b.nodes[kid[k]].b; map[nk] = k; nk++; }
            BvhNode dense = empty_node(); BvhNodeWide dw = empty_wide();
            put_boxes(dense, dw, boxes, nk);
            BvhNode n = empty_node();
            for (int a = 0; a < 3; a++) { n.origin[a] = dense.origin[a]; }
            n.step_x = dense.step_x; n.step_y = dense.step_y; n.step_z = dense.step_z;
            for (int a = 0; a < 3; a++) {
                uint32_t wl = 0xffffffffu, wh = 0u;
                for (int j = 0; j < nk; j++) {
                    const int k = map[j];
                    wl = (wl & ~(0xffu << (8 * k))) | (((dense.lo[a] >> (8 * j)) & 0xffu) << (8 * k));
                    wh = (wh & ~(0xffu << (8 * k))) | (((dense.hi[a] >> (8 * j)) & 0xffu) << (8 * k));
                }
                n.lo[a] = wl; n.hi[a] = wh;
            }
            const uint32_t tabL = order_table(kid[0], kid[1]), tabR = order_table(kid[2], kid[3]), tabT = order_table(L, R);
            uint32_t bx, by; std::memcpy(&bx, &n.step_x, 4); std::memcpy(&by, &n.step_y, 4);
            bx |= tabL | (tabR << 8); by |= tabT;
            std::memcpy(&n.step_x, &bx, 4); std::memcpy(&n.step_y, &by, 4);
            out[it.out] = n;
            for (int k = 0; k < 4; k++) {
                if (kid[k] < 0) continue;
                const TmpNode& c = b.nodes[kid[k]];
                if (c.left < 0) out[it.out].child[k] = leaf_code(c.first, c.count);
                else {
                    const int idx = (int)out.size();
                    out.push_back(empty_node());
                    out[it.out].child[k] = idx;
                    work4.push_back({kid[k], idx});
                }
            }
        }
        top_first(out, nullptr);
    }
    if (!nodes8_out) return;

    // ---- the same binary tree collapsed EIGHT-wide (BVH8 experiment): a node adopts descendants, largest box first, until it
    // has eight children; the children go into slots by octant (greedy assignment on the offset of the child's centre from the
    // node's centre along the slot's diagonal, after Ylitie et al. 2017), so a traversal order follows from the ray's direction
    // signs alone.
    std::vector<BvhNode8>& out8 = *nodes8_out;
    out8.clear();
    auto empty8 = [&]() {
        BvhNode8 n; std::memset(&n, 0, sizeof(n));
        n.exps = 1u | 1u << 8 | 1u << 16;
        for (int a = 0; a < 3; a++) { n.lo[a][0] = n.lo[a][1] = 0xffffffffu; n.hi[a][0] = n.hi[a][1] = 0u; }
        for (int k = 0; k < 8; k++) n.child[k] = leaf_code(0, 1);
        return n;
    };
    auto put_boxes8 = [&](BvhNode8& n, const Box* bx, const bool* used) {
        n.exps = 0;
        for (int a = 0; a < 3; a++) {
            float lo = 3.0e38f, hi = -3.0e38f;
            for (int k = 0; k < 8; k++) if (used[k]) { lo = std::min(lo, bx[k].lo[a] - pad); hi = std::max(hi, bx[k].hi[a] + pad); }
            const double org = lo, ext = (double)hi - (double)lo;
            int e = 1;
            if (ext > 0.0) { int ex; std::frexp(ext / 255.0, &ex); e = std::min(std::max(ex + 127, 1), 254); }
            while (e < 254 && org + 255.0 * std::ldexp(1.0, e - 127) < (double)hi) e++;
            const double step = std::ldexp(1.0, e - 127);
            n.origin[a] = lo;
            n.exps |= (uint32_t)e << (8 * a);
            for (int k = 0; k < 8; k++) {
                int ql = 255, qh = 0;  // unused slot: inverted
                if (used[k]) {
                    const double cl = (double)(bx[k].lo[a] - pad), ch = (double)(bx[k].hi[a] + pad);
                    ql = (int)std::floor((cl - org) / step); qh = (int)std::ceil((ch - org) / step);
                    ql = std::min(std::max(ql, 0), 255); qh = std::min(std::max(qh, 0), 255);
                    while (ql > 0 && org + ql * step > cl) ql--;
                    while (qh < 255 && org + qh * step < ch) qh++;
                }
                uint32_t& wl = n.lo[a][k >> 2]; uint32_t& wh = n.hi[a][k >> 2];
                const int sh = 8 * (k & 3);
                wl = (wl & ~(0xffu << sh)) | (uint32_t)ql << sh;
                wh = (wh & ~(0xffu << sh)) | (uint32_t)qh << sh;
            }
        }
    };
    out8.push_back(empty8());
    if (b.nodes[0].left < 0) {
        Box bx[8]; bool used[8] = {true, false, false, false, false, false, false, false};
        bx[0] = b.nodes[0].b;
        put_boxes8(out8[0], bx, used);
        out8[0].child[0] = leaf_code(b.nodes[0].first, b.nodes[0].count);
        return;
    }
    std::vector<std::pair<int, int>> work8;  // (binary node, output node)
    work8.push_back({0, 0});
    while (!work8.empty()) {
        const std::pair<int, int> it = work8.back(); work8.pop_back();
        int kids[8]; int nk = 0;
        kids[nk++] = b.nodes[it.first].left; kids[nk++] = b.nodes[it.first].right;
        while (nk < 8) {
            int best = -1; float ba = -1.0f;
            for (int k = 0; k < nk; k++)
                if (b.nodes[kids[k]].left >= 0) { float a = b.nodes[kids[k]].b.half_area(); if (a > ba) { ba = a; best = k; } }
            if (best < 0) break;
            const int t = kids[best];
            kids[best] = b.nodes[t].left; kids[nk++] = b.nodes[t].right;
        }
        // octant slots: repeatedly take the (child, free slot) pair with the largest projection of the child's offset on the slot's diagonal
        const Box& nb = b.nodes[it.first].b;
        float cx[3]; for (int a = 0; a < 3; a++) cx[a] = 0.5f * (nb.lo[a] + nb.hi[a]);
        int slot_of[8]; bool taken[8] = {false, false, false, false, false, false, false, false}, placed[8] = {false, false, false, false, false, false, false, false};
        for (int round = 0; round < nk; round++) {
            int bk = -1, bs = -1; float bc = -3.0e38f;
            for (int k = 0; k < nk; k++) {
                if (placed[k]) continue;
                const Box& cb = b.nodes[kids[k]].b;
                for (int sidx = 0; sidx < 8; sidx++) {
                    if (taken[sidx]) continue;
                    float c = 0.0f;
                    for (int a = 0; a < 3; a++) c += (0.5f * (cb.lo[a] + cb.hi[a]) - cx[a]) * (((sidx >> a) & 1) ? 1.0f : -1.0f);
                    if (c > bc) { bc = c; bk = k; bs = sidx; }
                }
            }
            slot_of[bk] = bs; placed[bk] = true; taken[bs] = true;
        }
        Box boxes[8]; bool used[8] = {false, false, false, false, false, false, false, false};
        for (int k = 0; k < nk; k++) { boxes[slot_of[k]] = b.nodes[kids[k]].b; used[slot_of[k]] = true; }
        put_boxes8(out8[it.second], boxes, used);
        for (int k = 0; k < nk; k++) {
            const TmpNode& c = b.nodes[kids[k]];
            if (c.left < 0) out8[it.second].child[slot_of[k]] = leaf_code(c.first, c.count);
            else {
                const int idx = (int)out8.size();
                out8.push_back(empty8());
                out8[it.second].child[slot_of[k]] = idx;
                work8.push_back({kids[k], idx});
            }
        }
    }
}

}  // namespace vpt
