// vote_sim — host model of ONE wave of the vote-scheduled traversal kernels (csrc/kernels_trace.hip k_trace_vote / k_trace_shadow,
// csrc/vote.hpp) running a ray stream over the product BVH: which kind of step the wave votes for, how many of its 64 lanes take part,
// and what the schedule costs in VALU wave-instructions per ray (step costs from profiles/r03_trace_isa_budget.md).  It exists to
// cost scheduling policies before a kernel family is spent on them; every policy must return the same hits as the first.
//   vote_sim tris.bin rays.bin kind      rays: m x 10 floats {o, tmin, d, tmax, tlim, expect-gid as float bits}; kind: closest | any
// Policies: 0 = the product's (weighted vote, fetch at 24 idle lanes); 1.. = postponed leaves (see Policy).  Test utility only.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../vulkan-path-tracer_amd/csrc/bvh_build.hpp"
using namespace vpt;
using vptfp::V3;

static const float kMiss = 3.0e38f;
static const int kIdle = 0x7fffffff, kDone = 0x7ffffffe, kEmpty = 0x7ffffffd;
struct Slab { V3 o, inv; bool nx, ny, nz; };
static float sinv(float d) { return (std::fabs(d) > 1e-30f) ? 1.0f / d : (std::signbit(d) ? -1e30f : 1e30f); }
static void entries(const BvhNode& n, const Slab& r, float tmin, float tlimit, float t[4]) {
    const float ax = n.step_x * r.inv.x, ay = n.step_y * r.inv.y, az = n.step_z * r.inv.z;
    const float bx = (n.origin[0] - r.o.x) * r.inv.x, by = (n.origin[1] - r.o.y) * r.inv.y, bz = (n.origin[2] - r.o.z) * r.inv.z;
    const uint32_t nxw = r.nx ? n.hi[0] : n.lo[0], fxw = r.nx ? n.lo[0] : n.hi[0];
    const uint32_t nyw = r.ny ? n.hi[1] : n.lo[1], fyw = r.ny ? n.lo[1] : n.hi[1];
    const uint32_t nzw = r.nz ? n.hi[2] : n.lo[2], fzw = r.nz ? n.lo[2] : n.hi[2];
    for (int k = 0; k < 4; k++) {
        auto B = [&](uint32_t w) { return (float)((w >> (8 * k)) & 0xffu); };
        float tn = std::fmax(std::fmax(std::fmaf(B(nxw), ax, bx), std::fmaf(B(nyw), ay, by)), std::fmax(std::fmaf(B(nzw), az, bz), tmin));
        float tf = std::fmin(std::fmin(std::fmaf(B(fxw), ax, bx), std::fmaf(B(fyw), ay, by)), std::fmin(std::fmaf(B(fzw), az, bz), tlimit));
        t[k] = (tn <= tf * 1.0000005f) ? tn : kMiss;
    }
}

static BvhNode clean4s(BvhNode n) {   // the split-order tree keeps its order tables in the mantissas of step_x / step_y
    uint32_t b; memcpy(&b, &n.step_x, 4); b &= 0x7f800000u; memcpy(&n.step_x, &b, 4);
    memcpy(&b, &n.step_y, 4); b &= 0x7f800000u; memcpy(&n.step_y, &b, 4);
    return n;
}
struct Policy {
    const char* name;
    int stash;          // leaves a lane may set aside while it goes on with inner nodes (0: the product)
    int w4;             // a node step wins when 4 x (lanes able to take one) > w4 x (lanes able to take a triangle step)
    int fetch_at;       // idle lanes that trigger a fetch step
    int node_extra, tri_extra;   // VALU a step costs more than the product's (stash bookkeeping)
    bool drain_first;   // a triangle step wins outright while some lane can do nothing else and holds a full stash
    int tri_per_step;   // triangles of ONE leaf a triangle step tests (1: the product)
    int pair;           // 1 (with pool 128): lane i owns slots i and i + 64 — two rays in its registers — and serves at most one of them per step
    int pool;           // ray slots per wave (64: a ray lives in a lane's registers, the product).  More: rays live in LDS slots and a step runs on
                        // up to 64 of the slots that want it (lanes are workers, not owners)
    int split_order = 0;   // 1: the split-order tree (bvh_build.hpp BvhBuildOptions::nodes4s): hit children in the order the binary splits give for the ray's octant, no distance sort
};

struct Lane {
    int cur = kIdle, sp = 0;
    int stash[4] = {kEmpty, kEmpty, kEmpty, kEmpty}; int ns = 0;
    std::vector<int> st;
    Slab s; V3 o, d; float tmin, tmax, tlim, best; int bgid; uint32_t expect; bool found; size_t rid;
};

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    std::vector<BvhTri> tris; std::vector<float> rays;
    { FILE* f = fopen(argv[1], "rb"); if (!f) return 3; fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); tris.resize(n / 48); if (fread(tris.data(), 48, tris.size(), f) != tris.size()) return 3; fclose(f); }
    { FILE* f = fopen(argv[2], "rb"); if (!f) return 3; fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); rays.resize(n / 4); if (fread(rays.data(), 4, rays.size(), f) != rays.size()) return 3; fclose(f); }
    const bool any = std::string(argv[3]) == "any";
    {
        std::vector<BvhTri> keep;
        for (const BvhTri& t : tris) if (!vptfp::triangle_degenerate(vptfp::v3(t.e1[0], t.e1[1], t.e1[2]), vptfp::v3(t.e2[0], t.e2[1], t.e2[2]))) keep.push_back(t);
        tris.swap(keep);
    }
    std::vector<BvhNode> nodes, nodes4s; std::vector<BvhNodeWide> wide; std::vector<BvhTri> leaf; int depth = 0;
    // builder study (round 6): SIM_BINS = SAH bins per axis (16: the product), SIM_SBVH=1 = spatial splits, SIM_STUDY=1 = only the product policy and the split-order one
    BvhBuildOptions opt; double sah = 0.0;
    opt.bins = getenv("SIM_BINS") ? atoi(getenv("SIM_BINS")) : 16; opt.spatial_splits = getenv("SIM_SBVH") && atoi(getenv("SIM_SBVH")) != 0;
    opt.nodes4s = &nodes4s; opt.sah_cost = &sah;
    build_bvh_ex(tris, nodes, wide, leaf, &depth, opt);
    const bool study = getenv("SIM_STUDY") && atoi(getenv("SIM_STUDY")) != 0;
    const size_t nrays = rays.size() / 10;
    printf("%s: %zu rays, %zu triangles, %zu nodes (split-order tree: %zu), depth %d, bins %d, sbvh %d, SAH cost %.3f\n", argv[3], nrays, leaf.size(), nodes.size(), nodes4s.size(), depth, opt.bins, (int)opt.spatial_splits, sah);
    const int node_cost = any ? 133 : 155, tri_cost = any ? 91 : 97, fetch_cost = 148;
    const Policy policies[] = {
        {"product (vote 2.0, fetch at 24)", 0, 8, 24, 0, 0, false, 1, 0, 64},
        {"product, fetch at 8", 0, 8, 8, 0, 0, false, 1, 0, 64},
        {"product, vote 1.0", 0, 4, 24, 0, 0, false, 1, 0, 64},
        {"stash 1, vote 2.0", 1, 8, 24, 14, 8, false, 1, 0, 64},
        {"stash 1, vote 1.0", 1, 4, 24, 14, 8, false, 1, 0, 64},
        {"stash 1, vote 4.0", 1, 16, 24, 14, 8, false, 1, 0, 64},
        {"stash 1, vote 2.0, drain first", 1, 8, 24, 14, 8, true, 1, 0, 64},
        {"stash 2, vote 2.0", 2, 8, 24, 20, 12, false, 1, 0, 64},
        {"stash 2, vote 4.0", 2, 16, 24, 20, 12, false, 1, 0, 64},
        {"stash 4, vote 4.0", 4, 16, 24, 28, 16, false, 1, 0, 64},
        {"stash 1, vote 2.0, fetch at 8", 1, 8, 8, 14, 8, false, 1, 0, 64},
        {"product, 2 triangles per step", 0, 8, 24, 0, 70, false, 2, 0, 64},
        {"2 triangles per step, vote 1.5", 0, 6, 24, 0, 70, false, 2, 0, 64},
        {"2 triangles per step, vote 1.0", 0, 4, 24, 0, 70, false, 2, 0, 64},
        {"2 triangles per step, vote 3.0", 0, 12, 24, 0, 70, false, 2, 0, 64},
        {"4 triangles per step, vote 1.0", 0, 4, 24, 0, 210, false, 4, 0, 64},
        {"pool 80, fetch at 16", 0, 8, 16, 20, 20, false, 1, 0, 80},
        {"pool 96, fetch at 24", 0, 8, 24, 20, 20, false, 1, 0, 96},
        {"pool 96, fetch at 32", 0, 8, 32, 20, 20, false, 1, 0, 96},
        {"pool 128, fetch at 32", 0, 8, 32, 20, 20, false, 1, 0, 128},
        {"pool 128, fetch at 48", 0, 8, 48, 20, 20, false, 1, 0, 128},
        {"pool 128, vote 4.0, fetch at 48", 0, 16, 48, 20, 20, false, 1, 0, 128},
        {"pool 192, fetch at 64", 0, 8, 64, 20, 20, false, 1, 0, 192},
        {"two rays per lane (registers), fetch at 48", 0, 8, 48, 28, 28, false, 1, 1, 128},
        {"two rays per lane, 2 triangles per step", 0, 8, 48, 28, 98, false, 2, 1, 128},
        // round 6: the split-order tree.  Its node step drops the five-exchange sorting network (25 VALU) and the four miss selects, and adds three table
        // tests (6), one mask of the two step words (2) and the three-exchange butterfly on the child codes (8 v_cndmask): 155 - 29 + 16 = 142 by the count, 134 if the
        // hit masks are permuted on the scalar unit
        {"split order, 2 triangles per step", 0, 8, 24, -13, 70, false, 2, 0, 64, 1},
        {"split order (scalar masks), 2 tri", 0, 8, 24, -21, 70, false, 2, 0, 64, 1},
        // the pair-wise collapse alone (children = the two children's children), nearest-first by entry distance as in the product, any-hit in slot order
        {"pair-wise collapse, 2 tri per step", 0, 8, 24, 0, 70, false, 2, 0, 64, 2},
    };
    std::vector<float> ref_t(nrays); std::vector<int> ref_g(nrays);
    for (size_t pi = 0; pi < sizeof(policies) / sizeof(policies[0]); pi++) {
        const Policy& P = policies[pi];
        if (study && !(pi == 0 || pi == 11 || P.split_order)) continue;
        if (P.split_order == 1 && any) continue;   // (an any-hit search has no order)
        std::vector<Lane> L((size_t)P.pool);
        const int pool = P.pool;
        size_t next = 0; bool exhausted = false;
        double n_node_steps = 0, n_tri_steps = 0, n_fetch_steps = 0, part_node = 0, part_tri = 0, visits = 0, tests = 0, idle_lane_steps = 0;
        size_t mismatches = 0;
        auto pop_or_done = [&](Lane& l) {
            if (!l.st.empty()) { l.cur = l.st.back(); l.st.pop_back(); }
            else if (l.ns > 0) { l.cur = l.stash[--l.ns]; l.stash[l.ns] = kEmpty; }
            else l.cur = kDone;
        };
        auto set_aside = [&](Lane& l) {   // a lane that arrived at a leaf sets it aside and goes on with an inner node, if it has room and something to go on with
            while (l.cur < 0 && l.ns < P.stash && !l.st.empty()) { l.stash[l.ns++] = l.cur; l.cur = l.st.back(); l.st.pop_back(); }
        };
        while (true) {
            int nn = 0, nl = 0, stuck = 0;
            for (Lane& l : L) {
                const bool busy = l.cur < kDone;
                const bool can_node = busy && l.cur >= 0, can_tri = busy && (l.cur < 0 || l.ns > 0);
                nn += can_node; nl += can_tri; stuck += busy && !can_node;
            }
            int busy_lanes = 0; for (Lane& l : L) busy_lanes += l.cur < kDone;
            const bool want_fetch = (!exhausted && pool - busy_lanes >= P.fetch_at) || busy_lanes == 0;
            if (want_fetch) {
                if (exhausted) break;
                n_fetch_steps++;
                for (Lane& l : L) {
                    if (l.cur == kDone) {
                        if (pi == 0) { ref_t[l.rid] = l.found ? l.best : -1.0f; ref_g[l.rid] = l.bgid; }
                        else if ((l.found ? l.best : -1.0f) != ref_t[l.rid] || (!any && l.bgid != ref_g[l.rid])) mismatches++;
                        l.cur = kIdle;
                    }
                    if (l.cur == kIdle && next < nrays) {
                        const float* q = &rays[next * 10];
                        l.rid = next++;
                        l.o = vptfp::v3(q[0], q[1], q[2]); l.d = vptfp::v3(q[4], q[5], q[6]); l.tmin = q[3]; l.tmax = q[7]; l.tlim = q[8];
                        memcpy(&l.expect, &q[9], 4);
                        l.s.o = l.o; l.s.inv = vptfp::v3(sinv(l.d.x), sinv(l.d.y), sinv(l.d.z)); l.s.nx = l.s.inv.x < 0; l.s.ny = l.s.inv.y < 0; l.s.nz = l.s.inv.z < 0;
                        l.best = any ? l.tlim : l.tmax; l.bgid = -1; l.found = false; l.st.clear(); l.ns = 0; l.cur = 0;
                    }
                }
                if (next >= nrays) exhausted = true;
                continue;
            }
            if (P.pair) {   // the vote counts LANES that could take each kind of step
                nn = 0; nl = 0;
                for (int i = 0; i < 64; i++) {
                    nn += (L[i].cur < kDone && L[i].cur >= 0) || (L[i + 64].cur < kDone && L[i + 64].cur >= 0);
                    nl += L[i].cur < 0 || L[i + 64].cur < 0;
                }
            }
            bool node_wins = 4 * std::min(nn, 64) > P.w4 * std::min(nl, 64);   // (a step serves 64 slots at most)
            if (pool > 64 && !P.pair) node_wins = (nn >= 64 && nl < 64) ? true : (nl >= 64 && nn < 64) ? false : 4 * nn > P.w4 * nl;
            if (P.drain_first && stuck > 0 && nl > 0) { bool full = false; for (Lane& l : L) full |= (l.cur < 0 && l.ns == P.stash); if (full && nn < 48) node_wins = false; }
            if (nl == 0) node_wins = true;
            if (nn == 0) node_wins = false;
            idle_lane_steps += pool - busy_lanes;
            if (node_wins) {
                n_node_steps++;
                std::vector<char> take(L.size(), 1);
                if (P.pair) {   // a lane serves one of its two slots
                    int lanes_busy = 0;
                    for (int i = 0; i < 64; i++) {
                        const bool a0 = L[i].cur < kDone && L[i].cur >= 0, a1 = L[i + 64].cur < kDone && L[i + 64].cur >= 0;
                        take[i] = a0; take[i + 64] = !a0 && a1; lanes_busy += a0 || a1;
                    }
                    part_node += lanes_busy;
                } else part_node += std::min(nn, 64);
                int served = 0;
                for (size_t li = 0; li < L.size(); li++) {
                    Lane& l = L[li];
                    if (!(l.cur < kDone && l.cur >= 0)) continue;
                    if (!take[li]) continue;
                    if (++served > 64) break;
                    visits++;
                    const BvhNode nd = P.split_order ? clean4s(nodes4s[l.cur]) : nodes[l.cur];
                    float t[4]; entries(nd, l.s, l.tmin, l.best, t);
                    int c[4] = {nd.child[0], nd.child[1], nd.child[2], nd.child[3]};
                    if (P.split_order == 1) {   // vote.hpp vote_node4s_step
                        uint32_t bx, by; memcpy(&bx, &nodes4s[l.cur].step_x, 4); memcpy(&by, &nodes4s[l.cur].step_y, 4);
                        const uint32_t oct = (uint32_t)l.s.nx | (uint32_t)l.s.ny << 1 | (uint32_t)l.s.nz << 2;
                        if ((bx >> oct) & 1u) { std::swap(t[0], t[1]); std::swap(c[0], c[1]); }
                        if ((bx >> (8 + oct)) & 1u) { std::swap(t[2], t[3]); std::swap(c[2], c[3]); }
                        if ((by >> oct) & 1u) { std::swap(t[0], t[2]); std::swap(c[0], c[2]); std::swap(t[1], t[3]); std::swap(c[1], c[3]); }
                        int first = -1;
                        for (int k = 0; k < 4; k++) if (t[k] < kMiss) { first = k; break; }
                        if (first < 0) pop_or_done(l);
                        else { for (int k = 3; k > first; k--) if (t[k] < kMiss) l.st.push_back(c[k]); l.cur = c[first]; }
                    } else if (any) {
                        int nxt = kIdle;
                        for (int k = 3; k >= 0; k--) if (t[k] < kMiss) { if (nxt != kIdle) l.st.push_back(nxt); nxt = c[k]; }
                        if (nxt != kIdle) l.cur = nxt; else pop_or_done(l);
                    } else {
                        for (int i = 0; i < 4; i++) for (int j = i + 1; j < 4; j++) if (t[j] < t[i]) { std::swap(t[i], t[j]); std::swap(c[i], c[j]); }
                        if (t[0] < kMiss) { if (t[3] < kMiss) l.st.push_back(c[3]); if (t[2] < kMiss) l.st.push_back(c[2]); if (t[1] < kMiss) l.st.push_back(c[1]); l.cur = c[0]; }
                        else pop_or_done(l);
                    }
                    set_aside(l);
                }
            } else {
                n_tri_steps++;
                std::vector<char> take(L.size(), 1);
                if (P.pair) {
                    int lanes_busy = 0;
                    for (int i = 0; i < 64; i++) {
                        const bool a0 = L[i].cur < 0, a1 = L[i + 64].cur < 0;
                        take[i] = a0; take[i + 64] = !a0 && a1; lanes_busy += a0 || a1;
                    }
                    part_tri += lanes_busy;
                } else part_tri += std::min(nl, 64);
                int served = 0;
                for (size_t li = 0; li < L.size(); li++) {
                    Lane& l = L[li];
                    if (!(l.cur < kDone && (l.cur < 0 || l.ns > 0))) continue;
                    if (!take[li]) continue;
                    if (++served > 64) break;
                    const bool own = l.cur < 0;
                    int code = own ? l.cur : l.stash[l.ns - 1];
                    bool stop = false;
                    for (int rep = 0; rep < P.tri_per_step && code != kEmpty && !stop; rep++) {
                        const uint32_t enc = (uint32_t)(~code); const int first = (int)(enc >> 3); const uint32_t more = enc & 7u;
                        const BvhTri& tr = leaf[first]; tests++;
                        float tt, u, v;
                        const bool hit = vptfp::ray_triangle(l.o, l.d, vptfp::v3(tr.v0[0], tr.v0[1], tr.v0[2]), vptfp::v3(tr.e1[0], tr.e1[1], tr.e1[2]), vptfp::v3(tr.e2[0], tr.e2[1], tr.e2[2]), l.tmin, l.tmax, &tt, &u, &v);
                        if (any) { if (hit && (tt < l.tlim || (tt == l.tlim && tr.gid < l.expect))) stop = true; }
                        else if (hit && (!l.found || tt < l.best || (tt == l.best && (int)tr.gid < l.bgid))) { l.best = tt; l.bgid = (int)tr.gid; l.found = true; }
                        code = more ? ~(int)((((uint32_t)first + 1u) << 3) | (more - 1u)) : kEmpty;
                    }
                    if (stop) { l.found = true; l.best = 1.0f; l.cur = kDone; l.ns = 0; l.st.clear(); continue; }
                    if (own) { if (code != kEmpty) l.cur = code; else { pop_or_done(l); set_aside(l); } }
                    else { if (code != kEmpty) l.stash[l.ns - 1] = code; else l.stash[--l.ns] = kEmpty; }
                }
            }
        }
        for (Lane& l : L) if (l.cur == kDone) { if (pi == 0) { ref_t[l.rid] = l.found ? l.best : -1.0f; ref_g[l.rid] = l.bgid; } else if ((l.found ? l.best : -1.0f) != ref_t[l.rid] || (!any && l.bgid != ref_g[l.rid])) mismatches++; }
        const double cost = n_node_steps * (node_cost + P.node_extra) + n_tri_steps * (tri_cost + P.tri_extra) + n_fetch_steps * fetch_cost;
        const double lane_instr = part_node * (node_cost + P.node_extra) + part_tri * (tri_cost + P.tri_extra) + n_fetch_steps * fetch_cost * (double)P.fetch_at;
        printf("%-36s VALU/ray %7.1f | node steps/ray %6.3f (%4.1f lanes) tri %6.3f (%4.1f lanes) fetch %5.3f | visits/ray %6.2f tests %5.2f | lane use %4.1f %% | idle lanes %4.1f | mismatches %zu\n", P.name,
               cost / nrays, n_node_steps / nrays, part_node / std::max(1.0, n_node_steps), n_tri_steps / nrays, part_tri / std::max(1.0, n_tri_steps), n_fetch_steps / nrays,
               visits / nrays, tests / nrays, 100.0 * lane_instr / (64.0 * cost), idle_lane_steps / std::max(1.0, n_node_steps + n_tri_steps), mismatches);
    }
    return 0;
}
