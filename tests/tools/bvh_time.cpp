// bvh_time — builds the product BVH (csrc/bvh_build.cpp) over tris.bin twice, multi-threaded and single-threaded, prints both wall times
// and an FNV-1a hash of each result (nodes + leaf-ordered triangles).  Test utility only (tests/test_bvh_host.py).
//   bvh_time tris.bin          tris: n x 12 dwords {v0,e1,e2,prim,inst,gid}
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../vulkan-path-tracer_amd/csrc/bvh_build.hpp"
using namespace vpt;
static unsigned long long fnv(const void* p, size_t n, unsigned long long h) {
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}
int main(int argc, char** argv) {
    if (argc < 2) return 2;
    std::vector<BvhTri> tris;
    { FILE* f = fopen(argv[1], "rb"); if (!f) return 3; fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); tris.resize(n / 48); if (fread(tris.data(), 48, tris.size(), f) != tris.size()) return 3; fclose(f); }
    for (int par = 1; par >= 0; par--) {
        std::vector<BvhNode> nodes; std::vector<BvhNodeWide> wide; std::vector<BvhTri> out; int depth = 0;
        double ms = 1e30;
        for (int rep = 0; rep < 3; rep++) {   // the fastest of three (the first build of a process pays the page faults of its allocations)
            const auto t0 = std::chrono::steady_clock::now();
            build_bvh(tris, nodes, wide, out, &depth, nullptr, false, par != 0);
            ms = std::min(ms, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }
        unsigned long long h = fnv(nodes.data(), nodes.size() * sizeof(BvhNode), 1469598103934665603ull);
        h = fnv(out.data(), out.size() * sizeof(BvhTri), h);
        printf("%s ms %.1f nodes %zu tris %zu depth %d hash %016llx\n", par ? "parallel" : "serial", ms, nodes.size(), out.size(), depth, h);
    }
    return 0;
}
