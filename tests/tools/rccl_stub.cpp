// rccl_stub — a file-backed stand-in for the five RCCL entry points libvpt_hip.so calls, for ONE purpose: running the library's
// real vpt_comm_init / vpt_comm_gather_shards / row re-interleave with world size 2 on a box that has one GPU (RCCL itself refuses
// two ranks on one device, so on such a box the N > 1 path of csrc/vpt_api.hip would otherwise never execute).  Test utility only:
// built on demand by tests/test_gpu_comm.py and put in front of librccl with LD_PRELOAD; nothing in the product links it.
//
// Rendezvous: a directory named by VPT_RCCL_STUB_DIR.  ncclGather = every rank copies its send buffer to the host and publishes
// it as <dir>/g<seq>_r<rank>.bin (write + rename); the root waits for all of them and copies each into recv + rank * count on the
// device.  That is the collective's contract (root receives rank r's count elements at offset r * count, in stream order).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct StubComm { int rank, world, device; unsigned seq; std::string dir; };

static std::string stub_dir() { const char* d = getenv("VPT_RCCL_STUB_DIR"); return d ? d : "/tmp"; }
static bool wait_for(const std::string& path) {
    for (int i = 0; i < 60000; i++) { if (access(path.c_str(), R_OK) == 0) return true; usleep(1000); }
    return false;
}

extern "C" {
ncclResult_t ncclGetVersion(int* v) { *v = NCCL_VERSION_CODE; return ncclSuccess; }
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "rccl_stub error"; }
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "stub-%d", (int)getpid());
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    StubComm* c = new StubComm();
    c->rank = rank; c->world = nranks; c->seq = 0; c->dir = stub_dir() + "/" + std::string(id.internal);
    if (hipGetDevice(&c->device) != hipSuccess) { delete c; return ncclUnhandledCudaError; }
    (void)!system(("mkdir -p '" + c->dir + "'").c_str());
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}
ncclResult_t ncclCommCount(const ncclComm_t comm, int* n) { *n = reinterpret_cast<const StubComm*>(comm)->world; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* r) { *r = reinterpret_cast<const StubComm*>(comm)->rank; return ncclSuccess; }
ncclResult_t ncclCommCuDevice(const ncclComm_t comm, int* d) { *d = reinterpret_cast<const StubComm*>(comm)->device; return ncclSuccess; }
ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete reinterpret_cast<StubComm*>(comm); return ncclSuccess; }
ncclResult_t ncclGather(const void* send, void* recv, size_t count, ncclDataType_t type, int root, ncclComm_t comm, hipStream_t stream) {
    StubComm* c = reinterpret_cast<StubComm*>(comm);
    if (type != ncclFloat32) return ncclInvalidArgument;
    const size_t bytes = count * 4;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    std::vector<char> host(bytes);
    if (hipMemcpy(host.data(), send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    char name[64];
    snprintf(name, sizeof(name), "/g%u_r%d.bin", c->seq, c->rank);
    const std::string path = c->dir + name, tmp = path + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f || fwrite(host.data(), 1, bytes, f) != bytes) return ncclSystemError;
    fclose(f);
    if (rename(tmp.c_str(), path.c_str()) != 0) return ncclSystemError;
    if (c->rank == root) {
        for (int r = 0; r < c->world; r++) {
            snprintf(name, sizeof(name), "/g%u_r%d.bin", c->seq, r);
            const std::string p = c->dir + name;
            if (!wait_for(p)) return ncclSystemError;
            FILE* g = fopen(p.c_str(), "rb");
            if (!g || fread(host.data(), 1, bytes, g) != bytes) return ncclSystemError;
            fclose(g);
            if (hipMemcpy(static_cast<char*>(recv) + (size_t)r * bytes, host.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
        }
    }
    c->seq++;
    return ncclSuccess;
}
}
