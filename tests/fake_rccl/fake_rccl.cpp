// fake_rccl.cpp -- TEST INFRASTRUCTURE, not part of the product.
//
// A stand-in for the five RCCL entry points librrt_hip.so binds (rrt_engine.hip: rccl_load), for ranks that are PROCESSES ON ONE
// HOST and may share one GPU.  Real RCCL refuses two ranks on the same device, and the build pool has one GPU per box, so without
// this the world > 1 branches of rrt_comm_init / rrt_gather / rrt_gather_fetch (rank * slab_bytes offsets, the size-check
// all-reduce, the self-describing slab tail) would never execute anywhere.  The collectives are host-staged through one POSIX
// shared-memory segment named by the unique id: device -> segment, barrier, segment -> device.  They block the calling host
// thread (stronger than RCCL's stream-asynchronous contract, never weaker) and every wait is bounded (30 s -> ncclSystemError).
//
// Built by tests/fake_rccl/build.py (g++ against the HIP runtime); selected with rrt_comm_use_library() / RRT_RCCL_LIB.
#include <hip/hip_runtime_api.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

extern "C" {
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef struct fakeComm *ncclComm_t;
}

namespace {
constexpr size_t DATA_BYTES = size_t(1) << 30;  // per-collective staging area (sparse: tmpfs pages appear when touched)
constexpr int MAX_WORLD = 16;
struct Header {
    std::atomic<uint32_t> arrived;     // sense-reversing barrier
    std::atomic<uint32_t> generation;
    std::atomic<uint32_t> attached;    // ranks that have mapped the segment
    uint32_t world;
};
size_t dtype_size(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        default: return 8;
    }
}
}  // namespace

struct fakeComm {
    int rank, world, fd;
    char name[128];
    unsigned char *base;
    Header *hdr;
    unsigned char *data;
};

namespace {
bool barrier(fakeComm *c) {
    Header *h = c->hdr;
    const uint32_t gen = h->generation.load(std::memory_order_acquire);
    if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->world) {
        h->arrived.store(0, std::memory_order_relaxed);
        h->generation.store(gen + 1, std::memory_order_release);
        return true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (h->generation.load(std::memory_order_acquire) == gen) {
        sched_yield();
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) return false;
    }
    return true;
}
}  // namespace

extern "C" {

const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "fake_rccl: HIP call failed";
        case ncclSystemError: return "fake_rccl: shared segment / peer time-out";
        case ncclInvalidArgument: return "fake_rccl: invalid argument";
        default: return "fake_rccl: internal error";
    }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return ncclInvalidArgument;
    memset(id->internal, 0, sizeof id->internal);
    unsigned long long r = 0;
    FILE *f = fopen("/dev/urandom", "rb");
    if (f) {
        if (fread(&r, sizeof r, 1, f) != 1) r = 0;
        fclose(f);
    }
    snprintf(id->internal, sizeof id->internal, "/rrt_fake_rccl_%d_%016llx", (int)getpid(), r);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int world, ncclUniqueId id, int rank) {
    if (!out || world < 1 || world > MAX_WORLD || rank < 0 || rank >= world) return ncclInvalidArgument;
    fakeComm *c = new fakeComm();
    c->rank = rank;
    c->world = world;
    id.internal[sizeof id.internal - 1] = 0;
    snprintf(c->name, sizeof c->name, "%s", id.internal);
    const size_t total = 4096 + DATA_BYTES;
    c->fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (c->fd < 0 || ftruncate(c->fd, (off_t)total) != 0) {
        delete c;
        return ncclSystemError;
    }
    c->base = (unsigned char *)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, c->fd, 0);
    if (c->base == MAP_FAILED) {
        close(c->fd);
        delete c;
        return ncclSystemError;
    }
    c->hdr = reinterpret_cast<Header *>(c->base);  // (a fresh segment is zero-filled: the atomics start at 0)
    c->data = c->base + 4096;
    c->hdr->world = (uint32_t)world;
    c->hdr->attached.fetch_add(1);
    const auto t0 = std::chrono::steady_clock::now();
    while (c->hdr->attached.load() < (uint32_t)world) {  // like ncclCommInitRank, returns when every rank has joined
        sched_yield();
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) return ncclSystemError;
    }
    if (!barrier(c)) return ncclSystemError;
    if (rank == 0) shm_unlink(c->name);  // everybody has it mapped: the name can go
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclSuccess;
    munmap(c->base, 4096 + DATA_BYTES);
    close(c->fd);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclComm_t c, hipStream_t stream) {
    if (!c || !send || !recv) return ncclInvalidArgument;
    const size_t bytes = count * dtype_size(dt);
    if (bytes * (size_t)c->world > DATA_BYTES) return ncclInvalidArgument;
    if (hipMemcpyAsync(c->data + (size_t)c->rank * bytes, send, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    if (hipMemcpyAsync(recv, c->data, bytes * (size_t)c->world, hipMemcpyHostToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;  // nobody overwrites the segment before everybody has read it
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t c, hipStream_t stream) {
    if (!c || !send || !recv || dt != ncclFloat64 || count > 4096) return ncclInvalidArgument;  // (librrt_hip.so reduces doubles only)
    const size_t bytes = count * sizeof(double);
    if (hipMemcpyAsync(c->data + (size_t)c->rank * bytes, send, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    double acc[4096];
    const double *all = reinterpret_cast<const double *>(c->data);
    for (size_t k = 0; k < count; ++k) {
        double a = all[k];
        for (int r = 1; r < c->world; ++r) {
            const double v = all[(size_t)r * count + k];
            a = op == ncclSum ? a + v : op == ncclProd ? a * v : op == ncclMax ? (v > a ? v : a) : (v < a ? v : a);
        }
        acc[k] = a;
    }
    if (hipMemcpyAsync(recv, acc, bytes, hipMemcpyHostToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    return ncclSuccess;
}

}  // extern "C"
