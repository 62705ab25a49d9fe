// film_reduce.cpp — the one exchange step of the multi-GPU path: every rank renders its own 16x16
// tiles into a full-size device film whose other pixels are zero (Film::merge_film_tile, src/core/film.rs:93-123,
// applied per rank), then the films are summed onto the root rank with ONE RCCL reduce over xGMI.
// One process per GPU; the 128-byte communicator id travels out of band (the host renderer's own launcher).
// RCCL is loaded on first use (dlopen), so single-GPU callers never need it.
#include <dlfcn.h>
#include "abi_guard.h"
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <string>

#include "../../include/pbrt_hip.h"
#include "scene.h"

namespace {
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};
Rccl g_rccl;
std::mutex g_rccl_mutex;
// Last error of the communicator entry points. Written from any thread under g_error_mutex; pbrt_hip_comm_last_error
// hands out a copy owned by the calling thread, so a concurrent writer cannot pull the buffer from under a reader.
std::mutex g_error_mutex;
std::string g_comm_error_text;
void set_comm_error(const std::string& text) {
    std::lock_guard<std::mutex> lock(g_error_mutex);
    g_comm_error_text = text;
}

template <class F>
bool bind(void* h, const char* name, F* fn) {
    *fn = reinterpret_cast<F>(dlsym(h, name));
    return *fn != nullptr;
}
// nullptr when RCCL cannot be loaded (the reason is in g_rccl.error)
Rccl* rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return &g_rccl;
    if (!g_rccl.error.empty()) return nullptr;
    // The RCCL that belongs to the HIP runtime this process is running on: a process may hold two ROCm stacks (a
    // Python host with PyTorch's bundled one besides /opt/rocm's), and RCCL on top of the other stack's runtime
    // fails in ncclCommInitRank. Look next to the loaded libamdhip64 first.
    void* h = nullptr;
    // a copy the process has already loaded (PyTorch's, the host renderer's) comes first: two RCCLs in one process
    // would each set up their own transport state
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
        if (h) break;
    }
    Dl_info info;
    if (!h && dladdr(reinterpret_cast<void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
        std::string dir(info.dli_fname);
        size_t slash = dir.rfind('/');
        if (slash != std::string::npos) {
            dir.resize(slash + 1);
            for (const char* name : {"librccl.so.1", "librccl.so"}) {
                h = dlopen((dir + name).c_str(), RTLD_NOW | RTLD_LOCAL);
                if (h) break;
            }
        }
    }
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        if (h) break;
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    }
    if (!h) {
        g_rccl.error = std::string("RCCL could not be loaded: ") + dlerror();
        return nullptr;
    }
    bool ok = bind(h, "ncclGetUniqueId", &g_rccl.GetUniqueId) && bind(h, "ncclCommInitRank", &g_rccl.CommInitRank) &&
              bind(h, "ncclReduce", &g_rccl.Reduce) && bind(h, "ncclAllReduce", &g_rccl.AllReduce) &&
              bind(h, "ncclCommDestroy", &g_rccl.CommDestroy) && bind(h, "ncclGetErrorString", &g_rccl.GetErrorString);
    if (!ok) {
        g_rccl.error = "RCCL is missing an entry point (ncclGetUniqueId / ncclCommInitRank / ncclReduce / ncclAllReduce)";
        dlclose(h);
        return nullptr;
    }
    g_rccl.handle = h;
    return &g_rccl;
}
}  // namespace

struct PbrtHipComm {
    PbrtHipContext* ctx = nullptr;
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0;
};

static_assert(sizeof(ncclUniqueId) == PBRT_HIP_COMM_ID_BYTES, "communicator id size");

extern "C" const char* pbrt_hip_comm_last_error(void) {
    static thread_local std::string mine;
    std::lock_guard<std::mutex> lock(g_error_mutex);
    mine = g_comm_error_text;
    return mine.c_str();
}

extern "C" int pbrt_hip_comm_unique_id(uint8_t id[PBRT_HIP_COMM_ID_BYTES]) try {
    if (!id) return PBRT_HIP_ERR_INVALID;
    Rccl* r = rccl();
    if (!r) {
        set_comm_error(g_rccl.error);
        return PBRT_HIP_ERR_DEVICE;
    }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) {
        set_comm_error("no HIP device visible: RCCL communicators need one GPU per process");
        return PBRT_HIP_ERR_NO_DEVICE;
    }
    ncclUniqueId uid;
    ncclResult_t e = r->GetUniqueId(&uid);
    if (e != ncclSuccess) {
        set_comm_error(std::string("ncclGetUniqueId: ") + r->GetErrorString(e));
        return PBRT_HIP_ERR_DEVICE;
    }
    std::memcpy(id, &uid, sizeof(uid));
    return PBRT_HIP_OK;
}
PB_ABI_CATCH

extern "C" int pbrt_hip_comm_create(PbrtHipContext* ctx, int32_t world, int32_t rank, const uint8_t id[PBRT_HIP_COMM_ID_BYTES],
                                    PbrtHipComm** out) try {
    if (out) *out = nullptr;
    if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) {
        set_comm_error("pbrt_hip_comm_create: bad argument");
        return PBRT_HIP_ERR_INVALID;
    }
    PB_ENTER(ctx);
    Rccl* r = rccl();
    if (!r) {
        set_comm_error(ctx->last_error = g_rccl.error);
        return PBRT_HIP_ERR_DEVICE;
    }
    if (hipSetDevice(ctx->device) != hipSuccess) {
        set_comm_error(ctx->last_error = "hipSetDevice failed");
        return PBRT_HIP_ERR_DEVICE;
    }
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    ncclComm_t comm = nullptr;
    ncclResult_t e = r->CommInitRank(&comm, world, uid, rank);
    if (e != ncclSuccess) {
        set_comm_error(ctx->last_error = std::string("ncclCommInitRank: ") + r->GetErrorString(e));
        return PBRT_HIP_ERR_DEVICE;
    }
    PbrtHipComm* c = new PbrtHipComm;
    c->ctx = ctx;
    c->comm = comm;
    c->world = world;
    c->rank = rank;
    *out = c;
    return PBRT_HIP_OK;
}
PB_ABI_CATCH

extern "C" void pbrt_hip_comm_destroy(PbrtHipComm* comm) {
    if (!comm) return;
    Rccl* r = rccl();
    {
        PB_LOCK(comm->ctx);  // not while a reduce of this context is in flight, and on the context's device
        (void)hipSetDevice(comm->ctx->device);
        // a lost context (scene.h): the abandoned stream may never drain and ncclCommDestroy waits for it — the communicator stays
        if (r && comm->comm && !comm->ctx->lost) (void)r->CommDestroy(comm->comm);
    }
    delete comm;
}

// Sum of the ranks' films, in place, on the context's stream; root >= 0: only that rank's buffer holds the sum
// afterwards (ncclReduce), root < 0: every rank's does (ncclAllReduce). Returns after the stream has drained.
extern "C" int pbrt_hip_film_reduce(PbrtHipComm* comm, float* d_film_xyzw, int64_t n_pixels, int32_t root) try {
    if (!comm || !d_film_xyzw || n_pixels < 0 || root >= comm->world) {
        set_comm_error("pbrt_hip_film_reduce: bad argument");
        return PBRT_HIP_ERR_INVALID;
    }
    Rccl* r = rccl();
    PbrtHipContext* ctx = comm->ctx;
    PB_ENTER(ctx);
    if (!r) return PBRT_HIP_ERR_DEVICE;
    if (hipSetDevice(ctx->device) != hipSuccess) return PBRT_HIP_ERR_DEVICE;
    if (n_pixels == 0) return PBRT_HIP_OK;
    const size_t count = (size_t)n_pixels * 4;
    ncclResult_t e = root >= 0 ? r->Reduce(d_film_xyzw, d_film_xyzw, count, ncclFloat32, ncclSum, root, comm->comm, ctx->stream)
                               : r->AllReduce(d_film_xyzw, d_film_xyzw, count, ncclFloat32, ncclSum, comm->comm, ctx->stream);
    if (e != ncclSuccess) {
        set_comm_error(ctx->last_error = std::string(root >= 0 ? "ncclReduce: " : "ncclAllReduce: ") + r->GetErrorString(e));
        return PBRT_HIP_ERR_DEVICE;
    }
    hipError_t he = hipStreamSynchronize(ctx->stream);
    if (he != hipSuccess) {
        set_comm_error(ctx->last_error = std::string("hipStreamSynchronize after the film reduce: ") + hipGetErrorString(he));
        return PBRT_HIP_ERR_DEVICE;
    }
    return PBRT_HIP_OK;
}
PB_ABI_CATCH
