// comm.cpp -- RCCL (xGMI) behind the C ABI: the communicator a non-Python host needs to run the one path with an
// exchange step, the firpfbch2 analyzer with its sub-bands sharded over the GPUs of a node.
//
// librccl.so.1 is bound at first use with dlopen, not at link time: the library (and every non-sharded object)
// loads on a machine without RCCL, and inside a PyTorch process the SONAME resolves to the copy torch already
// mapped, so the process has ONE RCCL.  A missing / failing RCCL is YAGI_ERR_DEVICE (status 7).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include "kernels.hpp"

namespace yagi {

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string why;
};

static Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) {
            const char *e = dlerror();
            r.why = e ? e : "dlopen failed";
            return;
        }
        auto sym = [&](const char *n) {
            void *p = dlsym(r.lib, n);
            if (!p && r.why.empty()) r.why = std::string("missing symbol ") + n;
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
        r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(sym("ncclCommUserRank"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return r;
}

static int rccl_ready() {
    Rccl &r = rccl();
    if (!r.lib || !r.why.empty()) return fail(YAGI_ERR_DEVICE, "RCCL unavailable: %s", r.why.c_str());
    return YAGI_OK;
}

#define YG_NCCL(expr)                                                                              \
    do {                                                                                           \
        ncclResult_t r_ = (expr);                                                                  \
        if (r_ != ncclSuccess)                                                                     \
            return ::yagi::fail(YAGI_ERR_DEVICE, "%s failed: %s", #expr, rccl().GetErrorString(r_)); \
    } while (0)

int comm_all_gather(Comm *c, const void *send, void *recv, size_t bytes_per_rank, hipStream_t st) {
    YG_TRY(rccl_ready());
    if (bytes_per_rank % 4) return fail(YAGI_ERR_CONFIG, "all-gather size must be a multiple of 4 bytes");
    YG_NCCL(rccl().AllGather(send, recv, bytes_per_rank / 4, ncclFloat32, static_cast<ncclComm_t>(c->nccl), st));
    return YAGI_OK;
}

}  // namespace yagi

using namespace yagi;

extern "C" {

int yagi_hip_comm_unique_id(unsigned char *id) try {
    if (!id) return fail(YAGI_ERR_CONFIG, "null pointer argument");
    YG_TRY(rccl_ready());
    static_assert(NCCL_UNIQUE_ID_BYTES == YAGI_HIP_COMM_ID_BYTES, "unique id size");
    ncclUniqueId u;
    YG_NCCL(rccl().GetUniqueId(&u));
    std::memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return YAGI_OK;
} catch (...) { return ::yagi::api_exception(); }

int yagi_hip_comm_create(const unsigned char *id, int rank, int nranks, yagi_hip_comm *out) try {
    if (!id || !out) return fail(YAGI_ERR_CONFIG, "null pointer argument");
    *out = nullptr;
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(YAGI_ERR_CONFIG, "bad rank %d of %d", rank, nranks);
    YG_TRY(rccl_ready());
    // released through yagi_hip_comm_destroy on every failure path below: communicator, stream and events alike
    struct Destroy { void operator()(yagi_hip_comm_s *p) const { (void)yagi_hip_comm_destroy(p); } };
    std::unique_ptr<yagi_hip_comm_s, Destroy> c(new (std::nothrow) yagi_hip_comm_s);
    if (!c) return fail(YAGI_ERR_INTERNAL, "out of memory");
    ncclUniqueId u;
    std::memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t comm = nullptr;
    YG_NCCL(rccl().CommInitRank(&comm, nranks, u, rank));          // collective: every rank of the id calls it
    c->nccl = comm;
    int cnt = 0, ur = -1;
    YG_NCCL(rccl().CommCount(comm, &cnt));
    YG_NCCL(rccl().CommUserRank(comm, &ur));
    c->rank = ur;
    c->nranks = cnt;
    YG_HIP(hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking));
    YG_HIP(hipEventCreateWithFlags(&c->done, hipEventDisableTiming));
    *out = c.release();
    return YAGI_OK;
} catch (...) { return ::yagi::api_exception(); }

int yagi_hip_comm_destroy(yagi_hip_comm c) try {
    if (!c) return YAGI_OK;
    if (c->st) (void)hipStreamSynchronize(c->st);
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    if (c->done) (void)hipEventDestroy(c->done);
    if (c->st) (void)hipStreamDestroy(c->st);
    if (c->nccl && rccl().CommDestroy) (void)rccl().CommDestroy(static_cast<ncclComm_t>(c->nccl));
    delete c;
    return YAGI_OK;
} catch (...) { return ::yagi::api_exception(); }

int yagi_hip_comm_rank(yagi_hip_comm c, int *rank, int *nranks) try {
    if (!c) return fail(YAGI_ERR_CONFIG, "null communicator");
    if (rank) *rank = c->rank;
    if (nranks) *nranks = c->nranks;
    return YAGI_OK;
} catch (...) { return ::yagi::api_exception(); }

int yagi_hip_comm_all_gather_dev(yagi_hip_comm c, const void *send_dev, void *recv_dev, size_t bytes_per_rank,
                                 yagi_stream_t s) try {
    if (!c) return fail(YAGI_ERR_CONFIG, "null communicator");
    if (bytes_per_rank == 0) return YAGI_OK;
    if (!send_dev || !recv_dev) return fail(YAGI_ERR_CONFIG, "null pointer argument");
    return comm_all_gather(c, send_dev, recv_dev, bytes_per_rank, to_stream(s));
} catch (...) { return ::yagi::api_exception(); }

}  // extern "C"
