// collective.hip -- the one exchange of the data-parallel job (SURVEY 8(b) `tsod_allgather_f32`, K17 / 8(e)): a thin wrapper
// over ncclAllGather (RCCL over xGMI) on the caller's stream, the communicator passed in.  The reference has no distributed
// code at all (grep over the whole tree), so there is no file:line to match; the record format it ships is
// the fixed-size [B_local, 300, 6] of tsod_detections_f32 (nets/rpn.py:65-69 pads to n_post, hence equal counts per rank).
//
// RCCL is bound at RUN TIME (dlopen + dlsym): libtsod.so has no link-time dependency on it, loads on a box without it, and
// every entry point here then returns TSOD_ERR_UNSUPPORTED.  When the process already holds an RCCL (torch.distributed's),
// dlopen by soname hands back that same copy.  No state besides the resolved function table.
#include "tsod_internal.h"
#include <dlfcn.h>
#include <string.h>
#include <mutex>

namespace {

// the slice of rccl.h this file needs (ABI-stable since NCCL 2.x): ncclResult_t is an int enum with ncclSuccess = 0,
// ncclUniqueId is 128 opaque bytes passed BY VALUE to ncclCommInitRank, ncclFloat32 = 7
struct UniqueId { char internal[128]; };
typedef int (*fn_get_unique_id)(UniqueId *);
typedef int (*fn_comm_init_rank)(void **, int, UniqueId, int);
typedef int (*fn_comm_destroy)(void *);
typedef int (*fn_all_gather)(const void *, void *, size_t, int, void *, hipStream_t);
constexpr int kNcclFloat32 = 7;

struct Rccl {
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_all_gather all_gather = nullptr;
    bool ok = false;
};

const Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        void *h = nullptr;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        // first: an RCCL this process has ALREADY mapped (torch.distributed's, whatever path it came from) - RTLD_NOLOAD only
        // succeeds for a library that is resident, so the process never ends up with two copies
        for (const char *name : names) {
            h = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
            if (h) break;
        }
        if (!h && dlsym(RTLD_DEFAULT, "ncclAllGather")) h = dlopen(nullptr, RTLD_NOW);   // loaded globally under another soname
        for (const char *name : names) {
            if (h) break;
            h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        }
        if (!h) return;
        r.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
        r.comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
        r.comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
        r.all_gather = (fn_all_gather)dlsym(h, "ncclAllGather");
        r.ok = r.get_unique_id && r.comm_init_rank && r.comm_destroy && r.all_gather;
    });
    return r;
}

}  // namespace

extern "C" int tsod_comm_unique_id(void *id128) {
    TSOD_REQUIRE(id128 != nullptr, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(rccl().ok, TSOD_ERR_UNSUPPORTED);
    return rccl().get_unique_id(static_cast<UniqueId *>(id128)) == 0 ? TSOD_OK : TSOD_ERR_LAUNCH;
}

extern "C" int tsod_comm_init_rank(void **comm, int32_t n_ranks, const void *id128, int32_t rank) {
    TSOD_REQUIRE(comm != nullptr && id128 != nullptr && n_ranks > 0 && rank >= 0 && rank < n_ranks, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(rccl().ok, TSOD_ERR_UNSUPPORTED);
    UniqueId id;
    memcpy(&id, id128, sizeof(id));
    *comm = nullptr;
    return rccl().comm_init_rank(comm, n_ranks, id, rank) == 0 && *comm != nullptr ? TSOD_OK : TSOD_ERR_LAUNCH;
}

extern "C" int tsod_comm_destroy(void *comm) {
    TSOD_REQUIRE(comm != nullptr, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(rccl().ok, TSOD_ERR_UNSUPPORTED);
    return rccl().comm_destroy(comm) == 0 ? TSOD_OK : TSOD_ERR_LAUNCH;
}

extern "C" int tsod_allgather_f32(void *comm, const float *send, float *recv, size_t count_per_rank, tsod_stream_t stream) {
    TSOD_REQUIRE(comm != nullptr && send != nullptr && recv != nullptr && count_per_rank > 0, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(tsod_aligned16(send) && tsod_aligned16(recv), TSOD_ERR_ALIGNMENT);
    TSOD_REQUIRE(rccl().ok, TSOD_ERR_UNSUPPORTED);
    return rccl().all_gather(send, recv, count_per_rank, kNcclFloat32, comm, tsod_stream(stream)) == 0 ? TSOD_OK : TSOD_ERR_LAUNCH;
}
