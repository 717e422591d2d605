// comm.hip -- the data-parallel gradient exchange of include/cnn_amd.h: RCCL (ncclAllReduce, fp32 sum, in place) over xGMI.
//
// The reference has no exchange step at all (single process); the coupling this reproduces is the
// batch mean inside the weight / bias gradients (cpu/src/conv2d.cpp:148,157, cpu/src/linear.cpp:62,70): every replica's
// kernels divide by their LOCAL batch, the flat gradient arena is summed over the replicas here, and cnn_sgd_update folds
// the remaining 1/G.
//
// librccl is bound at run time (dlopen by soname), not at link time: a process that already carries an RCCL -- PyTorch-ROCm
// bundles one under the same soname -- must keep exactly ONE copy, and single-GPU users of libcnn_amd.so need none.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "common.h"

using namespace cnn_amd;

namespace {
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommSplit)(ncclComm_t, int, int, ncclComm_t*, ncclConfig_t*) = nullptr;  // (optional: RCCL >= 2.18)
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    char why[256] = {0};
};

Rccl g_rccl;

Rccl* rccl() {
    static std::once_flag once;
    std::call_once(once, [] {
        Rccl& r = g_rccl;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            snprintf(r.why, sizeof(r.why), "librccl.so.1 not found (%s)", dlerror());
            return;
        }
        bool ok = true;
        auto sym = [&](const char* name) {
            void* p = dlsym(r.handle, name);
            if (!p) {
                ok = false;
                snprintf(r.why, sizeof(r.why), "librccl: symbol %s missing", name);
            }
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.CommCount = (decltype(r.CommCount))sym("ncclCommCount");
        r.CommUserRank = (decltype(r.CommUserRank))sym("ncclCommUserRank");
        r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
        r.Broadcast = (decltype(r.Broadcast))sym("ncclBroadcast");
        r.CommSplit = (decltype(r.CommSplit))dlsym(r.handle, "ncclCommSplit");  // may be absent: cnn_comm_split then reports it
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.GetVersion = (decltype(r.GetVersion))sym("ncclGetVersion");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        if (!ok) {
            dlclose(r.handle);
            r.handle = nullptr;
        }
    });
    return g_rccl.handle ? &g_rccl : nullptr;
}

const char* rccl_why() { return g_rccl.why[0] ? g_rccl.why : "librccl.so.1 could not be loaded"; }

#define CNN_RCCL_BIND(R)                                                                                  \
    Rccl* R = rccl();                                                                                     \
    if (!R) return fail(CNN_AMD_E_COMM, "RCCL unavailable: %s", rccl_why())

#define CNN_RCCL_CHECK(R, expr)                                                                           \
    do {                                                                                                  \
        ncclResult_t r__ = (expr);                                                                        \
        if (r__ != ncclSuccess) return fail(CNN_AMD_E_COMM + (int)r__, "%s failed: %s", #expr, R->GetErrorString(r__)); \
    } while (0)

static_assert(CNN_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "cnn_amd.h: CNN_COMM_ID_BYTES must match ncclUniqueId");
}  // namespace

extern "C" {

int cnn_comm_available(void) { return rccl() != nullptr; }

int cnn_comm_version(void) {
    Rccl* R = rccl();
    int v = 0;
    if (!R || R->GetVersion(&v) != ncclSuccess) return 0;
    return v;
}

int cnn_comm_unique_id(void* id_out) {
    CNN_REQUIRE(id_out != nullptr, "cnn_comm_unique_id: null pointer");
    CNN_RCCL_BIND(R);
    ncclUniqueId id;
    CNN_RCCL_CHECK(R, R->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return CNN_AMD_OK;
}

int cnn_comm_init_rank(void** comm, int world, int rank, const void* id_bytes) {
    CNN_REQUIRE(comm && id_bytes && world >= 1 && rank >= 0 && rank < world, "cnn_comm_init_rank: bad arguments (world %d rank %d)", world, rank);
    CNN_RCCL_BIND(R);
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof(id));
    ncclComm_t c = nullptr;
    CNN_RCCL_CHECK(R, R->CommInitRank(&c, world, id, rank));
    *comm = c;
    return CNN_AMD_OK;
}

int cnn_comm_init_all(void** comms, int ndev, const int* devices) {
    CNN_REQUIRE(comms && ndev >= 1, "cnn_comm_init_all: bad arguments");
    CNN_RCCL_BIND(R);
    static_assert(sizeof(ncclComm_t) == sizeof(void*), "communicator handles are pointers");
    CNN_RCCL_CHECK(R, R->CommInitAll(reinterpret_cast<ncclComm_t*>(comms), ndev, devices));
    return CNN_AMD_OK;
}

int cnn_comm_destroy(void* comm) {
    if (!comm) return CNN_AMD_OK;
    CNN_RCCL_BIND(R);
    CNN_RCCL_CHECK(R, R->CommDestroy(static_cast<ncclComm_t>(comm)));
    return CNN_AMD_OK;
}

int cnn_comm_info(void* comm, int* world, int* rank) {
    CNN_REQUIRE(comm != nullptr, "cnn_comm_info: null communicator");
    CNN_RCCL_BIND(R);
    if (world) CNN_RCCL_CHECK(R, R->CommCount(static_cast<ncclComm_t>(comm), world));
    if (rank) CNN_RCCL_CHECK(R, R->CommUserRank(static_cast<ncclComm_t>(comm), rank));
    return CNN_AMD_OK;
}

int cnn_comm_group_start(void) {
    CNN_RCCL_BIND(R);
    CNN_RCCL_CHECK(R, R->GroupStart());
    return CNN_AMD_OK;
}

int cnn_comm_group_end(void) {
    CNN_RCCL_BIND(R);
    CNN_RCCL_CHECK(R, R->GroupEnd());
    return CNN_AMD_OK;
}

// a second communicator over the same ranks (ncclCommSplit: every rank of `comm` calls it with the same color; key orders the ranks):
// collectives on different communicators are independent queues -- the small sync-BN reductions of BatchNorm2D (batchnorm2d.cpp:46-61,
// 129-147 couple the samples of the WHOLE batch) do not serialise behind the bucketed gradient exchange that uses `comm`
int cnn_comm_split(void* comm, int color, int key, void** new_comm) {
    CNN_REQUIRE(comm != nullptr && new_comm != nullptr, "cnn_comm_split: null pointer");
    CNN_RCCL_BIND(R);
    if (R->CommSplit == nullptr) return fail(CNN_AMD_E_COMM, "cnn_comm_split: this librccl has no ncclCommSplit");
    ncclComm_t c = nullptr;
    CNN_RCCL_CHECK(R, R->CommSplit(static_cast<ncclComm_t>(comm), color, key, &c, nullptr));
    *new_comm = c;
    return CNN_AMD_OK;
}

// `bytes` bytes at `buf` (device) of rank `root` to every rank, in place, enqueued on `stream` (e.g. rank 0's measured kernel choices:
// cnn_conv2d_tune_export / _import)
int cnn_comm_broadcast(void* comm, void* buf, size_t bytes, int root, void* stream) {
    CNN_REQUIRE(comm != nullptr && buf != nullptr && root >= 0, "cnn_comm_broadcast: bad arguments");
    if (bytes == 0) return CNN_AMD_OK;
    CNN_RCCL_BIND(R);
    CNN_RCCL_CHECK(R, R->Broadcast(buf, buf, bytes, ncclChar, root, static_cast<ncclComm_t>(comm), as_stream(stream)));
    publish_mark_stale(as_stream(stream));
    return CNN_AMD_OK;
}

// C1 of SURVEY.md section 2.1: ONE in-place fp32 sum over the flat gradient arena (or over one bucket of it)
int cnn_allreduce_grads(void* comm, float* grads, size_t n, void* stream) {
    CNN_REQUIRE(comm != nullptr && grads != nullptr, "cnn_allreduce_grads: null pointer");
    if (n == 0) return CNN_AMD_OK;
    // (measurement switch DP_SKIP_EXCHANGE=1, set on EVERY rank: the step without its all-reduces -- bench.py times K such steps behind its
    // timed regions and reports the difference as the step's exchange cost; the replicas diverge, nothing is compared afterwards)
    if (CNN_MEASURE_INT("DP_SKIP_EXCHANGE", 0) != 0) return CNN_AMD_OK;
    CNN_RCCL_BIND(R);
    CNN_RCCL_CHECK(R, R->AllReduce(grads, grads, n, ncclFloat32, ncclSum, static_cast<ncclComm_t>(comm), as_stream(stream)));
    publish_mark_stale(as_stream(stream));
    return CNN_AMD_OK;
}

}  // extern "C"
