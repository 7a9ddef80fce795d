// The one exchange step of the sharded Monte-Carlo run: the sum of the ranks' histograms, over RCCL.
//
// Keys of the histograms: vec_to_int(syndrome) (css_code.py:729) or the syndrome's weight; X errors against parity_check_c2, Z errors
// against parity_check_c1 (css_code.py:457-470).  Every sample is independent, so the shards never talk on the data path; this
// all-reduce of at most (r_1 + 1) + (r_2 + 1) uint64 bins (32 KiB at n = 4096) is the whole of the communication.  It is latency
// bound: the xGMI links' bandwidth does not matter at this size, one ncclAllReduce on the context's stream does.
//
// librccl is loaded on first use (dlopen), so that the library itself does not depend on it: a process that never creates a
// communicator never loads it.  The copy that is loaded is the one that sits beside the HIP runtime this library is bound to (see
// rccl_load: a process that imported torch first runs on torch's pair, any other on ROCm's).  Two ways in:
//   gf2_comm_create      one process per GPU (the bench's ranks): rank 0 makes an id (gf2_comm_unique_id), every rank receives it
//                        out of band (the launcher's rendezvous store) and joins with its context;
//   gf2_comm_create_all  one process, G contexts on G devices (ncclCommInitAll): no id to pass around.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include <rccl/rccl.h>

#include "gf2_internal.h"

namespace {

struct RcclApi {
    void* handle;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*);
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*GroupStart)();
    ncclResult_t (*GroupEnd)();
    const char* (*GetErrorString)(ncclResult_t);
    ncclResult_t (*GetVersion)(int*);
};

RcclApi g_rccl = {};
std::mutex g_rccl_mutex;            // entry points may be called from several host threads (one context each): one of them loads

int rccl_load() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return GF2_OK;
    // The RCCL that belongs to the HIP runtime THIS library runs on.  A process may hold two HIP runtimes -- ROCm's and the copy a
    // PyTorch wheel ships next to its own librccl -- and which of them libgf2hip is bound to depends on what was loaded first; a
    // communicator from the other runtime's RCCL cannot use this library's streams ("unhandled cuda error" from ncclCommInitAll).
    // So: look beside the libamdhip64 that hipGetDeviceCount resolves to, by full path, before the plain names.
    void* h = nullptr;
    Dl_info info;
    if (dladdr((void*)&hipGetDeviceCount, &info) && info.dli_fname) {
        const char* slash = strrchr(info.dli_fname, '/');
        if (slash) {
            const size_t dir_len = (size_t)(slash - info.dli_fname) + 1;
            for (const char* leaf : {"librccl.so.1", "librccl.so"}) {
                char path[4096];
                if (dir_len + strlen(leaf) + 1 > sizeof(path)) continue;
                memcpy(path, info.dli_fname, dir_len);
                strcpy(path + dir_len, leaf);
                h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
                if (h) break;
            }
        }
    }
    if (!h) {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* name : names) {
            h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
    }
    if (!h) GF2_FAIL(GF2_E_RCCL, "librccl.so.1 cannot be loaded: %s", dlerror());
    RcclApi api = {};
    api.handle = h;
#define GF2_SYM(field, name)                                                                       \
    *(void**)(&api.field) = dlsym(h, name);                                                        \
    if (!api.field) {                                                                              \
        dlclose(h);                                                                                \
        GF2_FAIL(GF2_E_RCCL, "librccl has no symbol %s", name);                                    \
    }
    GF2_SYM(GetUniqueId, "ncclGetUniqueId")
    GF2_SYM(CommInitRank, "ncclCommInitRank")
    GF2_SYM(CommInitAll, "ncclCommInitAll")
    GF2_SYM(CommDestroy, "ncclCommDestroy")
    GF2_SYM(AllReduce, "ncclAllReduce")
    GF2_SYM(GroupStart, "ncclGroupStart")
    GF2_SYM(GroupEnd, "ncclGroupEnd")
    GF2_SYM(GetErrorString, "ncclGetErrorString")
    GF2_SYM(GetVersion, "ncclGetVersion")
#undef GF2_SYM
    g_rccl = api;
    return GF2_OK;
}

#define GF2_RCCL(expr)                                                                                       \
    do {                                                                                                     \
        ncclResult_t gf2_nr_ = (expr);                                                                       \
        if (gf2_nr_ != ncclSuccess) {                                                                        \
            gf2_set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(gf2_nr_), __FILE__, __LINE__); \
            return GF2_E_RCCL;                                                                               \
        }                                                                                                    \
    } while (0)

}   // namespace

#define GF2_COMM_MAX_LOCAL 16
struct gf2_comm {
    int nlocal;                             // communicators (= contexts) this process holds
    int nranks;
    gf2_ctx* ctx[GF2_COMM_MAX_LOCAL];
    ncclComm_t comm[GF2_COMM_MAX_LOCAL];
};

extern "C" {

int gf2_comm_unique_id(void* id_out, size_t bytes) {
    if (!id_out || bytes < GF2_COMM_ID_BYTES) GF2_FAIL(GF2_E_ARG, "gf2_comm_unique_id: the id needs %d bytes", GF2_COMM_ID_BYTES);
    static_assert(sizeof(ncclUniqueId) == GF2_COMM_ID_BYTES, "GF2_COMM_ID_BYTES is RCCL's NCCL_UNIQUE_ID_BYTES");
    GF2_TRY(rccl_load());
    ncclUniqueId id;
    GF2_RCCL(g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return GF2_OK;
}

int gf2_comm_create(gf2_ctx* ctx, const void* id, int nranks, int rank, gf2_comm** comm_out) {
    if (!ctx || !id || !comm_out || nranks < 1 || rank < 0 || rank >= nranks) GF2_FAIL(GF2_E_ARG, "gf2_comm_create: bad argument");
    *comm_out = nullptr;
    GF2_TRY(rccl_load());
    GF2_TRY(gf2_ctx_activate(ctx));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    gf2_comm* c = (gf2_comm*)calloc(1, sizeof(gf2_comm));
    if (!c) GF2_FAIL(GF2_E_NOMEM, "gf2_comm_create: out of host memory");
    const ncclResult_t nr = g_rccl.CommInitRank(&c->comm[0], nranks, uid, rank);
    if (nr != ncclSuccess) {
        free(c);
        GF2_FAIL(GF2_E_RCCL, "ncclCommInitRank(rank %d of %d) failed: %s", rank, nranks, g_rccl.GetErrorString(nr));
    }
    c->nlocal = 1;
    c->nranks = nranks;
    c->ctx[0] = ctx;
    *comm_out = c;
    return GF2_OK;
}

int gf2_comm_create_all(gf2_ctx* const* ctxs, int count, gf2_comm** comm_out) {
    if (!ctxs || !comm_out || count < 1 || count > GF2_COMM_MAX_LOCAL) GF2_FAIL(GF2_E_ARG, "gf2_comm_create_all: bad argument");
    *comm_out = nullptr;
    int devs[GF2_COMM_MAX_LOCAL];
    for (int i = 0; i < count; ++i) {
        if (!ctxs[i]) GF2_FAIL(GF2_E_ARG, "gf2_comm_create_all: null context");
        devs[i] = ctxs[i]->device;
        for (int j = 0; j < i; ++j)
            if (devs[j] == devs[i]) GF2_FAIL(GF2_E_ARG, "gf2_comm_create_all: contexts %d and %d share device %d", j, i, devs[i]);
    }
    GF2_TRY(rccl_load());
    gf2_comm* c = (gf2_comm*)calloc(1, sizeof(gf2_comm));
    if (!c) GF2_FAIL(GF2_E_NOMEM, "gf2_comm_create_all: out of host memory");
    const ncclResult_t nr = g_rccl.CommInitAll(c->comm, count, devs);
    if (nr != ncclSuccess) {
        free(c);
        GF2_FAIL(GF2_E_RCCL, "ncclCommInitAll(%d devices) failed: %s", count, g_rccl.GetErrorString(nr));
    }
    c->nlocal = count;
    c->nranks = count;
    for (int i = 0; i < count; ++i) c->ctx[i] = ctxs[i];
    *comm_out = c;
    return GF2_OK;
}

int gf2_comm_size(const gf2_comm* comm, int* nranks_out, int* nlocal_out) {
    if (!comm) GF2_FAIL(GF2_E_ARG, "gf2_comm_size: null communicator");
    if (nranks_out) *nranks_out = comm->nranks;
    if (nlocal_out) *nlocal_out = comm->nlocal;
    return GF2_OK;
}

int gf2_hist_allreduce(gf2_comm* comm, uint64_t* const* hist_dev, int64_t nbins) {
    if (!comm || !hist_dev || nbins < 1) GF2_FAIL(GF2_E_ARG, "gf2_hist_allreduce: bad argument");
    for (int i = 0; i < comm->nlocal; ++i)
        if (!hist_dev[i]) GF2_FAIL(GF2_E_ARG, "gf2_hist_allreduce: null histogram %d", i);
    // one process with several communicators: the calls must sit in one group, or the first would wait for the others
    const bool grouped = comm->nlocal > 1;
    if (grouped) GF2_RCCL(g_rccl.GroupStart());
    for (int i = 0; i < comm->nlocal; ++i) {
        // a failure inside the group must not leave it open (every later collective of the process would hang): close it first
        if (gf2_ctx_activate(comm->ctx[i]) != GF2_OK) {
            if (grouped) (void)g_rccl.GroupEnd();
            return GF2_E_HIP;
        }
        const ncclResult_t nr =
            g_rccl.AllReduce(hist_dev[i], hist_dev[i], (size_t)nbins, ncclUint64, ncclSum, comm->comm[i], comm->ctx[i]->stream);
        if (nr != ncclSuccess) {
            if (grouped) (void)g_rccl.GroupEnd();
            GF2_FAIL(GF2_E_RCCL, "ncclAllReduce (communicator %d of %d) failed: %s", i, comm->nlocal, g_rccl.GetErrorString(nr));
        }
    }
    if (grouped) GF2_RCCL(g_rccl.GroupEnd());
    for (int i = 0; i < comm->nlocal; ++i) {
        GF2_TRY(gf2_ctx_activate(comm->ctx[i]));
        GF2_TRY(gf2_stream_wait(comm->ctx[i]->stream));
    }
    return GF2_OK;
}

int gf2_comm_destroy(gf2_comm* comm) {
    if (!comm) return GF2_OK;
    int rc = GF2_OK;
    for (int i = 0; i < comm->nlocal; ++i) {
        (void)hipSetDevice(comm->ctx[i]->device);
        (void)hipStreamSynchronize(comm->ctx[i]->stream);          // a collective of this communicator may still be on the stream
        if (comm->comm[i] && g_rccl.CommDestroy(comm->comm[i]) != ncclSuccess) rc = GF2_E_RCCL;
    }
    free(comm);
    if (rc != GF2_OK) gf2_set_error("ncclCommDestroy failed");
    return rc;
}

int gf2_rccl_version(int* version_out) {
    if (!version_out) GF2_FAIL(GF2_E_ARG, "gf2_rccl_version: null output");
    GF2_TRY(rccl_load());
    GF2_RCCL(g_rccl.GetVersion(version_out));
    return GF2_OK;
}

}   // extern "C"
