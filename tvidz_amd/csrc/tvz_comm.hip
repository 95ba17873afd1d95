// tvz_comm.hip — the multi-GPU exchange step of the corpus match behind the C ABI.
//
// One process per GPU; the corpus rows are sharded over the ranks, queries are replicated.  Per
// query batch every rank runs its local sweep + per-shard top-k (tvz_match.hip), ONE ncclAllGather
// moves the [Q][k+1][3] int32 blocks (k best hits + a totals row: <= 204 B per query and rank at
// k = 16, latency-bound on xGMI), and every rank runs the identical merge.  The reference has no
// counterpart: it scans one Postgres table in one Python process (inspector/db.py:83-91).
//
// RCCL is bound at run time with dlopen/dlsym - the copy the process has already mapped (e.g.
// PyTorch's) if there is one - so libtvz.so has no link-time dependency on it and a host language
// other than Python gets the whole multi-GPU path from this library alone.
#include <dlfcn.h>

#include <mutex>

#include "tvz_common.h"

namespace {

// ABI of the few RCCL entry points used (rccl.h: ncclUniqueId is 128 opaque bytes passed by value,
// ncclInt32 = 2, ncclSuccess = 0)
struct NcclId { char internal[TVZ_UNIQUE_ID_BYTES]; };
using NcclComm = void *;
constexpr int kNcclInt32 = 2;

struct Api {
    void *lib = nullptr;
    int (*GetUniqueId)(NcclId *) = nullptr;
    int (*CommInitRank)(NcclComm *, int, NcclId, int) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, NcclComm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};

Api g_api;                 // written once under g_api_mu, read-only afterwards
std::once_flag g_api_once;

void load_api() {
    const char *names[] = {"librccl.so", "librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)                       // a copy the process already mapped
        if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    const char *paths[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : paths)
        if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    g_api.lib = h;
    g_api.GetUniqueId = reinterpret_cast<decltype(g_api.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    g_api.CommInitRank = reinterpret_cast<decltype(g_api.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    g_api.CommDestroy = reinterpret_cast<decltype(g_api.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    g_api.AllGather = reinterpret_cast<decltype(g_api.AllGather)>(dlsym(h, "ncclAllGather"));
    g_api.GetErrorString = reinterpret_cast<decltype(g_api.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    g_api.ok = g_api.GetUniqueId && g_api.CommInitRank && g_api.CommDestroy && g_api.AllGather;
}

int need_api() {
    std::call_once(g_api_once, load_api);
    if (!g_api.ok)
        return tvz::fail(TVZ_ERR_COMM, "RCCL is not available (dlopen of librccl.so failed: %s)",
                         g_api.lib ? "symbols missing" : "library not found");
    return TVZ_OK;
}

int nccl_fail(const char *what, int rc) {
    return tvz::fail(TVZ_ERR_COMM, "%s failed: %s (%d)", what,
                     g_api.GetErrorString ? g_api.GetErrorString(rc) : "?", rc);
}

}  // namespace

struct tvz_comm {
    NcclComm comm = nullptr;
    int32_t n_ranks = 0, rank = 0, device = 0;
    // The collectives of ONE communicator are issued from whichever stream the caller passes (two
    // alternating streams in sharded.RcclShardedMatcher).  Their order is made explicit: a collective
    // on a stream other than the previous one's waits for an event recorded behind the previous one,
    // so the communicator never has two collectives in flight whatever RCCL does about stream changes.
    std::mutex mu;
    hipStream_t last_stream = nullptr;
    bool any = false;
    hipEvent_t last_ev = nullptr;       // created on first use
};

static int tvz_comm_unique_id_impl(void *out_id) {
    TVZ_REQUIRE(out_id != nullptr, "out_id is NULL");
    if (int rc = need_api()) return rc;
    NcclId id;
    if (int rc = g_api.GetUniqueId(&id)) return nccl_fail("ncclGetUniqueId", rc);
    memcpy(out_id, id.internal, TVZ_UNIQUE_ID_BYTES);
    return TVZ_OK;
}

static int tvz_comm_init_impl(tvz_comm **out, const void *unique_id, int32_t n_ranks, int32_t rank,
                              int32_t device) {
    TVZ_REQUIRE(out != nullptr && unique_id != nullptr, "NULL argument");
    TVZ_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks, "rank %d of %d", rank, n_ranks);
    if (int rc = need_api()) return rc;
    int n = 0;
    TVZ_HIP(hipGetDeviceCount(&n));
    TVZ_REQUIRE(device >= 0 && device < n, "device %d out of range (%d visible)", device, n);
    int prev = -1;
    (void)hipGetDevice(&prev);
    TVZ_HIP(hipSetDevice(device));                  // RCCL binds the communicator to the current device
    NcclId id;
    memcpy(id.internal, unique_id, TVZ_UNIQUE_ID_BYTES);
    tvz_comm *c = new tvz_comm();
    c->n_ranks = n_ranks;
    c->rank = rank;
    c->device = device;
    const int rc = g_api.CommInitRank(&c->comm, n_ranks, id, rank);
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (rc) {
        delete c;
        return nccl_fail("ncclCommInitRank", rc);
    }
    *out = c;
    return TVZ_OK;
}

static int tvz_comm_destroy_impl(tvz_comm *comm) {
    if (!comm) return TVZ_OK;
    if (comm->comm && g_api.ok) (void)g_api.CommDestroy(comm->comm);
    if (comm->last_ev) (void)hipEventDestroy(comm->last_ev);
    delete comm;
    return TVZ_OK;
}

static int tvz_match_sharded_impl(tvz_corpus *c, tvz_comm *comm, const double *d_queries,
                                  const int64_t *d_q_offsets, int32_t Q, int32_t max_query_len,
                                  int32_t min_match, const int32_t *d_exclude_ids, int32_t cap,
                                  int32_t k, int32_t *d_topk, int32_t *d_totals, void *d_workspace,
                                  size_t workspace_bytes, int32_t algo, void *hip_stream) {
    TVZ_REQUIRE(comm != nullptr && comm->comm != nullptr, "communicator is NULL");
    TVZ_REQUIRE(Q == 0 || d_topk != nullptr, "d_topk is NULL");
    if (Q == 0) return TVZ_OK;
    int32_t *gathered = nullptr;
    // local sweep + per-shard top-k into the workspace's own block
    if (int rc = tvz_match_topk_local(c, d_queries, d_q_offsets, Q, max_query_len, min_match,
                                      d_exclude_ids, cap, k, nullptr, d_workspace, workspace_bytes,
                                      comm->n_ranks, algo, hip_stream, &gathered))
        return rc;
    const int32_t *local = tvz_ws_local_block(d_workspace, Q, max_query_len, cap, k, comm->n_ranks);
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    const size_t count = (size_t)Q * (size_t)(k + 1) * 3;
    // RCCL enqueues on the communicator's device: make it current for the call (a host that drives
    // several GPUs from one thread may have another one selected)
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (prev != comm->device) TVZ_HIP(hipSetDevice(comm->device));
    int grc = 0;
    hipError_t herr = hipSuccess;
    {
        std::lock_guard<std::mutex> lk(comm->mu);
        if (!comm->last_ev) herr = hipEventCreateWithFlags(&comm->last_ev, hipEventDisableTiming);
        if (herr == hipSuccess && comm->any && comm->last_stream != st) herr = hipStreamWaitEvent(st, comm->last_ev, 0);
        if (herr == hipSuccess) {
            grc = g_api.AllGather(local, gathered, count, kNcclInt32, comm->comm, st);
            if (!grc) herr = hipEventRecord(comm->last_ev, st);
            comm->last_stream = st;
            comm->any = true;
        }
    }
    if (prev >= 0 && prev != comm->device) (void)hipSetDevice(prev);
    if (herr != hipSuccess) return tvz::fail(TVZ_ERR_HIP, "ordering the collective failed: %s", hipGetErrorString(herr));
    if (grc) return nccl_fail("ncclAllGather", grc);
    return tvz_topk_merge_ws(gathered, comm->n_ranks, Q, k, d_topk, d_totals, d_workspace, max_query_len, cap,
                             hip_stream);
}

TVZ_EXPORT int tvz_comm_unique_id(void *out_id) { TVZ_GUARDED(tvz_comm_unique_id_impl(out_id)); }

TVZ_EXPORT int tvz_comm_init(tvz_comm **out, const void *unique_id, int32_t n_ranks, int32_t rank,
                             int32_t device) {
    TVZ_GUARDED(tvz_comm_init_impl(out, unique_id, n_ranks, rank, device));
}

TVZ_EXPORT int tvz_comm_info(tvz_comm *comm, int32_t *n_ranks, int32_t *rank) {
    TVZ_REQUIRE(comm != nullptr, "communicator is NULL");
    if (n_ranks) *n_ranks = comm->n_ranks;
    if (rank) *rank = comm->rank;
    return TVZ_OK;
}

TVZ_EXPORT int tvz_comm_destroy(tvz_comm *comm) { TVZ_GUARDED(tvz_comm_destroy_impl(comm)); }

TVZ_EXPORT int tvz_match_sharded(tvz_corpus *c, tvz_comm *comm, const double *d_queries,
                                 const int64_t *d_q_offsets, int32_t Q, int32_t max_query_len,
                                 int32_t min_match, const int32_t *d_exclude_ids, int32_t cap,
                                 int32_t k, int32_t *d_topk, int32_t *d_totals, void *d_workspace,
                                 size_t workspace_bytes, int32_t algo, void *hip_stream) {
    TVZ_GUARDED(tvz_match_sharded_impl(c, comm, d_queries, d_q_offsets, Q, max_query_len, min_match, d_exclude_ids, cap, k, d_topk, d_totals, d_workspace, workspace_bytes, algo, hip_stream));
}
