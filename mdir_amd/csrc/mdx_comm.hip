// Multi-GPU exchange of per-shard partial scores over RCCL (xGMI), behind the C ABI.
//
// The reference has no distributed code; what is sharded is `scores = np.dot(vecs.T, qvecs)`
// (mdir/components/optim/score/cirscore.py:69): database rows are independent, rank g holds rows
// [lo_g, hi_g) as its own mdx_index and computes S_g [nq, n_g] alone.  Two exchanges put the pieces
// where `np.argsort(-scores, axis=0)` (cirscore.py:70) needs them:
//   mdx_allgather_scores  every rank ends with every block ("all-gather of per-shard partial scores",
//                         BASELINE.json north_star);
//   mdx_exchange_scores   every rank ends with ITS queries' rows of every block (all-to-all: 1/G of the
//                         bytes per rank, and the sort becomes G-way parallel).
// Both deliver blocks back to back in rank order -- exactly what mdx_rank_full_segments reads in place.
//
// RCCL is bound at run time (dlopen), not at link time: libmdx.so must load on a box without it and, inside
// a PyTorch process, must use the librccl.so.1 that process has already mapped (torch ships its own copy; two
// copies would each want their own HIP runtime state).  xGMI is point-to-point (7 links per GPU), so the
// uneven forms are grouped ncclSend / ncclRecv pairs -- one transfer per link, no ring.  The exchange calls make no HIP call of
// their own (tests/fake_rccl.c stands in for RCCL on a CPU box and runs them with host buffers, several ranks as threads).
#include <dlfcn.h>

#include <mutex>

#include <rccl/rccl.h>

#include "mdx_common.h"

namespace mdx {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

// nullptr + message when RCCL cannot be found.  Resolved ONCE per process under std::call_once: ranks-as-threads is an
// advertised mode (tests/fake_rccl.c, C hosts), so two threads may make their first communicator call together -- neither
// may see a half-filled table.  The failure text is kept and re-posted to every caller's thread-local error slot.
static const Rccl *rccl()
{
    static Rccl r;
    static std::once_flag once;
    static bool ok = false;
    static char why[256] = "";
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so"};
        for (const char *n : names)                     // the copy this process already has (PyTorch's), if any
            if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        for (const char *n : names)
            if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!r.handle) {
            const char *e = dlerror();
            snprintf(why, sizeof why, "mdx_comm: librccl.so.1 not found (%s)", e ? e : "?");
            return;
        }
#define MDX_SYM(field, name)                                                    \
    *(void **)(&r.field) = dlsym(r.handle, name);                               \
    if (!r.field) { snprintf(why, sizeof why, "mdx_comm: RCCL lacks %s", name); return; }
        MDX_SYM(GetUniqueId, "ncclGetUniqueId")
        MDX_SYM(CommInitRank, "ncclCommInitRank")
        MDX_SYM(CommDestroy, "ncclCommDestroy")
        MDX_SYM(AllGather, "ncclAllGather")
        MDX_SYM(Send, "ncclSend")
        MDX_SYM(Recv, "ncclRecv")
        MDX_SYM(GroupStart, "ncclGroupStart")
        MDX_SYM(GroupEnd, "ncclGroupEnd")
        MDX_SYM(GetErrorString, "ncclGetErrorString")
#undef MDX_SYM
        ok = true;
    });
    if (!ok) {
        set_error("%s", why);
        return nullptr;
    }
    return &r;
}

#define MDX_NCCL(call)                                                                              \
    do {                                                                                            \
        ncclResult_t r_ = (call);                                                                   \
        if (r_ != ncclSuccess) {                                                                    \
            ::mdx::set_error("%s failed: %s (%s:%d)", #call, R->GetErrorString(r_), __FILE__, __LINE__); \
            return MDX_ERR_RUNTIME;                                                                 \
        }                                                                                           \
    } while (0)

}  // namespace mdx

using namespace mdx;

struct mdx_comm {
    ncclComm_t comm;
    int nranks, rank, device;
};

extern "C" {

int mdx_comm_unique_id(void *id_host)
{
    MDX_CHECK_ARG(id_host, "mdx_comm_unique_id: NULL pointer");
    static_assert(MDX_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    const Rccl *R = rccl();
    if (!R) return MDX_ERR_RUNTIME;
    ncclUniqueId id;
    MDX_NCCL(R->GetUniqueId(&id));
    memcpy(id_host, id.internal, NCCL_UNIQUE_ID_BYTES);
    return MDX_OK;
}

int mdx_comm_init(mdx_comm **out, const void *id_host, int nranks, int rank)
{
    MDX_CHECK_ARG(out && id_host, "mdx_comm_init: NULL pointer");
    MDX_CHECK_ARG(nranks >= 1 && rank >= 0 && rank < nranks, "mdx_comm_init: rank %d of %d", rank, nranks);
    const Rccl *R = rccl();
    if (!R) return MDX_ERR_RUNTIME;
    ncclUniqueId id;
    memcpy(id.internal, id_host, NCCL_UNIQUE_ID_BYTES);
    mdx_comm *c = new mdx_comm();
    c->nranks = nranks;
    c->rank = rank;
    if (hipGetDevice(&c->device) != hipSuccess) c->device = -1;
    ncclResult_t r = R->CommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        set_error("mdx_comm_init: ncclCommInitRank(rank %d of %d) failed: %s", rank, nranks, R->GetErrorString(r));
        delete c;
        return MDX_ERR_RUNTIME;
    }
    *out = c;
    return MDX_OK;
}

int mdx_comm_destroy(mdx_comm *c)
{
    if (!c) return MDX_OK;
    const Rccl *R = rccl();
    if (!R) {                           // cannot happen after a successful mdx_comm_init in this process; never a silent leak
        delete c;
        return MDX_ERR_RUNTIME;
    }
    ncclResult_t r = R->CommDestroy(c->comm);
    delete c;
    if (r != ncclSuccess) {
        set_error("mdx_comm_destroy: ncclCommDestroy failed: %s", R->GetErrorString(r));
        return MDX_ERR_RUNTIME;
    }
    return MDX_OK;
}

int mdx_comm_info(const mdx_comm *c, int *nranks, int *rank)
{
    MDX_CHECK_ARG(c, "mdx_comm_info: NULL communicator");
    if (nranks) *nranks = c->nranks;
    if (rank) *rank = c->rank;
    return MDX_OK;
}

int mdx_query_bounds(int64_t nq, int nranks, int rank, int64_t *lo, int64_t *hi)
{
    MDX_CHECK_ARG(nq >= 0 && nranks >= 1 && rank >= 0 && rank < nranks && lo && hi, "mdx_query_bounds: bad arguments");
    const int64_t base = nq / nranks, rem = nq % nranks;
    *lo = rank * base + (rank < rem ? rank : rem);
    *hi = *lo + base + (rank < rem ? 1 : 0);
    return MDX_OK;
}

static int check_widths(const mdx_comm *c, const int64_t *widths, int64_t nq, const char *who)
{
    MDX_CHECK_ARG(c && widths, "%s: NULL pointer", who);
    MDX_CHECK_ARG(nq >= 0, "%s: nq=%lld", who, (long long)nq);
    if (c->device >= 0) {               // a communicator belongs to the device that was current at mdx_comm_init: work enqueued with
        int cur = -1;                   // another device current would go to the wrong GPU's stream and fail obscurely or hang
        if (hipGetDevice(&cur) == hipSuccess)
            MDX_CHECK_ARG(cur == c->device, "%s: the communicator was created on device %d, device %d is current", who, c->device, cur);
    }
    for (int g = 0; g < c->nranks; ++g) MDX_CHECK_ARG(widths[g] >= 0, "%s: widths[%d] negative", who, g);
    return MDX_OK;
}

int mdx_allgather_scores(mdx_comm *c, const float *local, int64_t nq, const int64_t *widths, float *all, void *stream)
{
    int rc = check_widths(c, widths, nq, "mdx_allgather_scores");
    if (rc != MDX_OK) return rc;
    MDX_CHECK_ARG(all && (local || widths[c->rank] == 0 || nq == 0), "mdx_allgather_scores: NULL buffer");
    const Rccl *R = rccl();
    if (!R) return MDX_ERR_RUNTIME;
    hipStream_t s = (hipStream_t)stream;
    bool even = true;
    for (int g = 1; g < c->nranks; ++g) even = even && widths[g] == widths[0];
    const size_t mine = (size_t)(nq * widths[c->rank]);
    if (even) {                                     // blocks of one size: the library's own all-gather
        if (mine) MDX_NCCL(R->AllGather(local, all, mine, ncclFloat, c->comm, s));
        return MDX_OK;
    }
    // shards differ (by a row, normally): one send + one receive per peer, each on its own link
    MDX_NCCL(R->GroupStart());
    int64_t off = 0;
    ncclResult_t first = ncclSuccess;
    for (int g = 0; g < c->nranks; ++g) {
        const size_t cnt = (size_t)(nq * widths[g]);
        // (the rank's own block travels the same way: a send and a receive to itself inside the group -- RCCL turns the
        // pair into a device copy -- so that the exchange makes no HIP call of its own)
        if (mine) { ncclResult_t r = R->Send(local, mine, ncclFloat, g, c->comm, s); if (first == ncclSuccess) first = r; }
        if (cnt) { ncclResult_t r = R->Recv(all + off, cnt, ncclFloat, g, c->comm, s); if (first == ncclSuccess) first = r; }
        off += (int64_t)cnt;
    }
    MDX_NCCL(R->GroupEnd());            // always closed, also after a failed call inside the group
    MDX_NCCL(first);
    return MDX_OK;
}

int mdx_exchange_scores(mdx_comm *c, const float *local, int64_t nq, const int64_t *widths, float *mine_out, void *stream)
{
    int rc = check_widths(c, widths, nq, "mdx_exchange_scores");
    if (rc != MDX_OK) return rc;
    const Rccl *R = rccl();
    if (!R) return MDX_ERR_RUNTIME;
    hipStream_t s = (hipStream_t)stream;
    int64_t qlo = 0, qhi = 0;
    mdx_query_bounds(nq, c->nranks, c->rank, &qlo, &qhi);
    const int64_t w_mine = widths[c->rank], nq_mine = qhi - qlo;
    MDX_CHECK_ARG((local || w_mine == 0 || nq == 0) && (mine_out || nq_mine == 0), "mdx_exchange_scores: NULL buffer");
    MDX_NCCL(R->GroupStart());
    int64_t off = 0;
    ncclResult_t first = ncclSuccess;
    for (int g = 0; g < c->nranks; ++g) {
        int64_t glo = 0, ghi = 0;
        mdx_query_bounds(nq, c->nranks, g, &glo, &ghi);
        const size_t send_cnt = (size_t)((ghi - glo) * w_mine);          // rows of g's queries in my block: contiguous
        const size_t recv_cnt = (size_t)(nq_mine * widths[g]);
        if (send_cnt) { ncclResult_t r = R->Send(local + glo * w_mine, send_cnt, ncclFloat, g, c->comm, s); if (first == ncclSuccess) first = r; }
        if (recv_cnt) { ncclResult_t r = R->Recv(mine_out + off, recv_cnt, ncclFloat, g, c->comm, s); if (first == ncclSuccess) first = r; }
        off += (int64_t)recv_cnt;
    }
    MDX_NCCL(R->GroupEnd());
    MDX_NCCL(first);
    return MDX_OK;
}

}  // extern "C"
