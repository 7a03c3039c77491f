// Direct-store exchange of per-shard partial scores (round 6): no separate collective.
//
// What is sharded is `scores = np.dot(vecs.T, qvecs)` (mdir/components/optim/score/cirscore.py:69), and the ranking that follows
// (`np.argsort(-scores, axis=0)`, cirscore.py:70) is split by QUERY: rank r ranks queries [qlo_r, qhi_r) against all rows.  With
// mdx_exchange_scores (mdx_comm.hip) every rank first writes its block S_g [nq, w_g] to its own memory and an all-to-all then
// moves the rows to their owners.  Here the similarity kernel's epilogue writes query q's run of scores STRAIGHT into row
// q - qlo_owner of the owner's receive buffer, at the columns of this shard, over xGMI (the owner's buffer is mapped into this
// process with hipIpcOpenMemHandle): the transfer is spread over the whole kernel, needs no collective launch, and the owner
// finds a DENSE [nq_mine, n_total] matrix -- mdx_rank_full, no peer blocks.  One flag per peer closes a step.
//
//   every rank:  mdx_p2p_create  -> 64-byte handle  -> (handles gathered by any means)  -> mdx_p2p_connect
//   every step:  mdx_scores_p2p (once per shard / chunk of this rank)   -> mdx_p2p_close_step -> mine [nq_mine, n_total]
//
// Memory of a rank, ONE hipMalloc (one handle): two receive buffers (steps alternate: a writer that is a step ahead fills the
// other one -- it has seen this rank's flag of the step before, which this rank raised after it had finished reading that
// buffer) | flags: one 64-byte line per peer | a status word.
//
// Ordering (LLVM AMDGPU memory model, gfx942/gfx950).  Writer: the routed stores are system-scope (sc0 sc1: written through,
// nothing of them stays dirty in an XCD's L2); the kernel's end waits for them; the close kernel, later on the same stream,
// raises the flag with a system-scope release store.  Reader: its close kernel spins on its own flag lines with system-scope
// acquire loads; the kernels launched after it (the ranking) begin with the agent-scope acquire every kernel begins with, which
// invalidates what its L2s may still hold of the buffer from two steps ago.  HARDWARE STATUS: verified with several rank
// processes on ONE GPU (same-device IPC; tests/test_gpu_round6.py) -- it has never run over xGMI; tools/preflight_ranks.py
// checks it on the first multi-GPU node before bench.py uses it, and the all-to-all stays the default.

#include "mdx_common.h"

struct mdx_p2p {
    int nranks, rank;
    int64_t nq, n_total, nq_cap;         // nq_cap = ceil(nq / nranks): rows of every rank's receive buffer
    int64_t buf_bytes, flags_off, status_off, total_bytes;
    char *base;                          // this rank's allocation
    char *peer[64];                      // every rank's allocation as mapped here (peer[rank] = base)
    bool opened[64];                     // mapped with hipIpcOpenMemHandle (to be closed)
    bool connected;
    float **routes;                      // device: [2][nq] row pointers (parity, query)
    uint32_t **peer_flags;               // device: [nranks] where MY flag lives in each peer's allocation
    uint32_t step;                       // steps closed so far
};

namespace mdx {

static void query_bounds(int64_t nq, int nranks, int r, int64_t *lo, int64_t *hi)
{
    const int64_t base = nq / nranks, rem = nq % nranks;
    *lo = r * base + (r < rem ? r : rem);
    *hi = *lo + base + (r < rem ? 1 : 0);
}

// mdx_index.hip: the row pointers of the step that is open
bool p2p_route(const mdx_p2p *p, int64_t nq, float *const **rows)
{
    if (!p || !p->connected || nq != p->nq) return false;
    *rows = p->routes + (int64_t)(p->step & 1u) * p->nq;
    return true;
}

constexpr uint32_t P2P_SPIN_TICKS = 2000000000u;        // 20 s of the 100 MHz constant clock: a peer that never arrives ends the wait

// lane r: raise my flag at peer r (system-scope release), then wait for peer r's flag here (system-scope acquire)
__global__ void p2p_close_kernel(uint32_t *const *peer_flags, uint32_t *my_flags, int nranks, int rank, uint32_t step, uint32_t *status)
{
    const int r = threadIdx.x;
    if (r >= nranks || r == rank) return;
    __hip_atomic_store(peer_flags[r], step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while ((int32_t)(__hip_atomic_load(my_flags + r * 16, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - step) < 0) {
        __builtin_amdgcn_s_sleep(8);
        if (__builtin_amdgcn_s_memrealtime() - t0 > P2P_SPIN_TICKS) {
            atomicOr(status, 1u << (r & 31));
            break;
        }
    }
}

}  // namespace mdx

using namespace mdx;

extern "C" {

int mdx_p2p_create(mdx_p2p **out, int nranks, int rank, int64_t nq, int64_t n_total, void *handle_host)
{
    MDX_CHECK_ARG(out && handle_host, "mdx_p2p_create: NULL pointer");
    MDX_CHECK_ARG(nranks >= 1 && nranks <= 64 && rank >= 0 && rank < nranks, "mdx_p2p_create: rank %d of %d", rank, nranks);
    MDX_CHECK_ARG(nq > 0 && nq <= 128 && n_total > 0, "mdx_p2p_create: nq=%lld (1..128) n_total=%lld", (long long)nq, (long long)n_total);
    static_assert(sizeof(hipIpcMemHandle_t) <= MDX_P2P_HANDLE_BYTES, "handle size");
    mdx_p2p *p = new mdx_p2p();
    memset(p, 0, sizeof *p);
    p->nranks = nranks;
    p->rank = rank;
    p->nq = nq;
    p->n_total = n_total;
    p->nq_cap = ceil_div(nq, (int64_t)nranks);
    p->buf_bytes = round_up(p->nq_cap * n_total * 4, 256);
    p->flags_off = 2 * p->buf_bytes;
    p->status_off = p->flags_off + 64 * 64;
    p->total_bytes = p->status_off + 256;
    hipError_t e = hipMalloc((void **)&p->base, (size_t)p->total_bytes);
    if (e != hipSuccess) {
        set_error("mdx_p2p_create: hipMalloc(%lld bytes) failed: %s", (long long)p->total_bytes, hipGetErrorString(e));
        delete p;
        return MDX_ERR_NOMEM;
    }
    memset(handle_host, 0, MDX_P2P_HANDLE_BYTES);
    e = hipMemset(p->base + p->flags_off, 0, (size_t)(p->total_bytes - p->flags_off));      // flags and status start at zero
    if (e != hipSuccess || hipMalloc((void **)&p->routes, sizeof(float *) * 2 * nq) != hipSuccess ||
        hipMalloc((void **)&p->peer_flags, sizeof(uint32_t *) * 64) != hipSuccess) {
        set_error("mdx_p2p_create: %s", e != hipSuccess ? hipGetErrorString(e) : "hipMalloc of the route tables failed");
        (void)hipGetLastError();
        (void)hipFree(p->base);
        if (p->routes) (void)hipFree(p->routes);
        if (p->peer_flags) (void)hipFree(p->peer_flags);
        delete p;
        return e != hipSuccess ? MDX_ERR_RUNTIME : MDX_ERR_NOMEM;
    }
    e = hipSuccess;
    if (nranks > 1) {
        hipIpcMemHandle_t h;
        e = hipIpcGetMemHandle(&h, p->base);
        if (e == hipSuccess) memcpy(handle_host, &h, sizeof h);
        else (void)hipGetLastError();           // a process that cannot export (no dmabuf IPC) can still be connected by pointers
    }
    p->peer[rank] = p->base;
    *out = p;
    if (e != hipSuccess) {
        set_error("mdx_p2p_create: hipIpcGetMemHandle failed: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 exported?); mdx_p2p_connect_ptrs still works",
                  hipGetErrorString(e));
        return MDX_ERR_RUNTIME;                 // *out is valid: the caller may connect by pointers or destroy it
    }
    return MDX_OK;
}

void *mdx_p2p_base(mdx_p2p *p) { return p ? (void *)p->base : nullptr; }

int64_t mdx_p2p_bytes(const mdx_p2p *p) { return p ? p->total_bytes : 0; }

static int p2p_finish_connect(mdx_p2p *p)
{
    // row pointers: query q belongs to rank o = owner(q); its row there is q - qlo_o, in buffer `parity`
    float *rows[2 * 128];
    for (int par = 0; par < 2; ++par)
        for (int o = 0; o < p->nranks; ++o) {
            int64_t lo, hi;
            query_bounds(p->nq, p->nranks, o, &lo, &hi);
            for (int64_t q = lo; q < hi; ++q)
                rows[par * p->nq + q] = (float *)(p->peer[o] + par * p->buf_bytes) + (q - lo) * p->n_total;
        }
    uint32_t *flags[64] = {nullptr};
    for (int r = 0; r < p->nranks; ++r) flags[r] = (uint32_t *)(p->peer[r] + p->flags_off) + p->rank * 16;
    MDX_HIP(hipMemcpy(p->routes, rows, sizeof(float *) * 2 * p->nq, hipMemcpyHostToDevice));
    MDX_HIP(hipMemcpy(p->peer_flags, flags, sizeof flags, hipMemcpyHostToDevice));
    p->connected = true;
    return MDX_OK;
}

int mdx_p2p_connect(mdx_p2p *p, const void *handles_host)
{
    MDX_CHECK_ARG(p && handles_host, "mdx_p2p_connect: NULL pointer");
    MDX_CHECK_ARG(!p->connected, "mdx_p2p_connect: already connected");
    for (int r = 0; r < p->nranks; ++r) {
        if (r == p->rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, (const char *)handles_host + (size_t)r * MDX_P2P_HANDLE_BYTES, sizeof h);
        void *ptr = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            set_error("mdx_p2p_connect: hipIpcOpenMemHandle of rank %d's buffer failed: %s", r, hipGetErrorString(e));
            (void)hipGetLastError();
            for (int k = 0; k < r; ++k)
                if (p->opened[k]) { (void)hipIpcCloseMemHandle(p->peer[k]); p->opened[k] = false; }
            return MDX_ERR_RUNTIME;
        }
        p->peer[r] = (char *)ptr;
        p->opened[r] = true;
    }
    return p2p_finish_connect(p);
}

int mdx_p2p_connect_ptrs(mdx_p2p *p, void *const *bases)
{
    MDX_CHECK_ARG(p && bases, "mdx_p2p_connect_ptrs: NULL pointer");
    MDX_CHECK_ARG(!p->connected, "mdx_p2p_connect_ptrs: already connected");
    for (int r = 0; r < p->nranks; ++r) {
        if (r == p->rank) continue;
        MDX_CHECK_ARG(bases[r], "mdx_p2p_connect_ptrs: rank %d's base is NULL", r);
        p->peer[r] = (char *)bases[r];
    }
    return p2p_finish_connect(p);
}

int mdx_p2p_close_step(mdx_p2p *p, float **mine, void *stream)
{
    MDX_CHECK_ARG(p && p->connected, "mdx_p2p_close_step: not connected");
    const uint32_t parity = p->step & 1u;
    p->step += 1;
    if (p->nranks > 1) {
        hipLaunchKernelGGL(p2p_close_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (uint32_t *const *)p->peer_flags,
                           (uint32_t *)(p->base + p->flags_off), p->nranks, p->rank, p->step, (uint32_t *)(p->base + p->status_off));
        MDX_LAUNCH_CHECK();
    }
    if (mine) *mine = (float *)(p->base + parity * p->buf_bytes);
    return MDX_OK;
}

int mdx_p2p_status(mdx_p2p *p, uint32_t *late_peers, void *stream)
{
    MDX_CHECK_ARG(p && late_peers, "mdx_p2p_status: NULL pointer");
    MDX_HIP(hipMemcpyAsync(late_peers, p->base + p->status_off, 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    MDX_HIP(hipStreamSynchronize((hipStream_t)stream));
    return MDX_OK;
}

int mdx_p2p_destroy(mdx_p2p *p)
{
    if (!p) return MDX_OK;
    hipError_t bad = hipSuccess;
    for (int r = 0; r < p->nranks; ++r)
        if (p->opened[r]) {
            hipError_t e = hipIpcCloseMemHandle(p->peer[r]);
            if (e != hipSuccess) bad = e;
        }
    hipError_t e = hipFree(p->base);
    if (e != hipSuccess) bad = e;
    if (p->routes) (void)hipFree(p->routes);
    if (p->peer_flags) (void)hipFree(p->peer_flags);
    delete p;
    if (bad != hipSuccess) {
        set_error("mdx_p2p_destroy: %s", hipGetErrorString(bad));
        (void)hipGetLastError();
        return MDX_ERR_RUNTIME;
    }
    return MDX_OK;
}

}  // extern "C"
