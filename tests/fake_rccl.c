/* TEST-ONLY stand-in for librccl.so.1 on a box without GPUs: the handful of RCCL entry points libmdx.so binds at run
 * time (mdir_amd/csrc/mdx_comm.hip), implemented over host memory with the "ranks" as THREADS of one process.  Built
 * with SONAME librccl.so.1 and loaded (RTLD_GLOBAL) by tests/test_comm_fake_rccl.py before libmdx looks for RCCL, so that
 * dlopen("librccl.so.1", RTLD_NOLOAD) finds it.  Semantics kept from NCCL: ncclSend / ncclRecv only enqueue inside
 * ncclGroupStart .. ncclGroupEnd and complete at ncclGroupEnd; a send to rank p matches the receive from this rank posted
 * by p (per pair in order); counts must agree; ncclAllGather needs equal counts.  The product never loads this file. */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef struct fake_comm { int rank, nranks; } *ncclComm_t;
enum { ncclSuccess = 0, ncclInvalidArgument = 4, ncclInvalidUsage = 5 };

#define MAXR 16
#define MAXOPS 64
static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t cv = PTHREAD_COND_INITIALIZER;
/* mailbox[src][dst]: sends posted by src for dst, not yet consumed */
static struct { const void *buf; size_t bytes; int taken; } box[MAXR][MAXR][MAXOPS];
static int nbox[MAXR][MAXR], nread[MAXR][MAXR];

typedef struct { int is_send, peer; const void *sbuf; void *rbuf; size_t bytes; ncclComm_t comm; } op_t;
static __thread op_t ops[MAXOPS];
static __thread int nops, depth;

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) { memset(id, 0x5A, sizeof *id); return ncclSuccess; }
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : (r == ncclInvalidUsage ? "invalid usage" : "invalid argument"); }

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    (void)id;
    if (nranks < 1 || nranks > MAXR || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    *comm = (ncclComm_t)malloc(sizeof **comm);
    (*comm)->rank = rank;
    (*comm)->nranks = nranks;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) { free(comm); return ncclSuccess; }
ncclResult_t ncclGroupStart(void) { if (depth++ == 0) nops = 0; return ncclSuccess; }

static size_t dsize(int dt) { return dt == 7 ? 4 : (dt == 8 ? 8 : 1); }      /* ncclFloat = 7, ncclDouble = 8 */

static ncclResult_t run_ops(void)
{
    ncclResult_t rc = ncclSuccess;
    pthread_mutex_lock(&mu);
    for (int i = 0; i < nops; ++i)                        /* post every send first: receives may need them in any order */
        if (ops[i].is_send) {
            const int me = ops[i].comm->rank, p = ops[i].peer;
            const int slot = nbox[me][p]++ % MAXOPS;
            box[me][p][slot].buf = ops[i].sbuf;
            box[me][p][slot].bytes = ops[i].bytes;
            box[me][p][slot].taken = 0;
        }
    pthread_cond_broadcast(&cv);
    for (int i = 0; i < nops; ++i)
        if (!ops[i].is_send) {
            const int me = ops[i].comm->rank, p = ops[i].peer;
            while (nread[p][me] >= nbox[p][me]) pthread_cond_wait(&cv, &mu);
            const int slot = nread[p][me]++ % MAXOPS;
            if (box[p][me][slot].bytes != ops[i].bytes) rc = ncclInvalidArgument;       /* count mismatch between the pair */
            else memcpy(ops[i].rbuf, box[p][me][slot].buf, ops[i].bytes);
            box[p][me][slot].taken = 1;
            pthread_cond_broadcast(&cv);
        }
    /* a send buffer must stay untouched until its receiver has copied it: wait for all of mine */
    for (int i = 0; i < nops; ++i)
        if (ops[i].is_send) {
            const int me = ops[i].comm->rank, p = ops[i].peer;
            for (;;) {
                int pending = 0;
                for (int s = 0; s < MAXOPS; ++s)
                    if (box[me][p][s].buf == ops[i].sbuf && box[me][p][s].bytes == ops[i].bytes && !box[me][p][s].taken && s < nbox[me][p]) pending = 1;
                if (!pending) break;
                pthread_cond_wait(&cv, &mu);
            }
        }
    pthread_mutex_unlock(&mu);
    nops = 0;
    return rc;
}

ncclResult_t ncclGroupEnd(void)
{
    if (depth <= 0) return ncclInvalidUsage;
    if (--depth > 0) return ncclSuccess;
    return run_ops();
}

static ncclResult_t add(int is_send, const void *sbuf, void *rbuf, size_t count, int dt, int peer, ncclComm_t comm)
{
    if (!comm || peer < 0 || peer >= comm->nranks || nops >= MAXOPS) return ncclInvalidArgument;
    const int solo = depth == 0;
    if (solo) nops = 0;
    ops[nops++] = (op_t){is_send, peer, sbuf, rbuf, count * dsize(dt), comm};
    return solo ? run_ops() : ncclSuccess;
}

ncclResult_t ncclSend(const void *buf, size_t count, int dt, int peer, ncclComm_t comm, void *stream) { (void)stream; return add(1, buf, 0, count, dt, peer, comm); }
ncclResult_t ncclRecv(void *buf, size_t count, int dt, int peer, ncclComm_t comm, void *stream) { (void)stream; return add(0, 0, buf, count, dt, peer, comm); }

ncclResult_t ncclAllGather(const void *sendbuf, void *recvbuf, size_t count, int dt, ncclComm_t comm, void *stream)
{
    ncclResult_t rc = ncclGroupStart();
    for (int p = 0; p < comm->nranks && rc == ncclSuccess; ++p) {
        rc = ncclSend(sendbuf, count, dt, p, comm, stream);
        if (rc == ncclSuccess) rc = ncclRecv((char *)recvbuf + (size_t)p * count * dsize(dt), count, dt, p, comm, stream);
    }
    ncclResult_t end = ncclGroupEnd();
    return rc != ncclSuccess ? rc : end;
}
