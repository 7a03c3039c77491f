// The first (register-streaming) similarity kernel, with its timing-only ablation switches.
// NOT part of libmdx.so any more: kept for tools/scores_ablate.hip, which measured why it lost
//  against the loader/consumer kernel that replaced it.
#pragma once
#include "mdx_scores_kernel.h"

namespace mdx {

constexpr int KBC = 4;          // k-blocks per LDS chunk  (64 k)
constexpr int WAVES = 4;

// ---------------------------------------------------------------------------
// similarity: one workgroup = NW waves (8 = two per SIMD), each wave owns R row tiles
// (16*R database rows) and ALL QT query tiles; the dimension is walked in chunks of 64.
//   - database: streamed once from HBM with non-temporal 16-B loads into a ring of NS
//               register sets, NS-1 chunks ahead of its use
//   - queries : chunk staged through LDS once per workgroup (double-buffered, one
//               barrier per chunk), read back as ds_read_b128 = 4 A operands
//   - fp32 MFMA 16x16x4: A = queries (M), B = database rows (N)
// Measured choices (tools/scores_ablate.hip): 8 waves sharing one query
// buffer, R = 1, NS = 3 and the explicit "set has landed" marker are each worth a few
// per cent; the kernel is bound by fp32 MFMA issue, with the query re-reads from L2
// and the database stream competing for the same per-CU load path.
// ---------------------------------------------------------------------------
// ABL != 0 builds timing-only ablations for tools/scores_ablate.hip (wrong results):
//   bit0: no database loads in the loop, bit1: no query staging in the loop,
//   bit2: no barrier, bit3: no MFMA, bit4: no explicit "set has landed" marker.
template <int QT, int R, int ABL = 0, bool CM = false, bool NT = true, int NS = 3, int WPS = 2, bool SPREAD = false, int NW = 8, int KC = KBC>
__global__ __launch_bounds__(NW * 64, WPS) void scores_kernel(const f32x4 *__restrict__ db,
                                                     const f32x4 *__restrict__ qtiles,
                                                     float *__restrict__ out, int64_t n, int KB,
                                                     int nq_valid, int64_t RTS = 0, unsigned long long *dbg = nullptr)
{
    // ABL bit5: in-kernel cycle stamps (diagnostic build only; sums per segment go to dbg)
    unsigned long long t_bar = 0, t_land = 0, t_issue = 0, t_mfma = 0, t_stq = 0, ts0 = 0, ts1 = 0;
#define MDX_STAMP(acc_)                                                              \
    if (ABL & 32) {                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                           \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts1)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                           \
        acc_ += ts1 - ts0; ts0 = ts1;                                                \
    }

    constexpr int CHUNK4 = QT * KC * 64;          // float4 per LDS buffer
    constexpr int NT_ = NW * 64;                              // threads per workgroup
    constexpr int COPIES = (CHUNK4 + NT_ - 1) / NT_;          // float4 per thread per chunk
    __shared__ f32x4 lds[2][CHUNK4];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int64_t rt0 = ((int64_t)blockIdx.x * NW + wave) * R;
    const int nchunks = KB / KC;

    const f32x4 *bp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) bp[r] = db + (CM ? (rt0 + r) * KC : (rt0 + r) * KB) * 64 + lane;
    const int64_t cstride = CM ? RTS * KC * 64 : KC * 64;     // float4 between consecutive chunks

    // per-thread source offsets of the query-chunk copy (chunk 0), in float4
    int qsrc[COPIES];
#pragma unroll
    for (int i = 0; i < COPIES; ++i) {
        const int e = (tid + i * NT_) < CHUNK4 ? (tid + i * NT_) : 0;
        const int tl = e >> 6, ln = e & 63;
        const int qt = tl / KC, kbc = tl % KC;
        qsrc[i] = (qt * KB + kbc) * 64 + ln;
    }

    f32x4 acc[R][QT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int q = 0; q < QT; ++q) acc[r][q] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Ring of NS statically indexed register sets for the database stream: while chunk c
    // is multiplied out of set c % NS, chunks c+1 .. c+NS-1 are in flight or landed, so
    // the only wait in steady state is for data requested NS-1 chunks (~5k cycles each) ago.
    f32x4 b[NS][R][KC], qreg[COPIES];

#define MDX_LOAD_B(set, c)                                                          \
    _Pragma("unroll") for (int r = 0; r < R; ++r)                                   \
        _Pragma("unroll") for (int kb = 0; kb < KC; ++kb)                          \
            b[set][r][kb] = NT ? __builtin_nontemporal_load(&bp[r][(c) * cstride + kb * 64]) \
                               : bp[r][(c) * cstride + kb * 64];
#define MDX_LOAD_Q(c)                                                               \
    _Pragma("unroll") for (int i = 0; i < COPIES; ++i)                              \
        qreg[i] = qtiles[qsrc[i] + (c) * KC * 64];
#define MDX_STORE_Q(buf)                                                            \
    _Pragma("unroll") for (int i = 0; i < COPIES; ++i)                              \
        if (CHUNK4 % NT_ == 0 || tid + i * NT_ < CHUNK4) lds[buf][tid + i * NT_] = qreg[i];
#define MDX_COMPUTE(buf, set)                                                       \
    _Pragma("unroll") for (int kb = 0; kb < KC; ++kb) {                            \
        f32x4 a[QT];                                                                \
        _Pragma("unroll") for (int q = 0; q < QT; ++q)                              \
            a[q] = lds[buf][(q * KC + kb) * 64 + lane];                            \
        _Pragma("unroll") for (int t = 0; t < 4; ++t)                               \
            _Pragma("unroll") for (int r = 0; r < R; ++r)                           \
                _Pragma("unroll") for (int q = 0; q < QT; ++q)                      \
                    acc[r][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(               \
                        a[q][t], b[set][r][kb][t], acc[r][q], 0, 0, 0);             \
    }

// one k-block of one chunk: A fragments from LDS, then (SPREAD) this k-block's share of
// the NEXT chunks' loads, then the MFMAs; sched_barrier keeps the loads where they are
// written so that they trickle into the memory pipe instead of arriving as one burst.
#define MDX_STEP(buf, set, kb, c, do_b, do_q)                                       \
    {                                                                               \
        f32x4 a[QT];                                                                \
        _Pragma("unroll") for (int q = 0; q < QT; ++q)                              \
            a[q] = lds[buf][(q * KC + kb) * 64 + lane];                            \
        if (do_b) {                                                                 \
            _Pragma("unroll") for (int r = 0; r < R; ++r)                           \
                b[(set + NS - 1) % NS][r][kb] =                                     \
                    NT ? __builtin_nontemporal_load(&bp[r][((c) + NS - 1) * cstride + kb * 64]) \
                       : bp[r][((c) + NS - 1) * cstride + kb * 64];                 \
        }                                                                           \
        if (do_q) {                                                                 \
            _Pragma("unroll") for (int i = kb; i < COPIES; i += KC)                \
                qreg[i] = qtiles[qsrc[i] + ((c) + 1) * KC * 64];                   \
        }                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                          \
        _Pragma("unroll") for (int t = 0; t < 4; ++t)                               \
            _Pragma("unroll") for (int r = 0; r < R; ++r)                           \
                _Pragma("unroll") for (int q = 0; q < QT; ++q)                      \
                    acc[r][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(               \
                        a[q][t], b[set][r][kb][t], acc[r][q], 0, 0, 0);             \
        __builtin_amdgcn_sched_barrier(0);                                          \
    }

    // prologue: chunks 0 .. NS-2 of the database, chunk 0 of the queries
    MDX_LOAD_Q(0);
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nchunks) { MDX_LOAD_B(s, s); }
    MDX_STORE_Q(0);

    // One barrier per chunk.  lds[x] is rewritten (MDX_STORE_Q) only after the barrier
    // of the step that follows its last read, so every wave is done reading it.
    for (int c0 = 0; c0 < nchunks; c0 += NS) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int c = c0 + s;
            if (c >= nchunks) break;
            const bool moreq = c + 1 < nchunks;
            if (SPREAD) {
                const bool moreb = c + NS - 1 < nchunks;
                __syncthreads();
                MDX_STEP(c & 1, s, 0, c, moreb, moreq);
                MDX_STEP(c & 1, s, 1, c, moreb, moreq);
                MDX_STEP(c & 1, s, 2, c, moreb, moreq);
                MDX_STEP(c & 1, s, 3, c, moreb, moreq);
                if (moreq) { MDX_STORE_Q((c + 1) & 1); }
                continue;
            }
            MDX_STAMP(t_stq)
            if (!(ABL & 4)) __syncthreads();
            MDX_STAMP(t_bar)
            // Make "set s has landed" explicit BEFORE new loads are issued: the empty asm
            // uses every register of the set, so the compiler's wait for them sits here,
            // where nothing younger is in flight (they were requested NS-1 chunks ago).
            // Without it hipcc's merged loop-header state makes the first MFMA wait for
            // the loads issued just above it (vmcnt(7) of 13), i.e. no prefetch at all.
            if (!(ABL & 16)) {
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int kb = 0; kb < KC; ++kb) asm volatile("" : "+v"(b[s][r][kb]));
            }
            MDX_STAMP(t_land)
            if (moreq && !(ABL & 2)) { MDX_LOAD_Q(c + 1); }
            if (c + NS - 1 < nchunks && !(ABL & 1)) { MDX_LOAD_B((s + NS - 1) % NS, c + NS - 1); }
            MDX_STAMP(t_issue)
            if (!(ABL & 8)) { MDX_COMPUTE(c & 1, s); }
            MDX_STAMP(t_mfma)
            if (moreq && !(ABL & 2)) { MDX_STORE_Q((c + 1) & 1); }
        }
    }
    if (ABL & 8) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int kb = 0; kb < KC; ++kb) acc[r][0] += b[s][r][kb];
    }
    if ((ABL & 32) && dbg && lane == 0) {
        unsigned long long *d = dbg + ((int64_t)blockIdx.x * NW + wave) * 8;
        d[0] = t_bar; d[1] = t_land; d[2] = t_issue; d[3] = t_mfma; d[4] = t_stq;
    }
#undef MDX_STAMP
#undef MDX_LOAD_B
#undef MDX_LOAD_Q
#undef MDX_STORE_Q
#undef MDX_COMPUTE
#undef MDX_STEP

    // C/D map of 16x16x4: reg i of lane l is (M = 4*(l>>4)+i, N = l&15)
    const int qrow = 4 * (lane >> 4);
    const int64_t col = lane & 15;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t row = (rt0 + r) * TILE_ROWS + col;
        if (row >= n) continue;
#pragma unroll
        for (int q = 0; q < QT; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int qi = q * 16 + qrow + i;
                if (qi < nq_valid) out[(int64_t)qi * n + row] = acc[r][q][i];
            }
    }
}


}  // namespace mdx
