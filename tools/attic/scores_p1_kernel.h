// NOT SHIPPED (tools/attic): first persistent design, ONE workgroup per CU with LDS-counter hand-offs; measured 3.5 ms
// against 2.65 ms for the per-row-block kernel (bit-identical results).  Not built by anything; kept for the record.
//
// Persistent form of the similarity kernel for big shards (one workgroup per CU for the whole launch).
//
// Stamps on the per-row-block kernel of mdx_scores_kernel.h (tools/scores_ablate.hip, 1 004 993 x 70 x 2048)
// showed where its last 13 % went: a workgroup lives 167 us, of which 14 us are the epilogue (accumulators
// through the ring, 35 KiB of stores, waiting for them) and the 512 workgroup slots are occupied 95 % of the
// launch (dispatch gaps + the last partial round of 7 852 workgroups); both workgroups of a CU start together
// and stay in phase, so their epilogues coincide and the MFMA pipes idle.  Here
//   * ONE workgroup per CU (grid = #CUs) walks a contiguous range of row tiles in blocks of 16 tiles:
//     no dispatch gaps, no partial last round (ranges differ by at most one tile, and a block of <= 8 tiles
//     costs half a block: the tiles are dealt one per consumer wave);
//   * 8 consumer waves (two per SIMD, 2 row tiles x all query tiles each) + 4 LDS-DMA loader waves;
//     the query tiles of a chunk are shared by 256 rows instead of 128 (half the L2 -> LDS query traffic);
//   * the loader -> consumer hand-off is a pair of LDS counters per wave instead of a workgroup barrier:
//     FULL[l] = chunks loader l has landed (written after its counted vmcnt), FREE[w] = chunks consumer w
//     has finished reading.  Waves drift apart, so the two consumers of a SIMD cover each other's
//     start-of-chunk LDS latency and epilogues, and the ring keeps streaming across row blocks: the first
//     chunks of block b+1 land while block b is in its epilogue;
//   * every consumer wave transposes its own accumulators through a private 2.3 KiB LDS patch (the ring is
//     never used as staging) and does not wait for its stores.
// Arithmetic is unchanged: every score is the k = 0..D-1 fma chain (oracle/chain.c); the leftover query tile
// (<= 8 queries, QR = 1) goes through v_mfma_f32_4x4x1 exactly as in scores_lc_kernel.
//
// All spins are bounded: a hand-off that never completes sets *err and the wave carries on (wrong scores,
// reported by mdx_scores as an error) instead of hanging the GPU.
#pragma once
#include "mdx_scores_kernel.h"

namespace mdx {

constexpr int P_CW = 8;             // consumer waves
constexpr int P_LW = 4;             // loader waves
constexpr int P_R = 2;              // row tiles per consumer wave
constexpr int P_BLOCK_TILES = P_CW * P_R;
constexpr int P_STG_W = 36;         // floats per staged query row: 32 rows + 4 (16-B aligned, conflict-free)
constexpr int P_STG_FLOATS = 16 * P_STG_W;
constexpr unsigned P_SPIN_LIMIT = 1u << 22;

template <int QT, int QR, int KC, int NSTAGE>
constexpr int p_lds_bytes()
{
    return NSTAGE * ((QT + QR) * KC + P_BLOCK_TILES * KC) * 1024 + P_CW * P_STG_FLOATS * 4 + 64;
}

template <int N>
__device__ __forceinline__ bool p_wait_min(const unsigned *words, unsigned need, unsigned *err)
{
    for (unsigned spins = 0;; ++spins) {
        unsigned m = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const unsigned v = __hip_atomic_load(words + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            m = v < m ? v : m;
        }
        if (__builtin_amdgcn_readfirstlane(m) >= need) return true;
        if (spins > P_SPIN_LIMIT) {
            if (err) *err = 1u;
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// MFMAs of one chunk (KC k-blocks) of one consumer wave: NR of its row tiles x QT query tiles (+ the leftover tile).
// Every fragment of the chunk is requested before the first MFMA (hipcc otherwise sinks some of the reads into
// the MFMA stream and waits for them a few instructions later: a lone wave then reached 69 % of the MFMA rate);
// `release` runs once all of them have landed -- the slot can be refilled while the MFMAs are still issuing.
template <int NR, int QT, int QR, int KC, typename Release>
__device__ __forceinline__ void p_chunk(f32x4 (&acc)[P_R][QT > 0 ? QT : 1], f32x4 &accl, const f32x4 *slot, int lane, int wave,
                                        int l_q, int l_boff, Release release)
{
    constexpr int QTILES = (QT + QR) * KC;
    constexpr int QTX = QT > 0 ? QT : 1;
    const f32x4 *qs = slot + lane;
    const f32x4 *bs = slot + (QTILES + wave * KC) * 64 + lane;
    f32x4 a[KC][QTX], b[KC][NR], al[4], bl[4];
#pragma unroll
    for (int kb = 0; kb < KC; ++kb) {
#pragma unroll
        for (int q = 0; q < QT; ++q) a[kb][q] = qs[(q * KC + kb) * 64];
#pragma unroll
        for (int r = 0; r < NR; ++r) b[kb][r] = bs[(r * P_CW * KC + kb) * 64];
    }
    auto load_left = [&](int kb) {          // leftover tile: the 16 k of the lane's query row / database row
        const f32x4 *ql = slot + (QT * KC + kb) * 64 + l_q;
        const f32x4 *bw = slot + (QTILES + wave * KC + kb) * 64 + l_boff;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) { al[gq] = ql[16 * gq]; bl[gq] = bw[16 * gq]; }
    };
    if constexpr (QR != 0) load_left(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kb = 0; kb < KC; ++kb) {
        if constexpr (QR == 0) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < NR; ++r)
#pragma unroll
                    for (int q = 0; q < QT; ++q) acc[r][q] = MmaF32::step(t, a[kb][q], b[kb][r], acc[r][q]);
        } else {
            constexpr int PIN = 0x0002 | 0x0004 | 0x0070 | 0x0380 | 0x0400;     // MFMA order pinned, see scores_lc_kernel
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int n_small = t == 0 ? 0 : (t == 3 ? 8 : 4);
                int done = 0;
#pragma unroll
                for (int r = 0; r < NR; ++r)
#pragma unroll
                    for (int q = 0; q < QT; ++q) {
                        acc[r][q] = MmaF32::step(t, a[kb][q], b[kb][r], acc[r][q]);
                        __builtin_amdgcn_sched_barrier(PIN);
                        const int due = ((r * QT + q + 1) * n_small) / (NR * QT);
#pragma unroll
                        for (; done < due; ++done) {
                            const int st = (t == 3 && done >= 4) ? 3 : t - 1, gq = done & 3;
                            accl = __builtin_amdgcn_mfma_f32_4x4x1f32(al[gq][st], bl[gq][st], accl, 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(PIN);
                        }
                    }
            }
            if (kb + 1 < KC) {          // the next k-block's leftover operands reuse the registers: they land under its first step
                __builtin_amdgcn_sched_barrier(0);
                load_left(kb + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (kb == KC - 2 || KC == 1) {
            // all LDS reads of the chunk are issued (QR: all but the last leftover operands -> released after the loop)
            if constexpr (QR == 0) {
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                release();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if constexpr (QR != 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        release();
    }
}

template <int QT, int QR, int KC, int NSTAGE, int DB_AUX = 2, int ABL = 0>   // ABL: timing-only ablations (tools/scores_ablate.hip)
__global__ __launch_bounds__((P_CW + P_LW) * 64, 3) void scores_p_kernel(const f32x4 *__restrict__ db,
                                                                          const f32x4 *__restrict__ qtiles,
                                                                          float *__restrict__ out, int64_t n, int64_t RT,
                                                                          int KB, int nq_valid, unsigned *err)
{
    constexpr int QTL = QT + QR;
    constexpr int QTILES = QTL * KC;
    constexpr int BTILES = P_BLOCK_TILES * KC;
    constexpr int STAGE_TILES = QTILES + BTILES;
    constexpr int PER_LOADER = (STAGE_TILES + P_LW - 1) / P_LW;
    static_assert((NSTAGE - 1) * PER_LOADER <= 63, "vmcnt is 6 bits");
    extern __shared__ __attribute__((aligned(16))) f32x4 ring[];        // [NSTAGE][STAGE_TILES][64] | staging | counters
    float *stg_all = (float *)(ring + NSTAGE * STAGE_TILES * 64);
    unsigned *full = (unsigned *)(stg_all + P_CW * P_STG_FLOATS);         // [P_LW]
    unsigned *freed = full + 4;                                           // [P_CW]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nchunks = KB / KC;
    // this workgroup's row tiles [t0, t1) and its blocks of 16
    const int64_t t0 = (RT * (int64_t)blockIdx.x) / gridDim.x, t1 = (RT * ((int64_t)blockIdx.x + 1)) / gridDim.x;
    const int nblocks = (int)((t1 - t0 + P_BLOCK_TILES - 1) / P_BLOCK_TILES);
    const int total = nblocks * nchunks;                                  // chunks of this workgroup

    if (tid < 12) full[tid] = 0;                                          // FULL[4] + FREE[8]
    __syncthreads();

    if (wave >= P_CW) {
        // ------------------------------------------------------------------ loader
        const int lw = wave - P_CW;
        auto issue = [&](int g) {
            const int blk = g / nchunks, c = g - blk * nchunks;
            const int64_t tb = t0 + (int64_t)blk * P_BLOCK_TILES;
            const int nt = (int)((t1 - tb) < P_BLOCK_TILES ? (t1 - tb) : P_BLOCK_TILES);
            f32x4 *slot = ring + (g % NSTAGE) * (STAGE_TILES * 64);
            if ((ABL & 7) >= 2) return;                                           // ablation: hand-off only, nothing loaded
#pragma unroll
            for (int t = 0; t < PER_LOADER; ++t) {
                const int i = (lw + t * P_LW) < STAGE_TILES ? (lw + t * P_LW) : (STAGE_TILES - 1);
                const f32x4 *src;
                if (i < QTILES) {
                    const int qt = i / KC, kbc = i % KC;
                    src = qtiles + ((int64_t)qt * KB + (int64_t)c * KC + kbc) * 64 + lane;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                     (__attribute__((address_space(3))) void *)(slot + i * 64), 16, 0, 0);
                } else {
                    const int j = i - QTILES;
                    int tile = j / KC;
                    const int kbc = j % KC;
                    tile = tile < nt ? tile : nt - 1;                      // absent tiles of a partial block: any valid tile
                    src = db + shard_tile(tb + tile, (int64_t)c * KC + kbc, KB) * 64 + lane;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                     (__attribute__((address_space(3))) void *)(slot + i * 64), 16, 0, DB_AUX);
                }
            }
        };
#pragma unroll
        for (int g = 0; g < NSTAGE - 1; ++g)
            if (g < total) issue(g);
        unsigned long long lt_vm = 0, lt_free = 0, lt_issue = 0, lt0 = 0, lt1 = 0;      // ABL & 8: diagnostic stamps
        auto lstamp = [&]() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; };
        const unsigned long long lt_begin = (ABL & 8) ? lstamp() : 0;
        for (int g = 0; g < total; ++g) {
            if (ABL & 8) lt0 = lstamp();
            const int younger = (total - 1 - g) < (NSTAGE - 2) ? (total - 1 - g) : (NSTAGE - 2);
            if (younger >= NSTAGE - 2 && NSTAGE > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * PER_LOADER) : "memory");
            else if (younger == 1 && NSTAGE > 3)     asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_LOADER) : "memory");
            else                                      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // relaxed LDS store behind the explicit counted wait above (a release would add vmcnt(0) and drain the ring)
            if (lane == 0) __hip_atomic_store(full + lw, (unsigned)(g + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (ABL & 8) { lt1 = lstamp(); lt_vm += lt1 - lt0; }
            if (g + NSTAGE - 1 < total) {
                // the slot of chunk g-1 is refilled: every consumer must have finished reading it
                if (g > 0 && (ABL & 7) != 3) p_wait_min<P_CW>(freed, (unsigned)g, err);
                asm volatile("" ::: "memory");
                if (ABL & 8) { lt0 = lstamp(); lt_free += lt0 - lt1; }
                issue(g + NSTAGE - 1);
                if (ABL & 8) lt_issue += lstamp() - lt0;
            }
        }
        if ((ABL & 8) && lane == 0 && err) {
            unsigned long long *d = (unsigned long long *)err + 8 + 256 * P_CW * 4 + ((int64_t)blockIdx.x * P_LW + lw) * 4;
            d[0] = lt_vm; d[1] = lt_free; d[2] = lt_issue; d[3] = lstamp() - lt_begin;
        }
        return;
    }

    // ---------------------------------------------------------------------- consumer
    float *stg = stg_all + wave * P_STG_FLOATS;
    const int l_q = 4 * ((lane >> 2) & 1) + (lane & 3);                  // leftover path: query row in the leftover tile
    const int l_row = 4 * (lane >> 3) + (lane & 3);                      // row among the wave's 32 rows
    const int l_boff = (l_row >> 4) * P_CW * KC * 64 + (l_row & 15);     // tile r*8+wave: r-stride = 8 tiles
    int g = 0;
    unsigned long long st_poll = 0, st_chunk = 0, st_epi = 0, st_a = 0, st_b = 0;      // ABL & 8: diagnostic stamps
    auto stamp = [&]() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; };
    const unsigned long long st_begin = (ABL & 8) ? stamp() : 0;
    for (int blk = 0; blk < nblocks; ++blk) {
        const int64_t tb = t0 + (int64_t)blk * P_BLOCK_TILES;
        const int nt = (int)((t1 - tb) < P_BLOCK_TILES ? (t1 - tb) : P_BLOCK_TILES);
        const int nr = __builtin_amdgcn_readfirstlane(nt > P_CW + wave ? 2 : (nt > wave ? 1 : 0));   // tiles wave, wave+8
        f32x4 acc[P_R][QT > 0 ? QT : 1];
#pragma unroll
        for (int r = 0; r < P_R; ++r)
#pragma unroll
            for (int q = 0; q < QT; ++q) acc[r][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 accl = (f32x4){0.f, 0.f, 0.f, 0.f};

        for (int c = 0; c < nchunks; ++c, ++g) {
            if (ABL & 8) st_a = stamp();
            if ((ABL & 7) != 3) p_wait_min<P_LW>(full, (unsigned)(g + 1), err);
            asm volatile("" ::: "memory");                              // no LDS read of the chunk above the poll
            if (ABL & 8) { st_b = stamp(); st_poll += st_b - st_a; }
            const f32x4 *slot = ring + (g % NSTAGE) * (STAGE_TILES * 64);
            // called once every LDS read of the chunk has returned: the slot may be refilled
            auto release = [&]() {
                if (lane == 0) __hip_atomic_store(freed + wave, (unsigned)(g + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            };
            if ((ABL & 7) == 1 || ((ABL & 7) == 4 && wave >= 4)) { acc[0][0] += slot[lane]; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); release(); }
            else if (nr == 2) p_chunk<2, QT, QR, KC>(acc, accl, slot, lane, wave, l_q, l_boff, release);
            else if (nr == 1) p_chunk<1, QT, QR, KC>(acc, accl, slot, lane, wave, l_q, l_boff, release);
            else release();
            if (ABL & 8) { st_a = stamp(); st_chunk += st_a - st_b; }
        }

        // epilogue: per query tile, the wave's two 16x16 results -> private patch [16 queries][32 rows] -> rows of
        // 16 consecutive scores per (query, tile) leave as 64-B runs; stores are not waited for
        if (ABL & 8) st_a = stamp();
        if (nr > 0) {
            const int fq = 4 * (lane >> 4), fj = lane & 15;
            const int64_t row_a = (tb + wave) * TILE_ROWS, row_b = (tb + P_CW + wave) * TILE_ROWS;
            const int64_t my_row = ((lane & 31) < 16 ? row_a : row_b) + (lane & 15);
            const bool row_ok = my_row < n && ((lane & 31) < 16 || nr == 2);
            // the output pointer advances by two query rows per store; the row stride is made opaque per block so
            // that the 40 (query, row) offsets are not hoisted out of the block loop into registers (they spilled)
            int64_t nn = n;
            asm volatile("" : "+s"(nn));
            float *po = out + (int64_t)(lane >> 5) * nn + my_row;
            int qi = lane >> 5;
#pragma unroll
            for (int q = 0; q < QT; ++q) {
#pragma unroll
                for (int r = 0; r < P_R; ++r)
#pragma unroll
                    for (int i = 0; i < 4; ++i) stg[(fq + i) * P_STG_W + r * 16 + fj] = acc[r][q][i];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const float v = stg[(2 * s + (lane >> 5)) * P_STG_W + (lane & 31)];
                    if (row_ok && qi < nq_valid) *po = v;
                    po += 2 * nn;
                    qi += 2;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // reads done before the patch is rewritten
            }
            if constexpr (QR != 0) {
#pragma unroll
                for (int v = 0; v < 4; ++v) stg[(4 * ((lane >> 2) & 1) + v) * P_STG_W + l_row] = accl[v];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float v = stg[(2 * s + (lane >> 5)) * P_STG_W + (lane & 31)];
                    if (row_ok && qi < nq_valid) *po = v;
                    po += 2 * nn;
                    qi += 2;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
        if (ABL & 8) st_epi += stamp() - st_a;
    }
    if ((ABL & 8) && lane == 0 && err) {
        unsigned long long *d = (unsigned long long *)err + 8 + ((int64_t)blockIdx.x * P_CW + wave) * 4;
        d[0] = st_poll; d[1] = st_chunk; d[2] = st_epi; d[3] = stamp() - st_begin;
    }
}

}  // namespace mdx
