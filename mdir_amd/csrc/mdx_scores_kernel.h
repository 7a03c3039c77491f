// The similarity kernel of libmdx.so (also built into tools/scores_ablate.hip for measurements).
#pragma once
#include "mdx_common.h"

namespace mdx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TILE_ROWS = 16;   // rows per tile  (MFMA N / M)
constexpr int TILE_K = 16;      // k per tile     (4 MFMA k-steps of 4)
constexpr int MAX_QT = 8;       // query tiles (of 16) per launch

// Element type of a shard.  A tile is always 64 lanes x 16 B; what the 16 bytes are and which
// MFMA consumes them is the only difference between the fp32 (exact chain) and the fp16
// (BASELINE.json configs[4]: "fp16 descriptors on CDNA4 fp16 MFMA") paths.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct MmaF32 {                 // v_mfma_f32_16x16x4_f32 x4: lane (g,j) element t = (row j, k 4t+g)
    static constexpr int KELEMS = 16;   // k per tile
    static constexpr int STEPS = 4;
    static __device__ __forceinline__ f32x4 step(int t, const f32x4 &a, const f32x4 &b, const f32x4 &c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[t], c, 0, 0, 0);
    }
};

struct MmaF16 {                 // v_mfma_f32_16x16x32_f16: lane (g,j) element e = (row j, k 8g+e), fp32 accumulate
    static constexpr int KELEMS = 32;
    static constexpr int STEPS = 1;
    static __device__ __forceinline__ f32x4 step(int, const f32x4 &a, const f32x4 &b, const f32x4 &c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

// ===========================================================================
// Loader / consumer form of the similarity kernel.
//
// In-kernel stamps on the kernel above (tools/scores_ablate.hip) show where it loses
// time: a wave spends about a third of its life BLOCKED ISSUING its global loads (a CU
// holds only a few tens of KiB of reads in flight, so later load instructions stall at
// issue), and a wave stalled on VMEM issue cannot issue MFMAs.  Here the roles are split:
//   waves 0-3  (one per SIMD)  consumers: LDS reads + fp32 MFMA only, never touch VMEM
//                              until the epilogue; each owns R row tiles x all QT query tiles
//   waves 4-7                  loaders: stream the NEXT chunks of the database tiles and
//                              of the query tiles into an LDS ring with LDS-DMA
//                              (global_load_lds_dwordx4: 1 KiB per instruction, lane-linear,
//                              exactly the tile format), then wait with a COUNTED vmcnt
// One raw s_barrier per chunk; ring of NSTAGE stages, a stage = (QT + 4R) x KC KiB.
// B_c = "stage c landed": loaders arrive after vmcnt says stage c is complete, consumers
// then read it; the slot of stage c-1 is refilled right after B_c (every consumer has
// finished chunk c-1 by then).  Accumulation order per output is unchanged (k ascending).
// ===========================================================================
template <int QT, int R, int KC, int NSTAGE, int DB_AUX = 0, bool STAMPS = false, typename MM = MmaF32>
__global__ __launch_bounds__(512, 2) void scores_lc_kernel(const f32x4 *__restrict__ db,
                                                           const f32x4 *__restrict__ qtiles,
                                                           float *__restrict__ out, int64_t n, int KB,
                                                           int nq_valid, unsigned long long *dbg = nullptr)
{
    constexpr int CW = 4;                           // consumer waves
    constexpr int LW = 4;                           // loader waves
    constexpr int QTILES = QT * KC;                 // KiB tiles of queries per stage
    constexpr int BTILES = CW * R * KC;             // KiB tiles of database per stage
    constexpr int STAGE_TILES = QTILES + BTILES;
    constexpr int PER_LOADER = (STAGE_TILES + LW - 1) / LW;   // uneven split: the last tile is loaded twice
    static_assert((NSTAGE - 1) * PER_LOADER <= 63, "vmcnt is 6 bits");
    extern __shared__ __attribute__((aligned(16))) f32x4 ring[];   // [NSTAGE][STAGE_TILES][64]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nchunks = KB / KC;
    const int64_t rt_wg = (int64_t)blockIdx.x * CW * R;            // first row tile of the workgroup
    // blockIdx.y = query pass: several full groups of QT query tiles in one launch (many queries
    // against a small database: one pass alone would not fill the chip)
    qtiles += (int64_t)blockIdx.y * QT * KB * 64;
    out += (int64_t)blockIdx.y * QT * TILE_ROWS * n;

    if (wave >= CW) {
        // ------------------------------------------------------------- loader
        const int lw = wave - CW;
        // this loader's tiles of a stage: i = lw, lw+LW, ...  (query tiles first)
        const f32x4 *src[PER_LOADER];
        int dst[PER_LOADER];
#pragma unroll
        for (int t = 0; t < PER_LOADER; ++t) {
            const int i = (lw + t * LW) < STAGE_TILES ? (lw + t * LW) : (STAGE_TILES - 1);
            dst[t] = i * 64;
            if (i < QTILES) {
                const int qt = i / KC, kbc = i % KC;
                src[t] = qtiles + ((int64_t)qt * KB + kbc) * 64 + lane;
            } else {
                const int j = i - QTILES;
                const int tile = j / KC, kbc = j % KC;              // tile = cw * R + r
                src[t] = db + ((rt_wg + tile) * KB + kbc) * 64 + lane;
            }
        }
        auto issue = [&](int c) {
            f32x4 *slot = ring + (c % NSTAGE) * (STAGE_TILES * 64);
#pragma unroll
            for (int t = 0; t < PER_LOADER; ++t) {
                // query tiles (re-read by every workgroup) keep the default cache policy; the
                // database stream may be marked non-temporal (DB_AUX = 2)
                if (DB_AUX != 0 && (lw + t * LW) >= QTILES)
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void *)(src[t] + (int64_t)c * KC * 64),
                        (__attribute__((address_space(3))) void *)(slot + dst[t]), 16, 0, DB_AUX);
                else
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void *)(src[t] + (int64_t)c * KC * 64),
                        (__attribute__((address_space(3))) void *)(slot + dst[t]), 16, 0, 0);
            }
        };
#pragma unroll
        for (int c = 0; c < NSTAGE - 1; ++c)
            if (c < nchunks) issue(c);
        for (int c = 0; c < nchunks; ++c) {
            // stage c must have landed: everything but the younger stages c+1 .. c+NSTAGE-2
            const int younger = (nchunks - 1 - c) < (NSTAGE - 2) ? (nchunks - 1 - c) : (NSTAGE - 2);
            if (younger >= NSTAGE - 2 && NSTAGE > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * PER_LOADER) : "memory");
            else if (younger == 1 && NSTAGE > 3)     asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_LOADER) : "memory");
            else                                      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                   // B_c
            if (c + NSTAGE - 1 < nchunks) issue(c + NSTAGE - 1);            // refill the slot of stage c-1
        }
        return;
    }

    // ----------------------------------------------------------------- consumer
    f32x4 acc[R][QT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int q = 0; q < QT; ++q) acc[r][q] = (f32x4){0.f, 0.f, 0.f, 0.f};

    unsigned long long t_wait = 0, t_work = 0, ts0 = 0, ts1 = 0;      // STAMPS: diagnostic build only
    if (STAMPS) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts0)::"memory");
    for (int c = 0; c < nchunks; ++c) {
        __builtin_amdgcn_s_barrier();                                       // B_c
        __builtin_amdgcn_sched_barrier(0);
        if (STAMPS) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts1)::"memory"); t_wait += ts1 - ts0; ts0 = ts1; }
        const f32x4 *slot = ring + (c % NSTAGE) * (STAGE_TILES * 64);
        const f32x4 *qs = slot + lane;
        const f32x4 *bs = slot + (QTILES + wave * R * KC) * 64 + lane;
#pragma unroll
        for (int kb = 0; kb < KC; ++kb) {
            f32x4 a[QT], b[R];
#pragma unroll
            for (int q = 0; q < QT; ++q) a[q] = qs[(q * KC + kb) * 64];
#pragma unroll
            for (int r = 0; r < R; ++r) b[r] = bs[(r * KC + kb) * 64];
#pragma unroll
            for (int t = 0; t < MM::STEPS; ++t)
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int q = 0; q < QT; ++q) acc[r][q] = MM::step(t, a[q], b[r], acc[r][q]);
        }
        // all LDS reads of this stage are consumed by the MFMAs above before the next barrier
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (STAMPS) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts1)::"memory"); t_work += ts1 - ts0; ts0 = ts1; }
    }
    if (STAMPS && dbg && lane == 0) { dbg[((int64_t)blockIdx.x * CW + wave) * 2] = t_wait; dbg[((int64_t)blockIdx.x * CW + wave) * 2 + 1] = t_work; }

    // Epilogue: the ring is free now (loaders have left, consumers are past their last read);
    // transpose the accumulators through LDS so that every query row of the workgroup's
    // 64*R database rows leaves as one contiguous run (full cache lines instead of 64-B pieces).
    constexpr int ROWS = CW * R * TILE_ROWS;        // database rows per workgroup
    constexpr int LDW = ROWS + 4;                   // +4: the four 16-lane groups hit different banks
    static_assert(QT * 16 * LDW * 4 <= NSTAGE * STAGE_TILES * 1024, "output staging must fit in the ring");
    __builtin_amdgcn_s_barrier();
    float *stage = (float *)ring;
    {
        const int qrow = 4 * (lane >> 4), col = lane & 15;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int q = 0; q < QT; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    stage[(q * 16 + qrow + i) * LDW + (wave * R + r) * TILE_ROWS + col] = acc[r][q][i];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int64_t row0 = rt_wg * TILE_ROWS;
    const int rows_valid = (int)((n - row0) < ROWS ? (n - row0) : ROWS);
    // queries of this group that exist (the last group of a launch may be partial)
    const int nq_here = (nq_valid - (int)blockIdx.y * QT * TILE_ROWS) < QT * TILE_ROWS ? (nq_valid - (int)blockIdx.y * QT * TILE_ROWS)
                                                                                   : QT * TILE_ROWS;
    for (int e = tid; e < nq_here * ROWS; e += CW * 64) {
        const int qi = e / ROWS, rr = e % ROWS;
        if (rr < rows_valid) out[(int64_t)qi * n + row0 + rr] = stage[qi * LDW + rr];
    }
}

}  // namespace mdx
