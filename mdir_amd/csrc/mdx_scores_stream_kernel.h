// Register-streaming similarity for the fp16 shard (MDX_F16 -- BASELINE.json configs[4]): the HBM-bound mode.
//
// Why another shape (numbers: profiles/r04_split3.md, "The fp16 shard and the epilogue").  With one MFMA per tile pair the
// ring kernel (mdx_scores_kernel.h) is nothing but a stream -- 4.12 GB of shard in, 0.28 GB of scores out -- and it reads at
// 5.1 TB/s where a plain stream of the shard gets 6.0-6.9.  The ring keeps the stream in flight in LDS, 64 KiB per CU beside
// the query tiles, refilled a whole stage at a time behind a workgroup-wide barrier: a shape made for the fp32 chain, whose
// MFMA waves must never stall on vector-memory issue.  Here the matrix pipe is nearly idle, so the roles are cut differently:
//   * every wave streams ITS OWN row tiles global -> VGPR (non-temporal 16-B loads, one fully coalesced KiB per instruction,
//     the tile format as it is), PF = 4 chunks of 32 k ahead, refilling a register slot the moment its MFMAs are issued: no
//     workgroup-wide hand-off for the shard at all, 4 waves x 3 workgroups per CU = 96 KiB per CU in flight in registers;
//   * the query tiles of the next four chunks go into a two-stage LDS ring (plain loads right after the barrier that opens
//     an iteration, ds_write_b128 at its end -- by then they are long back), one raw s_barrier per four chunks;
//   * no dedicated loader waves (with 12 waves per workgroup the allocation is 168 VGPRs and a single spilled register makes
//     the compiler drain the prefetch queue every iteration: its reload is a vector-memory load) and no LDS-DMA (issued by a
//     wave that also reads LDS it makes the compiler wait vmcnt(0) before the next ds_read);
//   * the prologue issues in the loop's order, pinned with sched_barrier: the compiler derives its counted vmcnt waits from
//     the issue order it sees on every way into the loop.
// A persistent variant that lets the stream run on into the next row block across the epilogue was built and is NOT faster
// (git show 47a9fe2:tools/attic/scores_stream_persistent_kernel.h: 0.79 against 0.76 ms; 768 workgroups x 10.2 blocks leave a 7 % tail, and
// what the epilogue costs is its scattered stores, not a stopped stream).  Operand format, MFMA and accumulators are the ring
// kernel's: results are bit-identical to it (tests/test_gpu_f16.py).
#pragma once
#include <type_traits>

#include "mdx_scores_kernel.h"

namespace mdx {

constexpr int STREAM_PF = 4;            // chunks of 32 k a wave keeps in flight = chunks per query stage
constexpr int STREAM_CW = 4;            // waves per workgroup: each owns R row tiles

template <int QT, int R>
constexpr int stream_lds_bytes()
{
    constexpr int ring = 2 * STREAM_PF * QT * 1024, epi = QT * 16 * (STREAM_CW * R * TILE_ROWS + 4) * 4;
    return ring > epi ? ring : epi;
}

// db: the fp16 shard's tiles (16 rows x 32 k, 1 KiB); NC = tiles per row tile = chunks of 32 k, a multiple of STREAM_PF;
// qtiles: [query tile][NC] KiB tiles of fp16 queries (retile_f16_kernel), already advanced to the launch's first query tile.
// One workgroup per row block of STREAM_CW * R row tiles; blockIdx.y = pass over groups of QT query tiles.
template <int QT, int R, int WGS>
__global__ __launch_bounds__(STREAM_CW * 64, WGS) void scores_f16_stream_kernel(const f32x4 *__restrict__ db, const f32x4 *__restrict__ qtiles,
                                                                               float *__restrict__ out, int64_t n, int NC, int nq_valid)
{
    constexpr int PF = STREAM_PF, CW = STREAM_CW;
    constexpr int STAGE_TILES = PF * QT;                            // [chunk of the stage][query tile]
    constexpr int PER_WAVE = (STAGE_TILES + CW - 1) / CW;           // query tiles of a stage this wave brings in (uneven: the last tile again)
    extern __shared__ __attribute__((aligned(16))) f32x4 ring[];   // [2][STAGE_TILES][64]; the output staging afterwards

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NIT = NC / PF;
    const int64_t rt_wg = row_block_of(blockIdx.x, gridDim.x) * CW * R;       // first row tile of the workgroup (XCD-contiguous order)
    qtiles += (int64_t)blockIdx.y * QT * NC * 64;
    out += (int64_t)blockIdx.y * QT * TILE_ROWS * n;

    // query stage s (chunks PF*s .. PF*s+PF-1) -> ring slot s & 1: this wave's tiles, through registers
    const f32x4 *qsrc[PER_WAVE];
    int qdst[PER_WAVE];
#pragma unroll
    for (int t = 0; t < PER_WAVE; ++t) {
        const int i = (wave + t * CW) < STAGE_TILES ? (wave + t * CW) : (STAGE_TILES - 1);
        const int g = i / QT, q = i % QT;
        qdst[t] = i * 64 + lane;
        qsrc[t] = qtiles + ((int64_t)q * NC + g) * 64 + lane;
    }
    f32x4 qreg[PER_WAVE];
    auto load_queries = [&](int s) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < PER_WAVE; ++t) qreg[t] = qsrc[t][(int64_t)s * PF * 64];
    };
    auto store_queries = [&](int s) __attribute__((always_inline)) {
        f32x4 *slot = ring + (s & 1) * (STAGE_TILES * 64);
#pragma unroll
        for (int t = 0; t < PER_WAVE; ++t) slot[qdst[t]] = qreg[t];
    };

    f32x4 acc[R][QT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int q = 0; q < QT; ++q) acc[r][q] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const f32x4 *dbp = db + (rt_wg + wave * R) * (int64_t)NC * 64 + lane;     // this wave's R row tiles: NC KiB each, back to back
    f32x4 raw[PF][R];
    auto fetch = [&](int j, int c) __attribute__((always_inline)) {            // chunk c of the wave's row tiles -> register slot j
#pragma unroll
        for (int r = 0; r < R; ++r) raw[j][r] = __builtin_nontemporal_load(dbp + ((int64_t)r * NC + c) * 64);
    };
    load_queries(0);                            // first, so that the wait for them leaves the shard loads below in flight
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < PF; ++j) {
        fetch(j, j);
        __builtin_amdgcn_sched_barrier(0);
    }
    store_queries(0);
    __builtin_amdgcn_sched_barrier(0);

    auto body = [&](int it, auto more) __attribute__((always_inline)) {
        constexpr bool MORE = decltype(more)::value;
        // B_it: every wave has written its part of stage `it` (and waited for the writes), and every wave has left stage
        // it-1, whose slot this iteration's writes go to
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MORE) load_queries(it + 1);
        __builtin_amdgcn_sched_barrier(0);      // the scheduler otherwise sinks every load of the iteration to its end
        const f32x4 *qs = ring + (it & 1) * (STAGE_TILES * 64) + lane;
#pragma unroll
        for (int j = 0; j < PF; ++j) {
#pragma unroll
            for (int q = 0; q < QT; ++q) {
                const f16x8 a = __builtin_bit_cast(f16x8, qs[(j * QT + q) * 64]);
#pragma unroll
                for (int r = 0; r < R; ++r)     // waits (counted vmcnt) for this chunk's loads only
                    acc[r][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(f16x8, raw[j][r]), acc[r][q], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MORE) fetch(j, (it + 1) * PF + j);                // the slot's MFMAs are issued: refill it at once
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (MORE) store_queries(it + 1);
    };
    for (int it = 0; it + 1 < NIT; ++it) body(it, std::true_type{});
    body(NIT - 1, std::false_type{});

    // Epilogue (as the ring kernels): transpose the accumulators through LDS so that every query row of the workgroup's
    // rows leaves as one contiguous run
    constexpr int ROWS = CW * R * TILE_ROWS;
    constexpr int LDW = ROWS + 4;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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
    const int left = nq_valid - (int)blockIdx.y * QT * TILE_ROWS;
    const int nq_here = left < QT * TILE_ROWS ? left : QT * TILE_ROWS;
    for (int e = tid; e < nq_here * ROWS; e += CW * 64) {
        const int qi = e / ROWS, rr = e % ROWS;
        if (rr < rows_valid) store_score<false>(out + (int64_t)qi * n + row0 + rr, stage[qi * LDW + rr]);
    }
}

}  // namespace mdx
