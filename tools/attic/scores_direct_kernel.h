// PARKED (round 4): built, bit-identical to the ring kernel, NOT faster -- numbers in profiles/r04_split3.md.  Kept for the
// record and for tools/split_ablate.hip.  What it taught: (1) with real operands every form of the split-precision kernel
// lands on 2.0-2.1 ms while with all-zero operands the same instruction streams run at their stream-only time (1.67 ms):
// the mode is POWER-bound (DVFS), not issue- or latency-bound, so the shape of the workgroup does not matter;
// (2) three ISA traps of a wave that both streams and multiplies: a single spilled register (its reload is a vector-memory
// load: the compiler drains the prefetch queue with vmcnt(0) every iteration), LDS-DMA issued by a wave that also reads LDS
// (vmcnt(0) before the next ds_read), and a prologue the scheduler has reshuffled (every counted wait in the loop then
// over-waits by chunks: the compiler merges the issue orders of both ways into the loop).
//
// Register-streaming similarity for the HBM-BOUND modes: MDX_F32_SPLIT3 on an fp32 shard and the fp16 shard.
//
// The LDS-ring kernels (mdx_scores_kernel.h, mdx_scores_split_kernel.h) keep the shard stream in flight in LDS: with the
// query tiles beside it the ring holds two stages (64 KiB of shard per CU) ahead of the one being multiplied, and a stage
// is only refilled after every wave of the workgroup has left it.  tools/split_ablate.hip: with the consumers idle that
// ring streams the 8.2 GB shard in 1.65 ms (5 TB/s) -- the ring depth, not HBM -- and any consumer time adds to the
// latency of the refill (the split-precision consumers: 2.0 ms).  That shape was made for the fp32 chain, whose MFMA waves
// must never stall on vector-memory issue; a mode that needs the matrix pipe for less than half of the launch can let its
// MFMA waves stall, so here the roles are cut differently:
//   consumer waves (CW = 8, two per SIMD)  stream THEIR OWN row tiles global -> VGPR (non-temporal 16-B loads, one fully
//       coalesced KiB per instruction, the tile format as it is), PF = 4 chunks of 32 k ahead: 128 KiB per CU in flight
//       in registers, each wave refilling a slot the moment it has consumed it -- no workgroup-wide hand-off for the
//       shard at all -- then split (fp32 shard) and multiply;
//   the QUERY tiles of the next four chunks go into a two-stage LDS ring (from the L2), every wave bringing in an eighth of
//       the stage: plain loads issued right after the barrier that opens an iteration, ds_write_b128 at its end (the
//       loads are long back by then), one raw s_barrier per four chunks.  Two things this shape avoids, both found in
//       the ISA: (i) dedicated loader waves make 12 waves = 3 per SIMD = 168 VGPRs, one register spilled -- and the
//       reload of a spill is a vector-memory load, for which the compiler drains the whole prefetch queue (vmcnt(0))
//       every iteration; 8 waves have 256; (ii) LDS-DMA issued by a wave that also READS LDS makes the compiler wait
//       vmcnt(0) before the next ds_read (it cannot tell the DMA's destination from the read's source).
// Same operand formats, same accumulators and the same epilogue as the ring kernels: results are bit-identical to them.
#pragma once
#include <type_traits>

#include "mdx_scores_split_kernel.h"       // mdir_amd/csrc (build with -I mdir_amd/csrc -I tools/attic)

namespace mdx {

struct DirectSplit3 {                   // fp32 shard, three bf16 pieces per operand, six products (mdx_scores_split_kernel.h)
    static constexpr bool TWO_ACC = false;
    static constexpr int TK = 2;        // KiB of shard per row tile and chunk of 32 k: the fp32 tiles (rt, 2c), (rt, 2c+1)
    static constexpr int NQP = 3;       // query piece arrays
    struct Db { u32x4 h, m, l; };
    static __device__ __forceinline__ void prepare(const f32x4 (&raw)[TK], Db &d) { split3(raw[0], raw[1], d.h, d.m, d.l); }
    static __device__ __forceinline__ f32x4 mma(const u32x4 (&q)[NQP], const Db &d, f32x4 a)
    {
        a = mfma_bf16(q[2], d.h, a);    // smallest terms first
        a = mfma_bf16(q[0], d.l, a);
        a = mfma_bf16(q[1], d.m, a);
        a = mfma_bf16(q[1], d.h, a);
        a = mfma_bf16(q[0], d.m, a);
        return mfma_bf16(q[0], d.h, a);
    }
};

struct DirectSplit2 {                   // fp32 shard, MDX_F32_SPLIT2: two fp16 pieces (scale 2^17 for unit-norm rows: timing harness), three products
    static constexpr int TK = 2;
    static constexpr int NQP = 2;
    static constexpr bool TWO_ACC = true;
    struct Db { u32x4 h, m; };
    static __device__ __forceinline__ void prepare(const f32x4 (&raw)[TK], Db &d) { split2(raw[0], raw[1], 131072.0f, d.h, d.m); }
    static __device__ __forceinline__ f32x4 mma(const u32x4 (&q)[NQP], const Db &d, f32x4 a) { return mfma_f16(q[0], d.h, a); }
    static __device__ __forceinline__ f32x4 mmx(const u32x4 (&q)[NQP], const Db &d, f32x4 a) { return mfma_f16(q[0], d.m, mfma_f16(q[1], d.h, a)); }
};

struct DirectF16 {                      // fp16 shard (MDX_F16): one v_mfma_f32_16x16x32_f16 per tile pair
    static constexpr bool TWO_ACC = false;
    static constexpr int TK = 1;
    static constexpr int NQP = 1;
    struct Db { f32x4 v; };
    static __device__ __forceinline__ void prepare(const f32x4 (&raw)[TK], Db &d) { d.v = raw[0]; }
    static __device__ __forceinline__ f32x4 mma(const u32x4 (&q)[NQP], const Db &d, f32x4 a)
    {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, q[0]), __builtin_bit_cast(f16x8, d.v), a, 0, 0, 0);
    }
};

constexpr int DIRECT_PF = 4;            // chunks of 32 k a consumer wave keeps in flight = chunks per query stage

template <int QT, int R, int CW, typename MM>
constexpr int direct_lds_bytes()
{
    constexpr int ring = 2 * DIRECT_PF * MM::NQP * QT * 1024, epi = QT * 16 * (CW * R * TILE_ROWS + 4) * 4;
    return ring > epi ? ring : epi;
}

// db: the shard's tiles; KB = KiB tiles per row tile (the index's KB: 16-k blocks of an fp32 shard, 32-k blocks of an fp16 one);
// qpieces: [piece][QT_total][NC] KiB tiles, NC = KB / TK chunks.  NC must be a multiple of DIRECT_PF and >= 2 * DIRECT_PF.
template <int QT, int R, int CW, typename MM, int ABL = 0, int WGS = 1>
__global__ __launch_bounds__(CW * 64, WGS) void scores_direct_kernel(const f32x4 *__restrict__ db, const u32x4 *__restrict__ qpieces,
                                                                   float *__restrict__ out, int64_t n, int KB, int QT_total,
                                                                   int qt_first, int nq_valid)
{
    constexpr int PF = DIRECT_PF, TK = MM::TK, NQP = MM::NQP;
    constexpr int STAGE_TILES = PF * NQP * QT;                      // [chunk of the stage][piece][query tile]
    constexpr int PER_WAVE = (STAGE_TILES + CW - 1) / CW;           // query tiles of a stage this wave brings in (uneven: the last tile again)
    extern __shared__ __attribute__((aligned(16))) f32x4 ring[];   // [2][STAGE_TILES][64]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NC = KB / TK, NIT = NC / PF;
    const int64_t rt_wg = row_block_of(blockIdx.x, gridDim.x) * CW * R;
    const int qt0 = qt_first + (int)blockIdx.y * QT;
    out += (int64_t)qt0 * TILE_ROWS * n;

    // query stage `it` -> ring slot it & 1: this wave's tiles, through registers
    const f32x4 *qsrc[PER_WAVE];
    int qdst[PER_WAVE];
#pragma unroll
    for (int t = 0; t < PER_WAVE; ++t) {
        const int i = (wave + t * CW) < STAGE_TILES ? (wave + t * CW) : (STAGE_TILES - 1);
        const int g = i / (NQP * QT), p = (i / QT) % NQP, q = i % QT;
        qdst[t] = i * 64 + lane;
        qsrc[t] = (const f32x4 *)qpieces + ((int64_t)(p * QT_total + qt0 + q) * NC + g) * 64 + lane;
    }
    f32x4 qreg[PER_WAVE];
    auto load_queries = [&](int it) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < PER_WAVE; ++t) qreg[t] = qsrc[t][(int64_t)it * PF * 64];
    };
    auto store_queries = [&](int it) __attribute__((always_inline)) {
        f32x4 *slot = ring + (it & 1) * (STAGE_TILES * 64);
#pragma unroll
        for (int t = 0; t < PER_WAVE; ++t) slot[qdst[t]] = qreg[t];
    };

    // ----------------------------------------------------------------- consumer
    constexpr bool TWO = MM::TWO_ACC;
    f32x4 acc[R][QT], acx[TWO ? R : 1][TWO ? QT : 1];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            acc[r][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (TWO) acx[r][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }

    const f32x4 *dbp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) dbp[r] = db + (rt_wg + wave * R + r) * (int64_t)KB * 64 + lane;
    f32x4 raw[PF][R][TK];
    auto fetch = [&](int j, int c) __attribute__((always_inline)) {            // chunk c of the wave's row tiles -> register slot j
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int t = 0; t < TK; ++t) raw[j][r][t] = __builtin_nontemporal_load(dbp[r] + (int64_t)(c * TK + t) * 64);
    };
    // The prologue issues in the loop's order (pinned): the compiler derives its counted waits from the issue order it
    // sees on BOTH ways into the loop, and a prologue it has reshuffled makes every wait in the loop over-wait by chunks
    if constexpr (ABL != 4 && ABL != 5) load_queries(0);    // first, so that the wait for them leaves the shard loads below in flight
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < PF; ++j) {
        fetch(j, j);
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (ABL != 4 && ABL != 5) store_queries(0);
    __builtin_amdgcn_sched_barrier(0);
    // Software pipeline inside the wave: while the MFMAs of chunk c run, the operands of chunk c+1 are prepared (the fp32
    // shard: 88 vector instructions of splitting per chunk, which otherwise sit in front of 60 MFMAs that wait for them --
    // and the two waves of a SIMD, released by the same barrier, would both split and then both multiply).
    typename MM::Db dcur[R];
#pragma unroll
    for (int r = 0; r < R; ++r) MM::prepare(raw[0][r], dcur[r]);
    fetch(0, PF);                               // NIT >= 2 (the host's condition for this kernel): chunk PF exists
    __builtin_amdgcn_sched_barrier(0);

    // TAIL: 0 = steady state, 1 = the iteration before the last (its last step has nothing left to fetch), 2 = the last
    auto body = [&](int it, auto tail) __attribute__((always_inline)) {
        constexpr int TAIL = decltype(tail)::value;
        // B_it: every wave has written its part of stage `it` (and waited for the writes), and every wave has left stage
        // it-1, whose slot this iteration's writes go to
        if constexpr (ABL != 4 && ABL != 5) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (TAIL < 2 && ABL != 4 && ABL != 5) load_queries(it + 1);
        __builtin_amdgcn_sched_barrier(0);      // the scheduler otherwise sinks every load of the iteration to its end
        const u32x4 *qs = (const u32x4 *)(ring + (it & 1) * (STAGE_TILES * 64)) + lane;
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            constexpr int dummy = 0;
            (void)dummy;
            const int jn = (j + 1) % PF;                                    // slot of chunk c+1
            const bool has_next = !(TAIL == 2 && j == PF - 1);
            const bool refill = TAIL == 0 || (TAIL == 1 && j < PF - 1);     // chunk c+1+PF exists
            typename MM::Db dnext[R];
            if (has_next) {
                if constexpr (ABL == 1) {                                   // timing only: raw bits as pieces
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        dnext[r].h = __builtin_bit_cast(u32x4, raw[jn][r][0]);
                        dnext[r].m = __builtin_bit_cast(u32x4, raw[jn][r][TK - 1]);
                        dnext[r].l = dnext[r].h ^ dnext[r].m;
                    }
                } else if constexpr (ABL == 3 || ABL == 4 || ABL == 5) {                // timing only: the stream and the barriers (4: not even those)
#pragma unroll
                    for (int r = 0; r < R; ++r) acc[r][0] += raw[jn][r][0] + raw[jn][r][TK - 1];
                } else {
#pragma unroll
                    for (int r = 0; r < R; ++r) MM::prepare(raw[jn][r], dnext[r]);  // waits (counted vmcnt) for chunk c+1's loads only
                }
            }
            if constexpr (ABL != 3 && ABL != 4 && ABL != 5) {
#pragma unroll
                for (int q = 0; q < QT; ++q) {
                    u32x4 qp[NQP];
#pragma unroll
                    for (int p = 0; p < NQP; ++p) qp[p] = qs[((j * NQP + p) * QT + q) * 64];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        acc[r][q] = MM::mma(qp, dcur[r], acc[r][q]);
                        if constexpr (TWO) acx[r][q] = MM::mmx(qp, dcur[r], acx[r][q]);      // split2: 1.54 ms against 1.55 for the ring kernel: parked too
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (refill) fetch(jn, (it + 1) * PF + j + 1);                   // the slot is free again: refill it
            __builtin_amdgcn_sched_barrier(0);                              // ... here: pinned (see above)
            if (has_next) {
#pragma unroll
                for (int r = 0; r < R; ++r) dcur[r] = dnext[r];
            }
        }
        if constexpr (TAIL < 2 && ABL != 4 && ABL != 5) store_queries(it + 1);
    };
    for (int it = 0; it + 2 < NIT; ++it) body(it, std::integral_constant<int, 0>{});
    body(NIT - 2, std::integral_constant<int, 1>{});
    body(NIT - 1, std::integral_constant<int, 2>{});

    // Epilogue (as the ring kernels): transpose the accumulators through LDS so that every query row of the workgroup's
    // rows leaves as one contiguous run
    constexpr int ROWS = CW * R * TILE_ROWS;
    constexpr int LDW = ROWS + 4;
    if constexpr (ABL == 5) {                   // timing only: no epilogue either (one store per lane keeps the sums alive)
        out[(int64_t)(tid % 70) * n + rt_wg * TILE_ROWS + tid / 70] = acc[0][0][0] + acc[R - 1][0][1];
        return;
    }
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
                    stage[(q * 16 + qrow + i) * LDW + (wave * R + r) * TILE_ROWS + col] = TWO ? acc[r][q][i] + acx[r][q][i] * (1.0f / 2048.0f) : acc[r][q][i];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int64_t row0 = rt_wg * TILE_ROWS;
    const int rows_valid = (int)((n - row0) < ROWS ? (n - row0) : ROWS);
    const int left = nq_valid - qt0 * TILE_ROWS;
    const int nq_here = left < QT * TILE_ROWS ? left : QT * TILE_ROWS;
    for (int e = tid; e < nq_here * ROWS; e += CW * 64) {
        const int qi = e / ROWS, rr = e % ROWS;
        if (rr < rows_valid) out[(int64_t)qi * n + row0 + rr] = stage[qi * LDW + rr];
    }
}

}  // namespace mdx
