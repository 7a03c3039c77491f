// PARKED (round 4): the PERSISTENT form of the register-streaming fp16 kernel -- built, bit-identical, NOT faster than the
// one-block-per-workgroup form that ships (mdir_amd/csrc/mdx_scores_stream_kernel.h): 0.79 against 0.76 ms at 1 M x 2048
// (768 workgroups x 10.2 row blocks leave a 7 % tail; prefetching across the epilogue gains nothing because what the
// epilogue costs is its scattered stores, not a stopped stream).  Kept for tools/split_ablate.hip.
//
// Persistent register-streaming similarity for an HBM-BOUND shard (the fp16 shard, MDX_F16 -- BASELINE.json configs[4]).
//
// Why another shape (numbers: profiles/r04_split3.md, "The fp16 shard and the epilogue").  With one MFMA per tile pair the
// ring kernel (mdx_scores_kernel.h) is nothing but a stream: 4.12 GB of shard in, 0.28 GB of scores out.  A plain stream of
// that shard reads at 6.0-6.9 TB/s; the ring kernel gets 5.1, and the whole distance is its EPILOGUE: every workgroup ends
// with an LDS transpose + its stores, all workgroups of a round end together, and chip-wide the stream stops while the stores
// drain (0.115 ms of a 0.85 ms launch; three workgroups per CU do not hide it).  Here the stream never stops:
//   * persistent workgroups (a few per CU) walk row blocks  b = blockIdx.x, + gridDim.x, ...;
//   * every wave streams ITS OWN row tiles global -> VGPR, PF = 4 chunks of 32 k ahead, refilling a register slot the moment
//     it has consumed it -- and the refill simply runs on into the wave's row tiles of the NEXT block, so when a block's
//     accumulators are stored its successor's first four chunks are already in flight;
//   * the query tiles of the next four chunks go into a two-stage LDS ring (plain loads + ds_write_b128 by all waves, one
//     raw s_barrier per four chunks); the query stream is periodic, so "the stage after the block's last" is stage 0 again;
//   * the epilogue stages ONE query tile at a time through a small LDS buffer of its own (it cannot borrow the ring: the ring
//     already holds the next block's first stage).
// One code path: the loop body is the same in every iteration (the pointer of a refill is a scalar select between this block
// and the next), so the compiler's counted vmcnt waits stay exact -- see the ISA traps in profiles/r04_split3.md.
// Operand formats, MFMA order and accumulators are the ring kernel's: results are bit-identical to it.
#pragma once
#include "mdx_scores_split_kernel.h"

namespace mdx {

struct StreamF16 {                      // fp16 shard (MDX_F16): tile = 16 rows x 32 k, one v_mfma_f32_16x16x32_f16 per tile pair
    static constexpr int TK = 1;        // KiB of shard per row tile and chunk of 32 k
    static constexpr int NQP = 1;       // query piece arrays
    static constexpr bool PIPE = false; // operands are used as loaded: nothing to prepare a chunk ahead
    struct Db { f32x4 v; };
    static __device__ __forceinline__ void prepare(const f32x4 (&raw)[TK], Db &d) { d.v = raw[0]; }
    static __device__ __forceinline__ f32x4 mma(const u32x4 (&q)[NQP], const Db &d, f32x4 a)
    {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, q[0]), __builtin_bit_cast(f16x8, d.v), a, 0, 0, 0);
    }
};

struct StreamSplit3 {                   // fp32 shard, three bf16 pieces per operand (mdx_scores_split_kernel.h); for the ablation harness
    static constexpr int TK = 2;
    static constexpr int NQP = 3;
    static constexpr bool PIPE = true;  // the split of chunk c+1 runs under the MFMAs of chunk c
    struct Db { u32x4 h, m, l; };
    static __device__ __forceinline__ void prepare(const f32x4 (&raw)[TK], Db &d) { split3(raw[0], raw[1], d.h, d.m, d.l); }
    static __device__ __forceinline__ f32x4 mma(const u32x4 (&q)[NQP], const Db &d, f32x4 a)
    {
        a = mfma_bf16(q[2], d.h, a);
        a = mfma_bf16(q[0], d.l, a);
        a = mfma_bf16(q[1], d.m, a);
        a = mfma_bf16(q[1], d.h, a);
        a = mfma_bf16(q[0], d.m, a);
        return mfma_bf16(q[0], d.h, a);
    }
};

constexpr int STREAMP_PF = 4;            // chunks of 32 k a wave keeps in flight = chunks per query stage

template <int QT, int R, int CW, typename MM>
constexpr int stream_persistent_lds_bytes()
{
    return 2 * STREAMP_PF * MM::NQP * QT * 1024 + TILE_ROWS * (CW * R * TILE_ROWS + 4) * 4;     // query ring + one query tile of output
}

// db: the shard's tiles; KB = KiB tiles per row tile; qpieces: [piece][QT_total][NC] KiB tiles, NC = KB / TK chunks of 32 k.
// NC must be a multiple of STREAMP_PF and >= 2 * STREAMP_PF.  nblocks = row blocks of CW * R row tiles; gridDim.x <= nblocks.
template <int QT, int R, int CW, typename MM, int WGS>
__global__ __launch_bounds__(CW * 64, WGS) void scores_stream_persistent_kernel(const f32x4 *__restrict__ db, const u32x4 *__restrict__ qpieces,
                                                                     float *__restrict__ out, int64_t n, int KB, int QT_total, int qt_first,
                                                                     int nq_valid, int nblocks)
{
    constexpr int PF = STREAMP_PF, TK = MM::TK, NQP = MM::NQP;
    constexpr int STAGE_TILES = PF * NQP * QT;                      // [chunk of the stage][piece][query tile]
    constexpr int PER_WAVE = (STAGE_TILES + CW - 1) / CW;           // query tiles of a stage this wave brings in (uneven: the last tile again)
    constexpr int ROWS = CW * R * TILE_ROWS, LDW = ROWS + 4;
    extern __shared__ __attribute__((aligned(16))) f32x4 ring[];   // [2][STAGE_TILES][64], then the output staging [16][LDW] floats
    float *ostage = (float *)(ring + 2 * STAGE_TILES * 64);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NC = KB / TK, NIT = NC / PF;
    const int qt0 = qt_first + (int)blockIdx.y * QT;
    out += (int64_t)qt0 * TILE_ROWS * n;
    const int left = nq_valid - qt0 * TILE_ROWS;
    const int nq_here = left < QT * TILE_ROWS ? left : QT * TILE_ROWS;

    // query stage s (chunks PF*s .. PF*s+PF-1) -> ring slot: this wave's tiles, through registers
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
    auto load_queries = [&](int s) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < PER_WAVE; ++t) qreg[t] = qsrc[t][(int64_t)s * PF * 64];
    };
    auto store_queries = [&](int par) __attribute__((always_inline)) {
        f32x4 *slot = ring + par * (STAGE_TILES * 64);
#pragma unroll
        for (int t = 0; t < PER_WAVE; ++t) slot[qdst[t]] = qreg[t];
    };

    f32x4 acc[R][QT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int q = 0; q < QT; ++q) acc[r][q] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // this wave's row tiles in block b; the last block of a workgroup "continues" into itself (a few KiB read twice, never used)
    auto tiles_of = [&](int b) __attribute__((always_inline)) { return db + ((int64_t)b * CW * R + wave * R) * (int64_t)KB * 64 + lane; };
    // walk = blockIdx.x, + gridDim.x, ... through row_block_of: an XCD's workgroups walk ONE contiguous range of row blocks
    int walk = (int)blockIdx.x;
    int blk = (int)row_block_of((unsigned)walk, (unsigned)nblocks);
    int nxt = walk + (int)gridDim.x < nblocks ? (int)row_block_of((unsigned)(walk + (int)gridDim.x), (unsigned)nblocks) : blk;
    const f32x4 *cur_p = tiles_of(blk), *nxt_p = tiles_of(nxt);
    f32x4 raw[PF][R][TK];
    auto fetch = [&](int j, const f32x4 *base, int c) __attribute__((always_inline)) {     // chunk c of the row tiles at `base` -> register slot j
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int t = 0; t < TK; ++t) raw[j][r][t] = __builtin_nontemporal_load(base + ((int64_t)r * KB + c * TK + t) * 64);
    };
    // The prologue issues in the loop's order, pinned (the compiler derives its counted waits from the issue order it sees
    // on every way into the loop)
    load_queries(0);                            // first, so that the wait for them leaves the shard loads below in flight
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < PF; ++j) {
        fetch(j, cur_p, j);
        __builtin_amdgcn_sched_barrier(0);
    }
    store_queries(0);
    __builtin_amdgcn_sched_barrier(0);
    typename MM::Db dcur[R];
    if constexpr (MM::PIPE) {                   // software pipeline: chunk c+1 is prepared while chunk c is multiplied
#pragma unroll
        for (int r = 0; r < R; ++r) MM::prepare(raw[0][r], dcur[r]);
        fetch(0, cur_p, PF);
        __builtin_amdgcn_sched_barrier(0);
    }

    int par = 0;                                // ring slot of the stage the next iteration multiplies
    for (;;) {
        for (int it = 0; it < NIT; ++it) {
            // B: every wave has written its part of this iteration's stage (and waited for the writes), and every wave has
            // left the previous stage, whose slot this iteration's writes go to
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            load_queries(it + 1 < NIT ? it + 1 : 0);            // the query stream is periodic: after the block's last stage, stage 0
            __builtin_amdgcn_sched_barrier(0);
            const u32x4 *qs = (const u32x4 *)(ring + par * (STAGE_TILES * 64)) + lane;
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                if constexpr (MM::PIPE) {
                    const int jn = (j + 1) % PF;                // register slot of the chunk after this one
                    typename MM::Db dnext[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) MM::prepare(raw[jn][r], dnext[r]);     // waits (counted vmcnt) for that chunk's loads only
#pragma unroll
                    for (int q = 0; q < QT; ++q) {
                        u32x4 qp[NQP];
#pragma unroll
                        for (int p = 0; p < NQP; ++p) qp[p] = qs[((j * NQP + p) * QT + q) * 64];
#pragma unroll
                        for (int r = 0; r < R; ++r) acc[r][q] = MM::mma(qp, dcur[r], acc[r][q]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // slot jn is free again: refill it with the chunk PF + 1 ahead -- of this block, or of the next one
                    const int cr = it * PF + j + 1 + PF;
                    const bool here = cr < NC;
                    fetch(jn, here ? cur_p : nxt_p, here ? cr : cr - NC);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < R; ++r) dcur[r] = dnext[r];
                } else {
                    typename MM::Db d[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) MM::prepare(raw[j][r], d[r]);          // waits (counted vmcnt) for this chunk's loads only
#pragma unroll
                    for (int q = 0; q < QT; ++q) {
                        u32x4 qp[NQP];
#pragma unroll
                        for (int p = 0; p < NQP; ++p) qp[p] = qs[((j * NQP + p) * QT + q) * 64];
#pragma unroll
                        for (int r = 0; r < R; ++r) acc[r][q] = MM::mma(qp, d[r], acc[r][q]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // the slot's MFMAs are issued: refill it with the chunk PF ahead -- of this block, or of the next one
                    const int cr = it * PF + j + PF;
                    const bool here = cr < NC;
                    fetch(j, here ? cur_p : nxt_p, here ? cr : cr - NC);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            store_queries(par ^ 1);
            par ^= 1;
        }
        // ---- the block's accumulators are complete; its successor's first chunks are in flight and its first query stage is
        // on its way into the ring.  One query tile at a time: accumulators -> LDS (transposed) -> contiguous runs per query row.
        const int64_t row0 = (int64_t)blk * ROWS;
        const int rows_valid = (int)((n - row0) < ROWS ? (n - row0) : ROWS);
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            __builtin_amdgcn_s_barrier();                       // the previous tile's reads of the staging buffer are done
            {
                const int qrow = 4 * (lane >> 4), col = lane & 15;
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int i = 0; i < 4; ++i) ostage[(qrow + i) * LDW + (wave * R + r) * TILE_ROWS + col] = acc[r][q][i];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int nq_tile = (nq_here - q * TILE_ROWS) < TILE_ROWS ? (nq_here - q * TILE_ROWS) : TILE_ROWS;
            for (int e = tid; e < nq_tile * ROWS; e += CW * 64) {
                const int qi = e / ROWS, rr = e % ROWS;
                if (rr < rows_valid) out[(int64_t)(q * TILE_ROWS + qi) * n + row0 + rr] = ostage[qi * LDW + rr];
            }
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (nxt == blk) break;                                  // that was this workgroup's last block
        walk += (int)gridDim.x;
        blk = nxt;
        nxt = walk + (int)gridDim.x < nblocks ? (int)row_block_of((unsigned)(walk + (int)gridDim.x), (unsigned)nblocks) : blk;
        cur_p = nxt_p;
        nxt_p = tiles_of(nxt);
    }
}

}  // namespace mdx
