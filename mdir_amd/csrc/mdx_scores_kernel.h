// The similarity kernel of libmdx.so: ONE form, no compile-time forks.  The timing-only variants that earlier rounds measured
// (no LDS-DMA beside the MFMAs, no ds_read, no epilogue, the same rows from the L2, the blocked shard order) are a patch on this
// file -- tools/ablate/scores_kernel_ablate.patch, regenerated from these sources by tools/ablate/make_patch.py and applied to a
// scratch copy by tools/scores_where.sh; round 6's two-loader-wave experiment is tools/ablate/scores_lw2.patch.
#pragma once
#include "mdx_common.h"

namespace mdx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TILE_ROWS = 16;   // rows per tile  (MFMA N / M)
constexpr int TILE_K = 16;      // k per tile     (4 MFMA k-steps of 4)
constexpr int MAX_QT = 8;       // query tiles (of 16) per launch

// Position (in KiB tiles) of tile (row tile rt, k-block kb) of a shard: row-tile-major.  A workgroup's chunk is then
// 8-16 pieces of KC KiB at a KB-KiB stride.  The alternative -- blocks of 16 row tiles with the k-block as the slow
// index, so that a chunk is ONE contiguous 32-KiB run -- was measured and is SLOWER (a pure stream of the shard
// 1.76 ms against 1.52 ms at 1 M x 2048: the strided pieces spread over more HBM channels at any instant).
__host__ __device__ __forceinline__ int64_t shard_tile(int64_t rt, int64_t kb, int64_t KB) { return rt * KB + kb; }

// Workgroup -> row block.  Consecutive workgroup ids go to different XCDs (id mod 8), each with its own L2; a workgroup's
// output is one run of 512-1 024 B per query row, 4 MB apart.  With block = id the runs that one L2 collects at about the
// same time are 8 blocks apart; giving each XCD a CONTIGUOUS range of row blocks makes them neighbours, so that they leave the
// L2 as longer runs (the same trick as the sort's tile order).
// sixteen zero bytes in device memory: what the row-major loader reads for k >= d (a padded chunk multiplied real row values
// by the zero query tiles before: 0 * Inf = NaN where np.dot has Inf -- ADVICE round 4)
__device__ __attribute__((aligned(16))) float mdx_zero16[4] = {0.f, 0.f, 0.f, 0.f};
__device__ __forceinline__ int64_t row_block_of(unsigned id, unsigned nblocks)
{
    const unsigned per = nblocks / 8, rem = nblocks % 8;          // XCD x takes `per` blocks, the first `rem` XCDs one more
    const unsigned x = id % 8, k = id / 8;
    return (int64_t)x * per + (x < rem ? x : rem) + k;
}

// A score leaves the kernel once and is next read by another kernel (the ranking).  NT: a non-temporal store, which does
// not allocate in the caches on its way out.  Measured (round 4, NT on / off, several processes each): the
// HBM-bound split-precision ring kernel 1.60-1.64 -> 1.39-1.46 ms with it -- its query pieces are re-read from the L2 by every
// workgroup, and 281 MB of output passing through the same caches pushes them out; the power-bound three-piece form, the exact
// chain (MFMA-bound) and the fp16 kernels: no change or 1 % worse.  So only the split kernels ask for it.
template <bool NT>
__device__ __forceinline__ void store_score(float *p, float v)
{
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

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
// In-kernel stamps on the single-role kernel of round 1 showed where it lost
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
//
// Leftover queries (QR = 1, fp32 shards with R = 2 only): 70 queries are 4 full MFMA tiles + 6.  A fifth
// 16-query tile would be 62 % padding (12.5 % of all MFMAs of the launch).  Instead the last query tile
// (<= 8 valid queries, stored in the same 1-KiB fragment format) is multiplied with
// v_mfma_f32_4x4x1_16B_f32: 16 independent 4x4 outer products per instruction = 2 groups of 4 queries x
// 8 groups of 4 rows = the wave's 32 database rows x 8 queries, ONE k per instruction (8 cycles against
// the 2 x 32 of the padded tile's two 16x16x4 MFMAs per 4 k).  A lane reads the 16 k of "its" query row
// and of "its" database row with four ds_read_b128 each (the four lane groups g of the tile format hold
// k = 4t+g) and issues the products in k order, so every output is still the k = 0..D-1 fma chain.
//
// RM = true (mdx_scores_rowmajor: a database that is multiplied ONCE, read where it lies): `db` is the caller's row-major
// fp32 matrix [n, ld] instead of tiles.  A loader lane (c, j) = (lane >> 4, lane & 15) fetches the 16 bytes of row
// 16 rt + j at k = 16 kb + 4 c, so a tile's KiB in LDS holds element w of lane (c, j) = k 4 c + w -- the tile format with the
// roles of lane group and element swapped.  The consumers put "their" operand together from four 4-byte reads 64 lanes
// apart (two ds_read2st64_b32; 64 different banks): lane (g, j), element t = k 4 t + g, as before -- same MFMAs, same k order,
// same bits.  Rows >= n read row n-1 (never stored), a piece past k = ld reads mdx_zero16 (zeros, as the tiled shard holds there).
//
// PIPE = true (fp32 tiles, KC = 2; round 5, the f64 GEMM's schedule of mdx_gram.hip): the consumer keeps TWO register sets of
// operands.  The operands of k-block 1 are read while k-block 0 multiplies; the stage hand-over -- lgkmcnt(0), the raw
// s_barrier B_{c+1}, the reads of the next stage's k-block 0 -- sits in front of the LAST step (k 12-15) of a stage's last
// k-block: the 8 + 8 MFMAs that follow need nothing from the new stage, so the matrix pipe runs through the barrier and through
// the LDS round trip that otherwise opens every chunk.  Same MFMAs, same k order per output: same bits.
//
// ROUTED = true (mdx_scores_p2p: the direct-store exchange of a row-sharded database, round 6): query q's run of scores does not go
// to out + q * n but to route[q] + col0 -- route[q] = row (q - first query of its owner) of the OWNER rank's receive buffer, mapped
// into this process (hipIpc), col0 = the shard's first global row.  The stores are system-scope (write-through: sc0 sc1), so that
// nothing of them lingers in this XCD's L2 when the kernel ends and the step's flag goes out.  Same accumulators, same bits.
template <int QT, int R, int KC, int NSTAGE, int DB_AUX = 0, typename MM = MmaF32, int QR = 0, int CWAVES = 4, bool RM = false,
          int PIPE = 0, int LWAVES = 4, bool ROUTED = false>
__global__ __launch_bounds__((CWAVES + LWAVES) * 64, ((R >= 4 || CWAVES > 4) ? 1 : 2)) void scores_lc_kernel(const f32x4 *__restrict__ db,
                                                           const f32x4 *__restrict__ qtiles,
                                                           float *__restrict__ out, int64_t n, int KB,
                                                           int nq_valid, int64_t ld = 0, float *const *__restrict__ route = nullptr,
                                                           int64_t col0 = 0)
{
    static_assert(!RM || (MM::STEPS == 4 && KC == 2), "row-major databases: fp32, two k-blocks per stage");
    constexpr int CW = CWAVES;                      // consumer waves (8: two per SIMD in ONE workgroup per CU, sharing the query stage)
    constexpr int LW = LWAVES;                      // loader waves
    static_assert(QR == 0 || (R == 2 && MM::STEPS == 4), "the 4x4x1 leftover path covers 32 fp32 rows per wave");
    constexpr int QTL = QT + QR;                    // query tiles in LDS: QT full ones + the leftover tile
    constexpr int QTILES = QTL * KC;                // KiB tiles of queries per stage
    constexpr int BTILES = CW * R * KC;             // KiB tiles of database per stage
    constexpr int STAGE_TILES = QTILES + BTILES;
    constexpr int PER_LOADER = (STAGE_TILES + LW - 1) / LW;   // uneven split: the last tile is loaded twice
    static_assert((NSTAGE - 1) * PER_LOADER <= 63, "vmcnt is 6 bits");
    extern __shared__ __attribute__((aligned(16))) f32x4 ring[];   // [NSTAGE][STAGE_TILES][64]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nchunks = KB / KC;
    const int64_t rt_wg = row_block_of(blockIdx.x, gridDim.x) * CW * R;       // first row tile of the workgroup
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
        int rm_k[PER_LOADER];            // RM: first k of this lane's 16 bytes inside a stage
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
                if constexpr (RM) {
                    const int64_t row = (rt_wg + tile) * TILE_ROWS + (lane & 15);
                    src[t] = (const f32x4 *)((const float *)db + (row < n ? row : n - 1) * ld);     // the row; k is added per stage
                    rm_k[t] = kbc * TILE_K + 4 * (lane >> 4);
                } else {
                    src[t] = db + shard_tile(rt_wg + tile, kbc, KB) * 64 + lane;
                }
            }
        }
        auto issue = [&](int c) {
            f32x4 *slot = ring + (c % NSTAGE) * (STAGE_TILES * 64);
#pragma unroll
            for (int t = 0; t < PER_LOADER; ++t) {
                // query tiles (re-read by every workgroup) keep the default cache policy and are stored densely;
                // the database stream may be marked non-temporal (DB_AUX = 2) and advances in the shard's order
                const bool is_db = (lw + t * LW) >= QTILES;
                const f32x4 *p = src[t] + (is_db ? shard_tile(0, (int64_t)c * KC, KB) : (int64_t)c * KC) * 64;
                if constexpr (RM) {
                    if (is_db) {
                        const int64_t k = (int64_t)c * KC * TILE_K + rm_k[t];
                        p = k + 4 <= ld ? (const f32x4 *)((const float *)src[t] + k) : (const f32x4 *)mdx_zero16;
                    }
                }
                if (DB_AUX != 0 && is_db)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p,
                                                     (__attribute__((address_space(3))) void *)(slot + dst[t]), 16, 0, DB_AUX);
                else
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p,
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
    f32x4 acc[R][QT > 0 ? QT : 1];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int q = 0; q < QT; ++q) acc[r][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // leftover path: lane = 4*block + i; block = 2*rg + qg: query 4*qg + i of the leftover tile (A operand),
    // database row 4*rg + i of the wave's 32 rows (B operand); D register v = query 4*qg + v, same row
    f32x4 accl = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int l_q = 4 * ((lane >> 2) & 1) + (lane & 3);         // query row inside the leftover tile
    const int l_row = 4 * (lane >> 3) + (lane & 3);             // row inside the wave's 32 rows
    const int l_boff = (l_row >> 4) * KC * 64 + (l_row & 15);   // f32x4 offset of (tile, row) inside the wave's tiles

    if constexpr (PIPE == 1) {
        static_assert(MM::STEPS == 4 && (KC == 2 || KC == 1) && !RM, "pipelined consumer: fp32 tiles, one or two k-blocks per stage");
        // Registers: a full second operand set does not fit beside the accumulators and the leftover operands (128 registers per
        // wave at two workgroups per CU), so only what a block's FIRST MFMAs need is read ahead -- its database operands bn[] and
        // query tile 0 (an0); query tiles 1.. are read into a[1..] during the previous block's last step, each right after the
        // last MFMA that used the register's old contents (that step multiplies tile 0 LAST, the next block's first step tile 0
        // FIRST: every read has >= 8 MFMAs = 256 cycles to come back).
        f32x4 a[QT > 0 ? QT : 1], b[R], bn[R], an0;
        f32x4 al[4], bl[4];
        constexpr int PIN = 0x0002 | 0x0004 | 0x0070 | 0x0380 | 0x0400;
        auto read_first = [&](const f32x4 *slot, int kb) __attribute__((always_inline)) {
            const f32x4 *bs = slot + (QTILES + wave * R * KC) * 64 + lane;
#pragma unroll
            for (int r = 0; r < R; ++r) bn[r] = bs[(r * KC + kb) * 64];
            an0 = slot[kb * 64 + lane];
        };
        auto read_left = [&](const f32x4 *slot, int kb) __attribute__((always_inline)) {
            if constexpr (QR != 0) {
                const f32x4 *ql = slot + (QT * KC + kb) * 64 + l_q;
                const f32x4 *bw = slot + (QTILES + wave * R * KC + kb) * 64 + l_boff;
#pragma unroll
                for (int g = 0; g < 4; ++g) { al[g] = ql[16 * g]; bl[g] = bw[16 * g]; }
            }
        };
        // One k-block.  `late` (in front of the last step): at a stage's end the hand-over, else nothing; then `nslot`/`nkb` = where
        // the NEXT block's operands lie (nullptr: no next block).  `first_here`: the next block's bn / an0 are read after step 0
        // (same stage: its tiles are there already) instead of inside `late`.
        auto block = [&](const f32x4 *nslot, int nkb, bool first_here, auto late) __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < R; ++r) b[r] = bn[r];
            a[0] = an0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t == 3) { __builtin_amdgcn_sched_barrier(0); late(); __builtin_amdgcn_sched_barrier(0); }
                const int n_small = QR == 0 ? 0 : (t == 0 ? 0 : (t == 3 ? 8 : 4));
                int done = 0, issued = 0;
#pragma unroll
                for (int qq = 0; qq < QT; ++qq) {
                    const int q = t == 3 ? (qq + 1) % QT : qq;          // last step: tile 0 last
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        acc[r][q] = MM::step(t, a[q], b[r], acc[r][q]);
                        __builtin_amdgcn_sched_barrier(PIN);
                        ++issued;
                        if constexpr (QR != 0) {
                            const int due = (issued * n_small) / (R * QT);
#pragma unroll
                            for (; done < due; ++done) {
                                const int st = (t == 3 && done >= 4) ? 3 : t - 1, g = done & 3;
                                accl = __builtin_amdgcn_mfma_f32_4x4x1f32(al[g][st], bl[g][st], accl, 0, 0, 0);
                                __builtin_amdgcn_sched_barrier(PIN);
                            }
                        }
                    }
                    if (t == 3 && q != 0 && nslot) {        // a[q] is dead: the next block's tile q
                        __builtin_amdgcn_sched_barrier(0);
                        a[q] = nslot[(q * KC + nkb) * 64 + lane];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (t == 0 && first_here) { __builtin_amdgcn_sched_barrier(0); read_first(nslot, nkb); __builtin_amdgcn_sched_barrier(0); }
            }
        };
        __builtin_amdgcn_s_barrier();                                       // B_0
        read_first(ring, 0);
#pragma unroll
        for (int q = 1; q < QT; ++q) a[q] = ring[(q * KC) * 64 + lane];
        for (int c = 0; c < nchunks; ++c) {
            const f32x4 *slot = ring + (c % NSTAGE) * (STAGE_TILES * 64);
            const f32x4 *next = ring + ((c + 1) % NSTAGE) * (STAGE_TILES * 64);
            const bool more = c + 1 < nchunks;
            if constexpr (KC == 2) {
                read_left(slot, 0);
                block(slot, 1, true, []() {});
            }
            read_left(slot, KC - 1);
            block(more ? next : nullptr, 0, false, [&]() __attribute__((always_inline)) {
                if (more) {
                    __builtin_amdgcn_s_waitcnt(0xC07F);                     // lgkmcnt(0): every read of this stage is back
                    __builtin_amdgcn_s_barrier();                           // B_{c+1}: the next stage has landed; this one may be refilled
                    read_first(next, 0);
                }
            });
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else
    for (int c = 0; c < nchunks; ++c) {
        __builtin_amdgcn_s_barrier();                                       // B_c
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 *slot = ring + (c % NSTAGE) * (STAGE_TILES * 64);
        const f32x4 *qs = slot + lane;
        const f32x4 *bs = slot + (QTILES + wave * R * KC) * 64 + lane;
#pragma unroll
        for (int kb = 0; kb < KC; ++kb) {
            f32x4 a[QT > 0 ? QT : 1], b[R];
#pragma unroll
            for (int q = 0; q < QT; ++q) a[q] = qs[(q * KC + kb) * 64];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if constexpr (RM) {
                    const float *bt = (const float *)(slot + (QTILES + (wave * R + r) * KC + kb) * 64) + 4 * (lane & 15) + (lane >> 4);
#pragma unroll
                    for (int t = 0; t < 4; ++t) b[r][t] = bt[t * 64];
                } else {
                    b[r] = bs[(r * KC + kb) * 64];
                }
            }
            if constexpr (QR == 0) {
#pragma unroll
                for (int t = 0; t < MM::STEPS; ++t)
#pragma unroll
                    for (int r = 0; r < R; ++r)
#pragma unroll
                        for (int q = 0; q < QT; ++q) acc[r][q] = MM::step(t, a[q], b[r], acc[r][q]);
            } else {
                f32x4 al[4], bl[4];         // [g]: element t = k 4t+g of the lane's query row / database row
                const f32x4 *ql = slot + (QT * KC + kb) * 64 + l_q;
                const f32x4 *bw = slot + (QTILES + wave * R * KC + kb) * 64 + l_boff;
#pragma unroll
                for (int g = 0; g < 4; ++g) { al[g] = ql[16 * g]; bl[g] = bw[16 * g]; }
                // MFMA order is pinned (LDS reads, VALU, SALU may still move): the one-k products of step t ride
                // between the 16x16x4 MFMAs of step t+1, so that their operands (read at the top of the k-block)
                // have landed long before; a small MFMA costs 8-9 cycles there, a dependent run of them 12.5 each
                constexpr int PIN = 0x0002 | 0x0004 | 0x0070 | 0x0380 | 0x0400;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int n_small = t == 0 ? 0 : (t == 3 ? 8 : 4);
                    int done = 0;
#pragma unroll
                    for (int r = 0; r < R; ++r)
#pragma unroll
                        for (int q = 0; q < QT; ++q) {
                            acc[r][q] = MM::step(t, a[q], b[r], acc[r][q]);
                            __builtin_amdgcn_sched_barrier(PIN);
                            const int due = ((r * QT + q + 1) * n_small) / (R * QT);
#pragma unroll
                            for (; done < due; ++done) {
                                const int st = (t == 3 && done >= 4) ? 3 : t - 1, g = done & 3;
                                accl = __builtin_amdgcn_mfma_f32_4x4x1f32(al[g][st], RM ? bl[st][g] : bl[g][st], accl, 0, 0, 0);
                                __builtin_amdgcn_sched_barrier(PIN);
                            }
                        }
                }
            }
        }
        // all LDS reads of this stage are consumed by the MFMAs above before the next barrier
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }

    // Epilogue: the ring is free now (loaders have left, consumers are past their last read);
    // transpose the accumulators through LDS so that every query row of the workgroup's
    // 64*R database rows leaves as one contiguous run (full cache lines instead of 64-B pieces).
    constexpr int ROWS = CW * R * TILE_ROWS;        // database rows per workgroup
    constexpr int LDW = ROWS + 4;                   // +4: the four 16-lane groups hit different banks
    static_assert((QT * 16 + QR * 8) * LDW * 4 <= NSTAGE * STAGE_TILES * 1024, "output staging must fit in the ring");
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
        if constexpr (QR != 0) {
#pragma unroll
            for (int v = 0; v < 4; ++v)
                stage[(QT * 16 + 4 * ((lane >> 2) & 1) + v) * LDW + wave * R * TILE_ROWS + l_row] = accl[v];
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int64_t row0 = rt_wg * TILE_ROWS;
    const int rows_valid = (int)((n - row0) < ROWS ? (n - row0) : ROWS);
    // queries of this group that exist (the last group of a launch may be partial)
    constexpr int QCAP = QT * TILE_ROWS + QR * 8;   // query rows staged (grid.y > 1 only with QR = 0)
    const int nq_here = (nq_valid - (int)blockIdx.y * QT * TILE_ROWS) < QCAP ? (nq_valid - (int)blockIdx.y * QT * TILE_ROWS) : QCAP;
    static_assert(ROWS % 64 == 0, "a wave's stores of one step belong to one query");
    for (int e = tid; e < nq_here * ROWS; e += CW * 64) {
        const int qi = e / ROWS, rr = e % ROWS;
        if constexpr (ROUTED) {
            // (qi is the same in all lanes of a wave: the row pointer is a scalar load)
            float *dst = route[__builtin_amdgcn_readfirstlane((int)blockIdx.y * QT * TILE_ROWS + qi)] + col0 + row0 + rr;
            if (rr < rows_valid) __hip_atomic_store(dst, stage[qi * LDW + rr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
            if (rr < rows_valid) store_score<false>(out + (int64_t)qi * n + row0 + rr, stage[qi * LDW + rr]);
        }
    }
}

}  // namespace mdx
