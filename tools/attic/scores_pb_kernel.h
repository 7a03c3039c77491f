// NOT SHIPPED (tools/attic): persistent form of the loader/consumer similarity kernel, measured 3-5 % SLOWER than the
// per-row-block kernel of mdir_amd/csrc/mdx_scores_kernel.h (2.71-2.75 against 2.60-2.63 ms at 1 004 993 x 70 x 2048,
// tools/scores_ablate.hip); bit-identical results.  Kept so that the experiment can be repeated.
//
// Persistent form of the loader/consumer similarity kernel for big fp32 shards.
//
// Stamps on the per-row-block kernel of mdx_scores_kernel.h (tools/scores_ablate.hip, 1 004 993 x 70 x 2048)
// showed where its time outside the MFMA loop goes: a workgroup lives 167 us, of which 14 us are the epilogue
// (accumulators through the ring, 35 KiB of stores, and the wave cannot retire before its stores are
// acknowledged -- under an 8 GB read stream that takes microseconds), and the 512 workgroup slots are
// occupied 95 % of the launch (dispatch gaps + the last partial round of 7 852 workgroups).  Both workgroups
// of a CU start together and stay in phase, so their epilogues coincide and the MFMA pipes idle.
//
// Here the SAME workgroup (4 MFMA consumer waves x 2 row tiles + 4 LDS-DMA loader waves, one barrier per
// chunk, ring of NSTAGE stages; two workgroups per CU) stays for the whole launch:
//   * grid G = 2 x #CUs; workgroup w takes row blocks (8 tiles) w, w+G, w+2G, ... and an even share (<= 8 tiles)
//     of what is left after the last full round: no dispatch gaps, no partial last round (a block of <= 4
//     tiles costs half a block: its tiles are dealt one per consumer wave);
//   * the chunk sequence is continuous across row blocks: loaders keep streaming, the first chunks of
//     block b+1 land while block b is still being multiplied;
//   * the ring is never used as staging: at the end of a block a consumer stores its accumulators straight
//     from registers (64-B runs per query and tile; neighbouring waves complete each other's cache lines in
//     L2) and does not wait for the stores.
// Arithmetic is unchanged: every score is the k = 0..D-1 fma chain (oracle/chain.c); the leftover query
// tile (<= 8 queries, QR = 1) goes through v_mfma_f32_4x4x1 exactly as in scores_lc_kernel.
//
// (A first persistent design -- ONE workgroup per CU, 8 consumers + 4 loaders, LDS counters instead of the
// barrier; tools/attic/scores_p1_kernel.h -- was built and measured at 3.5 ms against 2.65: with a single
// ring per CU every memory hiccup stalls all eight consumers, while two independent workgroups cover for
// each other.  The shard stream alone takes 1.5-1.8 ms of the 2.1 ms the MFMAs need, so the two must
// overlap almost perfectly.)
#pragma once
#include "mdx_scores_kernel.h"

namespace mdx {

constexpr int PB_CW = 4;             // consumer waves
constexpr int PB_LW = 4;             // loader waves
constexpr int PB_R = 2;              // row tiles per consumer wave
constexpr int PB_BLOCK_TILES = PB_CW * PB_R;

template <int QT, int QR, int KC, int NSTAGE>
constexpr int pb_lds_bytes()
{
    return NSTAGE * ((QT + QR) + PB_BLOCK_TILES) * KC * 1024;
}

// MFMAs of one chunk (KC k-blocks) of one consumer wave: NR of its row tiles (tiles wave and wave+4 of the block)
// x QT query tiles (+ the leftover tile)
template <int NR, int QT, int QR, int KC>
__device__ __forceinline__ void pb_chunk(f32x4 (&acc)[PB_R][QT > 0 ? QT : 1], f32x4 &accl, const f32x4 *slot, int lane, int wave,
                                         int l_q, int l_boff)
{
    constexpr int QTILES = (QT + QR) * KC;
    const f32x4 *qs = slot + lane;
    const f32x4 *bs = slot + (QTILES + wave * KC) * 64 + lane;
#pragma unroll
    for (int kb = 0; kb < KC; ++kb) {
        f32x4 a[QT > 0 ? QT : 1], b[NR];
#pragma unroll
        for (int q = 0; q < QT; ++q) a[q] = qs[(q * KC + kb) * 64];
#pragma unroll
        for (int r = 0; r < NR; ++r) b[r] = bs[(r * PB_CW * KC + kb) * 64];
        if constexpr (QR == 0) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < NR; ++r)
#pragma unroll
                    for (int q = 0; q < QT; ++q) acc[r][q] = MmaF32::step(t, a[q], b[r], acc[r][q]);
        } else {
            f32x4 al[4], bl[4];
            const f32x4 *ql = slot + (QT * KC + kb) * 64 + l_q;
            const f32x4 *bw = slot + (QTILES + wave * KC + kb) * 64 + l_boff;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) { al[gq] = ql[16 * gq]; bl[gq] = bw[16 * gq]; }
            constexpr int PIN = 0x0002 | 0x0004 | 0x0070 | 0x0380 | 0x0400;     // MFMA order pinned, see scores_lc_kernel
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int n_small = t == 0 ? 0 : (t == 3 ? 8 : 4);
                int done = 0;
#pragma unroll
                for (int r = 0; r < NR; ++r)
#pragma unroll
                    for (int q = 0; q < QT; ++q) {
                        acc[r][q] = MmaF32::step(t, a[q], b[r], acc[r][q]);
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
        }
    }
}

template <int QT, int QR, int KC, int NSTAGE, int DB_AUX = 2>
__global__ __launch_bounds__((PB_CW + PB_LW) * 64, 2) void scores_pb_kernel(const f32x4 *__restrict__ db,
                                                                             const f32x4 *__restrict__ qtiles,
                                                                             float *__restrict__ out, int64_t n, int64_t RT,
                                                                             int KB, int nq_valid)
{
    constexpr int QTL = QT + QR;
    constexpr int QTILES = QTL * KC;
    constexpr int BTILES = PB_BLOCK_TILES * KC;
    constexpr int STAGE_TILES = QTILES + BTILES;
    constexpr int PER_LOADER = (STAGE_TILES + PB_LW - 1) / PB_LW;
    static_assert((NSTAGE - 1) * PER_LOADER <= 63, "vmcnt is 6 bits");
    extern __shared__ __attribute__((aligned(16))) f32x4 ring[];        // [NSTAGE][STAGE_TILES][64]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nchunks = KB / KC;
    // Row blocks of this workgroup: full blocks of 8 tiles are dealt round-robin (block r*G + w in round r), so that
    // at any time the G workgroups stream one moving window of the shard, as the per-block kernel's grid does;
    // what is left after the last full round (< 8*G + 8 tiles) is cut into G even pieces of <= 8 tiles.
    const int64_t G = gridDim.x, w_id = blockIdx.x;
    const int rounds = (int)((RT / PB_BLOCK_TILES) / G);
    const int64_t rem0 = (int64_t)rounds * G * PB_BLOCK_TILES, rem = RT - rem0;
    const int64_t last0 = rem0 + (rem * w_id) / G, last1 = rem0 + (rem * (w_id + 1)) / G;
    const int nblocks = rounds + (last1 > last0 ? 1 : 0);
    const int total = nblocks * nchunks;                                  // chunks of this workgroup
    auto block_range = [&](int blk, int64_t &tb, int &nt) {
        if (blk < rounds) { tb = ((int64_t)blk * G + w_id) * PB_BLOCK_TILES; nt = PB_BLOCK_TILES; }
        else { tb = last0; nt = (int)(last1 - last0); }
    };

    if (wave >= PB_CW) {
        // ------------------------------------------------------------------ loader
        const int lw = wave - PB_CW;
        const f32x4 *src[PER_LOADER];
        int dst[PER_LOADER];
        auto block_sources = [&](int blk) {         // tiles i = lw, lw+LW, ... of a stage (query tiles first) for row block blk
            int64_t tb;
            int nt;
            block_range(blk, tb, nt);
#pragma unroll
            for (int t = 0; t < PER_LOADER; ++t) {
                const int i = (lw + t * PB_LW) < STAGE_TILES ? (lw + t * PB_LW) : (STAGE_TILES - 1);
                dst[t] = i * 64;
                if (i < QTILES) {
                    const int qt = i / KC, kbc = i % KC;
                    src[t] = qtiles + ((int64_t)qt * KB + kbc) * 64 + lane;
                } else {
                    const int j = i - QTILES;
                    int tile = j / KC;
                    const int kbc = j % KC;
                    tile = tile < nt ? tile : nt - 1;                      // absent tiles of a partial block: any valid tile
                    src[t] = db + shard_tile(tb + tile, kbc, KB) * 64 + lane;
                }
            }
        };
        int ib = 0, ic = 0;                                              // (block, chunk) of the next chunk to issue
        block_sources(0);
        auto issue = [&](int g) {
            f32x4 *slot = ring + (g % NSTAGE) * (STAGE_TILES * 64);
#pragma unroll
            for (int t = 0; t < PER_LOADER; ++t) {
                const bool is_db = (lw + t * PB_LW) >= QTILES;
                const f32x4 *p = src[t] + (is_db ? shard_tile(0, (int64_t)ic * KC, KB) : (int64_t)ic * KC) * 64;
                if (is_db)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p,
                                                     (__attribute__((address_space(3))) void *)(slot + dst[t]), 16, 0, DB_AUX);
                else
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p,
                                                     (__attribute__((address_space(3))) void *)(slot + dst[t]), 16, 0, 0);
            }
            if (++ic == nchunks) {
                ic = 0;
                if (++ib < nblocks) block_sources(ib);
            }
        };
#pragma unroll
        for (int g = 0; g < NSTAGE - 1; ++g)
            if (g < total) issue(g);
        for (int g = 0; g < total; ++g) {
            // stage g must have landed: everything but the younger stages g+1 .. g+NSTAGE-2
            const int younger = (total - 1 - g) < (NSTAGE - 2) ? (total - 1 - g) : (NSTAGE - 2);
            if (younger >= NSTAGE - 2 && NSTAGE > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * PER_LOADER) : "memory");
            else if (younger == 1 && NSTAGE > 3)     asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_LOADER) : "memory");
            else                                      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                   // B_g
            if (g + NSTAGE - 1 < total) issue(g + NSTAGE - 1);            // refill the slot of stage g-1
        }
        return;
    }

    // ---------------------------------------------------------------------- consumer
    const int l_q = 4 * ((lane >> 2) & 1) + (lane & 3);                  // leftover path: query row in the leftover tile
    const int l_row = 4 * (lane >> 3) + (lane & 3);                      // row among the wave's 32 rows
    const int l_boff = (l_row >> 4) * PB_CW * KC * 64 + (l_row & 15);    // tile r*4+wave: r-stride = 4 tiles
    int g = 0;
    for (int blk = 0; blk < nblocks; ++blk) {
        int64_t tb;
        int nt;
        block_range(blk, tb, nt);
        const int nr = __builtin_amdgcn_readfirstlane(nt > PB_CW + wave ? 2 : (nt > wave ? 1 : 0));   // tiles wave, wave+4
        f32x4 acc[PB_R][QT > 0 ? QT : 1];
#pragma unroll
        for (int r = 0; r < PB_R; ++r)
#pragma unroll
            for (int q = 0; q < QT; ++q) acc[r][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 accl = (f32x4){0.f, 0.f, 0.f, 0.f};

        for (int c = 0; c < nchunks; ++c, ++g) {
            __builtin_amdgcn_s_barrier();                                   // B_g
            __builtin_amdgcn_sched_barrier(0);
            const f32x4 *slot = ring + (g % NSTAGE) * (STAGE_TILES * 64);
            if (nr == 2) pb_chunk<2, QT, QR, KC>(acc, accl, slot, lane, wave, l_q, l_boff);
            else if (nr == 1) pb_chunk<1, QT, QR, KC>(acc, accl, slot, lane, wave, l_q, l_boff);
            // all LDS reads of this stage are consumed by the MFMAs above before the next barrier
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }

        // block epilogue: straight from the accumulators; lane (g4, j) holds queries 4*g4..4*g4+3 of row j of each tile
        if (nr > 0) {
            int64_t nn = n;
            asm volatile("" : "+s"(nn));                                   // keeps the 40 row offsets out of registers
            const int g4 = 4 * (lane >> 4), fj = lane & 15;
#pragma unroll
            for (int r = 0; r < PB_R; ++r) {
                const int64_t row = (tb + r * PB_CW + wave) * TILE_ROWS + fj;
                if (r < nr && row < n) {
                    float *po = out + (int64_t)g4 * nn + row;
#pragma unroll
                    for (int q = 0; q < QT; ++q) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (q * 16 + g4 + i < nq_valid) po[i * nn] = acc[r][q][i];
                        po += 16 * nn;
                    }
                }
            }
            if constexpr (QR != 0) {
                const int r = l_row >> 4;
                const int64_t row = (tb + r * PB_CW + wave) * TILE_ROWS + (l_row & 15);
                if (r < nr && row < n) {
                    const int q0 = QT * 16 + 4 * ((lane >> 2) & 1);
                    float *po = out + (int64_t)q0 * nn + row;
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        if (q0 + v < nq_valid) po[v * nn] = accl[v];
                }
            }
        }
    }
}

}  // namespace mdx
