// Split-precision similarity (MDX_F32_SPLIT3): the SAME fp32 shard as the exact kernel, multiplied on the bf16 MFMA.
//
// The exact kernel (mdx_scores_kernel.h) is bound by the fp32 MFMA rate (1/16 of the bf16 rate): 2.6 ms at
// 1 M x 70 x 2048, with the 8.2 GB shard stream needing ~1.5 ms.  Here every fp32 operand x is written as the sum of
// three bf16 pieces  x = h + m + l + e,  h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)  (round to nearest even, the
// subtractions exact), |e| <= 2^-27 |x|, and a product x*y is taken as the six piece products of order >= 2^-16:
//     x*y ~= hh + (hm + mh) + (hl + lh + mm)          (dropped: ml + lm + ll + e-terms, <= 2^-23 |x*y|)
// on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: 6/16 of the fp32 MFMA time, so the kernel is bound by the shard
// stream (HBM), not by the matrix pipe.  The database stays fp32 in HBM -- no second copy, no re-quantised shard: the
// consumer waves split the tile they have just read from LDS in registers (v_cvt_pk_bf16_f32 + shift/and + v_sub, 5.5
// vector instructions per element, issued in the shadow of the MFMAs); the queries are split once per call by the
// re-tiling kernel.  NOT the k-ordered fma chain: results differ from the exact kernel by the fp32 accumulation order
// inside and across the MFMAs (~1e-7 on unit vectors; tests bound it by 2e-6 = SUM_ORDER_TOL) -- a LABELLED second
// mode, the exact chain stays the default and the parity contract.
//
// Shapes.  A chunk = 32 k = the two fp32 tiles (rt, 2c), (rt, 2c+1) of a row tile.  Lane (g, j) of a row tile holds
// from them the 8 values k = 32c + 16*(e>>2) + 4*(e&3) + g, e = 0..7, of row j: those are "its" 8 k-slots of the
// 16x16x32 MFMA's B operand.  A dot product does not care in which order k is visited as long as both operands agree,
// so the query pieces are stored in the same slot order (retile_split3_kernel) and the shard format is untouched.
// Workgroup = CW consumer waves (R row tiles each) + 4 LDS-DMA loader waves, ring of NSTAGE stages, one raw s_barrier
// per chunk -- the protocol of scores_lc_kernel.  Stage = 3*QT query-piece tiles + 2*CW*R database tiles (KiB each).
#pragma once
#include "mdx_scores_kernel.h"

namespace mdx {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// two fp32 -> packed bf16 pair (round to nearest even; low half = a): v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t pk_bf16(float a, float b)
{
    bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, v);
}

// 8 fp32 (the lane's k-slots of one row tile and chunk) -> three packed bf16x8 pieces
__device__ __forceinline__ void split3(const f32x4 &x0, const f32x4 &x1, u32x4 &h, u32x4 &m, u32x4 &l)
{
    const float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float a = x[2 * p], b = x[2 * p + 1];
        const uint32_t hp = pk_bf16(a, b);
        const float ra = a - __uint_as_float(hp << 16), rb = b - __uint_as_float(hp & 0xFFFF0000u);       // exact
        const uint32_t mp = pk_bf16(ra, rb);
        const float sa = ra - __uint_as_float(mp << 16), sb = rb - __uint_as_float(mp & 0xFFFF0000u);     // exact
        h[p] = hp;
        m[p] = mp;
        l[p] = pk_bf16(sa, sb);
    }
}

__device__ __forceinline__ f32x4 mfma_bf16(const u32x4 &a, const u32x4 &b, const f32x4 &c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ---- MDX_F32_SPLIT2: two fp16 pieces with a scaled residual (block floating point) ----------------------------------------
// X = x * S (S a power of two that brings the matrix' largest magnitude into [2^13, 2^14): fp16's range, exactly);
// h = fp16(X) (round toward zero), m = fp16((X - h) * 2^11) (the residual is exact, the scaling too): |X - h - m / 2^11| <
// 2^-20 |X|.  A product is hh + (hm + mh) / 2^11 -- three MFMAs instead of six; the cross terms go to an accumulator of their
// own, which is scaled once at the end; the dropped mm / 2^22 is <= 2^-20 of the product.  What this costs against the
// three-piece form: elements more than 2^-27 below the matrix' largest lose relative precision (they are subnormal in
// fp16 after scaling) -- the error bound is relative to max|x| max|q|, not to each element.  Truncation (not rounding to
// nearest: v_cvt_pkrtz_f16_f32 packs two values in one instruction) leaves every operand short by up to 2^-20 of itself, so a
// score comes out short by up to 2^-19 of ITSELF: a bias proportional to the score, neutral for a ranking; the random part
// is an order of magnitude smaller.
typedef __fp16 half2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split2(const f32x4 &x0, const f32x4 &x1, float scale, u32x4 &h, u32x4 &m)
{
    const float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float a = x[2 * p] * scale, b = x[2 * p + 1] * scale;
        const half2_t hp = __builtin_amdgcn_cvt_pkrtz(a, b);
        const float ra = (a - (float)hp[0]) * 2048.0f, rb = (b - (float)hp[1]) * 2048.0f;       // exact
        const half2_t mp = __builtin_amdgcn_cvt_pkrtz(ra, rb);
        h[p] = __builtin_bit_cast(uint32_t, hp);
        m[p] = __builtin_bit_cast(uint32_t, mp);
    }
}

__device__ __forceinline__ f32x4 mfma_f16(const u32x4 &a, const u32x4 &b, const f32x4 &c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// S for a matrix whose largest finite magnitude has the fp32 bit pattern `max_bits`: 2^(13 - floor(log2 max)); 1 for an all-zero matrix
__host__ __device__ __forceinline__ float split2_scale(uint32_t max_bits)
{
    const int e = (int)((max_bits >> 23) & 0xFF);               // biased exponent of the maximum (0: zero / denormal)
    if (e == 0 || e == 0xFF) return 1.0f;
    int k = 13 - (e - 127);                                     // scaled maximum in [2^13, 2^14)
    k = k > 126 ? 126 : (k < -126 ? -126 : k);
    uint32_t bits = (uint32_t)(k + 127) << 23;
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

// largest finite |q - center| of the query matrix -> *cell (uint bit pattern; the caller zeroes it first); same addressing as the re-tiling
__global__ __launch_bounds__(256) void absmax_kernel(const float *__restrict__ src, int64_t rs, int64_t ks, int64_t nq, int64_t d,
                                                     const float *__restrict__ center, uint32_t *__restrict__ cell)
{
    const int64_t total = nq * d;
    uint32_t best = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / d, k = i % d;
        float x = src[row * rs + k * ks];
        if (center) x -= center[k];
        const uint32_t u = __float_as_uint(x) & 0x7FFFFFFFu;
        if (u < 0x7F800000u && u > best) best = u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t other = (uint32_t)__shfl_xor((int)best, o, 64);
        best = other > best ? other : best;
    }
    if ((threadIdx.x & 63) == 0 && best) atomicMax(cell, best);
}

// queries -> two arrays of fp16 tiles [piece][query tile][chunk] in the shard's k-slot order, scaled by split2_scale(*cell)
__global__ __launch_bounds__(256) void retile_split2_kernel(const float *__restrict__ src, int64_t rs, int64_t ks, int64_t nq,
                                                            int64_t d, const float *__restrict__ center, const uint32_t *__restrict__ cell,
                                                            u32x4 *__restrict__ tiles, int64_t QT_total, int64_t NC)
{
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= QT_total * NC) return;
    const int64_t qt = tile / NC, c = tile % NC;
    const int j = lane & 15, g = lane >> 4;
    const int64_t row = qt * TILE_ROWS + j;
    const float scale = split2_scale(*cell);
    f32x4 x0, x1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int64_t k = c * 32 + 16 * (e >> 2) + 4 * (e & 3) + g;
        float x = 0.0f;
        if (row < nq && k < d) {
            x = src[row * rs + k * ks];
            if (center) x -= center[k];
        }
        if (e < 4) x0[e] = x;
        else x1[e - 4] = x;
    }
    u32x4 h, m;
    split2(x0, x1, scale, h, m);
    tiles[((0 * QT_total + qt) * NC + c) * 64 + lane] = h;
    tiles[((1 * QT_total + qt) * NC + c) * 64 + lane] = m;
}

// queries -> three arrays of bf16 tiles [piece][query tile][chunk], 1 KiB each, lane (g, j) element e =
// piece(query 16*qt + j, k = 32c + 16*(e>>2) + 4*(e&3) + g); queries >= nq and k >= d read 0
__global__ __launch_bounds__(256) void retile_split3_kernel(const float *__restrict__ src, int64_t rs, int64_t ks, int64_t nq,
                                                            int64_t d, const float *__restrict__ center,
                                                            u32x4 *__restrict__ tiles, int64_t QT_total, int64_t NC)
{
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= QT_total * NC) return;
    const int64_t qt = tile / NC, c = tile % NC;
    const int j = lane & 15, g = lane >> 4;
    const int64_t row = qt * TILE_ROWS + j;
    f32x4 x0, x1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int64_t k = c * 32 + 16 * (e >> 2) + 4 * (e & 3) + g;
        float x = 0.0f;
        if (row < nq && k < d) {
            x = src[row * rs + k * ks];
            if (center) x -= center[k];
        }
        if (e < 4) x0[e] = x;
        else x1[e - 4] = x;
    }
    u32x4 h, m, l;
    split3(x0, x1, h, m, l);
    tiles[((0 * QT_total + qt) * NC + c) * 64 + lane] = h;
    tiles[((1 * QT_total + qt) * NC + c) * 64 + lane] = m;
    tiles[((2 * QT_total + qt) * NC + c) * 64 + lane] = l;
}

// NP = 3: three bf16 pieces, six products (MDX_F32_SPLIT3); NP = 2: two fp16 pieces with a scaled residual, three products
// (MDX_F32_SPLIT2; db_scale = the shard's S, q_cell = the device word holding the queries' largest magnitude)
template <int NP, int QT, int R, int NSTAGE, int CW, int DB_AUX>
__device__ __forceinline__ void split_kernel_body(f32x4 *ring, const f32x4 *__restrict__ db, const u32x4 *__restrict__ qpieces,
                                                  float *__restrict__ out, int64_t n, int KB, int QT_total, int qt_first, int nq_valid,
                                                  float db_scale, const uint32_t *__restrict__ q_cell)
{
    constexpr int LW = 4;                           // loader waves
    constexpr int QTILES = NP * QT;                 // KiB tiles of query pieces per stage: [piece][q]
    constexpr int BTILES = CW * R * 2;              // KiB tiles of database per stage: [wave][r][half]
    constexpr int STAGE_TILES = QTILES + BTILES;
    constexpr int PER_LOADER = (STAGE_TILES + LW - 1) / LW;
    static_assert((NSTAGE - 1) * PER_LOADER <= 63, "vmcnt is 6 bits");
    static_assert(NSTAGE * STAGE_TILES * 1024 <= 160 * 1024, "the ring must fit the CU's LDS");
    // ring: [NSTAGE][STAGE_TILES][64]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NC = KB / 2;                                          // chunks of 32 k (KB is a multiple of 4)
    const int64_t rt_wg = row_block_of(blockIdx.x, gridDim.x) * CW * R;            // first row tile of the workgroup
    const int qt0 = qt_first + (int)blockIdx.y * QT;                // first query tile of this workgroup (grid.y = pass over groups of QT tiles)
    out += (int64_t)qt0 * TILE_ROWS * n;

    if (wave >= CW) {
        // ------------------------------------------------------------- loader
        const int lw = wave - CW;
        const f32x4 *src[PER_LOADER];
        int dst[PER_LOADER], step[PER_LOADER];
#pragma unroll
        for (int t = 0; t < PER_LOADER; ++t) {
            const int i = (lw + t * LW) < STAGE_TILES ? (lw + t * LW) : (STAGE_TILES - 1);   // uneven split: the last tile twice
            dst[t] = i * 64;
            if (i < QTILES) {
                const int p = i / QT, q = i % QT;
                src[t] = (const f32x4 *)qpieces + ((int64_t)(p * QT_total + qt0 + q) * NC) * 64 + lane;
                step[t] = 64;                       // next chunk of the same (piece, query tile)
            } else {
                const int j = i - QTILES;
                const int tile = j >> 1, half = j & 1;              // tile = cw * R + r
                src[t] = db + shard_tile(rt_wg + tile, half, KB) * 64 + lane;
                step[t] = 128;                      // two fp32 tiles further along k
            }
        }
        auto issue = [&](int c) {
            f32x4 *slot = ring + (c % NSTAGE) * (STAGE_TILES * 64);
#pragma unroll
            for (int t = 0; t < PER_LOADER; ++t) {
                const bool is_db = (lw + t * LW) >= QTILES;
                const f32x4 *p = src[t] + (int64_t)c * step[t];
                // the shard is read once (non-temporal); the query pieces are re-read by every workgroup from the L2
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
            if (c < NC) issue(c);
        for (int c = 0; c < NC; ++c) {
            const int younger = (NC - 1 - c) < (NSTAGE - 2) ? (NC - 1 - c) : (NSTAGE - 2);
            if (younger >= NSTAGE - 2 && NSTAGE > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * PER_LOADER) : "memory");
            else if (younger == 1 && NSTAGE > 3)     asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_LOADER) : "memory");
            else                                      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                   // B_c: stage c has landed
            if (c + NSTAGE - 1 < NC) issue(c + NSTAGE - 1);                 // refill the slot of stage c-1
        }
        return;
    }

    // ----------------------------------------------------------------- consumer
    f32x4 acc[R][QT], acx[NP == 2 ? R : 1][NP == 2 ? QT : 1];      // acx: the cross terms hm + mh of the two-piece form (scaled at the end)
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            acc[r][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (NP == 2) acx[r][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }

    for (int c = 0; c < NC; ++c) {
        __builtin_amdgcn_s_barrier();                                       // B_c
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 *slot = ring + (c % NSTAGE) * (STAGE_TILES * 64);
        const u32x4 *qs = (const u32x4 *)slot + lane;
        const f32x4 *bs = slot + (QTILES + wave * R * 2) * 64 + lane;
        if constexpr (NP == 2) {
            u32x4 dh[R], dm[R];
#pragma unroll
            for (int r = 0; r < R; ++r) split2(bs[(2 * r) * 64], bs[(2 * r + 1) * 64], db_scale, dh[r], dm[r]);
#pragma unroll
            for (int q = 0; q < QT; ++q) {
                const u32x4 qh = qs[(0 * QT + q) * 64], qm = qs[(1 * QT + q) * 64];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    acx[r][q] = mfma_f16(qm, dh[r], acx[r][q]);
                    acx[r][q] = mfma_f16(qh, dm[r], acx[r][q]);
                    acc[r][q] = mfma_f16(qh, dh[r], acc[r][q]);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            continue;
        }
        u32x4 dh[R], dm[R], dl[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            split3(bs[(2 * r) * 64], bs[(2 * r + 1) * 64], dh[r], dm[r], dl[r]);
        }
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            const u32x4 qh = qs[(0 * QT + q) * 64], qm = qs[(1 * QT + q) * 64], ql = qs[(2 * QT + q) * 64];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                // smallest terms first: inside the accumulator's rounding they are kept best next to a young sum
                f32x4 a = acc[r][q];
                a = mfma_bf16(ql, dh[r], a);
                a = mfma_bf16(qh, dl[r], a);
                a = mfma_bf16(qm, dm[r], a);
                a = mfma_bf16(qm, dh[r], a);
                a = mfma_bf16(qh, dm[r], a);
                a = mfma_bf16(qh, dh[r], a);
                acc[r][q] = a;
            }
        }
        // all LDS reads of this stage are consumed before the next barrier lets the loaders refill it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }

    // Epilogue (as scores_lc_kernel): transpose the accumulators through the free ring so that every query row of the
    // workgroup's rows leaves as one contiguous run
    constexpr int ROWS = CW * R * TILE_ROWS;
    constexpr int LDW = ROWS + 4;
    static_assert(QT * 16 * LDW * 4 <= NSTAGE * STAGE_TILES * 1024, "output staging must fit in the ring");
    __builtin_amdgcn_s_barrier();
    float *stage = (float *)ring;
    {
        // two-piece form: (hh + cross / 2^11) / (S_db * S_q), every factor a power of two
        float unscale = 1.0f;
        if constexpr (NP == 2) unscale = 1.0f / (db_scale * split2_scale(*q_cell));
        const int qrow = 4 * (lane >> 4), col = lane & 15;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int q = 0; q < QT; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = acc[r][q][i];
                    if constexpr (NP == 2) v = (v + acx[r][q][i] * (1.0f / 2048.0f)) * unscale;
                    stage[(q * 16 + qrow + i) * LDW + (wave * R + r) * TILE_ROWS + col] = v;
                }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int64_t row0 = rt_wg * TILE_ROWS;
    const int rows_valid = (int)((n - row0) < ROWS ? (n - row0) : ROWS);
    const int left = nq_valid - qt0 * TILE_ROWS;
    const int nq_here = left < QT * TILE_ROWS ? left : QT * TILE_ROWS;
    for (int e = tid; e < nq_here * ROWS; e += CW * 64) {
        const int qi = e / ROWS, rr = e % ROWS;
        if (rr < rows_valid) store_score<true>(out + (int64_t)qi * n + row0 + rr, stage[qi * LDW + rr]);      // non-temporal: see store_score
    }
}

template <int QT, int R, int NSTAGE, int CW, int DB_AUX = 2>
__global__ __launch_bounds__(CW * 64 + 256, 1) void scores_split3_kernel(const f32x4 *__restrict__ db, const u32x4 *__restrict__ qpieces,
                                                                         float *__restrict__ out, int64_t n, int KB, int QT_total,
                                                                         int qt_first, int nq_valid)
{
    extern __shared__ __attribute__((aligned(16))) f32x4 split_ring[];
    split_kernel_body<3, QT, R, NSTAGE, CW, DB_AUX>(split_ring, db, qpieces, out, n, KB, QT_total, qt_first, nq_valid, 1.0f, nullptr);
}

template <int QT, int R, int NSTAGE, int CW, int DB_AUX = 2>
__global__ __launch_bounds__(CW * 64 + 256, 1) void scores_split2_kernel(const f32x4 *__restrict__ db, const u32x4 *__restrict__ qpieces,
                                                                         float *__restrict__ out, int64_t n, int KB, int QT_total,
                                                                         int qt_first, int nq_valid, float db_scale,
                                                                         const uint32_t *__restrict__ q_cell)
{
    extern __shared__ __attribute__((aligned(16))) f32x4 split_ring[];
    split_kernel_body<2, QT, R, NSTAGE, CW, DB_AUX>(split_ring, db, qpieces, out, n, KB, QT_total, qt_first, nq_valid, db_scale, q_cell);
}

}  // namespace mdx
