// float64 GEMMs of the whitening LEARNING step on the f64 matrix cores.
//
// Replaces the three dense products of mdir/external/cirtorch/utils/whiten.py:
//   Xcov = np.dot(Xc, Xc.T)      :22   (pcawhitenlearn)        -> mdx_gram_f64 (center = m)
//   S    = np.dot(df, df.T)      :42   (whitenlearn, pairs)     -> mdx_gram_f64
//   df   = np.dot(P, X - m)      :45                            -> mdx_project_f64
//   D    = np.dot(df, df.T)      :46                            -> mdx_gram_f64
// The reference learns in float64 (the low-variance directions of a 2048-d covariance sit below fp32
// noise), so these are f64 in, f64 accumulate: v_mfma_f64_16x16x4_f64.  The small dense factorisations
// (Cholesky, eig, inverse) stay on the host as in the reference.
//
// C[i][j] = sum_k At[k][i] * Bt[k][j], both operands K-MAJOR with the tile dimension contiguous: a tile row is one run of
// global memory and goes into LDS as it is (the first version read the K-contiguous operands -- descriptors [d, n],
// P [dout, d] -- in place and transposed 32-byte pieces into LDS: 4-way bank conflicts and 16 separate rows per wave-load;
// the Gram form, with two such operands, ran at half the rate of the projection with one).  So the K-contiguous inputs are
// transposed ONCE into the workspace (a streaming pass, ~0.1 ms for 2048 x 20 000; the Gram form's centring `X - m` is
// applied there).  Two kernels:
//   gemm_f64_lc_kernel   128 x 128 tiles of large problems (d, dout >= 1024): 4 MFMA waves + 8 LDS-DMA loader waves, a
//                        ring of four 16-k stages -- below, with the measurements that shaped it.  D = 2048, n = 20 000:
//                        projection 2.51-2.63 ms = 0.81-0.85 of the f64 matrix peak (round 3: 3.12 = 0.68; rocBLAS dgemm
//                        2.67-2.79 on the same box), Gram 1.70 ms = 0.67 of the peak for the triangle's tiles (round 3:
//                        2.14 = 0.53; 0.11 ms of it the transposed copy, 0.07 the ordered sum of the K ranges);
//   gemm_f64_kernel      everything else (64 x 64 tiles, 4 waves, registers -> double-buffered LDS, one barrier per 16 k;
//                        row stride = 32 words mod 64: the two 16-lane groups a half-wave reads together fall into
//                        different bank halves), and the odd last column of a projection.
// The Gram form is symmetric: only tiles on or above the diagonal are computed (SYRK-shaped: half the flops of a GEMM), long
// sums are cut into K ranges whose partial tiles are added in range order (a fixed order).
#include <type_traits>

#include "mdx_common.h"

namespace mdx {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x4u __attribute__((ext_vector_type(4), aligned(8)));    // one 32-byte access at 8-byte alignment

constexpr int GK = 16;

// [rows, cols] -> [cols, rows], optionally subtracting center[row] (the Gram form's centring)
__global__ __launch_bounds__(256) void transpose_f64_kernel(const double *__restrict__ src, int64_t rows, int64_t cols,
                                                            const double *__restrict__ center, double *__restrict__ dst, int64_t ldd)
{
    __shared__ double tile[32][33];
    const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int64_t row = r0 + r, col = c0 + tx;
        double v = 0.0;
        if (row < rows && col < cols) v = src[row * cols + col] - (center ? center[row] : 0.0);
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int64_t col = c0 + r, row = r0 + tx;
        if (col < cols && row < rows) dst[col * ldd + row] = tile[tx][r];
    }
}

// C[i][j] = sum_{k in range} at[k*lda + i] * (bt[k*ldb + j] - kcenter[k]);  SYM: only tiles with bi <= bj (at == bt).
// WT = rows / columns per wave (32 or 64): workgroup tile 2 WT x 2 WT.  At 64 x 64 a tile moves 1 KiB of operands per k
// for 8 Kflop, i.e. 9.8 TB/s of L2 -> LDS traffic at the f64 peak -- the first version's ceiling (50 TFLOP/s, whatever the
// loads looked like); 128 x 128 halves the bytes per flop.
template <bool SYM, int WT>
__global__ __launch_bounds__(256, WT == 64 ? 2 : 4) void gemm_f64_kernel(const double *__restrict__ at, int64_t lda, const double *__restrict__ bt,
                                                       int64_t ldb, const double *__restrict__ kcenter, double *__restrict__ out,
                                                       int64_t M, int64_t N, int64_t K, int64_t ksplit, int64_t ldc)
{
    constexpr int GM = 2 * WT, GN = 2 * WT, GLD = GM + 16;      // row stride = 32 words mod 64
    constexpr int NT = WT / 16;                                 // MFMA tiles per wave and side
    constexpr int QV = GM / 64;                                 // 32-byte quads per thread, operand and step
    __shared__ double As[2][GK][GLD], Bs[2][GK][GLD];
    int bi = blockIdx.y, bj = blockIdx.x;
    if (SYM) {
        // blockIdx.x counts only the tiles on or above the diagonal, row by row.  (A T x T grid whose lower half exits at
        // once leaves the XCDs -- workgroup id mod 8 -- with 10 to 24 of a row block's 136 tiles each: measured, half the
        // waves resident and 37 % MFMA occupancy against 72 % for the rectangular form.)
        const int T = (int)((M + 2 * WT - 1) / (2 * WT));
        int t = blockIdx.x;
        bi = 0;
        while (t >= T - bi) { t -= T - bi; ++bi; }
        bj = bi + t;
    }
    // blockIdx.z = K range: slice z of `out` ([ranges][M][N]; with one range `out` is the result itself)
    const int64_t kbeg = (int64_t)blockIdx.z * ksplit, kend = (kbeg + ksplit) < K ? (kbeg + ksplit) : K;
    out += (int64_t)blockIdx.z * M * ldc;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t row0 = (int64_t)bi * GM, col0 = (int64_t)bj * GN;

    // loader: thread -> (k row, 4 consecutive columns) of each operand's 16 x GM tile, QV times.  `fetch` only ISSUES the loads
    // (no arithmetic on what they return: the centring is applied in `park`, after the step's MFMAs -- round 3 subtracted
    // right behind the load, which put a full vmcnt(0) wait at the head of every step: 69 % matrix-pipe occupancy, PMC);
    // a step that lies inside the matrices (every step but the last of an edge tile) takes unchecked 32-byte loads.
    const bool diag = SYM && bi == bj;          // a diagonal tile multiplies its columns with themselves
    const bool interior = row0 + GM <= M && col0 + GN <= N;
    f64x4 ra[QV], rb[QV];
    double sub[QV];
    auto fetch = [&](int64_t k0) __attribute__((always_inline)) {
        if (interior && k0 + GK <= kend) {
#pragma unroll
            for (int q = 0; q < QV; ++q) {
                const int idx = tid + q * 256, pk = idx / (GM / 4), pj = (idx % (GM / 4)) * 4;
                const int64_t k = k0 + pk;
                ra[q] = *(const f64x4u *)(at + k * lda + row0 + pj);
                if (!diag) rb[q] = *(const f64x4u *)(bt + k * ldb + col0 + pj);
                sub[q] = kcenter ? kcenter[k] : 0.0;
            }
        } else {
#pragma unroll
            for (int q = 0; q < QV; ++q) {
                const int idx = tid + q * 256, pk = idx / (GM / 4), pj = (idx % (GM / 4)) * 4;
                const int64_t k = k0 + pk;
                sub[q] = (kcenter && k < kend) ? kcenter[k] : 0.0;     // rows past the K range stay exactly zero
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ra[q][e] = (k < kend && row0 + pj + e < M) ? at[k * lda + row0 + pj + e] : 0.0;
                    rb[q][e] = (k < kend && col0 + pj + e < N) ? bt[k * ldb + col0 + pj + e] : sub[q];   // columns past N: zero after centring
                }
            }
        }
    };
    auto park = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < QV; ++q) {
            const int idx = tid + q * 256, pk = idx / (GM / 4), pj = (idx % (GM / 4)) * 4;
            *(f64x4 *)&As[buf][pk][pj] = ra[q];
            *(f64x4 *)&Bs[buf][pk][pj] = diag ? ra[q] : rb[q] - sub[q];
        }
    };

    f64x4 acc[NT][NT];
#pragma unroll
    for (int mi = 0; mi < NT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = (f64x4){0.0, 0.0, 0.0, 0.0};

    fetch(kbeg);
    park(0);
    __syncthreads();
    const int kr = lane >> 4, c16 = lane & 15;
    int buf = 0;
    for (int64_t k0 = kbeg; k0 < kend; k0 += GK, buf ^= 1) {
        const bool more = k0 + GK < kend;
        if (more) fetch(k0 + GK);           // in flight under the MFMAs below
#pragma unroll
        for (int kk = 0; kk < GK / 4; ++kk) {
            double av[NT], bv[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                av[t] = As[buf][kk * 4 + kr][wm * WT + t * 16 + c16];
                bv[t] = Bs[buf][kk * 4 + kr][wn * WT + t * 16 + c16];
            }
#pragma unroll
            for (int mi = 0; mi < NT; ++mi)
#pragma unroll
                for (int ni = 0; ni < NT; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[mi], bv[ni], acc[mi][ni], 0, 0, 0);
        }
        if (more) park(buf ^ 1);            // the other buffer: its readers finished before the previous barrier
        __syncthreads();
    }
    // C/D of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int mi = 0; mi < NT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int64_t r = row0 + wm * WT + mi * 16 + (lane >> 4) + 4 * v;
                const int64_t c = col0 + wn * WT + ni * 16 + (lane & 15);
                if (r < M && c < N) {
                    out[r * ldc + c] = acc[mi][ni][v];
                    if (SYM && bi != bj && gridDim.z == 1) out[c * ldc + r] = acc[mi][ni][v];
                }
            }
}

// ===========================================================================
// Loader / consumer form of the same GEMM (128 x 128 tiles of large problems; the kernel above keeps the small and the
// odd-shaped ones).  Where the kernel above loses time (tools/gram_ablate.hip, D = 2048, n = 20 000, projection): 2.78 ms as
// it is, 2.49 without its global loads, 2.34 without its LDS writes + barrier, 2.29 with MFMAs alone (= 0.93 of the peak: the
// clock under f64 MFMA load is 2.24 of 2.4 GHz) -- a wave that loads, parks and waits at a workgroup-wide barrier every
// 16 k cannot keep a 64-cycle MFMA pipe fed.  So, as in the similarity kernel (mdx_scores_kernel.h):
//   waves 0-3 (one per SIMD)  consumers: ds_read + v_mfma_f64_16x16x4_f64 only, a 64 x 64 quadrant each; the operands of
//                             the next four k are read while the current four multiply, and the stage hand-over (one raw
//                             s_barrier per 16 k) sits BEFORE the last four k of a stage are multiplied: the MFMAs that
//                             follow it need nothing from the new stage, so the pipe runs through the barrier;
//   waves 4-11                loaders: one k row of a tile (128 doubles = 1 KiB) per global_load_lds_dwordx4, four stages of
//                             16 k in an LDS ring, counted vmcnt; no registers, no LDS writes by any wave.
// The centring of the projection (X - m) is applied by the consumers on the operand they have just read (the 16 centre
// values of a stage ride in the ring): the same differences the reference forms (whiten.py:45), no extra pass over X.
// Edges: a loader lane whose column pair lies past the matrix takes the last pair inside it (such columns only reach
// outputs that are never stored) and the B operand's k row is clamped to K-1; the A operand is always one of this file's
// transposed copies, whose rows K .. round_up(K, 16) are zero, so k past the range contributes exact zeros.  N is even
// here (a pair never straddles the end of a caller's row): the projection hands an odd last column to the register kernel.
// ===========================================================================
constexpr int LC_ROWB = (128 + 16) * 8;             // bytes per k row of an operand tile in LDS (row stride = 32 words mod 64)
// Shape of the ring, measured at D = 2048, n = 20 000 (tools/gram_ablate.hip -DMDX_GRAM_LW=.. -DMDX_GRAM_LC_GK=.. -DMDX_GRAM_LC_NSTAGE=..;
// projection, ms): loader waves 4 / 8 with 4 stages of 16 k: 2.60 / 2.52; 3 stages 2.60; 8 stages of 8 k 2.71 / 2.60; two
// workgroups of 4 + 1 or 4 + 2 waves per CU (2 stages of 16 k or 4 of 8 k, 74 KiB each): 3.2-3.5 / 2.9 -- the LDS-DMA issue
// of ONE wave does not keep a stage ahead, more issuing waves do; 12 waves of <= 168 registers are what a CU holds.
#ifndef MDX_GRAM_LW
#define MDX_GRAM_LW 8                   // loader waves
#endif
#ifndef MDX_GRAM_LC_GK
#define MDX_GRAM_LC_GK 16
#endif
#ifndef MDX_GRAM_LC_NSTAGE
#define MDX_GRAM_LC_NSTAGE 4
#endif
constexpr int LC_GK = MDX_GRAM_LC_GK;               // k per stage (8 or 16; the ranges and the zero rows stay multiples of GK = 16)
constexpr int LC_OPB = LC_GK * LC_ROWB;             // one operand of a stage
constexpr int LC_STAGEB = 2 * LC_OPB + 256;         // A, B, the stage's centre values
constexpr int LC_LW = MDX_GRAM_LW;
constexpr int LC_NSTAGE = MDX_GRAM_LC_NSTAGE;
constexpr int LC_WG_PER_CU = LC_LW <= 2 ? 2 : 1;
constexpr int LC_LDS = LC_NSTAGE * LC_STAGEB;       // 148 480 B: one workgroup per CU

// Workgroup -> (tile row, tile column, K range).  Workgroup ids go round the 8 XCDs (id mod 8), each with its own L2, and a
// CU holds one workgroup: the 32 workgroups an XCD runs at a time should share operand strips.  An XCD therefore takes a
// CONTIGUOUS run of the tile sequence (the row_block_of of the similarity kernel); in the rectangular form the sequence goes
// through blocks of 4 x 8 tiles (12 strips for 32 workgroups instead of 33), in the Gram form through the triangle row by row.
struct LcTile { int bi, bj, z; bool valid; };

template <bool SYM>
__device__ __forceinline__ LcTile lc_tile_of(unsigned id, unsigned total, int T_rows, int T_cols, int splits)
{
    const unsigned per = total / 8, rem = total % 8, x = id % 8, k = id / 8;
    int64_t L = (int64_t)x * per + (x < rem ? x : rem) + k;            // position in the tile sequence
    LcTile t;
    if (SYM) {
        const int tri = T_rows * (T_rows + 1) / 2;
        t.z = (int)(L / tri);
        int r = (int)(L % tri);
        t.bi = 0;
        while (r >= T_rows - t.bi) { r -= T_rows - t.bi; ++t.bi; }
        t.bj = t.bi + r;
        t.valid = t.z < splits;
    } else {
        constexpr int SBR = 4 * LC_WG_PER_CU;                           // super-blocks of SBR x 8 tiles: what an XCD's 32 CUs hold at a time
        const int sbc = (T_cols + 7) / 8;
        const int sb = (int)(L / (8 * SBR)), in = (int)(L % (8 * SBR));
        t.bi = (sb / sbc) * SBR + in / 8;
        t.bj = (sb % sbc) * 8 + in % 8;
        t.z = 0;
        t.valid = t.bi < T_rows && t.bj < T_cols;
    }
    return t;
}

template <bool SYM, bool CENTER>
__global__ __launch_bounds__((4 + LC_LW) * 64, LC_WG_PER_CU) void gemm_f64_lc_kernel(const double *__restrict__ at, int64_t lda, const double *__restrict__ bt,
                                                            int64_t ldb, const double *__restrict__ kcenter, double *__restrict__ out,
                                                            int64_t M, int64_t N, int64_t K, int64_t ksplit, int64_t ldc, int T_rows, int T_cols, int splits)
{
    constexpr int NSTAGE = LC_NSTAGE, LW = LC_LW, PER_LOADER = 2 * LC_GK / LW + (CENTER ? 1 : 0);
    static_assert((NSTAGE - 2) * PER_LOADER <= 63, "vmcnt is 6 bits");
    extern __shared__ __attribute__((aligned(16))) char lc_ring[];
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    const LcTile tl = lc_tile_of<SYM>(blockIdx.x, gridDim.x, T_rows, T_cols, splits);
    if (!tl.valid) return;                                              // whole workgroup: nobody waits at a barrier for it
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t kbeg = (int64_t)tl.z * ksplit, kend = (kbeg + ksplit) < K ? (kbeg + ksplit) : K;
    const int nsteps = (int)((kend - kbeg + LC_GK - 1) / LC_GK);
    const int64_t row0 = (int64_t)tl.bi * 128, col0 = (int64_t)tl.bj * 128;
    out += (int64_t)tl.z * M * ldc;

    if (wave >= 4) {
        // ------------------------------------------------------------- loader: k rows lw, lw + LW, ... of both operands
        const int lw = wave - 4;
        const int64_t ac = (row0 + 2 * lane) < M ? (row0 + 2 * lane) : (M - 1);      // (an odd M: the copy's rows are padded)
        const int64_t bc = (col0 + 2 * lane) < N ? (col0 + 2 * lane) : (N - 2);
        auto issue = [&](int c) __attribute__((always_inline)) {
            char *slot = lc_ring + (c % NSTAGE) * LC_STAGEB;
            const int64_t k0 = kbeg + (int64_t)c * LC_GK;
#pragma unroll
            for (int t = 0; t < LC_GK / LW; ++t) {
                const int kr = lw + LW * t;
                const int64_t ka = k0 + kr, kb = ka < K ? ka : K - 1;
                __builtin_amdgcn_global_load_lds((glb_void *)(at + ka * lda + ac), (lds_void *)(slot + kr * LC_ROWB), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((glb_void *)(bt + kb * ldb + bc), (lds_void *)(slot + LC_OPB + kr * LC_ROWB), 16, 0, 0);
            }
            if (CENTER) {                                               // 32 doubles from kcenter + k0 (16 used), by every loader wave alike
                const int64_t w = 2 * k0 + lane, wmax = 2 * K - 1;
                __builtin_amdgcn_global_load_lds((glb_void *)((const uint32_t *)kcenter + (w < wmax ? w : wmax)), (lds_void *)(slot + 2 * LC_OPB), 4, 0, 0);
            }
        };
#pragma unroll
        for (int c = 0; c < NSTAGE - 1; ++c)
            if (c < nsteps) issue(c);
        for (int c = 0; c < nsteps; ++c) {
            const int younger = (nsteps - 1 - c) < (NSTAGE - 2) ? (nsteps - 1 - c) : (NSTAGE - 2);     // stages issued after stage c
            if (NSTAGE >= 4 && younger >= 2)      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTAGE >= 4 ? 2 * PER_LOADER : 0) : "memory");
            else if (NSTAGE >= 3 && younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTAGE >= 3 ? PER_LOADER : 0) : "memory");
            else                   asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                               // B_c: stage c has landed, stage c-1 is read
            if (c + NSTAGE - 1 < nsteps) issue(c + NSTAGE - 1);
        }
        return;
    }

    // ----------------------------------------------------------------- consumer
    const int wm = wave >> 1, wn = wave & 1, kr = lane >> 4, c16 = lane & 15;
    const int a_rd = kr * LC_ROWB + (wm * 64 + c16) * 8, b_rd = LC_OPB + kr * LC_ROWB + (wn * 64 + c16) * 8, c_rd = 2 * LC_OPB + kr * 8;
    f64x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = (f64x4){0.0, 0.0, 0.0, 0.0};
    double av[2][4], bv[2][4], cen = 0.0;     // (one centre register: a block has formed its differences before the next value is read)
    auto read = [&](const char *slot, int kk, int set) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            av[set][t] = *(const double *)(slot + a_rd + kk * 4 * LC_ROWB + t * 128);
            bv[set][t] = *(const double *)(slot + b_rd + kk * 4 * LC_ROWB + t * 128);
        }
        if (CENTER) cen = *(const double *)(slot + c_rd + kk * 32);
    };
    // One block = the 16 MFMAs of four k (register set `set`), with `between` -- the reads of the NEXT four k into the other
    // set, at a stage's end the hand-over -- issued after the first four of them: whatever wait the compiler puts in front
    // of the block's first MFMA then only covers reads that were issued a whole block (~1 000 cycles) earlier, and the reads
    // issued here have twelve MFMAs to come back.  (Reads first, MFMAs after: the compiler waited lgkmcnt(0) at the loop
    // head -- behind the reads it had just issued; an LDS round trip per stage with an empty matrix pipe.)
    auto block = [&](int set, auto between) __attribute__((always_inline)) {
        double b[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) b[t] = CENTER ? bv[set][t] - cen : bv[set][t];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[0][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[set][0], b[ni], acc[0][ni], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        between();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 1; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[set][mi], b[ni], acc[mi][ni], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    // one stage: its k 0-3 are in register set 0 on entry; LAST = false leaves the next stage's k 0-3 there
    auto stage = [&](const char *slot, const char *next, auto last) __attribute__((always_inline)) {
        constexpr bool LAST = decltype(last)::value;
#pragma unroll
        for (int kk = 0; kk < LC_GK / 4; ++kk) {
            if (kk + 1 < LC_GK / 4) {
                block(kk & 1, [&]() __attribute__((always_inline)) { read(slot, kk + 1, (kk & 1) ^ 1); });
            } else {
                block(kk & 1, [&]() __attribute__((always_inline)) {
                    if constexpr (!LAST) {
                        __builtin_amdgcn_s_waitcnt(0xC07F);             // lgkmcnt(0): every read of this stage is back (the builtin, which the compiler's counter model reads)
                        __builtin_amdgcn_s_barrier();                   // B_{c+1}: the next stage has landed; this one may be refilled
                        read(next, 0, 0);                               // the twelve MFMAs that follow need nothing from it: the pipe runs through the barrier
                    }
                });
            }
        }
    };
    __builtin_amdgcn_s_barrier();                                       // B_0
    read(lc_ring, 0, 0);
    for (int c = 0; c + 1 < nsteps; ++c)
        stage(lc_ring + (c % NSTAGE) * LC_STAGEB, lc_ring + ((c + 1) % NSTAGE) * LC_STAGEB, std::false_type{});
    stage(lc_ring + ((nsteps - 1) % NSTAGE) * LC_STAGEB, nullptr, std::true_type{});
    // C/D of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int64_t r = row0 + wm * 64 + mi * 16 + (lane >> 4) + 4 * v;
                const int64_t cc = col0 + wn * 64 + ni * 16 + (lane & 15);
                if (r < M && cc < N) {
                    out[r * ldc + cc] = acc[mi][ni][v];
                    if (SYM && tl.bi != tl.bj && splits == 1) out[cc * ldc + r] = acc[mi][ni][v];
                }
            }
}

// out[i][j] = out[j][i] = part[0][i][j] + part[1][i][j] + ... (range order) for the tiles on or above the diagonal
// (blocks of 64 x 4; `tile` = the GEMM's tile size: a block below the diagonal of ITS tile grid was never written)
__global__ __launch_bounds__(256) void reduce_splits_kernel(const double *__restrict__ part, int splits, int64_t d, int tile,
                                                            double *__restrict__ out)
{
    const int64_t j = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63), i = (int64_t)blockIdx.y * 4 + (threadIdx.x >> 6);
    const int64_t ti = i / tile, tj = (int64_t)blockIdx.x * 64 / tile;
    if (ti > tj || j >= d || i >= d) return;
    double s = 0.0;
    for (int z = 0; z < splits; ++z) s += part[((int64_t)z * d + i) * d + j];
    out[i * d + j] = s;
    if (ti != tj) out[j * d + i] = s;
}

// K ranges of the Gram form: the 136 tiles of a 2048 x 2048 result are half a workgroup per CU with full-K loops; more,
// shorter workgroups fill the chip (at least 64 K-steps each, at most 16 ranges).  The count is chosen for whole rounds of
// the resident workgroups (256 of the loader/consumer kernel, one per CU): 16 ranges x 136 tiles = 8.5 rounds run as 9,
// 15 x 136 = 7.97 as 8.
static int gram_tile(int64_t d) { return d >= 1024 ? 128 : 64; }

static int gram_splits(int64_t d, int64_t n)
{
    const int64_t t = ceil_div(d, (int64_t)gram_tile(d)), tiles = t * (t + 1) / 2;
    const int64_t slots = gram_tile(d) == 128 ? 256 * LC_WG_PER_CU : 1024;
    int64_t max_s = n / (64 * GK) > 1 ? n / (64 * GK) : 1;
    if (max_s > 16) max_s = 16;
    if (tiles >= 4 * slots) return 1;
    int best = 1;
    double best_fill = 0.0;
    for (int64_t s = 1; s <= max_s; ++s) {
        const double fill = (double)(tiles * s) / (double)(ceil_div(tiles * s, slots) * slots);
        if (fill > best_fill + 0.02) { best_fill = fill; best = (int)s; }      // fewer ranges unless more fill the chip clearly better
    }
    return best;
}

static int64_t gram_partial_bytes(int64_t d, int64_t n)
{
    const int s = gram_splits(d, n);
    return s > 1 ? round_up((int64_t)s * d * d * 8, 256) : 0;
}

// Leading dimension of a transposed copy: a power-of-two row (2048 doubles = 16 KiB) puts the same 1-KiB column block of
// every k row into the same few memory channels -- both operands of the Gram form then queue on 4 channels of 16 and it ran
// at half the projection's rate; 256 bytes of padding per row rotate the channels.
static int64_t padded_ld(int64_t cols) { return round_up(cols, 4) + 32; }

// A transposed copy [K rows][padded_ld(cols)]: rows K .. round_up(K, 16) exist and are zero (the loader/consumer kernel
// reads whole stages of 16 k), then 1 KiB of slack (its clamped lanes never pass the last pair of a row; the slack is
// for the row-end reads of the register kernel's unchecked quads)
static int64_t transposed_bytes(int64_t k_rows, int64_t cols) { return round_up(padded_ld(cols) * round_up(k_rows, GK) * 8 + 1024, 256); }

static void launch_transpose(const double *src, int64_t rows, int64_t cols, const double *center, double *dst, hipStream_t s)
{
    const int64_t ld = padded_ld(rows), pad_rows = round_up(cols, GK) - cols;
    hipLaunchKernelGGL(transpose_f64_kernel, dim3((unsigned)ceil_div(cols, 32), (unsigned)ceil_div(rows, 32)), dim3(256), 0, s, src, rows,
                       cols, center, dst, ld);
    if (pad_rows) (void)hipMemsetAsync(dst + cols * ld, 0, (size_t)(pad_rows * ld * 8), s);
}

// > 64 KiB of dynamic LDS needs an opt-in per kernel and device
template <typename KERN>
static int lc_opt_in(KERN kern, bool *done)
{
    int dev = 0;
    MDX_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !done[dev]) {
        MDX_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LC_LDS));
        if (dev >= 0 && dev < 64) done[dev] = true;
    }
    return MDX_OK;
}

}  // namespace mdx

using namespace mdx;

extern "C" {

int64_t mdx_gram_f64_workspace(int64_t d, int64_t n)
{
    if (d <= 0 || n <= 0) return 0;
    return transposed_bytes(n, d) + gram_partial_bytes(d, n);          // the transposed (centred) input, then the partial tiles
}

int mdx_gram_f64(const double *a, int64_t d, int64_t n, const double *center, double *out, void *workspace, int64_t workspace_bytes,
                 void *stream)
{
    MDX_CHECK_ARG(a && out, "mdx_gram_f64: NULL pointer");
    MDX_CHECK_ARG(d > 0 && n > 0 && d <= (1ll << 21) && n < (1ll << 36), "mdx_gram_f64: d=%lld n=%lld", (long long)d, (long long)n);
    const int64_t need = mdx_gram_f64_workspace(d, n);
    if (!workspace || workspace_bytes < need) {
        set_error("mdx_gram_f64: workspace %lld B < required %lld B", (long long)workspace_bytes, (long long)need);
        return MDX_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const int64_t ld = padded_ld(d);
    double *at = (double *)workspace, *part = (double *)((char *)workspace + transposed_bytes(n, d));
    launch_transpose(a, d, n, center, at, s);                           // at [n, d] = (a - center)^T
    const int tile = gram_tile(d);
    const unsigned t = (unsigned)ceil_div(d, (int64_t)tile), t64 = (unsigned)ceil_div(d, (int64_t)64);
    const int splits = gram_splits(d, n);
    const int64_t ksplit = round_up(ceil_div(n, (int64_t)splits), GK);
    double *dst = splits > 1 ? part : out;
    if (tile == 128) {
        auto kern = gemm_f64_lc_kernel<true, false>;
        static bool opted[64];
        const int rc = lc_opt_in(kern, opted);
        if (rc != MDX_OK) return rc;
        hipLaunchKernelGGL(kern, dim3(t * (t + 1) / 2 * (unsigned)splits), dim3((4 + LC_LW) * 64), LC_LDS, s, (const double *)at, ld, (const double *)at, ld,
                           (const double *)nullptr, dst, d, d, n, ksplit, d, (int)t, (int)t, splits);
    } else {
        hipLaunchKernelGGL((gemm_f64_kernel<true, 32>), dim3(t * (t + 1) / 2, 1, (unsigned)splits), dim3(256), 0, s, (const double *)at, ld, (const double *)at,
                           ld, (const double *)nullptr, dst, d, d, n, ksplit, d);
    }
    if (splits > 1) hipLaunchKernelGGL(reduce_splits_kernel, dim3(t64, (unsigned)ceil_div(d, (int64_t)4)), dim3(256), 0, s, (const double *)part, splits, d, tile, out);
    MDX_LAUNCH_CHECK();
    return MDX_OK;
}

int64_t mdx_project_f64_workspace(int64_t dout, int64_t d)
{
    if (dout <= 0 || d <= 0) return 0;
    return transposed_bytes(d, dout);                                   // p transposed
}

// x [d, n] (float64, column = one descriptor): every column divided by (its L2 norm + eps), in place: `X / (norm(X, axis 0) + 1e-6)`
// of whitenapply (whiten.py:10-11) for float64 inputs.  A thread owns a column (rows are read coalesced across the columns).
__global__ __launch_bounds__(256) void l2n_cols_f64_kernel(double *__restrict__ x, int64_t d, int64_t n, double eps)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    double ss = 0.0;
    for (int64_t i = 0; i < d; ++i) ss += x[i * n + j] * x[i * n + j];
    const double den = sqrt(ss) + eps;
    for (int64_t i = 0; i < d; ++i) x[i * n + j] /= den;
}

int mdx_l2n_cols_f64(double *x, int64_t d, int64_t n, double eps, void *stream)
{
    MDX_CHECK_ARG(x && d > 0 && n > 0 && eps >= 0.0, "mdx_l2n_cols_f64: bad arguments");
    hipLaunchKernelGGL(l2n_cols_f64_kernel, dim3((unsigned)ceil_div(n, (int64_t)256)), dim3(256), 0, (hipStream_t)stream, x, d, n, eps);
    MDX_LAUNCH_CHECK();
    return MDX_OK;
}

int mdx_project_f64(const double *p, int64_t dout, int64_t d, const double *x, int64_t n, const double *center, double *out,
                    void *workspace, int64_t workspace_bytes, void *stream)
{
    MDX_CHECK_ARG(p && x && out, "mdx_project_f64: NULL pointer");
    MDX_CHECK_ARG(dout > 0 && d > 0 && n > 0 && dout < (1ll << 20) && d < (1ll << 20) && n < (1ll << 36), "mdx_project_f64: dout=%lld d=%lld n=%lld",
                  (long long)dout, (long long)d, (long long)n);
    const int64_t need = mdx_project_f64_workspace(dout, d);
    if (!workspace || workspace_bytes < need) {
        set_error("mdx_project_f64: workspace %lld B < required %lld B", (long long)workspace_bytes, (long long)need);
        return MDX_ERR_WORKSPACE;
    }
    MDX_CHECK_ARG(ceil_div(dout, (int64_t)64) < 65536, "mdx_project_f64: too many tiles");
    hipStream_t s = (hipStream_t)stream;
    double *pt = (double *)workspace;
    launch_transpose(p, dout, d, nullptr, pt, s);                       // pt [d, dout]
    if (dout >= 1024 && n >= 1024) {
        const int64_t ne = n & ~(int64_t)1;                             // the loader/consumer kernel moves column PAIRS; an odd last column: below
        const int T_rows = (int)ceil_div(dout, (int64_t)128), T_cols = (int)ceil_div(ne, (int64_t)128);
        const int64_t slots = (int64_t)ceil_div(T_rows, 4 * LC_WG_PER_CU) * ceil_div(T_cols, 8) * 32 * LC_WG_PER_CU;     // whole blocks of tiles
        MDX_CHECK_ARG(slots < (1ll << 31), "mdx_project_f64: too many tiles");
        static bool opted[2][64];
        int rc;
        if (center) {
            auto kern = gemm_f64_lc_kernel<false, true>;
            if ((rc = lc_opt_in(kern, opted[1])) != MDX_OK) return rc;
            hipLaunchKernelGGL(kern, dim3((unsigned)slots), dim3((4 + LC_LW) * 64), LC_LDS, s, (const double *)pt, padded_ld(dout), x, n, center, out, dout, ne, d, d,
                               n, T_rows, T_cols, 1);
        } else {
            auto kern = gemm_f64_lc_kernel<false, false>;
            if ((rc = lc_opt_in(kern, opted[0])) != MDX_OK) return rc;
            hipLaunchKernelGGL(kern, dim3((unsigned)slots), dim3((4 + LC_LW) * 64), LC_LDS, s, (const double *)pt, padded_ld(dout), x, n, center, out, dout, ne, d, d,
                               n, T_rows, T_cols, 1);
        }
        if (ne < n)
            hipLaunchKernelGGL((gemm_f64_kernel<false, 32>), dim3(1, (unsigned)ceil_div(dout, (int64_t)64)), dim3(256), 0, s, (const double *)pt,
                               padded_ld(dout), x + ne, n, center, out + ne, dout, (int64_t)1, d, d, n);
    } else {
        hipLaunchKernelGGL((gemm_f64_kernel<false, 32>), dim3((unsigned)ceil_div(n, (int64_t)64), (unsigned)ceil_div(dout, (int64_t)64)), dim3(256), 0, s,
                           (const double *)pt, padded_ld(dout), x, n, center, out, dout, n, d, d, n);
    }
    MDX_LAUNCH_CHECK();
    return MDX_OK;
}

}  // extern "C"
