// Database shard ("index") in MFMA-fragment order + the similarity kernel.
//
// Replaces `scores = np.dot(vecs.T, qvecs)` (mdir/components/optim/score/
// cirscore.py:69) and, with the whitening matrix as the "database", the projection
// of CirtorchWhiten.postprocess (mdir/components/data/wrapper.py:193-195).
//
// Data layout: rows are grouped in tiles of 16, the dimension in blocks
// of 16; tile (rt, kb) is 1 KiB = 64 lanes x float4, stored at ((rt*KB)+kb)*1024 B.
// Lane l = 16*g + j holds, in element t (0..3), value (row 16*rt+j, k 16*kb+4*t+g).
// One wave-wide 16-B load therefore fetches one fully coalesced KiB and element t
// of every lane is exactly the B (or A) operand of v_mfma_f32_16x16x4_f32 number t
// of that k-block, with k ascending inside each MFMA and across MFMAs -- so every
// score is the k = 0..D-1 fused-multiply-add chain stated in oracle/chain.c.
#include <stdarg.h>

#include "mdx_common.h"
#include "mdx_scores_kernel.h"
#include "mdx_scores_split_kernel.h"
#include "mdx_scores_stream_kernel.h"

namespace mdx {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

// ---------------------------------------------------------------------------
// Re-tiling: an fp32 matrix [rows, k] in either layout of the API -> fragment-order tiles (the index build, the queries
// of every call, the whitening matrix).  Rows >= n and k >= d read 0; `center[k]` (or nothing) is subtracted on the way.
//   fp32 tile = 16 rows x 16 k: lane (g, j), element t = (row j, k 16 kb + 4 t + g)
//   fp16 tile = 16 rows x 32 k: lane (g, j), element e = (row j, k 32 kb + 8 g + e), round-to-nearest-even
// Round 1-3 let every lane fetch its own elements from the source: a wave-load was 16 pieces of 16 B (row-major sources)
// or 4 pieces of 64 B (dimension-major) -- the build of the 8.2 GB shard ran at 0.86 TB/s read + write (19 ms; rocprofv3,
// profiles/r04_summary.md).  Here a workgroup reads a block of the source in whole 1-KiB runs (64 lanes x 16 B along the
// contiguous direction), parks it in LDS and every wave assembles tiles from there; the row stride of the LDS block makes the
// 64 lanes of an assembling read hit 64 different banks.  The tiles leave as before, one coalesced KiB per wave-store.
//   ROWMAJOR  (element (row, k) at src[row * d + k]):  block = 16 rows x 256 k
//   otherwise (element (row, k) at src[k * n + row]):  block = 32 k x 256 rows (two fp32 k blocks or one fp16 k block)
// ---------------------------------------------------------------------------
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));     // a 16-byte load at dword alignment (any n, d)
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <bool F16, bool ROWMAJOR>
__global__ __launch_bounds__(256) void retile_block_kernel(const float *__restrict__ src, int64_t n, int64_t d, const float *__restrict__ center,
                                                           f32x4 *__restrict__ tiles, int64_t RT, int64_t KB, uint32_t *__restrict__ absmax,
                                                           unsigned inner)
{
    // block id -> (bx, by) = (row tile, k span) or (row span, k block); the index that walks ALONG the source's contiguous
    // direction is the fast one: consecutive blocks read adjacent KiB of the same lines
    const unsigned bx = ROWMAJOR ? blockIdx.x / inner : blockIdx.x % inner, by = ROWMAJOR ? blockIdx.x % inner : blockIdx.x / inner;
    constexpr int TK = F16 ? 32 : 16;                       // k per tile
    constexpr int KBS = (!ROWMAJOR && !F16) ? 2 : 1;        // dimension-major fp32: two k blocks per block, so that a row tile's two tiles leave as one 2-KiB run
    constexpr int LINES = ROWMAJOR ? 16 : TK * KBS;         // 1-KiB runs of the source per block: rows, or k rows
    constexpr int LD = ROWMAJOR ? 260 : (F16 ? 258 : 272);  // floats per line in LDS (bank spread of the assembling reads)
    __shared__ __attribute__((aligned(16))) float blk[LINES * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    uint32_t best = 0;
    // ---- the block, in runs of 1 KiB: line `l`, floats 4 * lane .. 4 * lane + 3
    const int64_t line0 = ROWMAJOR ? (int64_t)bx * 16 : (int64_t)by * TK * KBS;      // first row / first k
    const int64_t col0 = (ROWMAJOR ? (int64_t)by : (int64_t)bx) * 256 + 4 * lane;   // first k / first row of this lane
    const int64_t nlines = ROWMAJOR ? n : d, ncols = ROWMAJOR ? d : n;
    const int64_t ld_src = ncols;
    float cen[4] = {0.f, 0.f, 0.f, 0.f};
    if (ROWMAJOR && center) {
#pragma unroll
        for (int e = 0; e < 4; ++e) cen[e] = col0 + e < d ? center[col0 + e] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < LINES / 4; ++i) {
        const int l = wave * (LINES / 4) + i;
        const int64_t line = line0 + l;
        float x[4] = {0.f, 0.f, 0.f, 0.f};
        if (line < nlines) {
            const float *p = src + line * ld_src + col0;
            if (col0 + 3 < ncols) {
                const f32x4u v = *(const f32x4u *)p;
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = v[e];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = col0 + e < ncols ? p[e] : 0.f;
            }
            if (center) {
                const float ck = ROWMAJOR ? 0.f : center[line];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (col0 + e < ncols) x[e] -= ROWMAJOR ? cen[e] : ck;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t u = __float_as_uint(x[e]) & 0x7FFFFFFFu;
            if (u < 0x7F800000u && u > best) best = u;
        }
        *(f32x2 *)&blk[l * LD + 4 * lane] = (f32x2){x[0], x[1]};
        *(f32x2 *)&blk[l * LD + 4 * lane + 2] = (f32x2){x[2], x[3]};
    }
    __syncthreads();
    // ---- tiles: 16 (fp16: 8) of them per block, 4 (2) per wave
    constexpr int TILES = ROWMAJOR ? 256 / TK : 16 * KBS;       // dimension-major: tile tl = (row tile tl / KBS, k block tl % KBS)
#pragma unroll
    for (int i = 0; i < TILES / 4; ++i) {
        const int tl = wave * (TILES / 4) + i;                          // tile of the block: along k (ROWMAJOR) or along rows
        const int rl = ROWMAJOR ? 0 : tl / KBS, kl = ROWMAJOR ? 0 : tl % KBS;       // dimension-major: row tile and k block inside the block
        const int64_t rt = ROWMAJOR ? (int64_t)bx : (int64_t)bx * 16 + rl;
        const int64_t kb = ROWMAJOR ? (int64_t)by * TILES + tl : (int64_t)by * KBS + kl;
        if (rt >= RT || kb >= KB) continue;
        f32x4 out;
        if constexpr (F16) {
            f16x8 h;
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = (_Float16)(ROWMAJOR ? blk[j * LD + tl * 32 + 8 * g + e] : blk[(8 * g + e) * LD + rl * 16 + j]);
            out = __builtin_bit_cast(f32x4, h);
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) out[t] = ROWMAJOR ? blk[j * LD + tl * 16 + 4 * t + g] : blk[(kl * 16 + 4 * t + g) * LD + rl * 16 + j];
        }
        tiles[(rt * KB + kb) * 64 + lane] = out;
    }
    if (absmax) {               // the shard's largest finite magnitude (index build only): the scale of MDX_F32_SPLIT2
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t other = (uint32_t)__shfl_xor((int)best, o, 64);
            best = other > best ? other : best;
        }
        // one device-wide atomic per wave would be 2 M of them on ONE address for a 1 M x 2048 shard (~10 ns each: 20 of the
        // build's 23 ms); a wave first looks at the cell -- a stale, smaller value only costs an atomic that changes nothing
        if (lane == 0 && best > *(volatile uint32_t *)absmax) atomicMax(absmax, best);
    }
}

// Loader/consumer kernel (4 MFMA waves + 4 LDS-DMA loader waves), chunks of 2 tiles along k,
// ring of 3 stages.  R = 2 row tiles per consumer (128-row workgroups, (QT+8)*6 KiB of LDS)
// for shards of >= 32 768 rows; R = 1 (64-row workgroups) below, so that small shards still
// spread over the CUs -- measured crossover (round 2; the harness is in the history at commit 47a9fe2).
constexpr int LC_KC = 2, LC_NSTAGE = 3;
#ifndef MDX_SCORES_PIPE_DEFAULT
#define MDX_SCORES_PIPE_DEFAULT 1      // profiles/r05_scores_schedule.md: -1.0 ... -1.2 % (bit-equal); MDX_SCORES_PIPE=0 keeps the round-4 schedule
#endif

// > 64 KiB of dynamic LDS needs an opt-in per kernel and device: done once, not on every launch
static int lds_opt_in(const void *kern, int lds, bool *done)
{
    int dev = 0;
    MDX_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !done[dev]) {
        MDX_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        if (dev >= 0 && dev < 64) done[dev] = true;
    }
    return MDX_OK;
}

// where a launch puts its scores: rows == nullptr: the plain [nq, n] matrix `out`; else query q's run goes to rows[q] + col0
// (device array of nq row pointers: mdx_scores_p2p)
struct Route { float *const *rows; int64_t col0; };

template <int QT, int R, typename MM, int QR = 0, bool RM = false>
static int launch_scores_lc(const f32x4 *db, const f32x4 *qt, float *out, int64_t n, int64_t RT,
                            int KB, int nq_valid, hipStream_t s, int passes = 1, int64_t ld = 0, Route route = Route{nullptr, 0})
{
    constexpr int lds = LC_NSTAGE * (QT + QR + 4 * R) * LC_KC * 1024;
    const int64_t blocks = ceil_div(RT, (int64_t)4 * R);
    if constexpr (MM::STEPS == 4 && !RM) {
        if (route.rows) {           // routed epilogue (the shipped schedule of each shape: PIPE for 128-row workgroups)
            auto kr = scores_lc_kernel<QT, R, LC_KC, LC_NSTAGE, 2, MM, QR, 4, false, (R == 2 ? 1 : 0), 4, true>;
            static bool opted_r[64];
            int rcr = lds_opt_in((const void *)kr, lds, opted_r);
            if (rcr != MDX_OK) return rcr;
            hipLaunchKernelGGL(kr, dim3((unsigned)blocks, (unsigned)passes), dim3(512), lds, s, db, qt, out, n, KB, nq_valid, ld, route.rows,
                               route.col0);
            return MDX_OK;
        }
    }
    if (route.rows) {
        set_error("routed scores need an fp32 tiled shard");
        return MDX_ERR_INVALID;
    }
    if constexpr (MM::STEPS == 4 && !RM && R == 2) {
        // the pipelined consumer (PIPE): MDX_SCORES_PIPE=0/1 picks the form per launch (A/B in one process: tools/chain_power_probe.py)
        const char *e = getenv("MDX_SCORES_PIPE");
        if (e ? e[0] == '1' : MDX_SCORES_PIPE_DEFAULT) {
            auto kp = scores_lc_kernel<QT, R, LC_KC, LC_NSTAGE, 2, MM, QR, 4, RM, 1>;
            static bool opted_p[64];
            int rcp = lds_opt_in((const void *)kp, lds, opted_p);
            if (rcp != MDX_OK) return rcp;
            hipLaunchKernelGGL(kp, dim3((unsigned)blocks, (unsigned)passes), dim3(512), lds, s, db, qt, out, n, KB, nq_valid, ld,
                               (float *const *)nullptr, (int64_t)0);
            return MDX_OK;
        }
    }
    auto kern = scores_lc_kernel<QT, R, LC_KC, LC_NSTAGE, 2, MM, QR, 4, RM>;   // 2 = non-temporal database stream
    static bool opted[64];
    int rc = lds_opt_in((const void *)kern, lds, opted);
    if (rc != MDX_OK) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks, (unsigned)passes), dim3(512), lds, s, db, qt, out, n, KB, nq_valid, ld,
                       (float *const *)nullptr, (int64_t)0);
    return MDX_OK;
}

// fp16 shards whose k range is a whole number of four-chunk stages (d_pad % 128 == 0: every descriptor size of the path) take the
// register-streaming kernel (mdx_scores_stream_kernel.h; bit-identical to the ring kernel, 6-10 % faster at 1 M x 2048);
// MDX_F16_RING=1 keeps the ring kernel (A/B, tests)
static bool f16_streams(int KB)
{
    static const bool ring_only = getenv("MDX_F16_RING") != nullptr;
    return !ring_only && KB % STREAM_PF == 0;
}

template <int QT, int R>
static int launch_f16_stream(const f32x4 *db, const f32x4 *qt, float *out, int64_t n, int64_t RT, int KB, int nq_valid, hipStream_t s, int passes)
{
    constexpr int WGS = QT <= 5 ? 3 : 2;
    auto kern = scores_f16_stream_kernel<QT, R, WGS>;
    constexpr int lds = stream_lds_bytes<QT, R>();
    static bool opted[64];
    int rc = lds_opt_in((const void *)kern, lds, opted);
    if (rc != MDX_OK) return rc;
    const int64_t blocks = ceil_div(RT, (int64_t)STREAM_CW * R);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks, (unsigned)passes), dim3(STREAM_CW * 64), lds, s, db, qt, out, n, KB, nq_valid);
    return MDX_OK;
}

// mode bit0: R = 2 (else 1), bit1: fp16 shard
template <int QT>
static int launch_qt(int mode, const f32x4 *db, const f32x4 *q, float *out, int64_t n, int64_t RT,
                     int KB, int nq_valid, hipStream_t s, int passes = 1, Route route = Route{nullptr, 0})
{
    switch (mode) {
        case 0: return launch_scores_lc<QT, 1, MmaF32>(db, q, out, n, RT, KB, nq_valid, s, passes, 0, route);
        case 1: return launch_scores_lc<QT, 2, MmaF32>(db, q, out, n, RT, KB, nq_valid, s, passes, 0, route);
        case 2:
            if (f16_streams(KB)) return launch_f16_stream<QT, 1>(db, q, out, n, RT, KB, nq_valid, s, passes);
            return launch_scores_lc<QT, 1, MmaF16>(db, q, out, n, RT, KB, nq_valid, s, passes);
        default:
            if (f16_streams(KB)) return launch_f16_stream<QT, 2>(db, q, out, n, RT, KB, nq_valid, s, passes);
            return launch_scores_lc<QT, 2, MmaF16>(db, q, out, n, RT, KB, nq_valid, s, passes);
    }
}

// `qt` query tiles of which the last holds <= 8 queries: qt-1 full tiles on the 16x16x4 MFMA + the leftover
// tile on v_mfma_f32_4x4x1 (fp32 shards with 128-row workgroups)
static int dispatch_leftover(int qt, const f32x4 *db, const f32x4 *q, float *out, int64_t n, int64_t RT, int KB,
                             int nq_valid, hipStream_t s, Route route = Route{nullptr, 0})
{
    switch (qt) {
        case 2: return launch_scores_lc<1, 2, MmaF32, 1>(db, q, out, n, RT, KB, nq_valid, s, 1, 0, route);
        case 3: return launch_scores_lc<2, 2, MmaF32, 1>(db, q, out, n, RT, KB, nq_valid, s, 1, 0, route);
        case 4: return launch_scores_lc<3, 2, MmaF32, 1>(db, q, out, n, RT, KB, nq_valid, s, 1, 0, route);
        case 5: return launch_scores_lc<4, 2, MmaF32, 1>(db, q, out, n, RT, KB, nq_valid, s, 1, 0, route);
        case 6: return launch_scores_lc<5, 2, MmaF32, 1>(db, q, out, n, RT, KB, nq_valid, s, 1, 0, route);
        case 7: return launch_scores_lc<6, 2, MmaF32, 1>(db, q, out, n, RT, KB, nq_valid, s, 1, 0, route);
        default: return launch_scores_lc<7, 2, MmaF32, 1>(db, q, out, n, RT, KB, nq_valid, s, 1, 0, route);
    }
}

static int dispatch_qt(int qt, int mode, const f32x4 *db, const f32x4 *q, float *out, int64_t n,
                       int64_t RT, int KB, int nq_valid, hipStream_t s, Route route = Route{nullptr, 0})
{
    switch (qt) {
        case 1: return launch_qt<1>(mode, db, q, out, n, RT, KB, nq_valid, s, 1, route);
        case 2: return launch_qt<2>(mode, db, q, out, n, RT, KB, nq_valid, s, 1, route);
        case 3: return launch_qt<3>(mode, db, q, out, n, RT, KB, nq_valid, s, 1, route);
        case 4: return launch_qt<4>(mode, db, q, out, n, RT, KB, nq_valid, s, 1, route);
        case 5: return launch_qt<5>(mode, db, q, out, n, RT, KB, nq_valid, s, 1, route);
        case 6: return launch_qt<6>(mode, db, q, out, n, RT, KB, nq_valid, s, 1, route);
        case 7: return launch_qt<7>(mode, db, q, out, n, RT, KB, nq_valid, s, 1, route);
        default: return launch_qt<8>(mode, db, q, out, n, RT, KB, nq_valid, s, 1, route);
    }
}

// The same kernels on a row-major database read where it lies (RM = true; mdx_scores_rowmajor)
template <int R>
static int dispatch_rowmajor(int qt, bool leftover, const float *db, int64_t ld, const f32x4 *q, float *out, int64_t n, int64_t RT, int KB,
                             int nq_valid, hipStream_t s)
{
    const f32x4 *d4 = (const f32x4 *)db;
    if constexpr (R == 2) {
        if (leftover) {
            switch (qt) {
                case 2: return launch_scores_lc<1, 2, MmaF32, 1, true>(d4, q, out, n, RT, KB, nq_valid, s, 1, ld);
                case 3: return launch_scores_lc<2, 2, MmaF32, 1, true>(d4, q, out, n, RT, KB, nq_valid, s, 1, ld);
                case 4: return launch_scores_lc<3, 2, MmaF32, 1, true>(d4, q, out, n, RT, KB, nq_valid, s, 1, ld);
                case 5: return launch_scores_lc<4, 2, MmaF32, 1, true>(d4, q, out, n, RT, KB, nq_valid, s, 1, ld);
                case 6: return launch_scores_lc<5, 2, MmaF32, 1, true>(d4, q, out, n, RT, KB, nq_valid, s, 1, ld);
                case 7: return launch_scores_lc<6, 2, MmaF32, 1, true>(d4, q, out, n, RT, KB, nq_valid, s, 1, ld);
                default: return launch_scores_lc<7, 2, MmaF32, 1, true>(d4, q, out, n, RT, KB, nq_valid, s, 1, ld);
            }
        }
    }
    switch (qt) {
        case 1: return launch_scores_lc<1, R, MmaF32, 0, true>(d4, q, out, n, RT, KB, nq_valid, s, 1, ld);
        case 2: return launch_scores_lc<2, R, MmaF32, 0, true>(d4, q, out, n, RT, KB, nq_valid, s, 1, ld);
        case 3: return launch_scores_lc<3, R, MmaF32, 0, true>(d4, q, out, n, RT, KB, nq_valid, s, 1, ld);
        case 4: return launch_scores_lc<4, R, MmaF32, 0, true>(d4, q, out, n, RT, KB, nq_valid, s, 1, ld);
        case 5: return launch_scores_lc<5, R, MmaF32, 0, true>(d4, q, out, n, RT, KB, nq_valid, s, 1, ld);
        case 6: return launch_scores_lc<6, R, MmaF32, 0, true>(d4, q, out, n, RT, KB, nq_valid, s, 1, ld);
        case 7: return launch_scores_lc<7, R, MmaF32, 0, true>(d4, q, out, n, RT, KB, nq_valid, s, 1, ld);
        default: return launch_scores_lc<8, R, MmaF32, 0, true>(d4, q, out, n, RT, KB, nq_valid, s, 1, ld);
    }
}

// Split-precision launches (mdx_scores_split_kernel.h): 8 consumer waves x R row tiles + 4 loader waves, one workgroup
// per CU, ring of 3 stages of (3 QT + 16 R) KiB.  Up to SPLIT_QT query tiles per workgroup; more queries = more passes (grid.y).
constexpr int SPLIT_CW = 8, SPLIT_NSTAGE = 3, SPLIT_QT = 5;

template <int QT, int R>
static int launch_split3(const f32x4 *db, const u32x4 *qp, float *out, int64_t n, int64_t RT, int KB, int QT_total,
                         int qt_first, int nq_valid, hipStream_t s, int passes)
{
    auto kern = scores_split3_kernel<QT, R, SPLIT_NSTAGE, SPLIT_CW>;
    constexpr int lds = SPLIT_NSTAGE * (3 * QT + 2 * SPLIT_CW * R) * 1024;
    static bool opted[64];
    int rc = lds_opt_in((const void *)kern, lds, opted);
    if (rc != MDX_OK) return rc;
    const int64_t blocks = ceil_div(RT, (int64_t)SPLIT_CW * R);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks, (unsigned)passes), dim3(SPLIT_CW * 64 + 256), lds, s, db, qp, out, n, KB,
                       QT_total, qt_first, nq_valid);
    return MDX_OK;
}

template <int R>
static int dispatch_split3(int qt, const f32x4 *db, const u32x4 *qp, float *out, int64_t n, int64_t RT, int KB, int QT_total,
                           int qt_first, int nq_valid, hipStream_t s, int passes)
{
    switch (qt) {
        case 1: return launch_split3<1, R>(db, qp, out, n, RT, KB, QT_total, qt_first, nq_valid, s, passes);
        case 2: return launch_split3<2, R>(db, qp, out, n, RT, KB, QT_total, qt_first, nq_valid, s, passes);
        case 3: return launch_split3<3, R>(db, qp, out, n, RT, KB, QT_total, qt_first, nq_valid, s, passes);
        case 4: return launch_split3<4, R>(db, qp, out, n, RT, KB, QT_total, qt_first, nq_valid, s, passes);
        default: return launch_split3<5, R>(db, qp, out, n, RT, KB, QT_total, qt_first, nq_valid, s, passes);
    }
}

template <int QT, int R>
static int launch_split2(const f32x4 *db, const u32x4 *qp, float *out, int64_t n, int64_t RT, int KB, int QT_total, int qt_first,
                         int nq_valid, float db_scale, const uint32_t *q_cell, hipStream_t s, int passes)
{
    auto kern = scores_split2_kernel<QT, R, SPLIT_NSTAGE, SPLIT_CW>;
    constexpr int lds = SPLIT_NSTAGE * (2 * QT + 2 * SPLIT_CW * R) * 1024;
    static bool opted[64];
    int rc = lds_opt_in((const void *)kern, lds, opted);
    if (rc != MDX_OK) return rc;
    const int64_t blocks = ceil_div(RT, (int64_t)SPLIT_CW * R);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks, (unsigned)passes), dim3(SPLIT_CW * 64 + 256), lds, s, db, qp, out, n, KB, QT_total,
                       qt_first, nq_valid, db_scale, q_cell);
    return MDX_OK;
}

template <int R>
static int dispatch_split2(int qt, const f32x4 *db, const u32x4 *qp, float *out, int64_t n, int64_t RT, int KB, int QT_total, int qt_first,
                           int nq_valid, float db_scale, const uint32_t *q_cell, hipStream_t s, int passes)
{
    switch (qt) {
        case 1: return launch_split2<1, R>(db, qp, out, n, RT, KB, QT_total, qt_first, nq_valid, db_scale, q_cell, s, passes);
        case 2: return launch_split2<2, R>(db, qp, out, n, RT, KB, QT_total, qt_first, nq_valid, db_scale, q_cell, s, passes);
        case 3: return launch_split2<3, R>(db, qp, out, n, RT, KB, QT_total, qt_first, nq_valid, db_scale, q_cell, s, passes);
        case 4: return launch_split2<4, R>(db, qp, out, n, RT, KB, QT_total, qt_first, nq_valid, db_scale, q_cell, s, passes);
        default: return launch_split2<5, R>(db, qp, out, n, RT, KB, QT_total, qt_first, nq_valid, db_scale, q_cell, s, passes);
    }
}

}  // namespace mdx

using namespace mdx;

struct mdx_index {
    f32x4 *tiles;
    int64_t n, d, d_pad, RT, RT_pad, KB, row_offset;
    int64_t bytes;
    int storage;        // MDX_F32 / MDX_F16
    bool owns;          // tiles came from hipMalloc here (mdx_index_create*) and are freed on destroy; false: the caller's memory
    uint32_t max_bits;  // fp32 shards: bit pattern of the largest finite |x| (read back once at creation): the scale of MDX_F32_SPLIT2
};

extern "C" {

int mdx_abi_version(void) { return MDX_ABI_VERSION; }

int mdx_capture_recover(void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone) {
        hipGraph_t g = nullptr;
        (void)hipStreamEndCapture(s, &g);       // an invalidated capture ends with an error and no graph
        if (g) (void)hipGraphDestroy(g);
    }
    (void)hipGetLastError();                    // the sticky (non-fatal) error of the failed capture
    (void)hipGetLastError();
    return MDX_OK;
}
const char *mdx_last_error(void) { return mdx::g_err; }

static int retile(const float *src, int64_t n, int64_t d, int layout, const float *center,
                  f32x4 *tiles, int64_t RT, int64_t KB, hipStream_t s, int storage = MDX_F32, uint32_t *absmax = nullptr)
{
    const bool rowmajor = layout != MDX_DIM_MAJOR, f16 = storage == MDX_F16;
    // blocks: 16 rows x 256 k (row-major) or 32 k x 256 rows (dimension-major: two fp32 k blocks or one fp16 k block)
    const int64_t gx = rowmajor ? RT : ceil_div(RT, (int64_t)16);
    const int64_t gy = rowmajor ? ceil_div(KB * (f16 ? 32 : 16), (int64_t)256) : (f16 ? KB : ceil_div(KB, (int64_t)2));
    MDX_CHECK_ARG(gx * gy < (1ll << 31), "matrix too large to re-tile in one launch");
    const dim3 grid((unsigned)(gx * gy)), block(256);
    const unsigned inner = (unsigned)(rowmajor ? gy : gx);
    if (f16 && rowmajor)       hipLaunchKernelGGL((retile_block_kernel<true, true>), grid, block, 0, s, src, n, d, center, tiles, RT, KB, absmax, inner);
    else if (f16)              hipLaunchKernelGGL((retile_block_kernel<true, false>), grid, block, 0, s, src, n, d, center, tiles, RT, KB, absmax, inner);
    else if (rowmajor)         hipLaunchKernelGGL((retile_block_kernel<false, true>), grid, block, 0, s, src, n, d, center, tiles, RT, KB, absmax, inner);
    else                       hipLaunchKernelGGL((retile_block_kernel<false, false>), grid, block, 0, s, src, n, d, center, tiles, RT, KB, absmax, inner);
    MDX_LAUNCH_CHECK();
    return MDX_OK;
}

int mdx_index_create(mdx_index **out, const float *src, int64_t n, int64_t d, int layout,
                     int64_t row_offset, void *stream)
{
    return mdx_index_create_ex(out, src, n, d, layout, row_offset, MDX_F32, stream);
}

static void index_geometry(mdx_index *ix, int64_t n, int64_t d, int storage)
{
    ix->n = n;
    ix->d = d;
    ix->storage = storage;
    ix->d_pad = round_up(d, 64);                                  // 4 fp32 or 2 fp16 k-blocks
    ix->KB = ix->d_pad / (storage == MDX_F16 ? 32 : TILE_K);
    ix->RT = ceil_div(n, TILE_ROWS);
    // every wave of every workgroup has a tile to read: 8 row tiles per workgroup, 16 for the split-precision kernel (fp32 shards)
    ix->RT_pad = round_up(ix->RT, storage == MDX_F32 ? 16 : 8);
    ix->bytes = ix->RT_pad * ix->KB * 1024;
    ix->max_bits = 0;
}

int64_t mdx_index_bytes(int64_t n, int64_t d, int storage)
{
    if (n <= 0 || d <= 0 || (storage != MDX_F32 && storage != MDX_F16)) return 0;
    mdx_index g;
    index_geometry(&g, n, d, storage);
    return g.bytes + 256;                                          // + one word behind the tiles: the |x| maximum
}

int mdx_index_create_ex(mdx_index **out, const float *src, int64_t n, int64_t d, int layout,
                        int64_t row_offset, int storage, void *stream)
{
    return mdx_index_create_in(out, src, n, d, layout, row_offset, storage, nullptr, 0, stream);
}

int mdx_index_create_in(mdx_index **out, const float *src, int64_t n, int64_t d, int layout, int64_t row_offset, int storage,
                        void *memory, int64_t memory_bytes, void *stream)
{
    MDX_CHECK_ARG(out && src, "mdx_index_create: NULL pointer");
    MDX_CHECK_ARG(storage == MDX_F32 || storage == MDX_F16, "mdx_index_create: storage %d", storage);
    MDX_CHECK_ARG(n > 0 && d > 0, "mdx_index_create: n=%lld d=%lld must be positive",
                  (long long)n, (long long)d);
    MDX_CHECK_ARG(layout == MDX_DIM_MAJOR || layout == MDX_ROW_MAJOR, "mdx_index_create: layout %d",
                  layout);
    mdx_index *ix = new mdx_index();
    index_geometry(ix, n, d, storage);
    ix->row_offset = row_offset;
    ix->owns = memory == nullptr;
    if (memory) {
        if (memory_bytes < ix->bytes + 256 || ((uintptr_t)memory & 255)) {
            set_error("mdx_index_create_in: %lld bytes at a 256-byte boundary needed (mdx_index_bytes), got %lld at %p", (long long)ix->bytes + 256,
                      (long long)memory_bytes, memory);
            delete ix;
            return MDX_ERR_WORKSPACE;
        }
        ix->tiles = (f32x4 *)memory;
    } else {
        hipError_t e = hipMalloc((void **)&ix->tiles, (size_t)ix->bytes + 256);
        if (e != hipSuccess) {
            set_error("mdx_index_create: hipMalloc(%lld bytes) failed: %s", (long long)ix->bytes,
                      hipGetErrorString(e));
            delete ix;
            return MDX_ERR_NOMEM;
        }
    }
    hipStream_t s = (hipStream_t)stream;
    uint32_t *cell = storage == MDX_F32 ? (uint32_t *)((char *)ix->tiles + ix->bytes) : nullptr;
    if (cell && hipMemsetAsync(cell, 0, 4, s) != hipSuccess) cell = nullptr;
    int rc = retile(src, n, d, layout, nullptr, ix->tiles, ix->RT_pad, ix->KB, s, storage, cell);
    if (rc == MDX_OK && cell) {
        // index creation is the call of this path that may synchronise: the maximum comes back with the build
        if (hipMemcpyAsync(&ix->max_bits, cell, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
            set_error("mdx_index_create: reading the shard's maximum failed: %s", hipGetErrorString(hipGetLastError()));
            rc = MDX_ERR_RUNTIME;
        }
    }
    if (rc != MDX_OK) {
        (void)hipStreamSynchronize(s);
        if (ix->owns) (void)hipFree(ix->tiles);
        delete ix;
        return rc;
    }
    // index creation is the one call of the ranking path that may synchronise: settle here how the sort ranks inside a
    // wave on this device (mdx_rank.hip), so that the enqueue-only ranking calls -- also captured ones -- find the answer
    (void)probe_lds_order(s);
    *out = ix;
    return MDX_OK;
}

int mdx_index_destroy(mdx_index *ix)
{
    if (!ix) return MDX_OK;
    const hipError_t e = ix->owns ? hipFree(ix->tiles) : hipSuccess;
    delete ix;
    if (e != hipSuccess) {
        set_error("mdx_index_destroy: hipFree failed: %s", hipGetErrorString(e));
        return MDX_ERR_RUNTIME;
    }
    return MDX_OK;
}

int mdx_index_info(const mdx_index *ix, int64_t *n, int64_t *d, int64_t *row_offset,
                   int64_t *device_bytes)
{
    MDX_CHECK_ARG(ix, "mdx_index_info: NULL index");
    if (n) *n = ix->n;
    if (d) *d = ix->d;
    if (row_offset) *row_offset = ix->row_offset;
    if (device_bytes) *device_bytes = ix->bytes;
    return MDX_OK;
}

int64_t mdx_scores_workspace(int64_t nq, int64_t d)
{
    if (nq <= 0 || d <= 0) return 0;
    return round_up(nq, TILE_ROWS) * round_up(d, 64) * 4;   // fp32 tiles; an fp16 shard uses half of it
}

static int scores_impl(const mdx_index *ix, const float *queries, int64_t nq, int qlayout, const float *center, float *scores,
                       Route route, void *workspace, int64_t workspace_bytes, void *stream);

int mdx_scores(const mdx_index *ix, const float *queries, int64_t nq, int qlayout,
               const float *center, float *scores, void *workspace, int64_t workspace_bytes,
               void *stream)
{
    MDX_CHECK_ARG(ix && queries && scores, "mdx_scores: NULL pointer");
    return scores_impl(ix, queries, nq, qlayout, center, scores, Route{nullptr, 0}, workspace, workspace_bytes, stream);
}

int mdx_scores_p2p(const mdx_index *ix, const float *queries, int64_t nq, int qlayout, const float *center, mdx_p2p *p2p,
                   void *workspace, int64_t workspace_bytes, void *stream)
{
    MDX_CHECK_ARG(ix && queries && p2p, "mdx_scores_p2p: NULL pointer");
    MDX_CHECK_ARG(ix->storage == MDX_F32, "mdx_scores_p2p: an fp32 shard is needed (this one is stored as fp16)");
    MDX_CHECK_ARG(nq > 0 && nq <= MAX_QT * TILE_ROWS, "mdx_scores_p2p: nq=%lld, 1..%d supported", (long long)nq, MAX_QT * TILE_ROWS);
    float *const *rows = nullptr;
    if (!p2p_route(p2p, nq, &rows)) {
        set_error("mdx_scores_p2p: the exchange is not connected or was created for another number of queries");
        return MDX_ERR_INVALID;
    }
    return scores_impl(ix, queries, nq, qlayout, center, nullptr, Route{rows, ix->row_offset}, workspace, workspace_bytes, stream);
}

static int scores_impl(const mdx_index *ix, const float *queries, int64_t nq, int qlayout, const float *center, float *scores,
                       Route route, void *workspace, int64_t workspace_bytes, void *stream)
{
    MDX_CHECK_ARG(nq > 0, "mdx_scores: nq=%lld must be positive", (long long)nq);
    MDX_CHECK_ARG(qlayout == MDX_DIM_MAJOR || qlayout == MDX_ROW_MAJOR, "mdx_scores: qlayout %d",
                  qlayout);
    const int64_t need = mdx_scores_workspace(nq, ix->d);
    if (!workspace || workspace_bytes < need) {
        set_error("mdx_scores: workspace %lld B < required %lld B", (long long)workspace_bytes,
                  (long long)need);
        return MDX_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    f32x4 *qtiles = (f32x4 *)workspace;
    const int64_t QT_total = ceil_div(nq, TILE_ROWS);
    int rc = retile(queries, nq, ix->d, qlayout, center, qtiles, QT_total, ix->KB, s, ix->storage);
    if (rc != MDX_OK) return rc;

    const bool small = ix->RT < 2048;                 // < 32 768 rows: 64-row workgroups
    int64_t qt_begin = 0;
    const int64_t full_passes = QT_total / MAX_QT;
    static const bool no_pass_grid = getenv("MDX_NO_PASS_GRID") != nullptr, no_query_split = getenv("MDX_NO_QUERY_SPLIT") != nullptr,
                      no_leftover = getenv("MDX_NO_LEFTOVER_MFMA") != nullptr;
    if (full_passes > 1 && full_passes < 65536 && !no_pass_grid && !route.rows) {
        // many queries: all full groups of MAX_QT query tiles in ONE launch (grid.y = group), so that a
        // small database still fills the chip (20 000 rows are 313 workgroups per group)
        const int mode = (small ? 0 : 1) | (ix->storage == MDX_F16 ? 2 : 0);
        rc = launch_qt<MAX_QT>(mode, ix->tiles, qtiles, scores, ix->n, ix->RT, (int)ix->KB,
                               (int)(full_passes * MAX_QT * TILE_ROWS), s, (int)full_passes);     // nq_valid = queries of the launch
        if (rc != MDX_OK) return rc;
        MDX_LAUNCH_CHECK();
        qt_begin = full_passes * MAX_QT;
    }
    if (qt_begin == 0 && QT_total <= MAX_QT && QT_total > 1 && ix->storage == MDX_F32 && !no_query_split) {
        // few queries against a small shard (rOxford5k alone: 70 x 4 993): 64-row workgroups taking all
        // query tiles are only RT/4 = 79 workgroups, each a 5-tile-long MFMA chain.  Give every workgroup
        // ONE query tile instead (grid.y = tile): 5x the workgroups, a fifth of the chain each.
        const int64_t blocks = ceil_div(ix->RT, (int64_t)4);
        if (small && blocks * QT_total <= 1024) {
            rc = launch_qt<1>(0, ix->tiles, qtiles, scores, ix->n, ix->RT, (int)ix->KB, (int)nq, s, (int)QT_total, route);
            if (rc != MDX_OK) return rc;
            MDX_LAUNCH_CHECK();
            return MDX_OK;
        }
    }
    for (int64_t qt0 = qt_begin; qt0 < QT_total; qt0 += MAX_QT) {
        const int qt = (int)((QT_total - qt0) < MAX_QT ? (QT_total - qt0) : MAX_QT);
        const int64_t q0 = qt0 * TILE_ROWS;
        const int nq_valid = (int)((nq - q0) < qt * TILE_ROWS ? (nq - q0) : qt * TILE_ROWS);
        const f32x4 *qp = qtiles + qt0 * ix->KB * 64;
        float *op = route.rows ? nullptr : scores + q0 * ix->n;
        const int mode = (small ? 0 : 1) | (ix->storage == MDX_F16 ? 2 : 0);
        const int tail = nq_valid - (qt - 1) * TILE_ROWS;      // queries in the last tile of this launch
        if (mode == 1 && qt >= 2 && tail <= 8 && !no_leftover)
            rc = dispatch_leftover(qt, ix->tiles, qp, op, ix->n, ix->RT, (int)ix->KB, nq_valid, s, route);
        else
            rc = dispatch_qt(qt, mode, ix->tiles, qp, op, ix->n, ix->RT, (int)ix->KB, nq_valid, s, route);
        if (rc != MDX_OK) return rc;
        MDX_LAUNCH_CHECK();
    }
    return MDX_OK;
}

int mdx_scores_rowmajor(const float *db, int64_t n, int64_t d, const float *queries, int64_t nq, int qlayout, const float *center,
                        float *scores, void *workspace, int64_t workspace_bytes, void *stream)
{
    MDX_CHECK_ARG(db && queries && scores, "mdx_scores_rowmajor: NULL pointer");
    MDX_CHECK_ARG(n > 0 && nq > 0, "mdx_scores_rowmajor: n=%lld nq=%lld must be positive", (long long)n, (long long)nq);
    // 16-byte LDS-DMA pieces are served from any 4-byte-aligned address (tools/align_probe.py: bit-exact at every offset)
    MDX_CHECK_ARG(d >= 4 && d % 4 == 0 && ((uintptr_t)db & 3) == 0,
                  "mdx_scores_rowmajor: d=%lld must be a multiple of 4 (rows are read in pieces of four values) and the matrix 4-byte aligned; "
                  "build an index for other shapes", (long long)d);
    MDX_CHECK_ARG(qlayout == MDX_DIM_MAJOR || qlayout == MDX_ROW_MAJOR, "mdx_scores_rowmajor: qlayout %d", qlayout);
    const int64_t need = mdx_scores_workspace(nq, d);
    if (!workspace || workspace_bytes < need) {
        set_error("mdx_scores_rowmajor: workspace %lld B < required %lld B", (long long)workspace_bytes, (long long)need);
        return MDX_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    f32x4 *qtiles = (f32x4 *)workspace;
    const int64_t QT_total = ceil_div(nq, TILE_ROWS), KB = round_up(d, 64) / TILE_K, RT = ceil_div(n, TILE_ROWS);
    int rc = retile(queries, nq, d, qlayout, center, qtiles, QT_total, KB, s, MDX_F32);
    if (rc != MDX_OK) return rc;
    const bool small = RT < 2048;                     // < 32 768 rows: 64-row workgroups
    for (int64_t qt0 = 0; qt0 < QT_total; qt0 += MAX_QT) {
        const int qt = (int)((QT_total - qt0) < MAX_QT ? (QT_total - qt0) : MAX_QT);
        const int64_t q0 = qt0 * TILE_ROWS;
        const int nq_valid = (int)((nq - q0) < qt * TILE_ROWS ? (nq - q0) : qt * TILE_ROWS);
        const int tail = nq_valid - (qt - 1) * TILE_ROWS;      // queries in the last tile of this launch
        const f32x4 *qp = qtiles + qt0 * KB * 64;
        float *op = scores + q0 * n;
        if (small) rc = dispatch_rowmajor<1>(qt, false, db, d, qp, op, n, RT, (int)KB, nq_valid, s);
        else       rc = dispatch_rowmajor<2>(qt, qt >= 2 && tail <= 8, db, d, qp, op, n, RT, (int)KB, nq_valid, s);
        if (rc != MDX_OK) return rc;
        MDX_LAUNCH_CHECK();
    }
    return MDX_OK;
}

int64_t mdx_scores_workspace_ex(int64_t nq, int64_t d, int compute)
{
    if (nq <= 0 || d <= 0) return 0;
    if (compute == MDX_F32_SPLIT3) return round_up(nq, TILE_ROWS) * round_up(d, 64) * 6;      // three bf16 pieces per element
    if (compute == MDX_F32_SPLIT2) return round_up(nq, TILE_ROWS) * round_up(d, 64) * 4 + 256;    // two fp16 pieces + the queries' |q| maximum
    return mdx_scores_workspace(nq, d);
}

int mdx_scores_ex(const mdx_index *ix, const float *queries, int64_t nq, int qlayout, const float *center, float *scores,
                  void *workspace, int64_t workspace_bytes, int compute, void *stream)
{
    if (compute == MDX_F32_CHAIN) return mdx_scores(ix, queries, nq, qlayout, center, scores, workspace, workspace_bytes, stream);
    MDX_CHECK_ARG(compute == MDX_F32_SPLIT3 || compute == MDX_F32_SPLIT2, "mdx_scores_ex: compute mode %d", compute);
    MDX_CHECK_ARG(ix && queries && scores, "mdx_scores_ex: NULL pointer");
    MDX_CHECK_ARG(ix->storage == MDX_F32, "mdx_scores_ex: the split-precision modes multiply an fp32 shard (this one is stored as fp16)");
    MDX_CHECK_ARG(nq > 0 && nq < (1 << 20), "mdx_scores_ex: nq=%lld", (long long)nq);
    MDX_CHECK_ARG(qlayout == MDX_DIM_MAJOR || qlayout == MDX_ROW_MAJOR, "mdx_scores_ex: qlayout %d", qlayout);
    const int64_t need = mdx_scores_workspace_ex(nq, ix->d, compute);
    if (!workspace || workspace_bytes < need) {
        set_error("mdx_scores_ex: workspace %lld B < required %lld B", (long long)workspace_bytes, (long long)need);
        return MDX_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    u32x4 *qp = (u32x4 *)workspace;
    const int64_t QT_total = ceil_div(nq, TILE_ROWS), NC = ix->KB / 2;
    const int64_t rs = qlayout == MDX_DIM_MAJOR ? 1 : ix->d, ks = qlayout == MDX_DIM_MAJOR ? nq : 1;
    const bool two = compute == MDX_F32_SPLIT2;
    uint32_t *q_cell = two ? (uint32_t *)((char *)workspace + need - 256) : nullptr;
    const float db_scale = two ? split2_scale(ix->max_bits) : 1.0f;
    if (two) {
        // the queries' largest |q - center| stays on the device: a reduction, then the re-tiling and the kernel's epilogue read it
        MDX_HIP(hipMemsetAsync(q_cell, 0, 4, s));
        const int64_t total = nq * ix->d;
        hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)(ceil_div(total, (int64_t)1024) < 1024 ? ceil_div(total, (int64_t)1024) : 1024)), dim3(256), 0, s,
                           queries, rs, ks, nq, ix->d, center, q_cell);
        hipLaunchKernelGGL(retile_split2_kernel, dim3((unsigned)ceil_div(QT_total * NC, (int64_t)4)), dim3(256), 0, s, queries, rs, ks, nq, ix->d,
                           center, (const uint32_t *)q_cell, qp, QT_total, NC);
    } else {
        hipLaunchKernelGGL(retile_split3_kernel, dim3((unsigned)ceil_div(QT_total * NC, (int64_t)4)), dim3(256), 0, s, queries, rs, ks, nq, ix->d,
                           center, qp, QT_total, NC);
    }
    MDX_LAUNCH_CHECK();
    const bool small = ix->RT < 4096;                 // < 65 536 rows: 128-row workgroups, so that the shard still spreads over the CUs
    const int64_t full = QT_total / SPLIT_QT, rem = QT_total % SPLIT_QT;
    auto go = [&](int qt, int64_t qt_first, int passes) -> int {
        if (two)
            return small ? dispatch_split2<1>(qt, ix->tiles, qp, scores, ix->n, ix->RT, (int)ix->KB, (int)QT_total, (int)qt_first, (int)nq, db_scale, q_cell, s, passes)
                         : dispatch_split2<2>(qt, ix->tiles, qp, scores, ix->n, ix->RT, (int)ix->KB, (int)QT_total, (int)qt_first, (int)nq, db_scale, q_cell, s, passes);
        return small ? dispatch_split3<1>(qt, ix->tiles, qp, scores, ix->n, ix->RT, (int)ix->KB, (int)QT_total, (int)qt_first, (int)nq, s, passes)
                     : dispatch_split3<2>(qt, ix->tiles, qp, scores, ix->n, ix->RT, (int)ix->KB, (int)QT_total, (int)qt_first, (int)nq, s, passes);
    };
    int rc = MDX_OK;
    for (int64_t g0 = 0; g0 < full && rc == MDX_OK; g0 += 32768)        // grid.y < 65 536
        rc = go(SPLIT_QT, g0 * SPLIT_QT, (int)((full - g0) < 32768 ? (full - g0) : 32768));
    if (rc == MDX_OK && rem) rc = go((int)rem, full * SPLIT_QT, 1);
    if (rc != MDX_OK) return rc;
    MDX_LAUNCH_CHECK();
    return MDX_OK;
}

}  // extern "C"
