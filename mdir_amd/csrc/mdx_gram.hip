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
// One GEMM kernel, C[i][j] = sum_k At[k][i] * Bt[k][j], both operands K-MAJOR with the tile dimension contiguous: a tile
// row is one 512-byte run of global memory and goes into LDS as it is (the first version read the K-contiguous operands --
// descriptors [d, n], P [dout, d] -- in place and transposed 32-byte pieces into LDS: 4-way bank conflicts and 16 separate
// rows per wave-load; the Gram form, with two such operands, ran at half the rate of the projection with one).  So the
// K-contiguous inputs are transposed ONCE into the workspace (a streaming pass, ~0.2 ms for 2048 x 20 000; the Gram form's
// centring `X - m` is applied there), then: workgroup = 4 waves = a 128 x 128 tile of the result (64 x 64 for small
// problems), K in steps of 16 through double-buffered LDS (row stride = 32 words mod 64: the two 16-lane groups a half-wave
// reads together fall into different bank halves), one barrier per step, the next step's global loads in flight under the
// current step's MFMAs; each wave owns 64 x 64 = 4 x 4 MFMA tiles.  The Gram form is symmetric: only tiles on or above the diagonal are computed (SYRK-shaped:
// half the flops of a GEMM), long sums are cut into K ranges whose partial tiles are added in range order (a fixed order).
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
                                                       int64_t M, int64_t N, int64_t K, int64_t ksplit)
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
    out += (int64_t)blockIdx.z * M * N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t row0 = (int64_t)bi * GM, col0 = (int64_t)bj * GN;

    // loader: thread -> (k row, 4 consecutive columns) of each operand's 16 x GM tile, QV times
    double ra[QV][4], rb[QV][4];
    auto quad = [&](const double *base, int64_t ld, int64_t k, int64_t c, int64_t cmax, double sub, double (&r)[4]) {
        if (k < kend && c + 3 < cmax) {
            const f64x4u v = *(const f64x4u *)(base + k * ld + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = v[e] - sub;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = (k < kend && c + e < cmax) ? base[k * ld + c + e] - sub : 0.0;
        }
    };
    auto fetch = [&](int64_t k0) {
#pragma unroll
        for (int q = 0; q < QV; ++q) {
            const int idx = tid + q * 256, pk = idx / (GM / 4), pj = (idx % (GM / 4)) * 4;
            const int64_t k = k0 + pk;
            quad(at, lda, k, row0 + pj, M, 0.0, ra[q]);
            if (SYM && bi == bj) {
#pragma unroll
                for (int e = 0; e < 4; ++e) rb[q][e] = ra[q][e];        // a diagonal tile multiplies its columns with themselves
            } else {
                quad(bt, ldb, k, col0 + pj, N, (kcenter && k < kend) ? kcenter[k] : 0.0, rb[q]);
            }
        }
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int q = 0; q < QV; ++q) {
            const int idx = tid + q * 256, pk = idx / (GM / 4), pj = (idx % (GM / 4)) * 4;
            *(f64x4 *)&As[buf][pk][pj] = (f64x4){ra[q][0], ra[q][1], ra[q][2], ra[q][3]};
            *(f64x4 *)&Bs[buf][pk][pj] = (f64x4){rb[q][0], rb[q][1], rb[q][2], rb[q][3]};
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
                    out[r * N + c] = acc[mi][ni][v];
                    if (SYM && bi != bj && gridDim.z == 1) out[c * N + r] = acc[mi][ni][v];
                }
            }
}

// out[i][j] = out[j][i] = part[0][i][j] + part[1][i][j] + ... (range order) for the tiles on or above the diagonal
// (blocks of 64 x 64; `tile` = the GEMM's tile size: a block below the diagonal of ITS tile grid was never written)
__global__ __launch_bounds__(256) void reduce_splits_kernel(const double *__restrict__ part, int splits, int64_t d, int tile,
                                                            double *__restrict__ out)
{
    const int64_t j = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63), i0 = (int64_t)blockIdx.y * 64 + (threadIdx.x >> 6) * 16;
    const int64_t ti = (int64_t)blockIdx.y * 64 / tile, tj = (int64_t)blockIdx.x * 64 / tile;
    if (ti > tj || j >= d) return;
    for (int64_t i = i0; i < i0 + 16 && i < d; ++i) {
        double s = 0.0;
        for (int z = 0; z < splits; ++z) s += part[((int64_t)z * d + i) * d + j];
        out[i * d + j] = s;
        if (ti != tj) out[j * d + i] = s;
    }
}

// K ranges of the Gram form: the 528 tiles of a 2048 x 2048 result are 2 workgroups per CU with full-K loops; more, shorter
// workgroups fill the chip evenly (at least 64 K-steps each, at most 16 ranges)
static int gram_tile(int64_t d) { return d >= 1024 ? 128 : 64; }

static int gram_splits(int64_t d, int64_t n)
{
    const int64_t t = ceil_div(d, (int64_t)gram_tile(d)), tiles = t * (t + 1) / 2;
    int64_t s = ceil_div((int64_t)2048, tiles);
    const int64_t max_by_k = n / (64 * GK) > 1 ? n / (64 * GK) : 1;
    s = s < 1 ? 1 : (s > 16 ? 16 : s);
    return (int)(s < max_by_k ? s : max_by_k);
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

static void launch_transpose(const double *src, int64_t rows, int64_t cols, const double *center, double *dst, hipStream_t s)
{
    hipLaunchKernelGGL(transpose_f64_kernel, dim3((unsigned)ceil_div(cols, 32), (unsigned)ceil_div(rows, 32)), dim3(256), 0, s, src, rows,
                       cols, center, dst, padded_ld(rows));
}

}  // namespace mdx

using namespace mdx;

extern "C" {

int64_t mdx_gram_f64_workspace(int64_t d, int64_t n)
{
    if (d <= 0 || n <= 0) return 0;
    return round_up(padded_ld(d) * n * 8, 256) + gram_partial_bytes(d, n);     // the transposed (centred) input, then the partial tiles
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
    double *at = (double *)workspace, *part = (double *)((char *)workspace + round_up(ld * n * 8, 256));
    launch_transpose(a, d, n, center, at, s);                           // at [n, d] = (a - center)^T
    const int tile = gram_tile(d);
    const unsigned t = (unsigned)ceil_div(d, (int64_t)tile), t64 = (unsigned)ceil_div(d, (int64_t)64);
    const int splits = gram_splits(d, n);
    const int64_t ksplit = round_up(ceil_div(n, (int64_t)splits), GK);
    double *dst = splits > 1 ? part : out;
    if (tile == 128)
        hipLaunchKernelGGL((gemm_f64_kernel<true, 64>), dim3(t * (t + 1) / 2, 1, (unsigned)splits), dim3(256), 0, s, (const double *)at, ld, (const double *)at,
                           ld, (const double *)nullptr, dst, d, d, n, ksplit);
    else
        hipLaunchKernelGGL((gemm_f64_kernel<true, 32>), dim3(t * (t + 1) / 2, 1, (unsigned)splits), dim3(256), 0, s, (const double *)at, ld, (const double *)at,
                           ld, (const double *)nullptr, dst, d, d, n, ksplit);
    if (splits > 1) hipLaunchKernelGGL(reduce_splits_kernel, dim3(t64, t64), dim3(256), 0, s, (const double *)part, splits, d, tile, out);
    MDX_LAUNCH_CHECK();
    return MDX_OK;
}

int64_t mdx_project_f64_workspace(int64_t dout, int64_t d)
{
    if (dout <= 0 || d <= 0) return 0;
    return round_up(padded_ld(dout) * d * 8, 256);                      // p transposed
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
    if (dout >= 1024 && n >= 1024)
        hipLaunchKernelGGL((gemm_f64_kernel<false, 64>), dim3((unsigned)ceil_div(n, (int64_t)128), (unsigned)ceil_div(dout, (int64_t)128)), dim3(256), 0, s,
                           (const double *)pt, padded_ld(dout), x, n, center, out, dout, n, d, d);
    else
        hipLaunchKernelGGL((gemm_f64_kernel<false, 32>), dim3((unsigned)ceil_div(n, (int64_t)64), (unsigned)ceil_div(dout, (int64_t)64)), dim3(256), 0, s,
                           (const double *)pt, padded_ld(dout), x, n, center, out, dout, n, d, d);
    MDX_LAUNCH_CHECK();
    return MDX_OK;
}

}  // extern "C"
