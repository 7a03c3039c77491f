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
// One kernel, two operand shapes.  Workgroup = 4 waves = a 64 x 64 tile of the result, K in steps of 16
// through LDS (k-major, row stride 80 doubles: the two 16-lane groups a half-wave reads together fall into
// different bank halves); each wave owns 32 x 32 = 2 x 2 MFMA tiles.  The next K-step's global loads are
// issued before the current step's MFMAs and parked in registers.  The Gram form is symmetric: only tiles
// on or above the diagonal are computed, each stored twice (SYRK-shaped: half the flops of a GEMM).
#include "mdx_common.h"

namespace mdx {

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int GM = 64, GN = 64, GK = 16, GLD = 80;

// MODE 0 (Gram):    C[i][j] = sum_k (a[i*n + k] - c[i]) * (a[j*n + k] - c[j])          a [d, n], C [d, d]
// MODE 1 (project): C[i][j] = sum_k a[i*kdim + k] * (b[k*n + j] - c[k])                a = P [m, kdim], b = X [kdim, n]
template <int MODE>
__global__ __launch_bounds__(256) void gemm_f64_kernel(const double *__restrict__ a, const double *__restrict__ b,
                                                       const double *__restrict__ center, double *__restrict__ out,
                                                       int64_t M, int64_t N, int64_t K)
{
    __shared__ double As[GK][GLD], Bs[GK][GLD];
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (MODE == 0 && bi > bj) return;                       // the mirror image of tile (bj, bi)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t row0 = (int64_t)bi * GM, col0 = (int64_t)bj * GN;

    // loader roles.  K-contiguous operand (a; and the second operand of the Gram form): thread -> (row r, 4 k).
    // N-contiguous operand (X of the projection): thread -> (k, 4 columns).
    const int lr = tid >> 2, lk = (tid & 3) * 4;
    const int pk = tid >> 4, pj = (tid & 15) * 4;
    const int64_t arow = row0 + lr, brow = col0 + lr;
    const double ca = (MODE == 0 && center && arow < M) ? center[arow] : 0.0;
    const double cb = (MODE == 0 && center && brow < N) ? center[brow] : 0.0;

    typedef double f64x4u __attribute__((ext_vector_type(4), aligned(8)));    // one 32-byte access at 8-byte alignment
    double ra[4], rb[4];
    // a row of 4 consecutive k (K-contiguous operand); whole quads take ONE 32-byte load instead of four guarded ones
    auto quad = [&](const double *base, int64_t row, int64_t rows, int64_t k, double c, double (&r)[4]) {
        if (row < rows && k + 3 < K) {
            const f64x4u v = *(const f64x4u *)(base + row * K + k);
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = v[e] - c;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = (row < rows && k + e < K) ? base[row * K + k + e] - c : 0.0;
        }
    };
    auto fetch = [&](int64_t k0) {
        quad(a, arow, M, k0 + lk, ca, ra);
        if (MODE == 0) {
            if (bi == bj) {
#pragma unroll
                for (int e = 0; e < 4; ++e) rb[e] = ra[e];              // a diagonal tile multiplies its rows with themselves
            } else {
                quad(a, brow, N, k0 + lk, cb, rb);
            }
        } else {
            const int64_t k = k0 + pk, j = col0 + pj;
            const double ck = (center && k < K) ? center[k] : 0.0;
            if (k < K && j + 3 < N) {
                const f64x4u v = *(const f64x4u *)(b + k * N + j);
#pragma unroll
                for (int e = 0; e < 4; ++e) rb[e] = v[e] - ck;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) rb[e] = (k < K && j + e < N) ? b[k * N + j + e] - ck : 0.0;
            }
        }
    };
    auto park = [&]() {
#pragma unroll
        for (int e = 0; e < 4; ++e) As[lk + e][lr] = ra[e];
        if (MODE == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) Bs[lk + e][lr] = rb[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) Bs[pk][pj + e] = rb[e];
        }
    };

    f64x4 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = (f64x4){0.0, 0.0, 0.0, 0.0};

    fetch(0);
    for (int64_t k0 = 0; k0 < K; k0 += GK) {
        __syncthreads();                    // every wave is done reading the previous step
        park();
        __syncthreads();
        if (k0 + GK < K) fetch(k0 + GK);    // in flight under the MFMAs below
        const int kr = lane >> 4, c16 = lane & 15;
#pragma unroll
        for (int kk = 0; kk < GK / 4; ++kk) {
            double av[2], bv[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                av[t] = As[kk * 4 + kr][wm * 32 + t * 16 + c16];
                bv[t] = Bs[kk * 4 + kr][wn * 32 + t * 16 + c16];
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[mi], bv[ni], acc[mi][ni], 0, 0, 0);
        }
    }
    // C/D of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int64_t r = row0 + wm * 32 + mi * 16 + (lane >> 4) + 4 * v;
                const int64_t c = col0 + wn * 32 + ni * 16 + (lane & 15);
                if (r < M && c < N) {
                    out[r * N + c] = acc[mi][ni][v];
                    if (MODE == 0 && bi != bj) out[c * N + r] = acc[mi][ni][v];
                }
            }
}

}  // namespace mdx

using namespace mdx;

extern "C" {

int mdx_gram_f64(const double *a, int64_t d, int64_t n, const double *center, double *out, void *stream)
{
    MDX_CHECK_ARG(a && out, "mdx_gram_f64: NULL pointer");
    MDX_CHECK_ARG(d > 0 && n > 0 && d < (1ll << 21), "mdx_gram_f64: d=%lld n=%lld", (long long)d, (long long)n);
    const unsigned t = (unsigned)ceil_div(d, GM);
    hipLaunchKernelGGL(gemm_f64_kernel<0>, dim3(t, t), dim3(256), 0, (hipStream_t)stream, a, (const double *)nullptr, center, out, d, d, n);
    MDX_LAUNCH_CHECK();
    return MDX_OK;
}

int mdx_project_f64(const double *p, int64_t dout, int64_t d, const double *x, int64_t n, const double *center, double *out,
                    void *stream)
{
    MDX_CHECK_ARG(p && x && out, "mdx_project_f64: NULL pointer");
    MDX_CHECK_ARG(dout > 0 && d > 0 && n > 0 && dout < (1ll << 21) && n < (1ll << 37), "mdx_project_f64: dout=%lld d=%lld n=%lld",
                  (long long)dout, (long long)d, (long long)n);
    hipLaunchKernelGGL(gemm_f64_kernel<1>, dim3((unsigned)ceil_div(n, GN), (unsigned)ceil_div(dout, GM)), dim3(256), 0,
                       (hipStream_t)stream, p, x, center, out, dout, n, d);
    MDX_LAUNCH_CHECK();
    return MDX_OK;
}

}  // extern "C"
