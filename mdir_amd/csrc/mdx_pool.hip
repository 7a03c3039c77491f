// Descriptor tail: global pooling (GeM / MAC / SPoC), L2 normalisation, multi-scale
// aggregation.  All HBM- (really launch-) bound streaming reductions: one wave per
// (image, channel) plane with 16-B loads, wavefront shuffles for the reduction.
//
// Replaces LF.gem / LF.mac / LF.spoc / LF.l2n (mdir/external/cirtorch/layers/
// functional.py:11-22,130-131) as used by ImageRetrievalNet.forward
// (cirtorch/networks/imageretrievalnet.py:108-112) and
// CirMultiscaleAggregation.aggregate_tensor (mdir/components/data/wrapper.py:109-119).
#include <math.h>

#include "mdx_common.h"

namespace mdx {

// x^p for x >= eps > 0.  Exact products for the exponents that occur untrained
// (p = 1, 2, 3, layers/pooling.py:38); exp2(p*log2 x) through the hardware
// transcendental units otherwise (relative error ~1e-6; tests/test_gpu_kernels.py::test_pool_l2n_golden, rtol 1e-5).
template <int MODE>
__device__ __forceinline__ float pow_pos(float x, float p)
{
    if (MODE == 1) return x;
    if (MODE == 2) return x * x;
    if (MODE == 3) return x * x * x;
    return __builtin_amdgcn_exp2f(p * __builtin_amdgcn_logf(x));
}

template <int KIND, int MODE>
__device__ __forceinline__ float pool_elem(float x, float p, float eps)
{
    if (KIND == MDX_POOL_GEM) return pow_pos<MODE>(fmaxf(x, eps), p);
    return x;
}

// one wave reduces one (image, channel) plane; every lane returns the pooled value
template <int KIND, int MODE>
__device__ __forceinline__ float pool_plane(const float *__restrict__ src, bool wide, int HW, float p, float inv_p, float eps,
                                            int lane)
{
    float acc = KIND == MDX_POOL_MAC ? -INFINITY : 0.0f;
    if (wide) {
        const float4 *s4 = (const float4 *)src;
        const int n4 = HW >> 2;
        for (int i = lane; i < n4; i += 64) {
            const float4 v = s4[i];
            const float a = pool_elem<KIND, MODE>(v.x, p, eps), b = pool_elem<KIND, MODE>(v.y, p, eps);
            const float c = pool_elem<KIND, MODE>(v.z, p, eps), d = pool_elem<KIND, MODE>(v.w, p, eps);
            if (KIND == MDX_POOL_MAC) acc = fmaxf(acc, fmaxf(fmaxf(a, b), fmaxf(c, d)));
            else acc += (a + b) + (c + d);
        }
    } else {
        for (int i = lane; i < HW; i += 64) {
            const float a = pool_elem<KIND, MODE>(src[i], p, eps);
            if (KIND == MDX_POOL_MAC) acc = fmaxf(acc, a);
            else acc += a;
        }
    }
    acc = KIND == MDX_POOL_MAC ? wave_max(acc) : wave_sum(acc);
    float r = acc;
    if (KIND != MDX_POOL_MAC) r = acc / (float)HW;
    if (KIND == MDX_POOL_GEM && MODE != 1) r = powf(r, inv_p);
    return r;
}

template <int KIND, int MODE>
__global__ __launch_bounds__(256) void pool_kernel(const float *__restrict__ feat, int64_t planes,
                                                   int HW, float p, float inv_p, float eps,
                                                   float *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t plane = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (plane >= planes) return;
    const bool wide = (HW & 3) == 0 && ((uintptr_t)feat & 15) == 0;
    const float r = pool_plane<KIND, MODE>(feat + plane * HW, wide, HW, p, inv_p, eps, lane);
    if (lane == 0) out[plane] = r;
}

// The S feature maps of an image pyramid ([B,C,H_s,W_s] each) pooled by ONE launch: plane index -> (scale, image,
// channel); out [S,B,C].  Same per-plane arithmetic as pool_kernel.
struct PoolMaps {
    const float *feat[8];
    int hw[8];
    int64_t first[9];           // first plane of scale s; first[S] = all planes
};

template <int KIND, int MODE>
__global__ __launch_bounds__(256) void pool_multi_kernel(PoolMaps maps, int S, float p, float inv_p, float eps,
                                                         float *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t plane = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (plane >= maps.first[S]) return;
    int s = 0;
    while (s + 1 < S && plane >= maps.first[s + 1]) ++s;
    const float *feat = maps.feat[s];
    const int HW = maps.hw[s];
    const bool wide = (HW & 3) == 0 && ((uintptr_t)feat & 15) == 0;
    const float r = pool_plane<KIND, MODE>(feat + (plane - maps.first[s]) * HW, wide, HW, p, inv_p, eps, lane);
    if (lane == 0) out[plane] = r;
}

// (A single-launch form of pool + L2N -- the workgroup that finishes an image last normalises it, hand-off through
// write-through stores, an arrival ticket and an agent-scope acquire -- was built, is bit-identical, and is SLOWER:
// 29 us against 12 us (batch 1) / 20 us (batch 4) for these two launches on a 2048 x 24 x 32 map:
// the hand-off through memory costs more than the kernel boundary it removes, as the MI355X
// guide's section 5.6 predicts.  In the history: git show 47a9fe2:tools/attic/pool_l2n_fused.hip.)

// one workgroup per row: x = (x + bias) / (||x + bias|| + eps)
__global__ __launch_bounds__(256) void l2n_rows_kernel(float *__restrict__ x, int64_t D,
                                                       const float *__restrict__ bias, float eps)
{
    __shared__ float part[4];
    float *row = x + (int64_t)blockIdx.x * D;
    const int tid = threadIdx.x;
    float ss = 0.0f;
    for (int64_t k = tid; k < D; k += 256) {
        float v = row[k];
        if (bias) v += bias[k];
        ss += v * v;
    }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) part[tid >> 6] = ss;
    __syncthreads();
    const float den = sqrtf((part[0] + part[1]) + (part[2] + part[3])) + eps;
    for (int64_t k = tid; k < D; k += 256) {
        float v = row[k];
        if (bias) v += bias[k];
        row[k] = v / den;
    }
}

struct ScalePtrs { const float *p[8]; };

// single workgroup: power mean over scales, then plain L2 normalisation (no eps)
__global__ __launch_bounds__(1024) void ms_aggregate_kernel(ScalePtrs sp, int S, int64_t D, float msp,
                                                            float inv_msp, float *__restrict__ out)
{
    __shared__ float part[16];
    const int tid = threadIdx.x;
    float ss = 0.0f;
    for (int64_t k = tid; k < D; k += 1024) {
        float a = 0.0f;
        for (int s = 0; s < S; ++s) a += (msp == 1.0f) ? sp.p[s][k] : powf(sp.p[s][k], msp);
        a = a / (float)S;
        if (msp != 1.0f) a = powf(a, inv_msp);
        out[k] = a;
        ss += a * a;
    }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) part[tid >> 6] = ss;
    __syncthreads();
    float tot = 0.0f;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += part[w];
    const float nrm = sqrtf(tot);
    for (int64_t k = tid; k < D; k += 1024) out[k] = out[k] / nrm;
}

// batched form: scale s is a [B,D] row-major matrix; one workgroup per image
__global__ __launch_bounds__(1024) void ms_aggregate_batch_kernel(ScalePtrs sp, int S, int64_t D, float msp, float inv_msp,
                                                                  float *__restrict__ out)
{
    __shared__ float part[16];
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.x;
    float *dst = out + b * D;
    float ss = 0.0f;
    for (int64_t k = tid; k < D; k += 1024) {
        float a = 0.0f;
        for (int s = 0; s < S; ++s) a += (msp == 1.0f) ? sp.p[s][b * D + k] : powf(sp.p[s][b * D + k], msp);
        a = a / (float)S;
        if (msp != 1.0f) a = powf(a, inv_msp);
        dst[k] = a;
        ss += a * a;
    }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) part[tid >> 6] = ss;
    __syncthreads();
    float tot = 0.0f;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += part[w];
    const float nrm = sqrtf(tot);
    for (int64_t k = tid; k < D; k += 1024) dst[k] = dst[k] / nrm;
}

// pooled [S,B,D] -> out [B,D]: L2N of every scale's row (eps added to the norm), power mean over the scales, plain
// renormalisation -- l2n_rows_kernel x S and ms_aggregate_batch_kernel in one launch, one workgroup per image.  The
// sums run in the order of those two kernels (the norms over 256 threads, the rest over 1024), so the result is
// bit-identical to the separate launches.
__global__ __launch_bounds__(1024) void l2n_aggregate_kernel(const float *__restrict__ pooled, int S, int64_t B, int64_t D,
                                                             float eps, float msp, float inv_msp, float *__restrict__ out)
{
    __shared__ float npart[8][4];
    __shared__ float part[16];
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.x;
    if (tid < 256) {
        for (int s = 0; s < S; ++s) {
            const float *row = pooled + ((int64_t)s * B + b) * D;
            float ss = 0.0f;
            for (int64_t k = tid; k < D; k += 256) {
                const float v = row[k];
                ss += v * v;
            }
            ss = wave_sum(ss);
            if ((tid & 63) == 0) npart[s][tid >> 6] = ss;
        }
    }
    __syncthreads();
    float den[8];
#pragma unroll
    for (int s = 0; s < 8; ++s)
        den[s] = s < S ? sqrtf((npart[s][0] + npart[s][1]) + (npart[s][2] + npart[s][3])) + eps : 1.0f;
    float *dst = out + b * D;
    float ss = 0.0f;
    for (int64_t k = tid; k < D; k += 1024) {
        float a = 0.0f;
#pragma unroll
        for (int s = 0; s < 8; ++s)
            if (s < S) {
                const float v = pooled[((int64_t)s * B + b) * D + k] / den[s];
                a += (msp == 1.0f) ? v : powf(v, msp);
            }
        a = a / (float)S;
        if (msp != 1.0f) a = powf(a, inv_msp);
        dst[k] = a;
        ss += a * a;
    }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) part[tid >> 6] = ss;
    __syncthreads();
    float tot = 0.0f;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += part[w];
    const float nrm = sqrtf(tot);
    for (int64_t k = tid; k < D; k += 1024) dst[k] = dst[k] / nrm;
}

template <int KIND>
static void launch_pool_multi(int mode, const PoolMaps &maps, int S, float p, float eps, float *out, hipStream_t s)
{
    const dim3 grid((unsigned)ceil_div(maps.first[S], 4)), blk(256);
    const float inv_p = 1.0f / p;
    switch (mode) {
        case 1: hipLaunchKernelGGL((pool_multi_kernel<KIND, 1>), grid, blk, 0, s, maps, S, p, inv_p, eps, out); break;
        case 2: hipLaunchKernelGGL((pool_multi_kernel<KIND, 2>), grid, blk, 0, s, maps, S, p, inv_p, eps, out); break;
        case 3: hipLaunchKernelGGL((pool_multi_kernel<KIND, 3>), grid, blk, 0, s, maps, S, p, inv_p, eps, out); break;
        default: hipLaunchKernelGGL((pool_multi_kernel<KIND, 0>), grid, blk, 0, s, maps, S, p, inv_p, eps, out); break;
    }
}

template <int KIND>
static void launch_pool(int mode, const float *feat, int64_t planes, int HW, float p, float eps,
                        float *out, hipStream_t s)
{
    const dim3 grid((unsigned)ceil_div(planes, 4)), blk(256);
    const float inv_p = 1.0f / p;
    switch (mode) {
        case 1: hipLaunchKernelGGL((pool_kernel<KIND, 1>), grid, blk, 0, s, feat, planes, HW, p, inv_p, eps, out); break;
        case 2: hipLaunchKernelGGL((pool_kernel<KIND, 2>), grid, blk, 0, s, feat, planes, HW, p, inv_p, eps, out); break;
        case 3: hipLaunchKernelGGL((pool_kernel<KIND, 3>), grid, blk, 0, s, feat, planes, HW, p, inv_p, eps, out); break;
        default: hipLaunchKernelGGL((pool_kernel<KIND, 0>), grid, blk, 0, s, feat, planes, HW, p, inv_p, eps, out); break;
    }
}

}  // namespace mdx

using namespace mdx;

extern "C" {

int mdx_l2n_rows(float *x, int64_t R, int64_t D, const float *bias, float eps, void *stream)
{
    MDX_CHECK_ARG(x, "mdx_l2n_rows: NULL pointer");
    MDX_CHECK_ARG(R > 0 && D > 0 && R < (1ll << 31), "mdx_l2n_rows: R=%lld D=%lld", (long long)R,
                  (long long)D);
    hipLaunchKernelGGL(l2n_rows_kernel, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, x, D,
                       bias, eps);
    MDX_LAUNCH_CHECK();
    return MDX_OK;
}

int mdx_pool_l2n(const float *feat, int B, int C, int H, int W, int kind, float p, float pool_eps,
                 float l2n_eps, float *out, void *stream)
{
    MDX_CHECK_ARG(feat && out, "mdx_pool_l2n: NULL pointer");
    MDX_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0, "mdx_pool_l2n: bad shape [%d,%d,%d,%d]", B, C, H, W);
    MDX_CHECK_ARG((int64_t)H * W < (1ll << 31), "mdx_pool_l2n: H*W too large");
    hipStream_t s = (hipStream_t)stream;
    const int64_t planes = (int64_t)B * C;
    const int HW = H * W;
    switch (kind) {
        case MDX_POOL_GEM: {
            MDX_CHECK_ARG(p > 0.0f && pool_eps > 0.0f, "mdx_pool_l2n: gem needs p > 0 and eps > 0");
            const int mode = p == 1.0f ? 1 : p == 2.0f ? 2 : p == 3.0f ? 3 : 0;
            launch_pool<MDX_POOL_GEM>(mode, feat, planes, HW, p, pool_eps, out, s);
            break;
        }
        case MDX_POOL_MAC: launch_pool<MDX_POOL_MAC>(1, feat, planes, HW, 1.0f, 0.0f, out, s); break;
        case MDX_POOL_SPOC: launch_pool<MDX_POOL_SPOC>(1, feat, planes, HW, 1.0f, 0.0f, out, s); break;
        default: MDX_CHECK_ARG(false, "mdx_pool_l2n: unknown pooling kind %d", kind);
    }
    MDX_LAUNCH_CHECK();
    if (l2n_eps >= 0.0f) return mdx_l2n_rows(out, B, C, nullptr, l2n_eps, stream);
    return MDX_OK;
}

// ---------------------------------------------------------------------------
// R-MAC (cirtorch/layers/functional.py:26-72): maxima over the whole map and over a grid of square regions, each
// L2-normalised over the channels (eps added to the norm), summed in region order.  The grid is the caller's (the reference
// computes it in float32 tensor arithmetic; the host restates that, mdir_amd/layers.py).
// Launch 1: one wave per (image, channel) plane -- the plane is read ONCE into registers (lane = column block), every region's
// maximum is a masked wave reduction: regmax[b][r][c].  Launch 2: one workgroup per image walks the regions in order: norm over
// the channels (workgroup reduction), v += max / (norm + eps).
// ---------------------------------------------------------------------------
constexpr int RMAC_MAX_REGIONS = 64;
struct RmacGrid { int i0[RMAC_MAX_REGIONS], j0[RMAC_MAX_REGIONS], h[RMAC_MAX_REGIONS], w[RMAC_MAX_REGIONS]; int n; };

// (also `roipool` of functional.py:75-121: KIND = the regional pooling -- max, mean or GeM over every region)
extern "C++" {
template <int KIND, int MODE>
__global__ __launch_bounds__(256) void roi_pool_kernel(const float *__restrict__ feat, int64_t planes, int C, int H, int W, RmacGrid grid,
                                                       float pw, float inv_p, float eps, float *__restrict__ regions_out)
{
    const int lane = threadIdx.x & 63;
    const int64_t plane = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (plane >= planes) return;
    const float *p = feat + plane * (int64_t)H * W;
    const int64_t b = plane / C, c = plane % C;
    for (int r = 0; r < grid.n; ++r) {
        const int i0 = grid.i0[r], j0 = grid.j0[r], rh = grid.h[r], rw = grid.w[r];
        float acc = KIND == MDX_POOL_MAC ? -INFINITY : 0.0f;
        for (int e = lane; e < rh * rw; e += 64) {           // (the map is L2-resident after region 0)
            const float v = pool_elem<KIND, MODE>(p[(i0 + e / rw) * W + j0 + e % rw], pw, eps);
            acc = KIND == MDX_POOL_MAC ? fmaxf(acc, v) : acc + v;
        }
        acc = KIND == MDX_POOL_MAC ? wave_max(acc) : wave_sum(acc);
        if (KIND != MDX_POOL_MAC) acc = acc / (float)(rh * rw);
        if (KIND == MDX_POOL_GEM && MODE != 1) acc = powf(acc, inv_p);
        if (lane == 0) regions_out[(b * grid.n + r) * C + c] = acc;
    }
}

static int fill_grid(RmacGrid *grid, const int32_t *regions, int nregions, int H, int W, const char *who)
{
    grid->n = nregions;
    for (int r = 0; r < nregions; ++r) {
        grid->i0[r] = regions[4 * r]; grid->j0[r] = regions[4 * r + 1]; grid->h[r] = regions[4 * r + 2]; grid->w[r] = regions[4 * r + 3];
        MDX_CHECK_ARG(grid->i0[r] >= 0 && grid->j0[r] >= 0 && grid->h[r] > 0 && grid->w[r] > 0 && grid->i0[r] + grid->h[r] <= H && grid->j0[r] + grid->w[r] <= W,
                      "%s: region %d = (%d, %d, %d, %d) outside the %d x %d map", who, r, grid->i0[r], grid->j0[r], grid->h[r], grid->w[r], H, W);
    }
    return MDX_OK;
}

template <int KIND>
static void launch_roi(int mode, const float *feat, int64_t planes, int C, int H, int W, const RmacGrid &grid, float p, float eps, float *out,
                       hipStream_t s)
{
    const dim3 g((unsigned)ceil_div(planes, (int64_t)4)), b(256);
    const float inv_p = 1.0f / p;
    switch (mode) {
        case 1: hipLaunchKernelGGL((roi_pool_kernel<KIND, 1>), g, b, 0, s, feat, planes, C, H, W, grid, p, inv_p, eps, out); break;
        case 2: hipLaunchKernelGGL((roi_pool_kernel<KIND, 2>), g, b, 0, s, feat, planes, C, H, W, grid, p, inv_p, eps, out); break;
        case 3: hipLaunchKernelGGL((roi_pool_kernel<KIND, 3>), g, b, 0, s, feat, planes, C, H, W, grid, p, inv_p, eps, out); break;
        default: hipLaunchKernelGGL((roi_pool_kernel<KIND, 0>), g, b, 0, s, feat, planes, C, H, W, grid, p, inv_p, eps, out); break;
    }
}
}   // extern "C++"

int mdx_roipool(const float *feat, int B, int C, int H, int W, const int32_t *regions, int nregions, int kind, float p, float pool_eps,
                float *out, void *stream)
{
    MDX_CHECK_ARG(feat && regions && out, "mdx_roipool: NULL pointer");
    MDX_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0 && (int64_t)H * W < (1ll << 31), "mdx_roipool: bad shape [%d,%d,%d,%d]", B, C, H, W);
    MDX_CHECK_ARG(nregions >= 1 && nregions <= RMAC_MAX_REGIONS, "mdx_roipool: %d regions, 1..%d supported", nregions, RMAC_MAX_REGIONS);
    RmacGrid grid;
    int rc = fill_grid(&grid, regions, nregions, H, W, "mdx_roipool");
    if (rc != MDX_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int64_t planes = (int64_t)B * C;
    switch (kind) {
        case MDX_POOL_GEM: {
            MDX_CHECK_ARG(p > 0.0f && pool_eps > 0.0f, "mdx_roipool: gem needs p > 0 and eps > 0");
            launch_roi<MDX_POOL_GEM>(p == 1.0f ? 1 : p == 2.0f ? 2 : p == 3.0f ? 3 : 0, feat, planes, C, H, W, grid, p, pool_eps, out, s);
            break;
        }
        case MDX_POOL_MAC: launch_roi<MDX_POOL_MAC>(1, feat, planes, C, H, W, grid, 1.0f, 0.0f, out, s); break;
        case MDX_POOL_SPOC: launch_roi<MDX_POOL_SPOC>(1, feat, planes, C, H, W, grid, 1.0f, 0.0f, out, s); break;
        default: MDX_CHECK_ARG(false, "mdx_roipool: unknown pooling kind %d", kind);
    }
    MDX_LAUNCH_CHECK();
    return MDX_OK;
}

// eps < 0: the region vectors are summed as they are (Rpool's `o.sum(1)`, pooling.py:91)
__global__ __launch_bounds__(256) void rmac_sum_kernel(const float *__restrict__ regmax, int C, int nreg, float eps, float *__restrict__ out)
{
    __shared__ float part[4];
    const int tid = threadIdx.x;
    const float *m = regmax + (int64_t)blockIdx.x * nreg * C;
    float *o = out + (int64_t)blockIdx.x * C;
    for (int r = 0; r < nreg; ++r) {
        float den = 1.0f;
        if (eps >= 0.0f) {                     // (uniform)
            float ss = 0.0f;
            for (int k = tid; k < C; k += 256) ss += m[r * (int64_t)C + k] * m[r * (int64_t)C + k];
            ss = wave_sum(ss);
            __syncthreads();                   // `part` of the previous region has been read by everybody
            if ((tid & 63) == 0) part[tid >> 6] = ss;
            __syncthreads();
            den = sqrtf((part[0] + part[1]) + (part[2] + part[3])) + eps;
        }
        for (int k = tid; k < C; k += 256) {
            const float v = m[r * (int64_t)C + k] / den;
            o[k] = r == 0 ? v : o[k] + v;       // a thread owns its channels: no hazard between regions
        }
    }
}

int mdx_region_sum(const float *vecs, int B, int nregions, int C, float l2n_eps, float *out, void *stream)
{
    MDX_CHECK_ARG(vecs && out, "mdx_region_sum: NULL pointer");
    MDX_CHECK_ARG(B > 0 && C > 0 && nregions >= 1, "mdx_region_sum: bad shape [%d,%d,%d]", B, nregions, C);
    hipLaunchKernelGGL(rmac_sum_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, vecs, C, nregions, l2n_eps, out);
    MDX_LAUNCH_CHECK();
    return MDX_OK;
}

int64_t mdx_rmac_workspace(int B, int C, int nregions)
{
    if (B <= 0 || C <= 0 || nregions <= 0) return 0;
    return (int64_t)B * nregions * C * 4;
}

int mdx_rmac(const float *feat, int B, int C, int H, int W, const int32_t *regions, int nregions, float eps, void *workspace,
             int64_t workspace_bytes, float *out, void *stream)
{
    MDX_CHECK_ARG(feat && regions && out, "mdx_rmac: NULL pointer");
    MDX_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0 && (int64_t)H * W < (1ll << 31), "mdx_rmac: bad shape [%d,%d,%d,%d]", B, C, H, W);
    MDX_CHECK_ARG(nregions >= 1 && nregions <= RMAC_MAX_REGIONS, "mdx_rmac: %d regions, 1..%d supported", nregions, RMAC_MAX_REGIONS);
    MDX_CHECK_ARG(eps >= 0.0f, "mdx_rmac: eps=%g", (double)eps);
    const int64_t need = mdx_rmac_workspace(B, C, nregions);
    if (!workspace || workspace_bytes < need) {
        set_error("mdx_rmac: workspace %lld B < required %lld B", (long long)workspace_bytes, (long long)need);
        return MDX_ERR_WORKSPACE;
    }
    RmacGrid grid;
    int rc = fill_grid(&grid, regions, nregions, H, W, "mdx_rmac");
    if (rc != MDX_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int64_t planes = (int64_t)B * C;
    launch_roi<MDX_POOL_MAC>(1, feat, planes, C, H, W, grid, 1.0f, 0.0f, (float *)workspace, s);
    hipLaunchKernelGGL(rmac_sum_kernel, dim3((unsigned)B), dim3(256), 0, s, (const float *)workspace, C, nregions, eps, out);
    MDX_LAUNCH_CHECK();
    return MDX_OK;
}

int mdx_pool_multi(const float *const *feats, int S, int B, int C, const int *H, const int *W, int kind, float p,
                   float pool_eps, float *pooled, void *stream)
{
    MDX_CHECK_ARG(feats && H && W && pooled, "mdx_pool_multi: NULL pointer");
    MDX_CHECK_ARG(S >= 1 && S <= 8, "mdx_pool_multi: S=%d not in 1..8", S);
    MDX_CHECK_ARG(B > 0 && C > 0, "mdx_pool_multi: bad batch/channels [%d,%d]", B, C);
    PoolMaps maps;
    maps.first[0] = 0;
    for (int s = 0; s < 8; ++s) {
        maps.feat[s] = s < S ? feats[s] : nullptr;
        maps.hw[s] = 0;
        if (s < S) {
            MDX_CHECK_ARG(feats[s], "mdx_pool_multi: map %d is NULL", s);
            MDX_CHECK_ARG(H[s] > 0 && W[s] > 0 && (int64_t)H[s] * W[s] < (1ll << 31), "mdx_pool_multi: bad map size %d x %d", H[s], W[s]);
            maps.hw[s] = H[s] * W[s];
        }
        maps.first[s + 1] = maps.first[s] + (s < S ? (int64_t)B * C : 0);
    }
    hipStream_t st = (hipStream_t)stream;
    switch (kind) {
        case MDX_POOL_GEM: {
            MDX_CHECK_ARG(p > 0.0f && pool_eps > 0.0f, "mdx_pool_multi: gem needs p > 0 and eps > 0");
            const int mode = p == 1.0f ? 1 : p == 2.0f ? 2 : p == 3.0f ? 3 : 0;
            launch_pool_multi<MDX_POOL_GEM>(mode, maps, S, p, pool_eps, pooled, st);
            break;
        }
        case MDX_POOL_MAC: launch_pool_multi<MDX_POOL_MAC>(1, maps, S, 1.0f, 0.0f, pooled, st); break;
        case MDX_POOL_SPOC: launch_pool_multi<MDX_POOL_SPOC>(1, maps, S, 1.0f, 0.0f, pooled, st); break;
        default: MDX_CHECK_ARG(false, "mdx_pool_multi: unknown pooling kind %d", kind);
    }
    MDX_LAUNCH_CHECK();
    return MDX_OK;
}

int mdx_l2n_aggregate(const float *pooled, int S, int64_t B, int64_t D, float l2n_eps, float msp, float *out,
                      void *stream)
{
    MDX_CHECK_ARG(pooled && out, "mdx_l2n_aggregate: NULL pointer");
    MDX_CHECK_ARG(S >= 1 && S <= 8, "mdx_l2n_aggregate: S=%d not in 1..8", S);
    MDX_CHECK_ARG(D > 0 && B > 0 && B < (1ll << 31), "mdx_l2n_aggregate: B=%lld D=%lld", (long long)B, (long long)D);
    MDX_CHECK_ARG(msp > 0.0f && l2n_eps >= 0.0f, "mdx_l2n_aggregate: msp=%g eps=%g", (double)msp, (double)l2n_eps);
    hipLaunchKernelGGL(l2n_aggregate_kernel, dim3((unsigned)B), dim3(1024), 0, (hipStream_t)stream, pooled, S, B, D, l2n_eps,
                       msp, 1.0f / msp, out);
    MDX_LAUNCH_CHECK();
    return MDX_OK;
}

int mdx_ms_aggregate_batch(const float *const *scale_mats, int S, int64_t B, int64_t D, float msp, float *out,
                           void *stream)
{
    MDX_CHECK_ARG(scale_mats && out, "mdx_ms_aggregate_batch: NULL pointer");
    MDX_CHECK_ARG(S >= 1 && S <= 8, "mdx_ms_aggregate_batch: S=%d not in 1..8", S);
    MDX_CHECK_ARG(D > 0 && B > 0 && B < (1ll << 31), "mdx_ms_aggregate_batch: B=%lld D=%lld", (long long)B, (long long)D);
    ScalePtrs sp;
    for (int s = 0; s < 8; ++s) {
        sp.p[s] = s < S ? scale_mats[s] : nullptr;
        MDX_CHECK_ARG(s >= S || sp.p[s], "mdx_ms_aggregate_batch: scale %d is NULL", s);
    }
    hipLaunchKernelGGL(ms_aggregate_batch_kernel, dim3((unsigned)B), dim3(1024), 0, (hipStream_t)stream, sp, S, D, msp,
                       1.0f / msp, out);
    MDX_LAUNCH_CHECK();
    return MDX_OK;
}

int mdx_ms_aggregate(const float *const *scale_vecs, int S, int64_t D, float msp, float *out,
                     void *stream)
{
    MDX_CHECK_ARG(scale_vecs && out, "mdx_ms_aggregate: NULL pointer");
    MDX_CHECK_ARG(S >= 1 && S <= 8, "mdx_ms_aggregate: S=%d not in 1..8", S);
    MDX_CHECK_ARG(D > 0, "mdx_ms_aggregate: D=%lld", (long long)D);
    ScalePtrs sp;
    for (int s = 0; s < 8; ++s) {
        sp.p[s] = s < S ? scale_vecs[s] : nullptr;
        MDX_CHECK_ARG(s >= S || sp.p[s], "mdx_ms_aggregate: scale %d is NULL", s);
    }
    hipLaunchKernelGGL(ms_aggregate_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, sp, S, D, msp,
                       1.0f / msp, out);
    MDX_LAUNCH_CHECK();
    return MDX_OK;
}

}  // extern "C"
