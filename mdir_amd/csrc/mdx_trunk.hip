// Trunk epilogue: inference batch-norm + residual add + ReLU of a convolution output in ONE
// in-place pass.
//
// The convolutions of the backbone stay on MIOpen (BASELINE.json north_star), but between them
// torchvision-style residual blocks (mirrored in mdir_amd/backbones.py; torchvision resnet.py
// Bottleneck.forward as kept by cirtorch/networks/imageretrievalnet.py:172-173) run three separate
// full-tensor kernels: batch_norm (read+write), `out += identity` (2 reads + write), relu
// (read+write).  On ResNet101 at 1024x768 that is 711 launches and a third of the trunk's GPU time,
// and the trunk is launch-bound on top.  HBM-bound: one read (+ one for the residual) and one write
// per element.
#include "mdx_common.h"

namespace mdx {

struct BnArgs {
    const float *mean, *var, *weight, *bias;
    float eps;
};

__device__ __forceinline__ void bn_coeffs(const BnArgs &a, int c, float &mean, float &scale, float &shift)
{
    mean = a.mean ? a.mean[c] : 0.0f;
    const float invstd = a.var ? 1.0f / sqrtf(a.var[c] + a.eps) : 1.0f;
    scale = a.weight ? invstd * a.weight[c] : invstd;
    shift = a.bias ? a.bias[c] : 0.0f;
}

// One workgroup = a run of one (image, channel) plane, so the four per-channel statistics are
// scalar loads and no thread divides anything.  VEC = 4: planes are a multiple of 4 long and 16-B
// aligned (float4 accesses); VEC = 1 covers odd plane sizes.  Each thread takes UNR elements, all
// loads issued before the first store.
constexpr int BN_UNR = 4;

template <int VEC, bool RES, bool RELU>
__global__ __launch_bounds__(256) void bn_act_kernel(float *__restrict__ x, const float *__restrict__ res,
                                                     int64_t plane0, unsigned hw_vec, unsigned C, BnArgs a)
{
    const int64_t plane = plane0 + blockIdx.y;
    float mean, scale, shift;
    bn_coeffs(a, (int)(plane % C), mean, scale, shift);
    const int64_t base = plane * hw_vec;
    const unsigned i0 = blockIdx.x * (256 * BN_UNR) + threadIdx.x;
    if (VEC == 4) {
        float4 v[BN_UNR], r[BN_UNR];
#pragma unroll
        for (int u = 0; u < BN_UNR; ++u) {
            const unsigned i = i0 + u * 256;
            if (i < hw_vec) {
                v[u] = ((const float4 *)x)[base + i];
                if (RES) r[u] = ((const float4 *)res)[base + i];
            }
        }
#pragma unroll
        for (int u = 0; u < BN_UNR; ++u) {
            const unsigned i = i0 + u * 256;
            if (i >= hw_vec) continue;
            float4 o;
            o.x = fmaf(v[u].x - mean, scale, shift) + (RES ? r[u].x : 0.f);
            o.y = fmaf(v[u].y - mean, scale, shift) + (RES ? r[u].y : 0.f);
            o.z = fmaf(v[u].z - mean, scale, shift) + (RES ? r[u].z : 0.f);
            o.w = fmaf(v[u].w - mean, scale, shift) + (RES ? r[u].w : 0.f);
            if (RELU) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
            ((float4 *)x)[base + i] = o;
        }
    } else {
#pragma unroll
        for (int u = 0; u < BN_UNR; ++u) {
            const unsigned i = i0 + u * 256;
            if (i >= hw_vec) continue;
            const float o = fmaf(x[base + i] - mean, scale, shift) + (RES ? res[base + i] : 0.0f);
            x[base + i] = RELU ? fmaxf(o, 0.f) : o;
        }
    }
}

template <int VEC>
static void launch_bn_act(float *x, const float *res, int64_t planes, unsigned hw_vec, unsigned C, const BnArgs &a,
                          bool relu, hipStream_t s)
{
    const unsigned gx = (unsigned)ceil_div((int64_t)hw_vec, (int64_t)(256 * BN_UNR));
    for (int64_t p0 = 0; p0 < planes; p0 += 65535) {
        const unsigned gy = (unsigned)((planes - p0) < 65535 ? (planes - p0) : 65535);
        const dim3 grid(gx, gy);
        if (res && relu) hipLaunchKernelGGL((bn_act_kernel<VEC, true, true>), grid, dim3(256), 0, s, x, res, p0, hw_vec, C, a);
        else if (res)    hipLaunchKernelGGL((bn_act_kernel<VEC, true, false>), grid, dim3(256), 0, s, x, res, p0, hw_vec, C, a);
        else if (relu)   hipLaunchKernelGGL((bn_act_kernel<VEC, false, true>), grid, dim3(256), 0, s, x, res, p0, hw_vec, C, a);
        else             hipLaunchKernelGGL((bn_act_kernel<VEC, false, false>), grid, dim3(256), 0, s, x, res, p0, hw_vec, C, a);
    }
}

}  // namespace mdx

using namespace mdx;

extern "C" int mdx_bn_act(float *x, const float *residual, int64_t N, int64_t C, int64_t HW, const float *mean,
                          const float *var, const float *weight, const float *bias, float eps, int relu,
                          void *stream)
{
    MDX_CHECK_ARG(x, "mdx_bn_act: NULL x");
    MDX_CHECK_ARG((mean == nullptr) == (var == nullptr), "mdx_bn_act: mean and var must both be given or both be NULL");
    MDX_CHECK_ARG(N > 0 && C > 0 && HW > 0, "mdx_bn_act: N=%lld C=%lld HW=%lld must be positive", (long long)N,
                  (long long)C, (long long)HW);
    MDX_CHECK_ARG(C < (1ll << 31) && HW < (1ll << 31), "mdx_bn_act: C or HW too large");
    MDX_CHECK_ARG(eps >= 0.0f, "mdx_bn_act: eps=%g must be >= 0", (double)eps);
    const BnArgs a{mean, var, weight, bias, eps};
    const bool vec = (HW % 4 == 0) && (((uintptr_t)x & 15) == 0) && (!residual || ((uintptr_t)residual & 15) == 0);
    if (vec) launch_bn_act<4>(x, residual, N * C, (unsigned)(HW / 4), (unsigned)C, a, relu != 0, (hipStream_t)stream);
    else     launch_bn_act<1>(x, residual, N * C, (unsigned)HW, (unsigned)C, a, relu != 0, (hipStream_t)stream);
    MDX_LAUNCH_CHECK();
    return MDX_OK;
}

// ---------------------------------------------------------------------------
// Input conversion on the device: uint8 HWC image -> normalised fp32 CHW tensor, i.e. the
// `pil2np | totensor | normalize` chain of the eval scenarios (mdir/components/data/transform/
// core_transforms.py:33-63; cirtorch's ToTensor + Normalize) with its exact fp32 operation order
// ((u / 255 - mean) / std, IEEE divisions), so the loader workers ship 1 byte per value instead of 4.
// ---------------------------------------------------------------------------
namespace mdx {

struct NormArgs { float mean[4], std[4]; };

template <int C>
__global__ __launch_bounds__(256) void u8_to_chw_kernel(const uint8_t *__restrict__ src, float *__restrict__ dst,
                                                        int64_t pixels_per_image, int64_t total_pixels, NormArgs a)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total_pixels; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / pixels_per_image, p = i - b * pixels_per_image;
        const uint8_t *s = src + i * C;
        float *d = dst + b * C * pixels_per_image + p;
#pragma unroll
        for (int c = 0; c < C; ++c) d[c * pixels_per_image] = ((float)s[c] / 255.0f - a.mean[c]) / a.std[c];
    }
}

}  // namespace mdx

extern "C" int mdx_u8_to_chw(const uint8_t *hwc, int64_t B, int64_t H, int64_t W, int C, const float *mean,
                             const float *std, float *out, void *stream)
{
    MDX_CHECK_ARG(hwc && out && mean && std, "mdx_u8_to_chw: NULL pointer");
    MDX_CHECK_ARG(B > 0 && H > 0 && W > 0, "mdx_u8_to_chw: B=%lld H=%lld W=%lld must be positive", (long long)B,
                  (long long)H, (long long)W);
    MDX_CHECK_ARG(C == 1 || C == 3, "mdx_u8_to_chw: C=%d (1 or 3 channels)", C);
    NormArgs a{};
    for (int c = 0; c < C; ++c) {
        MDX_CHECK_ARG(std[c] != 0.0f, "mdx_u8_to_chw: std[%d] is zero", c);
        a.mean[c] = mean[c];
        a.std[c] = std[c];
    }
    const int64_t px = H * W, total = B * px;
    const int64_t want = ceil_div(total, (int64_t)256);
    const unsigned grid = (unsigned)(want < 256 * 16 ? want : 256 * 16);
    if (C == 3) hipLaunchKernelGGL((u8_to_chw_kernel<3>), dim3(grid), dim3(256), 0, (hipStream_t)stream, hwc, out, px, total, a);
    else        hipLaunchKernelGGL((u8_to_chw_kernel<1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, hwc, out, px, total, a);
    MDX_LAUNCH_CHECK();
    return MDX_OK;
}

// ---------------------------------------------------------------------------
// Image down-scaling on the device: one pass (along the width or the height) of Pillow's separable resampling of
// 8-bit images -- what `img.thumbnail((imsize, imsize), Image.ANTIALIAS)` of imresize runs
// (cirtorch/datasets/datahelpers.py:48-50; Pillow src/libImaging/Resample.c ImagingResampleHorizontal_8bpc /
// ImagingResampleVertical_8bpc) -- with the taps precomputed on the host in Pillow's fixed point (2^22):
//     out = clip8((2^21 + sum_t src[first + t] * k[t]) >> 22)
// Integer arithmetic, so the thumbnail is Pillow's pixel for pixel; the loader workers then only decode.
// ---------------------------------------------------------------------------
namespace mdx {

template <int AXIS>
__global__ __launch_bounds__(256) void resample_u8_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int H,
                                                          int W, int C, int out_len, const int32_t *__restrict__ bounds,
                                                          const int32_t *__restrict__ k, int ksize, int64_t total)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    int64_t base, stride;
    int o;
    if (AXIS == 1) {            // [B,H,W,C] -> [B,H,out,C]
        const int c = (int)(idx % C);
        const int64_t r = idx / C;
        o = (int)(r % out_len);
        base = (r / out_len) * W * C + c;
        stride = C;
    } else {                    // [B,H,W,C] -> [B,out,W,C]
        const int64_t line = (int64_t)W * C;
        const int64_t e = idx % line, r = idx / line;
        o = (int)(r % out_len);
        base = (r / out_len) * H * line + e;
        stride = line;
    }
    const int first = bounds[2 * o], count = bounds[2 * o + 1];
    const int32_t *kk = k + (int64_t)o * ksize;
    const uint8_t *s = src + base + first * stride;
    int32_t acc = 1 << 21;
    for (int t = 0; t < count; ++t) acc += (int32_t)s[t * stride] * kk[t];
    acc >>= 22;
    dst[idx] = (uint8_t)(acc < 0 ? 0 : (acc > 255 ? 255 : acc));
}

}  // namespace mdx

extern "C" int mdx_resample_u8(const uint8_t *src, int64_t B, int H, int W, int C, int axis, int out_len,
                               const int32_t *bounds, const int32_t *k, int ksize, uint8_t *dst, void *stream)
{
    MDX_CHECK_ARG(src && dst && bounds && k, "mdx_resample_u8: NULL pointer");
    MDX_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C <= 4, "mdx_resample_u8: bad shape [%lld,%d,%d,%d]", (long long)B, H, W, C);
    MDX_CHECK_ARG((axis == 0 || axis == 1) && out_len > 0 && ksize > 0, "mdx_resample_u8: axis=%d out=%d ksize=%d", axis, out_len, ksize);
    const int64_t total = axis == 1 ? B * H * (int64_t)out_len * C : B * (int64_t)out_len * W * C;
    const dim3 grid((unsigned)ceil_div(total, (int64_t)256)), blk(256);
    if (axis == 1)
        hipLaunchKernelGGL((mdx::resample_u8_kernel<1>), grid, blk, 0, (hipStream_t)stream, src, dst, H, W, C, out_len, bounds, k, ksize, total);
    else
        hipLaunchKernelGGL((mdx::resample_u8_kernel<0>), grid, blk, 0, (hipStream_t)stream, src, dst, H, W, C, out_len, bounds, k, ksize, total);
    MDX_LAUNCH_CHECK();
    return MDX_OK;
}

// ---------------------------------------------------------------------------
// The image pyramid: `F.interpolate(x, scale_factor=s, mode='bilinear', align_corners=False)` of
// CirMultiscaleAggregation.preprocess (mdir/components/data/wrapper.py:104-107) and extract_ms
// (cirtorch/networks/imageretrievalnet.py:315) for every scale of the pyramid in ONE launch.  torch >= 1.6 semantics
// (SURVEY quirk Q6): output size floor(in * s); source coordinate (dst + 0.5) / s - 0.5 clamped at 0, computed in fp32
// with 1 / s rounded to fp32 first; the four neighbours blended as
//     h0 * (w0 * a + w1 * b) + h1 * (w0 * c + w1 * d),     h1 = frac(y), h0 = 1 - h1  (and likewise w).
// ---------------------------------------------------------------------------
namespace mdx {

struct PyramidLevels {
    float *out[8];
    int h[8], w[8];
    float rh[8], rw[8];         // 1 / scale as fp32
    int64_t first[9];           // first output element of level l
};

__global__ __launch_bounds__(256) void bilinear_pyramid_kernel(const float *__restrict__ src, int64_t planes, int H, int W,
                                                               PyramidLevels lv, int L)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= lv.first[L]) return;
    int l = 0;
    while (l + 1 < L && idx >= lv.first[l + 1]) ++l;
    const int64_t e = idx - lv.first[l];
    const int w = lv.w[l], h = lv.h[l];
    const int x = (int)(e % w), y = (int)((e / w) % h);
    const int64_t plane = e / ((int64_t)w * h);
    float fy = lv.rh[l] * ((float)y + 0.5f) - 0.5f, fx = lv.rw[l] * ((float)x + 0.5f) - 0.5f;
    fy = fy < 0.0f ? 0.0f : fy;
    fx = fx < 0.0f ? 0.0f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int yp = y0 < H - 1 ? 1 : 0, xp = x0 < W - 1 ? 1 : 0;
    const float h1 = fy - (float)y0, h0 = 1.0f - h1, w1 = fx - (float)x0, w0 = 1.0f - w1;
    const float *p = src + plane * (int64_t)H * W + (int64_t)y0 * W + x0;
    lv.out[l][e] = h0 * (w0 * p[0] + w1 * p[xp]) + h1 * (w0 * p[(int64_t)yp * W] + w1 * p[(int64_t)yp * W + xp]);
}

}  // namespace mdx

extern "C" int mdx_bilinear_pyramid(const float *src, int64_t B, int64_t C, int H, int W, int L, const double *scales,
                                    float *const *outs, void *stream)
{
    MDX_CHECK_ARG(src && scales && outs, "mdx_bilinear_pyramid: NULL pointer");
    MDX_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0, "mdx_bilinear_pyramid: bad shape [%lld,%lld,%d,%d]", (long long)B, (long long)C, H, W);
    MDX_CHECK_ARG(L >= 1 && L <= 8, "mdx_bilinear_pyramid: %d levels, 1..8 supported", L);
    mdx::PyramidLevels lv;
    lv.first[0] = 0;
    for (int l = 0; l < 8; ++l) {
        lv.out[l] = nullptr;
        lv.h[l] = lv.w[l] = 0;
        lv.rh[l] = lv.rw[l] = 1.0f;
        if (l < L) {
            MDX_CHECK_ARG(scales[l] > 0.0 && outs[l], "mdx_bilinear_pyramid: level %d: scale %g or NULL output", l, scales[l]);
            lv.out[l] = outs[l];
            lv.h[l] = (int)floor((double)H * scales[l]);
            lv.w[l] = (int)floor((double)W * scales[l]);
            MDX_CHECK_ARG(lv.h[l] > 0 && lv.w[l] > 0, "mdx_bilinear_pyramid: level %d is empty", l);
            lv.rh[l] = lv.rw[l] = (float)(1.0 / scales[l]);
        }
        lv.first[l + 1] = lv.first[l] + (l < L ? B * C * (int64_t)lv.h[l] * lv.w[l] : 0);
    }
    hipLaunchKernelGGL(mdx::bilinear_pyramid_kernel, dim3((unsigned)ceil_div(lv.first[L], (int64_t)256)), dim3(256), 0, (hipStream_t)stream,
                       src, B * C, H, W, lv, L);
    MDX_LAUNCH_CHECK();
    return MDX_OK;
}
