// CLAHE pre-processing of the paper's "CLAHE N/D" networks on the device.
//
// Replaces the `apply_clahe` transform of the scenarios -- mdir/components/data/transform/photometric_transforms.py:28-36
// -> functional.ImageClahe.apply (functional.py:106-129): on what `pil2np` makes of the image (float32 RGB in [0,1])
//     spc = (cv2.cvtColor(img, COLOR_RGB2LAB) + [0,128,128]) / [100,255,255]
//     spc[:,:,0] = cv2.createCLAHE(clipLimit, tileGridSize).apply((spc[:,:,0]*255).astype(uint8)) / 255
//     img = cv2.cvtColor(spc*[100,255,255] - [0,128,128], COLOR_LAB2RGB)
// followed by `totensor | normalize`.  OpenCV is third-party arithmetic that is neither under /root/reference nor in the
// build image: the kernels restate OpenCV 4's published algorithms (imgproc/src/clahe.cpp: padding rule of
// CLAHE_Impl::apply, CLAHE_CalcLut_Body, CLAHE_Interpolation_Body; color_lab.cpp: RGB2Lab_f / Lab2RGB_f, sRGB gamma, D65)
// with exact transfer functions where OpenCV interpolates spline tables.  PARITY UNPINNED against OpenCV itself (no copy of
// it exists here to generate vectors with); pinned against oracle.apply_clahe_rgb, the same restatement in numpy.
//
// Two launches per batch (round 4; three before, 0.18 ms per batch of four 1024 x 768 images, most of it the transcendental
// functions of RGB -> Lab evaluated twice per pixel):
//   tile    one workgroup per (image, tile): RGB -> Lab of the tile's pixels ONCE -- sRGB -> linear through a 256-entry table
//           the workgroup builds in LDS with the same powf (the input has 256 values: identical results) -- the uint8
//           lightness into the LDS histogram and, for the image's own pixels, L8 [B,H,W] and the chroma (a, b) [B,H,W,2] fp32 to
//           the workspace; then clip, spread, prefix -> the tile's LUT u8 [B,ty,tx,256]
//   apply   L8 + (a, b) + LUTs -> normalised fp32 CHW   (bilinear LUT blend, Lab -> RGB, (x - mean) / std)
// The tile kernel must finish before any pixel can be finished (a pixel blends the LUTs of four tiles): the launch boundary is
// the grid-wide dependency, not an accident.
#include "mdx_common.h"

// Every product and sum below is rounded on its own, as in OpenCV's scalar code (and in the numpy restatement): hipcc's
// default contraction would fuse a*b + c*d into an fma and move the interpolation's exact .5 ties (HIP's __fmul_rn /
// __fadd_rn are plain operators inlined from a header compiled WITH contraction, so they do not stop it: the helpers below do).
#pragma clang fp contract(off)

namespace mdx {

// defined HERE, under the pragma (the HIP header versions are inlined with the contraction flags of their own scope)
__device__ __forceinline__ float mul_(float a, float b) { return a * b; }
__device__ __forceinline__ float add_(float a, float b) { return a + b; }
__device__ __forceinline__ float sub_(float a, float b) { return a - b; }
__device__ __forceinline__ float div_(float a, float b) { return a / b; }

struct Lab { float L, a, b; };

__device__ __forceinline__ float srgb_to_linear(float c)
{
    return c <= 0.04045f ? c / 12.92f : powf((c + 0.055f) / 1.055f, 2.4f);
}

__device__ __forceinline__ float lab_f(float t) { return t > 0.008856f ? cbrtf(t) : 7.787f * t + (float)(16.0 / 116.0); }

// the uint8 lightness CLAHE sees: ((L + 0) / 100 * 255).astype(uint8) -- truncation (functional.py:117)
__device__ __forceinline__ uint8_t lab_l8(float L) { return (uint8_t)(int)mul_(div_(L, 100.0f), 255.0f); }

__device__ __forceinline__ int reflect101(int i, int n)
{
    if (n == 1) return 0;
    const int period = 2 * (n - 1);
    i = (i < 0 ? -i : i) % period;
    return i >= n ? period - i : i;
}

// RGB2Lab_f with the sRGB -> linear step looked up (lin[v] = srgb_to_linear(v / 255), the same call on the same 256 inputs)
__device__ __forceinline__ Lab rgb8_to_lab_lut(const float *lin, uint8_t r8, uint8_t g8, uint8_t b8)
{
    const float r = lin[r8], g = lin[g8], b = lin[b8];
    constexpr float m00 = (float)(0.412453 / 0.950456), m01 = (float)(0.357580 / 0.950456), m02 = (float)(0.180423 / 0.950456);
    constexpr float m10 = 0.212671f, m11 = 0.715160f, m12 = 0.072169f;
    constexpr float m20 = (float)(0.019334 / 1.088754), m21 = (float)(0.119193 / 1.088754), m22 = (float)(0.950227 / 1.088754);
    const float x = add_(add_(mul_(r, m00), mul_(g, m01)), mul_(b, m02));
    const float y = add_(add_(mul_(r, m10), mul_(g, m11)), mul_(b, m12));
    const float z = add_(add_(mul_(r, m20), mul_(g, m21)), mul_(b, m22));
    const float fx = lab_f(x), fy = lab_f(y), fz = lab_f(z);
    Lab o;
    o.L = y > 0.008856f ? sub_(mul_(116.0f, fy), 16.0f) : mul_(903.3f, y);
    o.a = mul_(500.0f, sub_(fx, fy));
    o.b = mul_(200.0f, sub_(fy, fz));
    return o;
}

constexpr int CLAHE_TILE_THREADS = 512;

// one workgroup per (image, tile): Lab of the tile's pixels (of the virtually padded plane), histogram of their uint8 lightness,
// clip, spread, cumulative LUT; the image's own pixels leave their lightness and chroma for the apply kernel
__global__ __launch_bounds__(CLAHE_TILE_THREADS) void clahe_tile_kernel(const uint8_t *__restrict__ rgb, int H, int W, int tiles_x, int tiles_y,
                                                                       int tile_w, int tile_h, int clip, uint8_t *__restrict__ l8,
                                                                       float2 *__restrict__ ab, uint8_t *__restrict__ luts)
{
    __shared__ float lin[256];
    __shared__ int hist[256];
    __shared__ int scan[256];
    __shared__ int s_clipped;
    const int tid = threadIdx.x;
    const int tile = blockIdx.x % (tiles_x * tiles_y), img = blockIdx.x / (tiles_x * tiles_y);
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int64_t base = (int64_t)img * H * W;
    if (tid < 256) {
        lin[tid] = srgb_to_linear((float)tid / 255.0f);
        hist[tid] = 0;
    }
    if (tid == 0) s_clipped = 0;
    __syncthreads();
    const int area = tile_w * tile_h;
    for (int e = tid; e < area; e += CLAHE_TILE_THREADS) {
        const int py = ty * tile_h + e / tile_w, px = tx * tile_w + e % tile_w;      // in the padded plane
        const int y = reflect101(py, H), x = reflect101(px, W);
        const int64_t pix = base + (int64_t)y * W + x;
        const Lab v = rgb8_to_lab_lut(lin, rgb[3 * pix], rgb[3 * pix + 1], rgb[3 * pix + 2]);
        const uint8_t l = lab_l8(v.L);
        atomicAdd(&hist[l], 1);
        if (py < H && px < W) {                     // the image's own pixel (every one lies in exactly one tile): keep what apply needs
            l8[pix] = l;
            ab[pix] = make_float2(v.a, v.b);
        }
    }
    __syncthreads();
    if (tid >= 256) return;                         // the 256 bins: one thread each (no barrier below is reached by the others)
    int v = hist[tid];
    if (clip > 0) {
        if (v > clip) {
            atomicAdd(&s_clipped, v - clip);
            v = clip;
        }
    }
    // the remaining barriers are among the first 256 threads only: a named sub-group does not exist, so they are waves 0-3 of
    // the workgroup synchronising through LDS with s_barrier -- which counts ALL live waves; the returned waves have ended
    __syncthreads();
    if (clip > 0) {
        const int clipped = s_clipped;
        const int batch = clipped / 256, residual = clipped - batch * 256;
        v += batch;
        if (residual != 0) {
            const int step = 256 / residual > 1 ? 256 / residual : 1;
            // bins 0, step, 2 step, ... get one more, `residual` of them at most (and none beyond bin 255)
            if (tid % step == 0 && tid / step < residual) v += 1;
        }
    }
    // inclusive prefix over the 256 bins (Hillis-Steele in LDS: 8 steps)
    scan[tid] = v;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const int add = tid >= o ? scan[tid - o] : 0;
        __syncthreads();
        scan[tid] += add;
        __syncthreads();
    }
    const float lut_scale = div_(255.0f, (float)area);
    const float r = rintf(mul_((float)scan[tid], lut_scale));          // saturate_cast<uchar>: round half to even, clamp
    luts[(int64_t)blockIdx.x * 256 + tid] = (uint8_t)fminf(fmaxf(r, 0.0f), 255.0f);
}

struct ClaheNorm { float mean[3], std[3]; };

// linear -> sRGB on the OUTPUT side: c^(1/2.4) through the hardware's exp2 / log2 (a few ulp; nothing is truncated after it --
// the value goes straight into (x - mean) / std -- and OpenCV itself interpolates a spline table here).  The INPUT side keeps
// powf: its result is truncated to the uint8 lightness CLAHE sees, and it is evaluated 256 times per workgroup, not per pixel.
__device__ __forceinline__ float linear_to_srgb_fast(float c)
{
    return c <= 0.0031308f ? c * 12.92f : 1.055f * __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(c) * (float)(1.0 / 2.4)) - 0.055f;
}

// one pixel: CLAHE_Interpolation_Body on its lightness, then Lab -> RGB -> (x - mean) / std
__device__ __forceinline__ void clahe_finish_pixel(int x, int y, int v, float2 chroma, const uint8_t *__restrict__ luts_img, int tiles_x, int tiles_y,
                                                   int tile_w, int tile_h, const ClaheNorm &nrm, uint8_t &l_eq, float (&o)[3])
{
    // inv_tw = 1.0f / tileSize.width as an IEEE division (a plain `/` may become v_rcp_f32: one ulp off flips exact .5 ties below)
    const float txf = sub_(mul_((float)x, div_(1.0f, (float)tile_w)), 0.5f);
    const float tyf = sub_(mul_((float)y, div_(1.0f, (float)tile_h)), 0.5f);
    int tx1 = (int)floorf(txf), ty1 = (int)floorf(tyf);
    const float xa = sub_(txf, (float)tx1), ya = sub_(tyf, (float)ty1);
    const float xa1 = sub_(1.0f, xa), ya1 = sub_(1.0f, ya);
    int tx2 = tx1 + 1, ty2 = ty1 + 1;
    tx1 = tx1 < 0 ? 0 : tx1;
    ty1 = ty1 < 0 ? 0 : ty1;
    tx2 = tx2 > tiles_x - 1 ? tiles_x - 1 : tx2;
    ty2 = ty2 > tiles_y - 1 ? tiles_y - 1 : ty2;
    const uint8_t *lt = luts_img + v;
    const float l11 = lt[(ty1 * tiles_x + tx1) * 256], l12 = lt[(ty1 * tiles_x + tx2) * 256];
    const float l21 = lt[(ty2 * tiles_x + tx1) * 256], l22 = lt[(ty2 * tiles_x + tx2) * 256];
    const float top = add_(mul_(l11, xa1), mul_(l12, xa)), bot = add_(mul_(l21, xa1), mul_(l22, xa));
    const float res = add_(mul_(top, ya1), mul_(bot, ya));
    const float l_new8 = fminf(fmaxf(rintf(res), 0.0f), 255.0f);
    l_eq = (uint8_t)l_new8;                         // the equalised lightness, kept for the caller (1 byte per pixel)
    // back through the reference's normalised space: spc = (lab + [0,128,128]) / [100,255,255]; spc[0] = clahe / 255;
    // lab' = spc * [100,255,255] - [0,128,128]   (float32 steps as written there)
    const float L = mul_(div_(l_new8, 255.0f), 100.0f);
    const float a = sub_(mul_(div_(add_(chroma.x, 128.0f), 255.0f), 255.0f), 128.0f);
    const float b = sub_(mul_(div_(add_(chroma.y, 128.0f), 255.0f), 255.0f), 128.0f);
    // Lab2RGB_f
    constexpr float lthresh = (float)(0.008856 * 903.3), fthresh = (float)(7.787 * 0.008856 + 16.0 / 116.0);
    float fy, yy;
    if (L <= lthresh) {
        yy = div_(L, 903.3f);
        fy = add_(mul_(7.787f, yy), (float)(16.0 / 116.0));
    } else {
        fy = div_(add_(L, 16.0f), 116.0f);
        yy = mul_(mul_(fy, fy), fy);
    }
    const float fx = add_(div_(a, 500.0f), fy), fz = sub_(fy, div_(b, 200.0f));
    const float xx = fx <= fthresh ? div_(sub_(fx, (float)(16.0 / 116.0)), 7.787f) : mul_(mul_(fx, fx), fx);
    const float zz = fz <= fthresh ? div_(sub_(fz, (float)(16.0 / 116.0)), 7.787f) : mul_(mul_(fz, fz), fz);
    constexpr float k00 = (float)(3.240479 * 0.950456), k01 = -1.53715f, k02 = (float)(-0.498535 * 1.088754);
    constexpr float k10 = (float)(-0.969256 * 0.950456), k11 = 1.875991f, k12 = (float)(0.041556 * 1.088754);
    constexpr float k20 = (float)(0.055648 * 0.950456), k21 = -0.204043f, k22 = (float)(1.057311 * 1.088754);
    float c[3];
    c[0] = add_(add_(mul_(xx, k00), mul_(yy, k01)), mul_(zz, k02));
    c[1] = add_(add_(mul_(xx, k10), mul_(yy, k11)), mul_(zz, k12));
    c[2] = add_(add_(mul_(xx, k20), mul_(yy, k21)), mul_(zz, k22));
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const float s = linear_to_srgb_fast(fminf(fmaxf(c[ch], 0.0f), 1.0f));
        o[ch] = div_(sub_(s, nrm.mean[ch]), nrm.std[ch]);         // totensor (no scaling of a float image) | normalize
    }
}

// VEC = 4: a thread finishes four consecutive pixels of a row (W % 4 == 0): one 4-byte lightness load, two 16-byte chroma
// loads, a 4-byte and three 16-byte stores; VEC = 1: any width
template <int VEC>
__global__ __launch_bounds__(256) void clahe_apply_kernel(const float2 *__restrict__ ab, const uint8_t *__restrict__ l8,
                                                          const uint8_t *__restrict__ luts, int H, int W, int tiles_x, int tiles_y,
                                                          int tile_w, int tile_h, ClaheNorm nrm, uint8_t *__restrict__ l8_out,
                                                          float *__restrict__ out)
{
    const int64_t hw = (int64_t)H * W;
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC;
    const int img = blockIdx.y;
    if (i >= hw) return;
    const int y = (int)(i / W), x = (int)(i % W);
    const int64_t pix = (int64_t)img * hw + i;
    const uint8_t *luts_img = luts + (int64_t)img * tiles_x * tiles_y * 256;
    uint8_t v[VEC], leq[VEC];
    float2 chroma[VEC];
    float o[VEC][3];
    if constexpr (VEC == 4) {
        const uint32_t w = *(const uint32_t *)(l8 + pix);
        const float4 c01 = *(const float4 *)(ab + pix), c23 = *(const float4 *)(ab + pix + 2);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (uint8_t)(w >> (8 * k));
        chroma[0] = make_float2(c01.x, c01.y);
        chroma[1] = make_float2(c01.z, c01.w);
        chroma[2] = make_float2(c23.x, c23.y);
        chroma[3] = make_float2(c23.z, c23.w);
    } else {
        v[0] = l8[pix];
        chroma[0] = ab[pix];
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) clahe_finish_pixel(x + k, y, v[k], chroma[k], luts_img, tiles_x, tiles_y, tile_w, tile_h, nrm, leq[k], o[k]);
    if constexpr (VEC == 4) {
        *(uint32_t *)(l8_out + pix) = (uint32_t)leq[0] | ((uint32_t)leq[1] << 8) | ((uint32_t)leq[2] << 16) | ((uint32_t)leq[3] << 24);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
            *(float4 *)(out + ((int64_t)img * 3 + ch) * hw + i) = make_float4(o[0][ch], o[1][ch], o[2][ch], o[3][ch]);
    } else {
        l8_out[pix] = leq[0];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) out[((int64_t)img * 3 + ch) * hw + i] = o[0][ch];
    }
}

static void clahe_geometry(int H, int W, int tiles_x, int tiles_y, int *tile_w, int *tile_h)
{
    int eh = H, ew = W;
    if (!(W % tiles_x == 0 && H % tiles_y == 0)) {      // OpenCV pads BOTH sides as soon as one does not divide
        eh = H + (tiles_y - H % tiles_y);
        ew = W + (tiles_x - W % tiles_x);
    }
    *tile_w = ew / tiles_x;
    *tile_h = eh / tiles_y;
}

}  // namespace mdx

using namespace mdx;

extern "C" {

int64_t mdx_clahe_workspace(int64_t B, int64_t H, int64_t W, int tiles_x, int tiles_y)
{
    if (B <= 0 || H <= 0 || W <= 0 || tiles_x <= 0 || tiles_y <= 0) return 0;
    return 2 * round_up(B * H * W, 256) + round_up(B * (int64_t)tiles_x * tiles_y * 256, 256) + 8 * B * H * W;     // + chroma (a, b) fp32
}

int mdx_clahe_u8_to_chw(const uint8_t *rgb, int64_t B, int64_t H, int64_t W, int clip_limit, int tiles_x, int tiles_y,
                        const float *mean, const float *std, void *workspace, int64_t workspace_bytes, float *out, void *stream)
{
    MDX_CHECK_ARG(rgb && out && mean && std, "mdx_clahe_u8_to_chw: NULL pointer");
    MDX_CHECK_ARG(B > 0 && H > 0 && W > 0 && H < (1 << 24) && W < (1 << 24) && B < 65536, "mdx_clahe_u8_to_chw: bad sizes");
    MDX_CHECK_ARG(tiles_x >= 1 && tiles_y >= 1 && tiles_x <= 256 && tiles_y <= 256 && clip_limit >= 0, "mdx_clahe_u8_to_chw: bad grid or clip limit");
    const int64_t need = mdx_clahe_workspace(B, H, W, tiles_x, tiles_y);
    if (!workspace || workspace_bytes < need) {
        set_error("mdx_clahe_u8_to_chw: workspace %lld B < required %lld B", (long long)workspace_bytes, (long long)need);
        return MDX_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    uint8_t *l8 = (uint8_t *)workspace, *luts = l8 + round_up(B * H * W, 256);
    uint8_t *l8_out = luts + round_up(B * (int64_t)tiles_x * tiles_y * 256, 256);
    float2 *ab = (float2 *)(l8_out + round_up(B * H * W, 256));
    int tw = 0, th = 0;
    clahe_geometry((int)H, (int)W, tiles_x, tiles_y, &tw, &th);
    const int area = tw * th;
    int clip = 0;
    if (clip_limit > 0) {
        clip = (int)((double)clip_limit * area / 256);          // static_cast<int>(clipLimit_ * tileSizeTotal / histSize)
        clip = clip > 1 ? clip : 1;
    }
    hipLaunchKernelGGL(clahe_tile_kernel, dim3((unsigned)(B * tiles_x * tiles_y)), dim3(CLAHE_TILE_THREADS), 0, s, rgb, (int)H, (int)W, tiles_x,
                       tiles_y, tw, th, clip, l8, ab, luts);
    ClaheNorm nrm;
    for (int c = 0; c < 3; ++c) {
        nrm.mean[c] = mean[c];
        nrm.std[c] = std[c];
    }
    // four pixels per thread where every row (and with it every plane: H * W, the workspace regions' 256-byte starts, the
    // caller's `out`) keeps 16-byte alignment
    if (W % 4 == 0 && ((uintptr_t)out & 15) == 0)
        hipLaunchKernelGGL(clahe_apply_kernel<4>, dim3((unsigned)ceil_div(H * W / 4, 256), (unsigned)B), dim3(256), 0, s, (const float2 *)ab,
                           (const uint8_t *)l8, (const uint8_t *)luts, (int)H, (int)W, tiles_x, tiles_y, tw, th, nrm, l8_out, out);
    else
        hipLaunchKernelGGL(clahe_apply_kernel<1>, dim3((unsigned)ceil_div(H * W, 256), (unsigned)B), dim3(256), 0, s, (const float2 *)ab,
                           (const uint8_t *)l8, (const uint8_t *)luts, (int)H, (int)W, tiles_x, tiles_y, tw, th, nrm, l8_out, out);
    MDX_LAUNCH_CHECK();
    return MDX_OK;
}

}  // extern "C"
