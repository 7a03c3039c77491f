/* mdx.h -- C ABI of libmdx.so: the MI355X (gfx950) implementation of the
 * descriptor-extraction-and-ranking hot path of jenicek/mdir + cirtorch.
 *
 * This is the drop-in boundary.  Every entry point replaces one statement (or a
 * short run of statements) of the reference; the reference location is cited
 * per function (paths relative to the upstream repository).  The reference is
 * pure Python, so "what its FFI would bind" is a ctypes stub -- INTEGRATION.md
 * shows it for each call site.
 *
 * Conventions
 *  - All data pointers are DEVICE pointers (hipMalloc / torch CUDA tensors) unless
 *    a parameter says "host".  The caller owns every buffer; the library allocates
 *    nothing persistent except inside an mdx_index (explicit create/destroy).
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls
 *    only enqueue work and do not synchronise the device, with two exceptions:
 *    mdx_index_create / _ex / _destroy allocate / free (creating an fp32 shard also waits
 *    for its build: see there), and creation also waits for a
 *    one-off probe kernel (a few hundred microseconds per device and process) that
 *    settles how the sort ranks inside a wave; a process that ranks WITHOUT ever
 *    creating an index runs that probe in its first mdx_rank_* / mdx_topk call instead
 *    (and waits for it once) -- unless that stream is being captured, in which case
 *    nothing synchronises and the probe-free kernels are recorded.  Everything but
 *    index creation / destruction and mdx_comm_init / _destroy is safe under graph capture.
 *  - Every function returns MDX_OK (0) or a negative mdx_status; the message of
 *    the last failure on the calling thread is mdx_last_error().  Nothing aborts.
 *  - One host thread per device at a time; handles are not internally locked.
 *  - Matrices are dense fp32.  "row-major [a,b]" means element (i,j) at i*b+j.
 */
#ifndef MDX_H
#define MDX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: round 4's additions (mdx_index_bytes, mdx_index_create_in, mdx_scores_rowmajor, mdx_scores_ex / _workspace_ex,
 * mdx_rank_positions) and the stream synchronisation inside mdx_index_create* for fp32 shards (the shard maximum is read
 * back): a host built against version 1 must not load this library unnoticed. */
/* 3: round 6's additions (the direct-store exchange: mdx_p2p_* and mdx_scores_p2p); nothing of version 2 changed. */
#define MDX_ABI_VERSION 3

typedef enum mdx_status {
    MDX_OK = 0,
    MDX_ERR_INVALID = -1,   /* bad argument (NULL, negative size, unsupported value) */
    MDX_ERR_RUNTIME = -2,   /* a HIP runtime call failed (message has hipGetErrorString) */
    MDX_ERR_NOMEM = -3,     /* device allocation failed */
    MDX_ERR_WORKSPACE = -4  /* caller-provided workspace too small */
} mdx_status;

/* Layout of a descriptor matrix handed to the library. */
typedef enum mdx_layout {
    MDX_DIM_MAJOR = 0,  /* [D,N]: the reference's `vecs` (imageretrievalnet.py:291) */
    MDX_ROW_MAJOR = 1   /* [N,D]: one descriptor per row */
} mdx_layout;

/* Global pooling kinds (cirtorch/networks/imageretrievalnet.py:32-37; rmac is out
 * of scope). */
/* Storage type of a shard.  MDX_F32 is the exact path (k-ordered fp32 fma chain).  MDX_F16
 * stores descriptors (and the queries of a call) as IEEE fp16 and multiplies them on the
 * fp16 MFMA with fp32 accumulation (BASELINE.json configs[4]); scores then carry fp16 input
 * rounding (~1e-3 relative) -- a separate, looser parity contract (tests/test_gpu_f16.py). */
typedef enum mdx_storage { MDX_F32 = 0, MDX_F16 = 1 } mdx_storage;

typedef enum mdx_pool_kind { MDX_POOL_GEM = 0, MDX_POOL_MAC = 1, MDX_POOL_SPOC = 2 } mdx_pool_kind;

int mdx_abi_version(void);
const char *mdx_last_error(void);
/* After a stream capture that was INVALIDATED (a call that is illegal while capturing: a synchronisation, a host read): ends the
 * capture if `stream` is still in it (the half-built graph is destroyed) and clears the runtime's per-thread last-error, which
 * otherwise makes the next -- perfectly legal -- launch check of the host framework report "operation failed due to a previous
 * error during capture".  The host of this path captures one graph per input shape (mdir_amd/graphs.py) and stays eager for a shape
 * whose capture was refused; without this call it could not.  No reference counterpart (the reference has no graphs). */
int mdx_capture_recover(void *stream);

/* ---------------------------------------------------------------- extraction */

/* Global pooling of a feature-map batch followed by L2 normalisation over channels.
 *   feat [B,C,H,W] row-major  ->  out [B,C]
 * Replaces `self.norm(self.pool(o))` of ImageRetrievalNet.forward
 * (cirtorch/networks/imageretrievalnet.py:108) = LF.gem / LF.mac / LF.spoc
 * (cirtorch/layers/functional.py:11-22) then LF.l2n (functional.py:130-131).
 *   p, pool_eps: GeM exponent and clamp (ignored for mac/spoc).
 *   l2n_eps    : added to the norm; pass a NEGATIVE value to skip normalisation. */
int mdx_pool_l2n(const float *feat, int B, int C, int H, int W, int kind, float p,
                 float pool_eps, float l2n_eps, float *out, void *stream);

/* R-MAC pooling: feat [B,C,H,W] row-major  ->  out [B,C] (NOT yet L2-normalised as a whole: the network's `self.norm` follows).
 * Replaces `LF.rmac(x, L, eps)` (cirtorch/layers/functional.py:26-72; module RMAC, layers/pooling.py:50-60):
 *     v = l2n(max over the map);  for every region of the grid: v += l2n(max over the region)      (l2n: eps added to the norm)
 *   regions : HOST array [nregions][4] int32 = (row0, col0, height, width), the whole map first, then the grid in the
 *             reference's order (levels 1..L, rows of centres outer, columns inner); the host restates the reference's float32
 *             grid arithmetic (mdir_amd/layers.py: rmac_regions).  At most 64 regions (L = 3 gives 15-51).
 *   workspace: mdx_rmac_workspace(B, C, nregions) bytes of device memory (the regions' maxima). */
int64_t mdx_rmac_workspace(int B, int C, int nregions);
int mdx_rmac(const float *feat, int B, int C, int H, int W, const int32_t *regions, int nregions, float eps,
             void *workspace, int64_t workspace_bytes, float *out, void *stream);

/* Regional pooling: feat [B,C,H,W]  ->  out [B, nregions, C]: the pooling `kind` (p, pool_eps as in mdx_pool_l2n) of every
 * region, no normalisation.  Replaces `LF.roipool(x, rpool, L, eps)` (cirtorch/layers/functional.py:75-121) inside `Rpool.forward`
 * (layers/pooling.py:62-95: `regional: True`); regions as in mdx_rmac. */
int mdx_roipool(const float *feat, int B, int C, int H, int W, const int32_t *regions, int nregions, int kind, float p,
                float pool_eps, float *out, void *stream);

/* vecs [B, nregions, C]  ->  out [B, C] = the sum over the regions in order; l2n_eps >= 0: every region vector is L2-normalised
 * (eps added to the norm) before it is added (R-MAC, functional.py:62-70); l2n_eps < 0: summed as they are (`o.sum(1)`,
 * layers/pooling.py:91). */
int mdx_region_sum(const float *vecs, int B, int nregions, int C, float l2n_eps, float *out, void *stream);

/* The S feature maps of one image pyramid (or of a batch of B equal-sized images), pooled by ONE launch:
 *   feats[s] [B,C,H[s],W[s]] row-major  ->  pooled [S,B,C]   (no normalisation)
 * Replaces the S calls of `self.pool(o)` (imageretrievalnet.py:108; LF.gem / LF.mac / LF.spoc,
 * functional.py:11-22) that CirMultiscaleAggregation (mdir/components/data/wrapper.py:104-107) and extract_ms
 * (imageretrievalnet.py:315-318) cause, one per scale.  Per plane the arithmetic of mdx_pool_l2n. */
int mdx_pool_multi(const float *const *feats, int S, int B, int C, const int *H, const int *W, int kind,
                   float p, float pool_eps, float *pooled, void *stream);

/* Descriptor tail of a pyramid in ONE launch:  pooled [S,B,D] -> out [B,D]
 *   per scale  v_s = pooled[s,b,:] / (||pooled[s,b,:]||_2 + l2n_eps)          LF.l2n, functional.py:130-131
 *   then       out = (sum_s v_s^msp / S)^(1/msp);  out /= ||out||_2 (no eps)  wrapper.py:112-117, imageretrievalnet.py:319-322
 * Bit-identical to mdx_l2n_rows on every scale followed by mdx_ms_aggregate_batch. */
int mdx_l2n_aggregate(const float *pooled, int S, int64_t B, int64_t D, float l2n_eps, float msp, float *out,
                      void *stream);

/* In place: x[r,:] = (x[r,:] + bias) / (||x[r,:] + bias||_2 + eps) for R rows of
 * length D; bias may be NULL.  LF.l2n (functional.py:130-131); with bias it is the
 * tail of the in-network whitening `self.norm(self.whiten(o))`
 * (imageretrievalnet.py:111-112). */
int mdx_l2n_rows(float *x, int64_t R, int64_t D, const float *bias, float eps, void *stream);

/* Trunk epilogue, in place on a convolution output x[N,C,H*W] (NCHW, contiguous):
 *   x = act( (x - mean[c]) * weight[c] / sqrt(var[c] + eps) + bias[c]  (+ residual) ),  act = ReLU or identity
 * i.e. inference `bn(x)`, `out += identity`, `relu(out)` of a residual block (mdir_amd/backbones.py; the
 * torchvision Bottleneck/BasicBlock forward kept by cirtorch/networks/imageretrievalnet.py:172-173) as one
 * pass.  weight / bias / residual may be NULL; mean and var may both be NULL (no normalisation: with only
 * `bias` given this is the `conv bias + ReLU` of a VGG / AlexNet layer).  mean, var, weight, bias: C floats. */
int mdx_bn_act(float *x, const float *residual, int64_t N, int64_t C, int64_t HW, const float *mean,
               const float *var, const float *weight, const float *bias, float eps, int relu, void *stream);

/* 1x1 convolution with its epilogue, one kernel (stride 1, no padding, no groups, no conv bias):
 *   out[b,co,p] = act( (sum_ci w[co,ci] * x[b,ci,p] - mean[co]) * weight[co] / sqrt(var[co] + eps) + bias[co]  (+ residual[b,co,p]) )
 * = `self.bn1(self.conv1(x))` + relu and `self.bn3(self.conv3(out)); out += identity; relu` of the torchvision Bottleneck
 * that cirtorch keeps as `features` (cirtorch/networks/imageretrievalnet.py:172-173; mdir_amd/backbones.py): the GEMM on
 * the f32 matrix cores (v_mfma_f32_32x32x2_f32), the arithmetic of mdx_bn_act applied to the accumulators on their way out.
 *   x [N,Cin,HW], out / residual [N,Cout,HW] (NCHW, contiguous; out must not alias x);  Cin % 16 == 0, Cout % 64 == 0
 *   wt [Cin,Cout]: the weights TRANSPOSED, made once per convolution by mdx_conv1x1_transpose_weights(w [Cout,Cin])
 *   mean/var, weight, bias, residual: optional as in mdx_bn_act.
 * Accumulation order: ci ascending from +0 (a fixed order; the library convolution's differs by fp32 rounding). */
int mdx_conv1x1_transpose_weights(const float *w, int64_t Cout, int64_t Cin, float *wt, void *stream);
int mdx_conv1x1_bn_act(const float *x, const float *wt, int64_t N, int64_t Cin, int64_t Cout, int64_t HW, const float *mean,
                       const float *var, const float *weight, const float *bias, float eps, const float *residual, int relu,
                       float *out, void *stream);

/* Input conversion: uint8 images [B,H,W,C] (C = 1 or 3, interleaved) -> fp32 [B,C,H,W] with
 *   out = (u / 255 - mean[c]) / std[c]            (fp32, IEEE divisions, this operation order)
 * = the `pil2np | totensor | normalize` transform chain of the scenarios (mdir/components/data/transform/
 * core_transforms.py:33-63) moved behind the host-to-device copy.  mean, std: HOST arrays of C floats. */
int mdx_u8_to_chw(const uint8_t *hwc, int64_t B, int64_t H, int64_t W, int C, const float *mean,
                  const float *std, float *out, void *stream);

/* The paper's CLAHE pre-processing + input conversion: uint8 RGB images [B,H,W,3] -> normalised fp32 [B,3,H,W]
 * = the `pil2np | apply_clahe[:clip[:lab[:grid]]] | totensor | normalize` chain of the CLAHE networks' checkpoints
 * (mdir/components/data/transform/photometric_transforms.py:28-36 -> functional.ImageClahe.apply, functional.py:106-129):
 * RGB -> Lab, CLAHE (clip limit, tiles_x x tiles_y grid) on the uint8 lightness, Lab -> RGB, (x - mean) / std.
 * The reference calls OpenCV for all three steps; this restates OpenCV 4's published algorithms (clahe.cpp, color_lab.cpp)
 * and is pinned against the same restatement in numpy (oracle.apply_clahe_rgb) only -- PARITY UNPINNED against OpenCV, which
 * is absent from the build image.  mean, std: HOST arrays of 3 floats.  workspace: device scratch of
 * mdx_clahe_workspace(B, H, W, tiles_x, tiles_y) bytes (the uint8 lightness plane [B,H,W], the per-tile look-up tables
 * [B, tiles_y, tiles_x, 256], the equalised lightness plane [B,H,W]; each region starts at a multiple of 256 bytes and is
 * readable by the caller afterwards; then the chroma (a, b) of RGB -> Lab as fp32 [B,H,W,2], so that the colour conversion's
 * transcendental functions run once per pixel).  Two launches: one workgroup per (image, tile) -- Lab, LDS histogram, clip,
 * spread, LUT -- then one thread per pixel. */
int64_t mdx_clahe_workspace(int64_t B, int64_t H, int64_t W, int tiles_x, int tiles_y);
int mdx_clahe_u8_to_chw(const uint8_t *rgb, int64_t B, int64_t H, int64_t W, int clip_limit, int tiles_x, int tiles_y,
                        const float *mean, const float *std, void *workspace, int64_t workspace_bytes, float *out, void *stream);

/* The scaled copies of an image batch, every level in ONE launch:  src [B,C,H,W] fp32 -> outs[l] [B,C,floor(H*s_l),floor(W*s_l)]
 * = `F.interpolate(x, scale_factor=s, mode='bilinear', align_corners=False)` of CirMultiscaleAggregation.preprocess
 * (mdir/components/data/wrapper.py:104-107) and extract_ms (cirtorch/networks/imageretrievalnet.py:315), torch >= 1.6
 * semantics (source coordinate (dst + 0.5) / s - 0.5 clamped at 0, fp32).  scales: HOST array of L doubles (levels with
 * s = 1 are the caller's own tensor and are not passed); outs: HOST array of L device pointers. */
int mdx_bilinear_pyramid(const float *src, int64_t B, int64_t C, int H, int W, int L, const double *scales,
                         float *const *outs, void *stream);

/* One pass of the image down-scaling, along the width (axis 1: [B,H,W,C] -> [B,H,out_len,C]) or the height
 * (axis 0: -> [B,out_len,W,C]) of uint8 images:
 *   dst[o] = clip8((2^21 + sum_{t < count[o]} src[first[o] + t] * k[o*ksize + t]) >> 22)
 * = ImagingResampleHorizontal_8bpc / ImagingResampleVertical_8bpc of Pillow (src/libImaging/Resample.c), the
 * arithmetic of `img.thumbnail((imsize, imsize), Image.ANTIALIAS)` in imresize
 * (cirtorch/datasets/datahelpers.py:48-50) as called by ImagesFromList.__getitem__
 * (cirtorch/datasets/genericdataset.py:63-64).  bounds int32 [out_len,2] = (first, count) and the fixed-point
 * taps k int32 [out_len,ksize] (round(w * 2^22), Pillow's precompute_coeffs + normalize_coeffs_8bpc) are DEVICE
 * arrays computed once per (source length, out_len) by the host (mdir_amd/resample.py).  Width pass first, then
 * height, as Pillow orders them; the result equals Pillow's pixel for pixel. */
int mdx_resample_u8(const uint8_t *src, int64_t B, int H, int W, int C, int axis, int out_len,
                    const int32_t *bounds, const int32_t *k, int ksize, uint8_t *dst, void *stream);

/* JPEG decoding, host half + device half.  Together they replace `Image.open(f).convert('RGB')` of pil_loader
 * (cirtorch/datasets/datahelpers.py:24-31) for baseline JPEG files, with the arithmetic of libjpeg(-turbo) as Pillow
 * configures it (Huffman decoding; jidctint.c islow IDCT; jdsample.c fancy upsampling; jdcolor.c YCbCr -> RGB), bit for bit.
 *   mdx_jpeg_probe          HOST.  Geometry of the file; info->supported = 0 for what stays with the host decoder
 *                           (arithmetic, 12-bit, lossless, CMYK / RGB-coded, other sampling factors).  Sequential and
 *                           progressive Huffman files are covered.  The file is UNTRUSTED input: every table index,
 *                           code length, component / table id, spectral band and block index derived from its bytes is
 *                           range-checked, a Huffman table whose codes do not fit their lengths is refused as libjpeg
 *                           refuses it (jdhuff.c), and a frame header that announces more picture than the file can hold
 *                           (size < width * height / 512 bytes: one bit per block) is "unsupported", so nobody sizes a
 *                           buffer from it.  These two functions run under ASan + UBSan over mutated and hand-made
 *                           hostile files in tests/test_fuzz_asan.py (`make -C mdir_amd/csrc -f Makefile.asan`).
 *   mdx_jpeg_coefficients   HOST (no device call; thread-safe, so loader threads run it in parallel).  Entropy decoding
 *                           of all scans (an error for a stream whose data runs out inside a scan: leave it to Pillow):
 *                           coef [nblocks][64] int16, quantised, natural order, component after component, every
 *                           component's blocks row by row over whole MCUs; quant [3][64] uint16, natural order.
 *   mdx_jpeg_pixels         DEVICE.  coef / quant as above (device copies) -> rgb uint8 [height, width, 3].
 *                           planes: device scratch of nblocks * 64 bytes. */
typedef struct {
    int32_t width, height;          /* image size */
    int32_t ncomp;                  /* 1 (grey) or 3 (YCbCr) */
    int32_t hsamp[3], vsamp[3];     /* sampling factors */
    int32_t blocks_w[3], blocks_h[3];   /* 8x8 blocks per row / column of every component (whole MCUs) */
    int32_t supported;
    int64_t block_offset[3];        /* first block of the component in coef */
    int64_t nblocks;                /* blocks of all components */
} mdx_jpeg_info;
int mdx_jpeg_probe(const uint8_t *file, int64_t size, mdx_jpeg_info *info);
int mdx_jpeg_coefficients(const uint8_t *file, int64_t size, int16_t *coef, int64_t coef_blocks, uint16_t *quant);
int mdx_jpeg_pixels(const int16_t *coef, const uint16_t *quant, const mdx_jpeg_info *info, uint8_t *planes,
                    uint8_t *rgb, void *stream);

/* Multi-scale aggregation of S per-scale descriptors of one image:
 *   out[k] = v[k] / ||v||,  v[k] = (sum_s vecs[s][k]^msp / S)^(1/msp)     (no eps)
 * Replaces CirMultiscaleAggregation.aggregate_tensor
 * (mdir/components/data/wrapper.py:109-119) and the tail of extract_ms
 * (cirtorch/networks/imageretrievalnet.py:319-322).
 *   scale_vecs: HOST array of S device pointers, each D floats (1 <= S <= 8). */
int mdx_ms_aggregate(const float *const *scale_vecs, int S, int64_t D, float msp, float *out,
                     void *stream);

/* The same for a batch: scale s is a [B,D] row-major matrix (one row per image), out is [B,D]; ONE launch
 * (a workgroup per image) instead of one mdx_ms_aggregate per image. */
int mdx_ms_aggregate_batch(const float *const *scale_mats, int S, int64_t B, int64_t D, float msp, float *out,
                           void *stream);

/* ------------------------------------------------------------------- index  */

/* A resident shard of database descriptors, re-laid out once into MFMA-fragment
 * order (DESIGN.md "Data layout").  Also used for a whitening matrix P, whose rows
 * then play the part of database rows. */
typedef struct mdx_index mdx_index;

/* Build a shard from n descriptors of dimension d.  `src` is a DEVICE pointer in
 * the given layout.  `row_offset` is the global id of the shard's first row
 * (added to ids by nothing here -- kept for the caller, see mdx_index_info).
 * Allocates n_pad*d_pad*4 bytes of device memory.  An fp32 shard's creation waits for
 * the build on `stream` (the shard's largest magnitude -- the block exponent of
 * MDX_F32_SPLIT2 -- is reduced inside the re-tiling pass and read back here, once);
 * an fp16 shard's only if the build fails. */
int mdx_index_create(mdx_index **out, const float *src, int64_t n, int64_t d, int layout,
                     int64_t row_offset, void *stream);
/* Same with an explicit storage type (mdx_storage).  `src` is fp32 in both cases. */
int mdx_index_create_ex(mdx_index **out, const float *src, int64_t n, int64_t d, int layout,
                        int64_t row_offset, int storage, void *stream);
/* The same in memory the caller provides (and keeps alive until mdx_index_destroy, which then frees nothing): `memory` =
 * mdx_index_bytes(n, d, storage) bytes of device memory at a 256-byte boundary.  hipMalloc + hipFree of an 8 GB shard cost
 * ~190 ms per create / destroy pair -- seventy times the re-tiling itself; a host that builds an index per evaluation hands
 * in memory from its own pool (PyTorch's caching allocator in mdir_amd/ops.py). */
int64_t mdx_index_bytes(int64_t n, int64_t d, int storage);
int mdx_index_create_in(mdx_index **out, const float *src, int64_t n, int64_t d, int layout, int64_t row_offset,
                        int storage, void *memory, int64_t memory_bytes, void *stream);
int mdx_index_destroy(mdx_index *index);
/* n, d, row_offset and device bytes held. */
int mdx_index_info(const mdx_index *index, int64_t *n, int64_t *d, int64_t *row_offset,
                   int64_t *device_bytes);

/* Bytes of scratch mdx_scores needs for nq queries of dimension d. */
int64_t mdx_scores_workspace(int64_t nq, int64_t d);

/* Similarity of nq queries against every row of the shard:
 *   scores[q, i] = sum_k queries(q,k) * db(i,k)      scores row-major [nq, n]
 * i.e. the TRANSPOSE of `np.dot(vecs.T, qvecs)` (mdir/components/optim/score/
 * cirscore.py:69; cirtorch/examples/test.py:240,250), one row per query.
 * Accumulation order is fixed: k ascending, one fused multiply-add per k from +0
 * (oracle/chain.c states it; fp32 MFMA executes exactly that chain).
 *   queries : device, layout `qlayout` ([d,nq] dim-major as the reference's qvecs,
 *             or [nq,d] row-major)
 *   center  : optional device vector [d] subtracted from every query first
 *             (the `v - m` of CirtorchWhiten.postprocess, wrapper.py:194); NULL = none
 *   workspace: device scratch of at least mdx_scores_workspace(nq, d) bytes */
int mdx_scores(const mdx_index *index, const float *queries, int64_t nq, int qlayout,
               const float *center, float *scores, void *workspace, int64_t workspace_bytes,
               void *stream);

/* The same similarity for a database that is multiplied ONCE -- the literal `scores = np.dot(vecs.T, qvecs)` of one
 * evaluation (cirscore.py:69) -- read where it lies: db [n, d] row-major fp32 on the device, no index, no second copy of it
 * (an index of 1 M x 2048 is another 8.2 GB and 3 ms of re-tiling).  Same kernels, same k order: the scores are bit-identical
 * to mdx_scores on an index of the same rows.  d must be a multiple of 4 (rows are fetched in pieces of four values, from any
 * 4-byte-aligned address; MDX_ERR_INVALID otherwise: build an index).  queries / center / scores / workspace (mdx_scores_workspace(nq, d)) as
 * in mdx_scores. */
int mdx_scores_rowmajor(const float *db, int64_t n, int64_t d, const float *queries, int64_t nq, int qlayout,
                        const float *center, float *scores, void *workspace, int64_t workspace_bytes, void *stream);

/* How mdx_scores_ex multiplies an fp32 shard.
 *   MDX_F32_CHAIN   the exact path of mdx_scores (k-ordered fp32 fma chain on the fp32 MFMA; the parity contract).
 *   MDX_F32_SPLIT3  split precision, a LABELLED second mode for the same statement (cirscore.py:69) on the SAME shard
 *                   (nothing is re-quantised or copied): every fp32 operand is written as three bf16 pieces
 *                   x = h + m + l (round to nearest even, residuals exact; the shard's tiles are split in registers by
 *                   the waves that multiply them, the queries once per call) and a product as the six piece products
 *                   hh + hm + mh + hl + lh + mm on v_mfma_f32_16x16x32_bf16 with fp32 accumulation -- dropped terms
 *                   <= 2^-23 of a product, below fp32's own rounding.  6/16 of the fp32 MFMA time: the kernel is bound
 *                   by the shard stream (HBM) instead of the matrix pipe.  NOT bit-equal to the chain: the accumulation
 *                   order differs (measured 8e-7 at most over the 70 M scores of the headline workload; tests/test_gpu_round4.py
 *                   bounds it by 2e-6 -- the summation-order bound bench.py holds the reference's own BLAS path to -- for
 *                   scores up to ~0.5 and by 1e-6 + 4e-6 |s| in general: at a self-match, s = 1, ANY fp32 evaluation of a
 *                   2048-term dot product is ~2e-6 from the exact value, the chain included).  fp32 range (bf16 has
 *                   fp32's exponent); an infinite operand gives NaN.
 *   MDX_F32_SPLIT2  a second labelled mode, for when the matrix' dynamic range is ordinary (L2-normalised descriptors: the path's
 *                   own data).  Block floating point: each matrix is scaled by a power of two that brings its largest magnitude
 *                   into fp16's range (the shard's maximum is read once at mdx_index_create, the queries' by a reduction on the
 *                   device), every operand is two fp16 pieces X = h + m / 2^11 (round toward zero, residual exact and scaled:
 *                   |X - h - m / 2^11| < 2^-20 |X|), a product is hh + (hm + mh) / 2^11 on v_mfma_f32_16x16x32_f16 -- THREE
 *                   products instead of six, the cross terms in an accumulator of their own -- and the result is unscaled
 *                   exactly.  Worst case for unit vectors 2^-20 sum |x_k q_k| <= 1e-6 on top of fp32 accumulation; elements
 *                   more than 2^-27 below their matrix' largest lose relative precision (the bound is relative to
 *                   max|x| max|q|, not to each element -- use MDX_F32_SPLIT3 for wide-range data).  Half the matrix work of
 *                   SPLIT3: the kernel runs at its stream's speed (measured 1.4-1.55 ms against 1.92-2.1 and 2.6 for the exact chain). */
typedef enum mdx_compute { MDX_F32_CHAIN = 0, MDX_F32_SPLIT3 = 1, MDX_F32_SPLIT2 = 2 } mdx_compute;

/* mdx_scores with an explicit compute mode; workspace of at least mdx_scores_workspace_ex(nq, d, compute) bytes
 * (MDX_F32_SPLIT3: 6 bytes per padded query element; MDX_F32_SPLIT2: 4 + 256 bytes).  Both need an MDX_F32 shard. */
int64_t mdx_scores_workspace_ex(int64_t nq, int64_t d, int compute);
int mdx_scores_ex(const mdx_index *index, const float *queries, int64_t nq, int qlayout, const float *center,
                  float *scores, void *workspace, int64_t workspace_bytes, int compute, void *stream);

/* ------------------------------------------------------------------ ranking */

int64_t mdx_rank_workspace(int64_t n, int64_t nq);

/* Full descending ranking of every query row:
 *   ranks[q, r] = id of the r-th best database row for query q    int64 [nq, n]
 * the transpose of `np.argsort(-scores, axis=0)` (cirscore.py:70).  Order: larger
 * score first, equal scores by ascending id, -0 == +0, NaN last (numpy puts NaN
 * last too; its order inside a run of equal scores is unspecified).
 * `id_offset` is added to every id (shard offset / global ids). */
int mdx_rank_full(const float *scores, int64_t n, int64_t nq, int64_t id_offset, int64_t *ranks,
                  void *workspace, int64_t workspace_bytes, void *stream);

/* The same ranking when row q of the scores lies in pieces: block g is a [nq, widths[g]] row-major matrix and
 * row q of the problem is its rows q side by side, g = 0 .. nblocks-1 (<= 32), n = sum of the widths.  These are
 * the peer blocks the multi-GPU exchange delivers ("all-gather of per-shard partial scores", BASELINE.json
 * north_star): the first pass reads them in place, no re-blocked [nq, n] copy is made.  Workspace of
 * mdx_rank_workspace(n, nq).  Ids are column positions in the concatenation + id_offset. */
int mdx_rank_full_segments(const float *const *blocks, const int64_t *widths, int nblocks, int64_t nq,
                           int64_t id_offset, int64_t *ranks, void *workspace, int64_t workspace_bytes,
                           void *stream);

/* First k entries of mdx_rank_full per query, with their scores:
 *   top_ids [nq,k] int64, top_scores [nq,k] fp32 (either may be NULL).
 * For k << n this is a radix SELECT (the k-th key found digit by digit from histograms, candidates
 * compacted in id order, only those sorted) -- about a third of the time of the full ranking at
 * 1M rows; same order, same tie rule.  Same workspace size as mdx_rank_full. */
int mdx_topk(const float *scores, int64_t n, int64_t nq, int64_t k, int64_t id_offset,
             int64_t *top_ids, float *top_scores, void *workspace, int64_t workspace_bytes,
             void *stream);

/* Rank position of labelled database ids without materialising the ranking:
 *   pos[t] = #{i : score[q,i] ranks strictly before id t's score under the order above}
 * for t in [offsets[q], offsets[q+1]).  Gives compute_map (cirtorch/utils/
 * evaluate.py:80-81) exactly what `np.arange(N)[np.in1d(ranks[:,q], ids)]` yields,
 * unsorted.  ids int64 [total], offsets int64 [nq+1] (CSR), pos int64 [total];
 * ids are LOCAL to this score matrix (0 <= id < n).  id_scores fp32 [total]
 * receives scores[q, ids[t]] (it is also the kernel's scratch, so it is required). */
int mdx_rank_of(const float *scores, int64_t n, int64_t nq, const int64_t *ids,
                const int64_t *offsets, int64_t total, float *id_scores, int64_t *pos,
                void *stream);

/* Positions of labelled ids inside a GIVEN ranking (the literal form of evaluate.py:80-81,
 * `np.arange(N)[np.in1d(ranks[:, q], ids)]`, for every query in one pass):
 *   pos[t] = p with ranks[q * ld + p] == ids[t], or -1 when the id does not occur among the n entries of row q,
 * t in [offsets[q], offsets[q+1]).  ranks int64, one row of n ids per query at a stride of ld >= n elements (the
 * [Q, N] matrix mdx_rank_full writes, or its first columns); ids int64 [total] >= 0, unique within a query. */
int mdx_rank_positions(const int64_t *ranks, int64_t n, int64_t nq, int64_t ld, const int64_t *ids,
                       const int64_t *offsets, int64_t total, int64_t *pos, void *stream);

/* Scores of given ids: out[t] = scores[q, ids[t]] for t in the CSR range of q. */
int mdx_gather_scores(const float *scores, int64_t n, int64_t nq, const int64_t *ids,
                      const int64_t *offsets, int64_t total, float *out, void *stream);

/* Count, per labelled id given by its SCORE (possibly living on another shard),
 * how many rows of THIS score matrix rank strictly before it:
 *   cnt[t] += #{i : (score[q,i], i + id_offset) precedes (ref_scores[t], ref_ids[t])}
 * The multi-GPU form of mdx_rank_of: every shard adds its partial count, the sum
 * over shards is the global position.  cnt int64 [total] is ACCUMULATED into. */
int mdx_rank_count(const float *scores, int64_t n, int64_t nq, int64_t id_offset,
                   const float *ref_scores, const int64_t *ref_ids, const int64_t *offsets,
                   int64_t total, int64_t *cnt, void *stream);

/* ------------------------------------------------- whitening learning (float64) */

/* The dense products of whitenlearn / pcawhitenlearn (mdir/external/cirtorch/utils/whiten.py:14-53), which the
 * reference runs in float64 on float64 descriptors; here on the f64 matrix cores (v_mfma_f64_16x16x4_f64, f64
 * accumulation).  Cholesky / eig / inverse of the D x D results stay on the host, as in the reference.
 *
 * Gram matrix of the rows of a dimension-major matrix:
 *   a [d, n] row-major, center [d] or NULL  ->  out [d, d],  out[i][j] = sum_k (a[i][k] - center[i]) * (a[j][k] - center[j])
 * = `np.dot(Xc, Xc.T)` with `Xc = X - m` (whiten.py:21-22), `np.dot(df, df.T)` (whiten.py:42 and :46).  Only the
 * tiles on or above the diagonal are computed and each is stored twice: the result is exactly symmetric.  workspace:
 * mdx_gram_f64_workspace(d, n) bytes of device scratch -- the centred input transposed to [n, d] (both operands of the GEMM
 * are then read in 1-KiB runs), then the partial results of up to 16 K ranges, which are added in range order: a fixed
 * summation order, whatever the schedule.  That is at most 8 * (n_pad * d_pad + 16 * d^2) bytes: 0.84 GB at d = 2048,
 * n = 20 000 (0.33 GB of transposed input + 0.50 GB for 15 ranges) -- size the scratch from the function, not by guess. */
int64_t mdx_gram_f64_workspace(int64_t d, int64_t n);
int mdx_gram_f64(const double *a, int64_t d, int64_t n, const double *center, double *out, void *workspace,
                 int64_t workspace_bytes, void *stream);

/* Projection of centred descriptors:
 *   p [dout, d] row-major, x [d, n] row-major, center [d] or NULL  ->  out [dout, n] = p . (x - center)
 * = `df = np.dot(P, X-m)` (whiten.py:45).  workspace: mdx_project_f64_workspace(dout, d) bytes of device scratch (p
 * transposed, so that both operands are read in 1-KiB runs).  An odd n costs one more short launch (the large-tile kernel
 * moves column pairs). */
int64_t mdx_project_f64_workspace(int64_t dout, int64_t d);
/* x [d, n] float64, in place: every COLUMN divided by (its L2 norm + eps) = `X / (np.linalg.norm(X, ord=2, axis=0, keepdims=True)
 * + 1e-6)` of whitenapply (whiten.py:10-11) when it is handed float64 `P` (then the reference computes in float64). */
int mdx_l2n_cols_f64(double *x, int64_t d, int64_t n, double eps, void *stream);
int mdx_project_f64(const double *p, int64_t dout, int64_t d, const double *x, int64_t n, const double *center,
                    double *out, void *workspace, int64_t workspace_bytes, void *stream);

/* ------------------------------------------------- multi-GPU exchange (RCCL over xGMI) */

/* The reference is single-process; what is sharded here is `scores = np.dot(vecs.T, qvecs)` (cirscore.py:69):
 * database rows are independent, so rank g of G (one process per GPU) keeps rows [lo_g, hi_g) as its own mdx_index
 * and computes its block S_g [nq, w_g] with mdx_scores alone.  The two calls below move the blocks to where
 * `np.argsort(-scores, axis=0)` (cirscore.py:70) needs them; both deliver blocks BACK TO BACK in rank order (block g
 * row-major, widths[g] columns), which is the form mdx_rank_full_segments reads in place.
 *
 * RCCL is bound at run time (dlopen of librccl.so.1 -- the copy the process already holds, e.g. PyTorch's); a box
 * without it gets MDX_ERR_RUNTIME from these calls and nothing else changes.  Communicator set-up follows RCCL:
 * ONE rank calls mdx_comm_unique_id and hands the MDX_COMM_ID_BYTES bytes to the others by any means (the Python
 * host uses its torch.distributed group), then EVERY rank calls mdx_comm_init with its device selected
 * (hipSetDevice); the call is collective.  Exchange calls only enqueue work on `stream`. */
typedef struct mdx_comm mdx_comm;
#define MDX_COMM_ID_BYTES 128
int mdx_comm_unique_id(void *id_host);
int mdx_comm_init(mdx_comm **out, const void *id_host, int nranks, int rank);
int mdx_comm_destroy(mdx_comm *comm);
int mdx_comm_info(const mdx_comm *comm, int *nranks, int *rank);

/* Queries [lo, hi) that rank `rank` of `nranks` ranks (sorts) under the query split: contiguous, sizes differ by <= 1. */
int mdx_query_bounds(int64_t nq, int nranks, int rank, int64_t *lo, int64_t *hi);

/* "All-gather of per-shard partial scores" (BASELINE.json north_star):
 *   local [nq, widths[rank]] on every rank  ->  all = G blocks back to back, block g = S_g [nq, widths[g]], on every rank.
 * widths: HOST array of nranks column counts (the shard sizes; the same on every rank).  Equal widths go out as one
 * ncclAllGather, unequal ones as a grouped send/receive per peer (one per xGMI link). */
int mdx_allgather_scores(mdx_comm *comm, const float *local, int64_t nq, const int64_t *widths, float *all, void *stream);

/* The query-split form (1/G of the bytes per rank; the ranking becomes G-way parallel): rank r keeps queries
 * [qlo_r, qhi_r) = mdx_query_bounds(nq, G, r) and receives their rows of every block:
 *   local [nq, widths[rank]]  ->  mine = G blocks back to back, block g = S_g[qlo_r:qhi_r, :]  ([nq_mine, widths[g]]).
 * mdx_rank_full_segments(blocks, widths, G, nq_mine, 0, ...) then yields the rankings of this rank's queries with
 * GLOBAL row ids. */
int mdx_exchange_scores(mdx_comm *comm, const float *local, int64_t nq, const int64_t *widths, float *mine, void *stream);

/* ------------------------------------------------- direct-store exchange (no collective; hipIpc + xGMI stores) */

/* The third form of the same exchange (round 6): the similarity kernel itself writes query q's run of scores into row
 * q - qlo_owner of the OWNER's receive buffer, at the columns of this shard -- over xGMI, into memory the owner has shared with
 * hipIpcGetMemHandle.  The transfer is spread over the kernel's run time, there is no collective launch, and the owner finds a
 * DENSE [nq_mine, n_total] matrix (mdx_rank_full; no peer blocks).  A step is closed by one flag per peer.  Shards the same
 * statement as above, `np.dot(vecs.T, qvecs)` (cirscore.py:69), for the query-split `np.argsort(-scores, axis=0)` (cirscore.py:70).
 *
 *   every rank, once:  mdx_p2p_create(&p, G, r, nq, n_total, handle)   (allocates 2 receive buffers of ceil(nq/G) x n_total fp32)
 *                      gather the G handles (MDX_P2P_HANDLE_BYTES each, rank order) by any means
 *                      mdx_p2p_connect(p, handles)                        (maps the peers' buffers; not a collective)
 *   every step:        mdx_scores_p2p(index, queries, nq, ..., p, ...)  once per shard (or row chunk) this rank holds
 *                      mdx_p2p_close_step(p, &mine, stream)             mine = this rank's queries x all rows, valid until the
 *                                                                       step after next is opened by any peer
 * Every rank must run the same steps with the same nq.  nq <= 128; fp32 shards; the shard's first global row is the index's
 * row_offset.  HARDWARE STATUS: exercised with several rank processes on ONE GPU (same-device IPC); it has not run over xGMI --
 * tools/preflight_ranks.py checks it on a multi-GPU node before bench.py uses it, and mdx_exchange_scores stays the default.
 * Needs HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC) in the environment of every rank process. */
typedef struct mdx_p2p mdx_p2p;
#define MDX_P2P_HANDLE_BYTES 64
/* handle_host: MDX_P2P_HANDLE_BYTES bytes, written.  MDX_ERR_RUNTIME with *out VALID when the buffer cannot be exported
 * (then only mdx_p2p_connect_ptrs can connect it). */
int mdx_p2p_create(mdx_p2p **out, int nranks, int rank, int64_t nq, int64_t n_total, void *handle_host);
/* handles_host: nranks x MDX_P2P_HANDLE_BYTES bytes in rank order (this rank's own entry is not read). */
int mdx_p2p_connect(mdx_p2p *p2p, const void *handles_host);
/* Ranks that live in ONE process (threads, tests): the peers' mdx_p2p_base pointers instead of handles (host array of nranks). */
int mdx_p2p_connect_ptrs(mdx_p2p *p2p, void *const *bases);
void *mdx_p2p_base(mdx_p2p *p2p);
int64_t mdx_p2p_bytes(const mdx_p2p *p2p);
/* mdx_scores with the routed epilogue: scores[q, i] goes to row q - qlo_owner(q), column row_offset(index) + i of owner(q)'s
 * receive buffer of the open step.  Same kernels, same k order, same bits as mdx_scores.  workspace: mdx_scores_workspace(nq, d). */
int mdx_scores_p2p(const mdx_index *index, const float *queries, int64_t nq, int qlayout, const float *center, mdx_p2p *p2p,
                   void *workspace, int64_t workspace_bytes, void *stream);
/* Enqueues: raise this rank's flag of the step at every peer, wait for every peer's.  *mine = device pointer to this rank's
 * [ceil(nq/G), n_total] buffer of the step (its first nq_mine rows are the queries mdx_query_bounds gives this rank). */
int mdx_p2p_close_step(mdx_p2p *p2p, float **mine, void *stream);
/* Synchronises `stream` and reads the status word: bit r set = a wait for peer r gave up after 20 s (results of that step are
 * undefined).  0 = every step so far was closed by every peer. */
int mdx_p2p_status(mdx_p2p *p2p, uint32_t *late_peers, void *stream);
int mdx_p2p_destroy(mdx_p2p *p2p);

#ifdef __cplusplus
}
#endif
#endif /* MDX_H */
