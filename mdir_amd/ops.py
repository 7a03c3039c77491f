"""Operators of the hot path as calls into libmdx.so on torch CUDA (ROCm) tensors.

torch is plumbing only: device memory, the current HIP stream, and (elsewhere)
torch.distributed.  Every function here hands raw device pointers to the C ABI
(include/mdx.h); nothing computes with torch ops and nothing falls back to the CPU.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import MDX_DIM_MAJOR, MDX_ROW_MAJOR, POOL_KINDS, check

_vp = ctypes.c_void_p


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """Handle of torch's current HIP stream (launches are enqueued there; nothing synchronises)."""
    if _raw_stream is not None:           # a few hundred ns instead of building a torch.cuda.Stream object
        return _vp(_raw_stream(torch.cuda.current_device()))
    return _vp(torch.cuda.current_stream().cuda_stream)


class _Here:
    """No-op context: the tensor's device is already the current one (the common case; a real
    ``torch.cuda.device`` guard costs a few microseconds per launch, and the trunk epilogue launches ~100 times per image)."""
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_HERE = _Here()


def _on(t):
    """Context in which ``t``'s device is current, so that ``_stream()`` and the launch use the device the pointers
    live on (a caller may pass ``device=cuda:1`` without ``set_device``; the reference's torch ops follow the tensor)."""
    idx = t.device.index
    if idx is None or idx == torch.cuda.current_device():
        return _HERE
    return torch.cuda.device(idx)


def _dev(t, dtype, what):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("%s must be a CUDA/ROCm tensor: the MI355X path has no CPU fallback" % what)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (what, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % what)
    return _vp(t.data_ptr())


def _layout(t, layout, what):
    """(n, d) of a 2-D descriptor matrix under the given layout name."""
    if t.dim() != 2:
        raise ValueError("%s must be 2-D" % what)
    if layout in ("DN", "dim_major", MDX_DIM_MAJOR):
        return t.shape[1], t.shape[0], MDX_DIM_MAJOR
    if layout in ("ND", "row_major", MDX_ROW_MAJOR):
        return t.shape[0], t.shape[1], MDX_ROW_MAJOR
    raise ValueError("unknown layout %r" % (layout,))


def _workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


# ------------------------------------------------------------------ extraction

def pool_l2n(feat, kind="gem", p=3.0, pool_eps=1e-6, l2n_eps=1e-6):
    """[B,C,H,W] feature maps -> [B,C] pooled (+ L2-normalised unless l2n_eps is None).

    ``self.norm(self.pool(o))`` of cirtorch/networks/imageretrievalnet.py:108."""
    if feat.dim() != 4:
        raise ValueError("feature map must be [B,C,H,W]")
    fp = _dev(feat, torch.float32, "feature map")
    B, C, H, W = feat.shape
    out = torch.empty((B, C), dtype=torch.float32, device=feat.device)
    with _on(feat):
        check(_lib.lib().mdx_pool_l2n(fp, B, C, H, W, POOL_KINDS[kind], float(p), float(pool_eps),
                                      -1.0 if l2n_eps is None else float(l2n_eps), _vp(out.data_ptr()),
                                      _stream()), "mdx_pool_l2n")
    return out


def rmac(feat, regions, eps=1e-6):
    """R-MAC pooling ``[B,C,H,W] -> [B,C]`` over the given regions (``[(row0, col0, height, width), ...]``, the whole map
    first): ``sum_r l2n(max over region r)`` -- ``LF.rmac`` (functional.py:26-72) through ``mdx_rmac``."""
    fp = _dev(feat, torch.float32, "feature map")
    if feat.dim() != 4:
        raise ValueError("feature map must be [B,C,H,W]")
    B, C, H, W = feat.shape
    n = len(regions)
    if not 1 <= n <= 64:
        raise ValueError("1..64 regions supported, got %d" % n)
    flat = (ctypes.c_int32 * (4 * n))(*[int(v) for reg in regions for v in reg])
    out = torch.empty((B, C), dtype=torch.float32, device=feat.device)
    ws = _workspace(_lib.lib().mdx_rmac_workspace(B, C, n), feat.device)
    with _on(feat):
        check(_lib.lib().mdx_rmac(fp, B, C, H, W, flat, n, float(eps), _vp(ws.data_ptr()), ws.numel(), _vp(out.data_ptr()), _stream()),
              "mdx_rmac")
    return out


def roipool(feat, regions, kind="gem", p=3.0, pool_eps=1e-6):
    """Regional pooling ``[B,C,H,W] -> [B,R,C]``: the pooling ``kind`` of every region ``(row0, col0, height, width)``, no
    normalisation -- ``LF.roipool`` (functional.py:75-121) through ``mdx_roipool``."""
    fp = _dev(feat, torch.float32, "feature map")
    if feat.dim() != 4:
        raise ValueError("feature map must be [B,C,H,W]")
    B, C, H, W = feat.shape
    n = len(regions)
    if not 1 <= n <= 64:
        raise ValueError("1..64 regions supported, got %d" % n)
    flat = (ctypes.c_int32 * (4 * n))(*[int(v) for reg in regions for v in reg])
    out = torch.empty((B, n, C), dtype=torch.float32, device=feat.device)
    with _on(feat):
        check(_lib.lib().mdx_roipool(fp, B, C, H, W, flat, n, POOL_KINDS[kind], float(p), float(pool_eps), _vp(out.data_ptr()), _stream()),
              "mdx_roipool")
    return out


def region_sum(vecs, l2n_eps=None):
    """``[B,R,C] -> [B,C]``: sum over the regions in order; ``l2n_eps`` not None: each region vector L2-normalised first."""
    vp = _dev(vecs, torch.float32, "region vectors")
    if vecs.dim() != 3:
        raise ValueError("region vectors must be [B,R,C]")
    B, R, C = vecs.shape
    out = torch.empty((B, C), dtype=torch.float32, device=vecs.device)
    with _on(vecs):
        check(_lib.lib().mdx_region_sum(vp, B, R, C, -1.0 if l2n_eps is None else float(l2n_eps), _vp(out.data_ptr()), _stream()),
              "mdx_region_sum")
    return out


def l2n_rows_(x, bias=None, eps=1e-6):
    """In place ``(x + bias) / (||x + bias|| + eps)`` per row of a [R,D] matrix."""
    xp = _dev(x, torch.float32, "x")
    if x.dim() != 2:
        raise ValueError("x must be [R,D]")
    bp = _dev(bias, torch.float32, "bias") if bias is not None else None
    if bias is not None and bias.numel() != x.shape[1]:
        raise ValueError("bias length %d != D %d" % (bias.numel(), x.shape[1]))
    with _on(x):
        check(_lib.lib().mdx_l2n_rows(xp, x.shape[0], x.shape[1], bp, float(eps), _stream()), "mdx_l2n_rows")
    return x


def ms_aggregate(vecs, msp=1.0):
    """Per-scale descriptors (list of [D] or [D,1] tensors) -> aggregated [D].

    mdir/components/data/wrapper.py:109-119."""
    if not 1 <= len(vecs) <= 8:
        raise ValueError("1..8 scales supported, got %d" % len(vecs))
    flat = [v.reshape(-1) for v in vecs]
    D = flat[0].numel()
    ptrs = (ctypes.c_void_p * len(flat))()
    for i, v in enumerate(flat):
        if v.numel() != D:
            raise ValueError("scale %d has %d elements, expected %d" % (i, v.numel(), D))
        ptrs[i] = _dev(v, torch.float32, "scale descriptor").value
    out = torch.empty(D, dtype=torch.float32, device=flat[0].device)
    with _on(flat[0]):
        check(_lib.lib().mdx_ms_aggregate(ptrs, len(flat), D, float(msp), _vp(out.data_ptr()), _stream()),
              "mdx_ms_aggregate")
    return out


def ms_aggregate_batch(mats, msp=1.0):
    """Per-scale descriptor MATRICES (list of ``[B,D]`` tensors, one row per image) -> aggregated ``[B,D]`` in one
    launch (``mdx_ms_aggregate_batch``)."""
    if not 1 <= len(mats) <= 8:
        raise ValueError("1..8 scales supported, got %d" % len(mats))
    B, D = mats[0].shape
    ptrs = (ctypes.c_void_p * len(mats))()
    for i, v in enumerate(mats):
        if tuple(v.shape) != (B, D):
            raise ValueError("scale %d is %s, expected %s" % (i, tuple(v.shape), (B, D)))
        ptrs[i] = _dev(v, torch.float32, "scale descriptors").value
    out = torch.empty((B, D), dtype=torch.float32, device=mats[0].device)
    with _on(mats[0]):
        check(_lib.lib().mdx_ms_aggregate_batch(ptrs, len(mats), B, D, float(msp), _vp(out.data_ptr()), _stream()),
              "mdx_ms_aggregate_batch")
    return out


def pool_multi(feats, kind="gem", p=3.0, pool_eps=1e-6):
    """The feature maps of one pyramid (list of ``[B,C,H_s,W_s]``, same B and C) -> pooled ``[S,B,C]`` in ONE launch
    (``mdx_pool_multi``); no normalisation."""
    if not 1 <= len(feats) <= 8:
        raise ValueError("1..8 scales supported, got %d" % len(feats))
    if feats[0].dim() != 4:
        raise ValueError("feature maps must be [B,C,H,W]")
    B, C = feats[0].shape[:2]
    S = len(feats)
    ptrs = (ctypes.c_void_p * S)()
    hs, ws = (ctypes.c_int * S)(), (ctypes.c_int * S)()
    for i, f in enumerate(feats):
        if f.dim() != 4 or tuple(f.shape[:2]) != (B, C):
            raise ValueError("map %d is %s, expected [%d,%d,H,W]" % (i, tuple(f.shape), B, C))
        ptrs[i] = _dev(f, torch.float32, "feature map").value
        hs[i], ws[i] = f.shape[2], f.shape[3]
    out = torch.empty((S, B, C), dtype=torch.float32, device=feats[0].device)
    with _on(feats[0]):
        check(_lib.lib().mdx_pool_multi(ptrs, S, B, C, hs, ws, POOL_KINDS[kind], float(p), float(pool_eps),
                                        _vp(out.data_ptr()), _stream()), "mdx_pool_multi")
    return out


def l2n_aggregate(pooled, l2n_eps=1e-6, msp=1.0):
    """Pooled ``[S,B,D]`` -> aggregated descriptors ``[B,D]`` in ONE launch (``mdx_l2n_aggregate``): L2N of every
    scale's row, power mean over the scales, renormalisation."""
    if pooled.dim() != 3 or not 1 <= pooled.shape[0] <= 8:
        raise ValueError("pooled must be [S,B,D] with 1..8 scales")
    S, B, D = pooled.shape
    out = torch.empty((B, D), dtype=torch.float32, device=pooled.device)
    with _on(pooled):
        check(_lib.lib().mdx_l2n_aggregate(_dev(pooled, torch.float32, "pooled"), S, B, D, float(l2n_eps), float(msp),
                                           _vp(out.data_ptr()), _stream()), "mdx_l2n_aggregate")
    return out


def u8_to_chw(images, mean, std):
    """uint8 ``[B,H,W,C]`` device images -> normalised fp32 ``[B,C,H,W]`` (``mdx_u8_to_chw``):
    ``(u / 255 - mean) / std``, the scenarios' ``pil2np | totensor | normalize``."""
    if not (images.is_cuda and images.dtype == torch.uint8 and images.dim() == 4 and images.is_contiguous()):
        raise ValueError("u8_to_chw expects a contiguous uint8 [B,H,W,C] CUDA/ROCm tensor (no CPU fallback)")
    b, h, w, c = images.shape
    if len(mean) != c or len(std) != c:
        raise ValueError("mean / std need %d values" % c)
    out = torch.empty((b, c, h, w), dtype=torch.float32, device=images.device)
    if images.numel() == 0:
        return out
    arr = ctypes.c_float * c
    with _on(images):
        check(_lib.lib().mdx_u8_to_chw(images.data_ptr(), b, h, w, c, arr(*[float(v) for v in mean]),
                                       arr(*[float(v) for v in std]), out.data_ptr(), _stream()), "mdx_u8_to_chw")
    return out


def clahe_u8_to_chw(images, clip_limit, grid, mean, std, return_intermediates=False):
    """uint8 RGB ``[B,H,W,3]`` device images -> CLAHE on the Lab lightness -> normalised fp32 ``[B,3,H,W]``
    (``mdx_clahe_u8_to_chw``): the scenarios' ``pil2np | apply_clahe | totensor | normalize``.  ``grid``: int or
    ``(tiles_x, tiles_y)``.  ``return_intermediates``: also the uint8 lightness ``[B,H,W]``, the LUTs
    ``[B,tiles_y,tiles_x,256]`` and the equalised lightness ``[B,H,W]`` the kernels left in the workspace (tests)."""
    if not (images.is_cuda and images.dtype == torch.uint8 and images.dim() == 4 and images.is_contiguous() and images.shape[3] == 3):
        raise ValueError("clahe_u8_to_chw expects a contiguous uint8 [B,H,W,3] CUDA/ROCm tensor (no CPU fallback)")
    b, h, w, _ = images.shape
    tx, ty = (int(grid), int(grid)) if not isinstance(grid, (tuple, list)) else (int(grid[0]), int(grid[1]))
    if len(mean) != 3 or len(std) != 3:
        raise ValueError("mean / std need 3 values")
    out = torch.empty((b, 3, h, w), dtype=torch.float32, device=images.device)
    need = _lib.lib().mdx_clahe_workspace(b, h, w, tx, ty)
    ws = _workspace(need, images.device)
    if images.numel():
        arr = ctypes.c_float * 3
        with _on(images):
            check(_lib.lib().mdx_clahe_u8_to_chw(images.data_ptr(), b, h, w, int(clip_limit), tx, ty, arr(*[float(v) for v in mean]),
                                                 arr(*[float(v) for v in std]), ws.data_ptr(), ws.numel(), out.data_ptr(), _stream()),
                  "mdx_clahe_u8_to_chw")
    if return_intermediates:
        plane, tables = -(-(b * h * w) // 256) * 256, -(-(b * ty * tx * 256) // 256) * 256
        return (out, ws[:b * h * w].view(b, h, w), ws[plane:plane + b * ty * tx * 256].view(b, ty, tx, 256),
                ws[plane + tables:plane + tables + b * h * w].view(b, h, w))
    return out


def bilinear_pyramid(x, scales):
    """``[F.interpolate(x, scale_factor=s, mode="bilinear", align_corners=False) for s in scales]`` for an fp32
    ``[B,C,H,W]`` device tensor, all levels in ONE launch (``mdx_bilinear_pyramid``); a scale of exactly 1 returns ``x``."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()):
        raise ValueError("bilinear_pyramid expects a contiguous fp32 [B,C,H,W] CUDA/ROCm tensor (no CPU fallback)")
    import math
    b, c, h, w = x.shape
    todo = [(i, float(s)) for i, s in enumerate(scales) if float(s) != 1.0]
    out = [x] * len(scales)
    if not todo:
        return out
    if len(todo) > 8:
        raise ValueError("at most 8 scaled levels")
    sc = (ctypes.c_double * len(todo))(*[s for _, s in todo])
    ptrs = (ctypes.c_void_p * len(todo))()
    for k, (i, s) in enumerate(todo):
        out[i] = torch.empty((b, c, int(math.floor(h * s)), int(math.floor(w * s))), dtype=torch.float32, device=x.device)
        ptrs[k] = out[i].data_ptr()
    with _on(x):
        check(_lib.lib().mdx_bilinear_pyramid(x.data_ptr(), b, c, h, w, len(todo), sc, ptrs, _stream()), "mdx_bilinear_pyramid")
    return out


def resample_u8(images, axis, bounds, taps):
    """One pass of Pillow's 8-bit resampling on uint8 ``[B,H,W,C]`` device images: along the width (``axis=1``) or the
    height (``axis=0``), with device taps ``bounds`` int32 ``[out,2]`` and ``taps`` int32 ``[out,ksize]``
    (``mdx_resample_u8``; the taps come from ``mdir_amd.resample``)."""
    if not (images.is_cuda and images.dtype == torch.uint8 and images.dim() == 4 and images.is_contiguous()):
        raise ValueError("resample_u8 expects a contiguous uint8 [B,H,W,C] CUDA/ROCm tensor (no CPU fallback)")
    if axis not in (0, 1):
        raise ValueError("axis must be 0 (height) or 1 (width)")
    b, h, w, c = images.shape
    out_len, ksize = taps.shape
    if tuple(bounds.shape) != (out_len, 2) or bounds.dtype != torch.int32 or taps.dtype != torch.int32 \
            or bounds.device != images.device or taps.device != images.device:
        raise ValueError("bounds / taps must be int32 [out,2] / [out,ksize] on the images' device")
    out = torch.empty((b, h, out_len, c) if axis == 1 else (b, out_len, w, c), dtype=torch.uint8, device=images.device)
    with _on(images):
        check(_lib.lib().mdx_resample_u8(images.data_ptr(), b, h, w, c, axis, out_len, _dev(bounds, torch.int32, "bounds"),
                                         _dev(taps, torch.int32, "taps"), ksize, out.data_ptr(), _stream()), "mdx_resample_u8")
    return out


def bn_act_(x, running_mean, running_var, weight=None, bias=None, eps=1e-5, residual=None, relu=True):
    """In place on a convolution output ``x [N,C,H,W]``: inference batch-norm, ``+ residual``, ReLU
    in one pass (``mdx_bn_act``); returns ``x``.  Called ~100 times per image by a launch-bound trunk,
    so the checks are kept to what protects the raw pointers."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()):
        raise ValueError("bn_act_ expects a contiguous fp32 [N,C,H,W] CUDA/ROCm tensor (no CPU fallback)")
    n, c, h, w = x.shape
    ptrs = []
    for name, t in (("running_mean", running_mean), ("running_var", running_var), ("weight", weight), ("bias", bias)):
        if t is None:
            ptrs.append(None)
            continue
        if t.numel() != c or t.dtype != torch.float32 or t.device != x.device or not t.is_contiguous():
            raise ValueError("%s must be %d contiguous fp32 values on %s" % (name, c, x.device))
        ptrs.append(t.data_ptr())
    if (ptrs[0] is None) != (ptrs[1] is None):
        raise ValueError("running_mean and running_var must both be given or both be None")
    rp = None
    if residual is not None:
        if residual.shape != x.shape or residual.dtype != torch.float32 or residual.device != x.device \
                or not residual.is_contiguous():
            raise ValueError("residual must be contiguous fp32 and shaped like x")
        rp = residual.data_ptr()
    if x.numel() == 0:
        return x
    with _on(x):
        check(_lib.lib().mdx_bn_act(x.data_ptr(), rp, n, c, h * w, ptrs[0], ptrs[1], ptrs[2], ptrs[3], float(eps),
                                    1 if relu else 0, _stream()), "mdx_bn_act")
    return x


def conv1x1_transpose_weights(weight):
    """``[Cout, Cin(,1,1)]`` convolution weights -> the ``[Cin, Cout]`` copy ``conv1x1_bn_act`` reads (made once)."""
    w = weight.reshape(weight.shape[0], -1)
    wp = _dev(w, torch.float32, "weight")
    wt = torch.empty((w.shape[1], w.shape[0]), dtype=torch.float32, device=w.device)
    with _on(w):
        check(_lib.lib().mdx_conv1x1_transpose_weights(wp, w.shape[0], w.shape[1], wt.data_ptr(), _stream()),
              "mdx_conv1x1_transpose_weights")
    return wt


def conv1x1_supported(cin, cout):
    return cin % 16 == 0 and cout % 64 == 0


def conv1x1_bn_act(x, weight_t, running_mean, running_var, weight=None, bias=None, eps=1e-5, residual=None, relu=True):
    """1x1 convolution + inference batch-norm (+ residual) (+ ReLU) in one kernel (``mdx_conv1x1_bn_act``).
    ``x [N,Cin,H,W]`` contiguous fp32; ``weight_t [Cin,Cout]`` from ``conv1x1_transpose_weights``; returns ``[N,Cout,H,W]``."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()):
        raise ValueError("conv1x1_bn_act expects a contiguous fp32 [N,C,H,W] CUDA/ROCm tensor (no CPU fallback)")
    n, cin, h, w = x.shape
    if weight_t.dim() != 2 or weight_t.shape[0] != cin or weight_t.dtype != torch.float32 or weight_t.device != x.device \
            or not weight_t.is_contiguous():
        raise ValueError("weight_t must be contiguous fp32 [Cin=%d, Cout] on %s" % (cin, x.device))
    cout = weight_t.shape[1]
    ptrs = []
    for name, t in (("running_mean", running_mean), ("running_var", running_var), ("weight", weight), ("bias", bias)):
        if t is None:
            ptrs.append(None)
            continue
        if t.numel() != cout or t.dtype != torch.float32 or t.device != x.device or not t.is_contiguous():
            raise ValueError("%s must be %d contiguous fp32 values on %s" % (name, cout, x.device))
        ptrs.append(t.data_ptr())
    rp = None
    if residual is not None:
        if tuple(residual.shape) != (n, cout, h, w) or residual.dtype != torch.float32 or residual.device != x.device \
                or not residual.is_contiguous():
            raise ValueError("residual must be contiguous fp32 [%d,%d,%d,%d]" % (n, cout, h, w))
        rp = residual.data_ptr()
    out = torch.empty((n, cout, h, w), dtype=torch.float32, device=x.device)
    if out.numel() == 0:
        return out
    with _on(x):
        check(_lib.lib().mdx_conv1x1_bn_act(x.data_ptr(), weight_t.data_ptr(), n, cin, cout, h * w, ptrs[0], ptrs[1], ptrs[2],
                                            ptrs[3], float(eps), rp, 1 if relu else 0, out.data_ptr(), _stream()),
              "mdx_conv1x1_bn_act")
    return out


# ----------------------------------------------------------------------- index

class DescriptorIndex:
    """A resident, re-tiled shard of descriptors (``mdx_index``).

    ``vecs`` is a device tensor, ``[D,N]`` (layout "DN", the reference's
    ``extract_vectors`` output) or ``[N,D]`` (layout "ND")."""

    def __init__(self, vecs, layout="DN", row_offset=0, storage="f32"):
        """``storage="f16"`` keeps the shard (and each call's queries) in fp16 and uses the fp16
        MFMA with fp32 accumulation: half the HBM bytes, ~1e-3 relative score error."""
        self._h = None
        self.storage = storage
        vp = _dev(vecs, torch.float32, "vecs")
        n, d, lay = _layout(vecs, layout, "vecs")
        self._h = ctypes.c_void_p()
        self.device = vecs.device
        self.n, self.d, self.row_offset = n, d, int(row_offset)
        with torch.cuda.device(self.device):
            # the tiles live in PyTorch's caching allocator: hipMalloc + hipFree of an 8 GB shard cost ~190 ms per index, a
            # block of the pool nothing after the first use
            need = _lib.lib().mdx_index_bytes(n, d, _lib.STORAGE[storage])
            self._tiles = torch.empty(need, dtype=torch.uint8, device=self.device)
            check(_lib.lib().mdx_index_create_in(ctypes.byref(self._h), vp, n, d, lay, self.row_offset, _lib.STORAGE[storage],
                                                 _vp(self._tiles.data_ptr()), need, _stream()), "mdx_index_create_in")
            # the source tensor may be freed by the caller right after: finish the re-tiling first
            torch.cuda.current_stream().synchronize()

    @property
    def device_bytes(self):
        b = ctypes.c_int64()
        check(_lib.lib().mdx_index_info(self._h, None, None, None, ctypes.byref(b)), "mdx_index_info")
        return b.value

    def scores(self, queries, qlayout="DN", center=None, out=None, compute="chain"):
        """fp32 ``[nq, n]``: row q = similarities of query q to every shard row.

        The transpose of ``np.dot(vecs.T, qvecs)`` (cirscore.py:69).  ``compute="chain"`` (default): the exact k-ordered
        fp32 fma chain; ``"split3"``: the labelled split-precision mode on the same fp32 shard (three bf16 pieces per
        operand, six products on the bf16 MFMA, fp32 accumulation: HBM-bound instead of fp32-MFMA-bound; scores within
        the summation-order bound 2e-6 of the chain, ``include/mdx.h`` ``MDX_F32_SPLIT3``); ``"split2"``: the second labelled
        mode, block floating point with two fp16 pieces and three products (``MDX_F32_SPLIT2``): for data of ordinary dynamic
        range (L2-normalised descriptors), 0.63 of the exact kernel's time."""
        if self._h is None:
            raise RuntimeError("index is closed")
        nq, d, lay = _layout(queries, qlayout, "queries")
        if d != self.d:
            raise ValueError("query dimension %d != index dimension %d" % (d, self.d))
        qp = _dev(queries, torch.float32, "queries")
        cp = _dev(center, torch.float32, "center") if center is not None else None
        if center is not None and center.numel() != d:
            raise ValueError("center has %d elements, expected %d" % (center.numel(), d))
        if out is None:
            out = torch.empty((nq, self.n), dtype=torch.float32, device=self.device)
        elif tuple(out.shape) != (nq, self.n):
            raise ValueError("out must be [%d,%d]" % (nq, self.n))
        if compute not in _lib.COMPUTE:
            raise ValueError("compute %r (one of %s)" % (compute, sorted(_lib.COMPUTE)))
        mode = _lib.COMPUTE[compute]
        if mode != _lib.MDX_F32_CHAIN and self.storage != "f32":
            raise ValueError("compute=%r multiplies an fp32 shard; this one is stored as %s" % (compute, self.storage))
        need = _lib.lib().mdx_scores_workspace_ex(nq, d, mode)
        ws = _workspace(need, self.device)
        with torch.cuda.device(self.device):
            check(_lib.lib().mdx_scores_ex(self._h, qp, nq, lay, cp, _dev(out, torch.float32, "out"),
                                           _vp(ws.data_ptr()), need, mode, _stream()), "mdx_scores_ex")
        return out

    def scores_p2p(self, queries, p2p, qlayout="DN", center=None):
        """The similarity of ``scores`` with the ROUTED epilogue (``mdx_scores_p2p``): query q's scores against this shard's
        rows go straight to row ``q - qlo_owner`` of the owner rank's receive buffer, at columns ``row_offset ..`` -- the
        direct-store exchange of a row-sharded database (:class:`P2P`).  Nothing is returned: ``p2p.close_step()`` hands
        out this rank's ``[nq_mine, n_total]`` matrix once every rank has written its part."""
        if self._h is None:
            raise RuntimeError("index is closed")
        nq, d, lay = _layout(queries, qlayout, "queries")
        if d != self.d:
            raise ValueError("query dimension %d != index dimension %d" % (d, self.d))
        if self.storage != "f32":
            raise ValueError("the direct-store exchange multiplies an fp32 shard; this one is stored as %s" % self.storage)
        qp = _dev(queries, torch.float32, "queries")
        cp = _dev(center, torch.float32, "center") if center is not None else None
        need = _lib.lib().mdx_scores_workspace(nq, d)
        ws = _workspace(need, self.device)
        with torch.cuda.device(self.device):
            check(_lib.lib().mdx_scores_p2p(self._h, qp, nq, lay, cp, p2p._h, _vp(ws.data_ptr()), need, _stream()), "mdx_scores_p2p")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            h, self._h = self._h, None
            check(_lib.lib().mdx_index_destroy(h), "mdx_index_destroy")
            # back to the pool -- after every kernel that may still read the tiles, on whatever stream it was launched
            # (what the library's own hipFree guaranteed by being device-synchronous; here without its ~100 ms)
            if self._tiles is not None and self._tiles.is_cuda:
                torch.cuda.synchronize(self._tiles.device)
            self._tiles = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# --------------------------------------------------------------------- ranking

def rank_workspace_bytes(n, nq):
    return _lib.lib().mdx_rank_workspace(n, nq)


def rank_full(scores, id_offset=0, out=None, workspace=None):
    """int64 ``[nq, n]``: row q = ids best to worst (transpose of cirscore.py:70)."""
    sp = _dev(scores, torch.float32, "scores")
    nq, n = scores.shape
    need = rank_workspace_bytes(n, nq)
    ws = workspace if workspace is not None else _workspace(need, scores.device)
    if out is None:
        out = torch.empty((nq, n), dtype=torch.int64, device=scores.device)
    with torch.cuda.device(scores.device):
        check(_lib.lib().mdx_rank_full(sp, n, nq, int(id_offset), _dev(out, torch.int64, "ranks"),
                                       _vp(ws.data_ptr()), ws.numel(), _stream()), "mdx_rank_full")
    return out


MAX_RANK_SEGMENTS = 32


def rank_full_segments(blocks, id_offset=0, out=None, workspace=None):
    """``rank_full`` of scores that lie in column blocks: ``blocks[g]`` is ``[nq, w_g]`` and row q of the problem is
    the blocks' rows q side by side (the peer blocks of the multi-GPU exchange).  No concatenated copy is made:
    the first pass of the sort reads the blocks in place (``mdx_rank_full_segments``)."""
    if not 1 <= len(blocks) <= MAX_RANK_SEGMENTS:
        raise ValueError("1..%d blocks supported, got %d" % (MAX_RANK_SEGMENTS, len(blocks)))
    nq = blocks[0].shape[0]
    ptrs = (ctypes.c_void_p * len(blocks))()
    widths = (ctypes.c_int64 * len(blocks))()
    for g, b in enumerate(blocks):
        if b.dim() != 2 or b.shape[0] != nq:
            raise ValueError("block %d is %s, expected [%d, w]" % (g, tuple(b.shape), nq))
        ptrs[g] = _dev(b, torch.float32, "score block").value
        widths[g] = b.shape[1]
    n = sum(int(b.shape[1]) for b in blocks)
    dev = blocks[0].device
    ws = workspace if workspace is not None else _workspace(rank_workspace_bytes(n, nq), dev)
    if out is None:
        out = torch.empty((nq, n), dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        check(_lib.lib().mdx_rank_full_segments(ptrs, widths, len(blocks), nq, int(id_offset), _dev(out, torch.int64, "ranks"),
                                                _vp(ws.data_ptr()), ws.numel(), _stream()), "mdx_rank_full_segments")
    return out


def scores_rowmajor(db, queries, qlayout="DN", center=None, out=None):
    """fp32 ``[nq, n]`` similarities of ``queries`` against the rows of ``db`` -- a row-major ``[n, d]`` fp32 CUDA matrix that
    is multiplied ONCE and read where it lies (``mdx_scores_rowmajor``): no index build, no second copy of the database.
    Bit-identical to ``DescriptorIndex(db, "ND").scores(queries, qlayout, center)`` (same kernels, same k order).
    ``d`` must be a multiple of 4 (rows are fetched in 16-byte pieces); other shapes: build a :class:`DescriptorIndex`."""
    dp = _dev(db, torch.float32, "db")
    if db.dim() != 2:
        raise ValueError("db: a 2-d [n, d] matrix")
    n, d = db.shape
    nq, dq, lay = _layout(queries, qlayout, "queries")
    if dq != d:
        raise ValueError("query dimension %d != database dimension %d" % (dq, d))
    qp = _dev(queries, torch.float32, "queries")
    cp = _dev(center, torch.float32, "center") if center is not None else None
    if center is not None and center.numel() != d:
        raise ValueError("center has %d elements, expected %d" % (center.numel(), d))
    if out is None:
        out = torch.empty((nq, n), dtype=torch.float32, device=db.device)
    elif tuple(out.shape) != (nq, n):
        raise ValueError("out must be [%d,%d]" % (nq, n))
    need = _lib.lib().mdx_scores_workspace(nq, d)
    ws = _workspace(need, db.device)
    with torch.cuda.device(db.device):
        check(_lib.lib().mdx_scores_rowmajor(dp, n, d, qp, nq, lay, cp, _dev(out, torch.float32, "out"), _vp(ws.data_ptr()), need,
                                             _stream()), "mdx_scores_rowmajor")
    return out


def topk(scores, k, id_offset=0, workspace=None):
    """(ids int64 [nq,k], scores fp32 [nq,k]) of the k best rows per query."""
    sp = _dev(scores, torch.float32, "scores")
    nq, n = scores.shape
    need = rank_workspace_bytes(n, nq)
    ws = workspace if workspace is not None else _workspace(need, scores.device)
    ids = torch.empty((nq, k), dtype=torch.int64, device=scores.device)
    vals = torch.empty((nq, k), dtype=torch.float32, device=scores.device)
    with torch.cuda.device(scores.device):
        check(_lib.lib().mdx_topk(sp, n, nq, int(k), int(id_offset), _vp(ids.data_ptr()), _vp(vals.data_ptr()),
                                  _vp(ws.data_ptr()), ws.numel(), _stream()), "mdx_topk")
    return ids, vals


def _csr(id_lists, device):
    arrays = [np.asarray(ids, dtype=np.int64).reshape(-1) for ids in id_lists]
    offsets = np.zeros(len(arrays) + 1, dtype=np.int64)
    if arrays:
        np.cumsum([len(a) for a in arrays], out=offsets[1:])
    flat = np.concatenate(arrays) if arrays else np.empty(0, dtype=np.int64)
    both = torch.from_numpy(np.concatenate([flat, offsets])).to(device)      # one copy
    return both[:len(flat)], both[len(flat):], [int(o) for o in offsets]


def rank_of(scores, id_lists):
    """Zero-based rank positions of the given ids, per query, without sorting.

    ``id_lists[q]`` = database ids of query q.  Returns (positions, id_scores) as
    flat device tensors plus the CSR offsets (python list)."""
    sp = _dev(scores, torch.float32, "scores")
    nq, n = scores.shape
    if len(id_lists) != nq:
        raise ValueError("need one id list per query")
    ids_t, off_t, offsets = _csr(id_lists, scores.device)
    total = ids_t.numel()
    pos = torch.zeros(total, dtype=torch.int64, device=scores.device)
    sc = torch.empty(total, dtype=torch.float32, device=scores.device)
    if total:
        flat = np.concatenate([np.asarray(ids, dtype=np.int64).reshape(-1) for ids in id_lists])
        if int(flat.min()) < 0 or int(flat.max()) >= n:
            raise IndexError("labelled id out of range [0,%d)" % n)
        with torch.cuda.device(scores.device):
            check(_lib.lib().mdx_rank_of(sp, n, nq, _vp(ids_t.data_ptr()), _vp(off_t.data_ptr()), total,
                                         _vp(sc.data_ptr()), _vp(pos.data_ptr()), _stream()), "mdx_rank_of")
    return pos, sc, offsets


def rank_positions(ranks, id_lists):
    """Positions of the given ids inside a ranking: ``ranks`` int64 ``[Q, N]`` on the device (rows contiguous; a row stride
    larger than N, e.g. the first columns of a wider matrix, is fine), ``id_lists[q]`` = non-negative ids, unique within a
    query.  Returns ``(pos, offsets)``: flat int64 device tensor aligned with the concatenated lists (-1 where the id does not
    occur in row q) and the CSR offsets.  ``np.arange(N)[np.in1d(ranks[:, q], ids)]`` of evaluate.py:80-81 in one pass."""
    if not (isinstance(ranks, torch.Tensor) and ranks.is_cuda and ranks.dtype == torch.int64 and ranks.dim() == 2):
        raise ValueError("ranks: a 2-d int64 CUDA tensor [Q, N]")
    if ranks.shape[1] != 1 and ranks.stride(1) != 1:        # (the stride of a one-element row means nothing)
        raise ValueError("ranks: every query's row must be contiguous (pass the [Q, N] matrix, not a copy of its transpose)")
    nq, n = ranks.shape
    if len(id_lists) != nq:
        raise ValueError("need one id list per query")
    ids_t, off_t, offsets = _csr(id_lists, ranks.device)
    total = ids_t.numel()
    pos = torch.empty(total, dtype=torch.int64, device=ranks.device)
    if total:
        with torch.cuda.device(ranks.device):
            check(_lib.lib().mdx_rank_positions(_vp(ranks.data_ptr()), n, nq, ranks.stride(0) if nq > 1 else n, _vp(ids_t.data_ptr()),
                                                _vp(off_t.data_ptr()), total, _vp(pos.data_ptr()), _stream()), "mdx_rank_positions")
    return pos, offsets


def gather_scores(scores, ids_t, off_t):
    sp = _dev(scores, torch.float32, "scores")
    nq, n = scores.shape
    out = torch.empty(ids_t.numel(), dtype=torch.float32, device=scores.device)
    if ids_t.numel():
        with torch.cuda.device(scores.device):
            check(_lib.lib().mdx_gather_scores(sp, n, nq, _dev(ids_t, torch.int64, "ids"),
                                               _dev(off_t, torch.int64, "offsets"), ids_t.numel(),
                                               _vp(out.data_ptr()), _stream()), "mdx_gather_scores")
    return out


def rank_count_(cnt, scores, id_offset, ref_scores, ref_ids, off_t):
    """cnt[t] += number of rows of this shard's ``scores`` that rank before
    (ref_scores[t], ref_ids[t]); the per-shard term of a global rank position."""
    sp = _dev(scores, torch.float32, "scores")
    nq, n = scores.shape
    if ref_ids.numel():
        with torch.cuda.device(scores.device):
            check(_lib.lib().mdx_rank_count(sp, n, nq, int(id_offset), _dev(ref_scores, torch.float32, "ref_scores"),
                                            _dev(ref_ids, torch.int64, "ref_ids"), _dev(off_t, torch.int64, "offsets"),
                                            ref_ids.numel(), _dev(cnt, torch.int64, "cnt"), _stream()),
                  "mdx_rank_count")
    return cnt


# ---------------------------------------------------------- whitening learning

def gram_f64(a, center=None):
    """``(a - center) @ (a - center).T`` for a float64 ``[d, n]`` device matrix -> ``[d, d]`` (exactly symmetric):
    the ``np.dot(df, df.T)`` / ``np.dot(Xc, Xc.T)`` of cirtorch/utils/whiten.py:22,42,46 on the f64 matrix cores."""
    ap = _dev(a, torch.float64, "a")
    if a.dim() != 2:
        raise ValueError("a must be [d, n]")
    d, n = a.shape
    cp = _dev(center, torch.float64, "center") if center is not None else None
    if center is not None and center.numel() != d:
        raise ValueError("center has %d elements, expected %d" % (center.numel(), d))
    out = torch.empty((d, d), dtype=torch.float64, device=a.device)
    need = _lib.lib().mdx_gram_f64_workspace(d, n)
    ws = _workspace(need, a.device)
    with _on(a):
        check(_lib.lib().mdx_gram_f64(ap, d, n, cp, _vp(out.data_ptr()), _vp(ws.data_ptr()), need, _stream()), "mdx_gram_f64")
    return out


def l2n_cols_f64_(x, eps=1e-6):
    """In place: every column of a float64 ``[d, n]`` matrix divided by (its L2 norm + eps) (``mdx_l2n_cols_f64``)."""
    xp = _dev(x, torch.float64, "x")
    if x.dim() != 2:
        raise ValueError("x must be [d, n]")
    with _on(x):
        check(_lib.lib().mdx_l2n_cols_f64(xp, x.shape[0], x.shape[1], float(eps), _stream()), "mdx_l2n_cols_f64")
    return x


def project_f64(p, x, center=None):
    """``p @ (x - center)`` for float64 ``p [dout, d]``, ``x [d, n]``, ``center [d]`` -> ``[dout, n]``
    (``np.dot(P, X-m)``, whiten.py:45)."""
    pp = _dev(p, torch.float64, "p")
    xp = _dev(x, torch.float64, "x")
    if p.dim() != 2 or x.dim() != 2 or p.shape[1] != x.shape[0]:
        raise ValueError("p [dout, d] and x [d, n] expected, got %s and %s" % (tuple(p.shape), tuple(x.shape)))
    cp = _dev(center, torch.float64, "center") if center is not None else None
    if center is not None and center.numel() != x.shape[0]:
        raise ValueError("center has %d elements, expected %d" % (center.numel(), x.shape[0]))
    out = torch.empty((p.shape[0], x.shape[1]), dtype=torch.float64, device=x.device)
    need = _lib.lib().mdx_project_f64_workspace(p.shape[0], p.shape[1])
    ws = _workspace(need, x.device)
    with _on(x):
        check(_lib.lib().mdx_project_f64(pp, p.shape[0], p.shape[1], xp, x.shape[1], cp, _vp(out.data_ptr()), _vp(ws.data_ptr()), need,
                                         _stream()), "mdx_project_f64")
    return out


# ------------------------------------------------------------ multi-GPU exchange

def query_bounds(nq, nranks, rank):
    """Queries ``[lo, hi)`` of ``rank`` under the query split (``mdx_query_bounds``)."""
    lo, hi = ctypes.c_int64(), ctypes.c_int64()
    check(_lib.lib().mdx_query_bounds(int(nq), int(nranks), int(rank), ctypes.byref(lo), ctypes.byref(hi)), "mdx_query_bounds")
    return lo.value, hi.value


class Comm:
    """RCCL communicator of libmdx.so (``mdx_comm``): the exchange of per-shard partial scores through the C ABI.

    ``Comm.from_process_group(device)`` builds one over an initialised ``torch.distributed`` group (rank 0's unique id is
    broadcast through that group -- the only use made of it); ``Comm(id_bytes, nranks, rank)`` takes an id that
    travelled by other means (``Comm.unique_id()`` on one rank)."""

    def __init__(self, id_bytes, nranks, rank, device=None):
        self._h = None
        self.nranks, self.rank = int(nranks), int(rank)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if len(id_bytes) != 128:
            raise ValueError("a communicator id is 128 bytes")
        buf = (ctypes.c_char * 128).from_buffer_copy(bytes(id_bytes))
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            check(_lib.lib().mdx_comm_init(ctypes.byref(h), buf, self.nranks, self.rank), "mdx_comm_init")
        self._h = h

    @staticmethod
    def unique_id():
        buf = (ctypes.c_char * 128)()
        check(_lib.lib().mdx_comm_unique_id(buf), "mdx_comm_unique_id")
        return bytes(buf.raw)

    @classmethod
    def from_process_group(cls, device, group=None):
        import torch.distributed as dist
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        box = [cls.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(box[0], world, rank, device)

    def _widths(self, widths):
        if len(widths) != self.nranks:
            raise ValueError("need one width per rank")
        return (ctypes.c_int64 * self.nranks)(*[int(w) for w in widths])

    def allgather_scores(self, local, widths):
        """``local [nq, widths[rank]]`` -> list of the G blocks ``[nq, widths[g]]`` (views of one buffer, back to back)."""
        nq = local.shape[0]
        if local.shape[1] != widths[self.rank]:
            raise ValueError("local block is %s, widths[%d] = %d" % (tuple(local.shape), self.rank, widths[self.rank]))
        out = torch.empty(nq * int(sum(widths)), dtype=torch.float32, device=local.device)
        with torch.cuda.device(local.device):
            check(_lib.lib().mdx_allgather_scores(self._h, _dev(local, torch.float32, "local scores"), nq, self._widths(widths),
                                                  _vp(out.data_ptr()), _stream()), "mdx_allgather_scores")
        return self._blocks(out, nq, widths)

    def exchange_scores(self, local, widths):
        """``local [nq, widths[rank]]`` -> ``(blocks [nq_mine, widths[g]] of MY queries, (qlo, qhi))``."""
        nq = local.shape[0]
        if local.shape[1] != widths[self.rank]:
            raise ValueError("local block is %s, widths[%d] = %d" % (tuple(local.shape), self.rank, widths[self.rank]))
        qlo, qhi = query_bounds(nq, self.nranks, self.rank)
        out = torch.empty(max(1, (qhi - qlo) * int(sum(widths))), dtype=torch.float32, device=local.device)
        with torch.cuda.device(local.device):
            check(_lib.lib().mdx_exchange_scores(self._h, _dev(local, torch.float32, "local scores"), nq, self._widths(widths),
                                                 _vp(out.data_ptr()), _stream()), "mdx_exchange_scores")
        return self._blocks(out, qhi - qlo, widths), (qlo, qhi)

    @staticmethod
    def _blocks(buf, rows, widths):
        blocks, o = [], 0
        for w in widths:
            blocks.append(buf[o:o + rows * int(w)].view(rows, int(w)))
            o += rows * int(w)
        return blocks

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            h, self._h = self._h, None
            check(_lib.lib().mdx_comm_destroy(h), "mdx_comm_destroy")

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _DeviceMemory:
    """Library-owned device memory as something ``torch.as_tensor`` accepts (the CUDA array interface)."""

    def __init__(self, ptr, shape, owner):
        self.__cuda_array_interface__ = {"shape": tuple(int(x) for x in shape), "typestr": "<f4", "data": (int(ptr), False),
                                         "version": 2, "strides": None}
        self._owner = owner             # keeps the exchange (and with it the allocation) alive as long as the view is


class P2P:
    """The direct-store exchange of per-shard partial scores (``mdx_p2p_*``, include/mdx.h; round 6): every rank's
    similarity kernel writes its scores straight into the receive buffers of the ranks that own the queries, over xGMI
    (buffers shared with hipIpc); a step is closed by one flag per peer; the owner ranks a DENSE ``[nq_mine, n_total]`` matrix.

    ``P2P.from_process_group(nq, n_total, device)`` builds and connects one over an initialised ``torch.distributed`` group
    (the 64-byte handles are gathered through it -- the only use made of it); ``P2P(nranks, rank, nq, n_total)`` +
    ``connect(handles)`` / ``connect_local(peers)`` take handles that travelled by other means / ranks living in this
    process.  Every rank must run the same steps: ``index.scores_p2p(queries, p2p)`` for each of its shards or chunks, then
    ``close_step()``."""

    def __init__(self, nranks, rank, nq, n_total, device=None):
        self._h = None
        self.nranks, self.rank, self.nq, self.n_total = int(nranks), int(rank), int(nq), int(n_total)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.qlo, self.qhi = query_bounds(self.nq, self.nranks, self.rank)
        buf = (ctypes.c_char * 64)()
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            status = _lib.lib().mdx_p2p_create(ctypes.byref(h), self.nranks, self.rank, self.nq, self.n_total, buf)
        self._h = h if h.value else None
        self.exportable = status == 0
        if status != 0 and self._h is None:
            check(status, "mdx_p2p_create")
        self.handle = bytes(buf.raw)
        self.connected = False

    @classmethod
    def from_process_group(cls, nq, n_total, device, group=None):
        import torch.distributed as dist
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        me = cls(world, rank, nq, n_total, device)
        handles = [None] * world
        if world > 1:
            dist.all_gather_object(handles, (me.handle, me.exportable), group=group)
            if not all(ok for _, ok in handles):
                me.close()
                raise _lib.MdxError("a rank could not export its receive buffer (hipIpcGetMemHandle): is HSA_ENABLE_IPC_MODE_LEGACY=0 set?")
            me.connect([h for h, _ in handles])
        else:
            me.connect([me.handle])
        return me

    def connect(self, handles):
        if len(handles) != self.nranks or any(len(h) != 64 for h in handles):
            raise ValueError("need one 64-byte handle per rank")
        blob = (ctypes.c_char * (64 * self.nranks)).from_buffer_copy(b"".join(bytes(h) for h in handles))
        with torch.cuda.device(self.device):
            check(_lib.lib().mdx_p2p_connect(self._h, blob), "mdx_p2p_connect")
        self.connected = True

    def connect_local(self, peers):
        """Ranks that live in ONE process: ``peers`` = the P2P objects of all ranks, in rank order."""
        bases = (ctypes.c_void_p * self.nranks)(*[_lib.lib().mdx_p2p_base(q._h) for q in peers])
        with torch.cuda.device(self.device):
            check(_lib.lib().mdx_p2p_connect_ptrs(self._h, bases), "mdx_p2p_connect_ptrs")
        self.connected = True

    def close_step(self):
        """Enqueue the end of a step (raise my flag at every peer, wait for theirs) on the current stream; returns the
        ``[nq_mine, n_total]`` similarities of MY queries against ALL rows -- a view of the receive buffer, valid until the
        step after next (clone it to keep it longer)."""
        mine = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            check(_lib.lib().mdx_p2p_close_step(self._h, ctypes.byref(mine), _stream()), "mdx_p2p_close_step")
        rows = self.qhi - self.qlo
        if rows == 0:
            return torch.empty((0, self.n_total), dtype=torch.float32, device=self.device)
        return torch.as_tensor(_DeviceMemory(mine.value, (rows, self.n_total), self), device=self.device)

    def late_peers(self):
        """Synchronises the current stream; bit r set = a wait for peer r gave up (20 s): that step's result is undefined."""
        word = ctypes.c_uint32()
        with torch.cuda.device(self.device):
            check(_lib.lib().mdx_p2p_status(self._h, ctypes.byref(word), _stream()), "mdx_p2p_status")
        return int(word.value)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            h, self._h = self._h, None
            torch.cuda.synchronize(self.device)
            check(_lib.lib().mdx_p2p_destroy(h), "mdx_p2p_destroy")

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

