"""Image down-scaling of the loader, moved behind the host-to-device copy.

The reference's loader ends with ``img.thumbnail((imsize, imsize), Image.ANTIALIAS)``
(``cirtorch/datasets/datahelpers.py:48-50``, called by ``ImagesFromList.__getitem__``,
``genericdataset.py:63-64``): an aspect-preserving LANCZOS down-scale that never enlarges.  For a worker
process that is about as much CPU time as the JPEG decode before it.  Here the workers ship the decoded
uint8 image and the device resamples it with Pillow's own integer arithmetic (``mdx_resample_u8``), so the
thumbnail is Pillow's pixel for pixel:

* :func:`thumbnail_size` -- the size Pillow's ``thumbnail`` chooses (``Image.py`` ``preserve_aspect_ratio``);
* :func:`on_device` -- whether the device takes an image: Pillow resizes in ONE LANCZOS step only when the
  image shrinks by less than 4x (``reducing_gap=2.0``: larger ratios are first reduced by an integer factor)
  and is not a 100:1 strip; everything else stays with Pillow in the worker;
* :func:`lanczos_taps` -- Pillow's ``precompute_coeffs`` + ``normalize_coeffs_8bpc`` (``Resample.c``) for one
  axis, with libm's ``sin`` as Pillow uses it; cached per (source length, target length);
* :class:`DeviceThumbnail` -- width pass, then height pass, as Pillow orders them.
"""
import functools
import math

import numpy as np
import torch

from . import ops

PRECISION_BITS = 32 - 8 - 2
LANCZOS_SUPPORT = 3.0
REDUCING_GAP = 2.0          # Image.thumbnail's default


def thumbnail_size(width, height, imsize):
    """``(w, h)`` after ``thumbnail((imsize, imsize))``, or ``None`` if the image is left alone."""
    x = y = int(math.floor(imsize))
    if x >= width and y >= height:
        return None
    aspect = width / height

    def closest(number, err):
        lo, hi = math.floor(number), math.ceil(number)
        return max(lo if err(lo) <= err(hi) else hi, 1)       # min(floor, ceil, key=err): floor wins ties

    if x / y >= aspect:
        x = closest(y * aspect, lambda n: abs(aspect - n / y))
    else:
        y = closest(x / aspect, lambda n: 0 if n == 0 else abs(aspect - x / n))
    return x, y


_PILLOW_OK = None


def pillow_agrees():
    """One-time self-check against the INSTALLED Pillow: the size rule, the one-step rule and the 22-bit taps above restate
    Pillow 12's ``thumbnail`` (``reducing_gap=2.0``); another Pillow may shrink in a different size or in two steps.  A few
    small images are thumbnailed both ways on the host (numpy with this module's taps = the arithmetic ``mdx_resample_u8``
    runs); on any difference the device route is switched off for the process and the loader resizes with Pillow, as the
    reference does."""
    global _PILLOW_OK
    if _PILLOW_OK is None:
        try:
            from PIL import Image
            rng = np.random.default_rng(0)
            ok = True
            for (w, h, imsize) in ((61, 47, 40), (50, 90, 33), (120, 31, 64)):
                img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
                pil = Image.fromarray(img)
                pil.thumbnail((imsize, imsize), getattr(Image, "LANCZOS", None) or Image.Resampling.LANCZOS)
                size = thumbnail_size(w, h, imsize)
                ok = ok and size == pil.size and np.array_equal(_host_resample(img, size), np.asarray(pil))
            _PILLOW_OK = bool(ok)
        except Exception:
            _PILLOW_OK = False
        if not _PILLOW_OK:
            import warnings
            warnings.warn("mdir_amd.resample: the installed Pillow thumbnails differently from the restated rule; "
                          "images are resized by Pillow on the host")
    return _PILLOW_OK


def _host_resample(img, size):
    """numpy form of the two ``mdx_resample_u8`` passes (width, then height) on one uint8 ``[H,W,C]`` image."""
    out = img
    for axis, target in ((1, size[0]), (0, size[1])):
        if out.shape[axis] == target:
            continue
        bounds, taps = lanczos_taps(out.shape[axis], target)
        x = np.moveaxis(out.astype(np.int64), axis, 0)
        res = np.empty((target,) + x.shape[1:], dtype=np.uint8)
        for o, (lo, cnt) in enumerate(bounds):
            acc = np.tensordot(taps[o, :cnt].astype(np.int64), x[lo:lo + cnt], axes=(0, 0)) + (1 << (PRECISION_BITS - 1))
            res[o] = np.clip(acc >> PRECISION_BITS, 0, 255)
        out = np.moveaxis(res, 0, axis)
    return out


def on_device(width, height, imsize):
    """Target ``(w, h)`` when the device should make this thumbnail, else ``None`` (nothing to do, a case Pillow
    handles in more than one LANCZOS step, or an installed Pillow this module does not restate: ``pillow_agrees``)."""
    size = thumbnail_size(width, height, imsize)
    if size is None or size == (width, height):
        return None
    if not pillow_agrees():
        return None
    if int(width / size[0] / REDUCING_GAP) > 1 or int(height / size[1] / REDUCING_GAP) > 1:
        return None             # Image.resize first reduces by an integer factor
    if height > width * 100 and size[1] < height:
        return None             # Image.resize resamples such strips height first, in two calls
    return size


def _sinc(x):
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


@functools.lru_cache(maxsize=256)
def lanczos_taps(in_size, out_size):
    """``(bounds int32 [out,2], taps int32 [out,ksize])`` for resampling ``in_size`` samples to ``out_size``."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = LANCZOS_SUPPORT * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    inv = 1.0 / filterscale
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    taps = np.zeros((out_size, ksize), dtype=np.int32)
    one = float(1 << PRECISION_BITS)
    for o in range(out_size):
        center = (o + 0.5) * scale
        first = max(int(center - support + 0.5), 0)
        count = min(int(center + support + 0.5), in_size) - first
        weights, total = [], 0.0
        for t in range(count):
            x = (t + first - center + 0.5) * inv
            w = _sinc(x) * _sinc(x / 3) if -LANCZOS_SUPPORT <= x < LANCZOS_SUPPORT else 0.0
            weights.append(w)
            total += w
        for t, w in enumerate(weights):
            if total != 0.0:
                w /= total
            taps[o, t] = int(w * one - 0.5) if w < 0 else int(w * one + 0.5)
        bounds[o] = (first, count)
    return bounds, taps


class DeviceThumbnail:
    """``uint8 [B,H,W,C]`` device images -> their ``thumbnail((imsize, imsize), LANCZOS)``, on the device.  Images the
    loader already shrank (or that need no shrinking) pass through untouched."""

    def __init__(self, imsize):
        self.imsize = imsize
        self._taps = {}

    def _device_taps(self, in_size, out_size, device):
        key = (in_size, out_size, str(device))
        if key not in self._taps:
            bounds, taps = lanczos_taps(in_size, out_size)
            self._taps[key] = (torch.from_numpy(bounds).to(device), torch.from_numpy(taps).to(device))
        return self._taps[key]

    def __call__(self, images):
        _, h, w, _ = images.shape
        size = on_device(w, h, self.imsize) if self.imsize is not None else None
        if size is None:
            return images
        if size[0] != w:
            images = ops.resample_u8(images, 1, *self._device_taps(w, size[0], images.device))
        if size[1] != h:
            images = ops.resample_u8(images, 0, *self._device_taps(h, size[1], images.device))
        return images
