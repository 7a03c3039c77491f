"""JPEG decoding split between a loader thread and the device.

The reference decodes with ``Image.open(f).convert('RGB')`` (``cirtorch/datasets/datahelpers.py:24-31``): Pillow ->
libjpeg(-turbo).  Here a loader thread only undoes the entropy coding (``mdx_jpeg_coefficients``: the serial part) and
ships the quantised DCT coefficients -- 2 bytes each, no more than the decoded pixels -- and the device does the rest with
libjpeg's integer arithmetic (``mdx_jpeg_pixels``: dequantisation, islow IDCT, fancy chroma upsampling, YCbCr -> RGB), so the
image is Pillow's pixel for pixel.  Files the device path does not cover (progressive, CMYK, ...; ``mdx_jpeg_probe``) and
every other format stay with Pillow.
"""
import ctypes

import numpy as np
import torch

from . import _lib, ops


class JpegCoefficients:
    """What a loader thread hands over for one JPEG file: geometry + quantised coefficients (+ an optional crop box)."""

    __slots__ = ("info", "coef", "quant", "box")

    def __init__(self, info, coef, quant, box=None):
        self.info, self.coef, self.quant, self.box = info, coef, quant, box

    @property
    def size(self):
        return self.info.width, self.info.height

    def pin_memory(self):
        self.coef, self.quant = self.coef.pin_memory(), self.quant.pin_memory()
        return self


MAX_PIXELS = 89_478_485         # Pillow's Image.MAX_IMAGE_PIXELS: beyond it Pillow warns (and refuses at twice that)


def entropy_decode(data, box=None, max_pixels=MAX_PIXELS):
    """``bytes`` of a JPEG file -> :class:`JpegCoefficients`, or ``None`` when the file is left to Pillow.  Host only
    (``libmdx.so`` makes no device call here), thread-safe, releases the GIL.  The file is untrusted: the coefficient buffer
    (128 B per 8x8 block) is sized from its frame header only after the probe has checked that a file of this length can
    hold such a picture (``mdx_jpeg_probe``: one bit per block at least) and never for more than ``max_pixels``."""
    lib = _lib.lib()
    buf = np.frombuffer(data, dtype=np.uint8)
    if buf.size == 0:
        return None
    info = _lib.JpegInfo()
    ops.check(lib.mdx_jpeg_probe(buf.ctypes.data, buf.size, ctypes.byref(info)), "mdx_jpeg_probe")
    if not info.supported or info.width * info.height > max_pixels:
        return None
    coef = torch.empty((info.nblocks, 64), dtype=torch.int16)
    quant = torch.empty((3, 64), dtype=torch.int16)            # uint16 bit patterns
    if lib.mdx_jpeg_coefficients(buf.ctypes.data, buf.size, coef.data_ptr(), info.nblocks, quant.data_ptr()) != 0:
        return None                                             # corrupt stream: let Pillow report (or repair) it
    return JpegCoefficients(info, coef, quant, pil_box(box))


def pil_box(box):
    """The integer box ``Image.crop`` makes of ``box``: every corner through Python's ``round`` (PIL ``Image._crop``)."""
    return tuple(int(round(v)) for v in box) if box else None


def box_on_device(box, width, height):
    """A crop box (already rounded) the device can take as a slice: inside the image (PIL pads boxes that stick out)."""
    if not box:
        return True
    x1, y1, x2, y2 = box
    return 0 <= x1 < x2 <= width and 0 <= y1 < y2 <= height


def pixels(item, device):
    """:class:`JpegCoefficients` -> uint8 ``[1,H,W,3]`` on ``device`` (cropped to the item's box)."""
    info = item.info
    coef = item.coef.to(device, non_blocking=True)
    quant = item.quant.to(device, non_blocking=True)
    planes = torch.empty(info.nblocks * 64, dtype=torch.uint8, device=device)
    rgb = torch.empty((1, info.height, info.width, 3), dtype=torch.uint8, device=device)
    with ops._on(rgb):
        ops.check(_lib.lib().mdx_jpeg_pixels(coef.data_ptr(), quant.data_ptr(), ctypes.byref(info), planes.data_ptr(), rgb.data_ptr(),
                                             ops._stream()), "mdx_jpeg_pixels")
    if item.box:
        x1, y1, x2, y2 = (int(v) for v in item.box)
        rgb = rgb[:, y1:y2, x1:x2].contiguous()
    return rgb
