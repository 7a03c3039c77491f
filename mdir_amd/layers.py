"""Pooling and normalisation layers with the cirtorch names and state-dict keys.

Drop-in for ``mdir/external/cirtorch/layers/{functional,pooling,normalization}.py``
(the global poolings ``mac``/``spoc``/``gem`` :11-22, ``l2n`` :130-131; modules
``MAC``/``SPoC``/``GeM`` pooling.py:14-47, ``L2N`` normalization.py:10-20).  The
arithmetic runs in the HIP library (``mdx_pool_l2n`` / ``mdx_l2n_rows``); there is
no torch-op or CPU fallback.  ``rmac`` / ``RMAC`` (functional.py:26-72, pooling.py:50-60; round 5) complete the
``POOLING`` registry of imageretrievalnet.py:32-37, ``roipool`` / ``Rpool`` (functional.py:75-121, pooling.py:62-95; round 5)
the ``regional: True`` networks; the losses are out of scope.
"""
import functools
import math

import torch
import torch.nn as nn
from torch.nn.parameter import Parameter

from . import ops


def _as_scalar(p):
    return float(p.detach().reshape(-1)[0]) if isinstance(p, torch.Tensor) else float(p)


def _pool(x, kind, p=3.0, eps=1e-6):
    out = ops.pool_l2n(x.contiguous(), kind, p, eps, l2n_eps=None)
    return out.reshape(x.shape[0], x.shape[1], 1, 1)


def mac(x):
    return _pool(x, "mac")


def spoc(x):
    return _pool(x, "spoc")


def gem(x, p=3, eps=1e-6):
    return _pool(x, "gem", _as_scalar(p), eps)


@functools.lru_cache(maxsize=256)
def rmac_regions(H, W, L=3):
    """``[(row0, col0, height, width), ...]`` of R-MAC on an ``H x W`` map: the whole map, then the grid of
    functional.py:26-72 in its order.  The reference computes the grid with float32 TENSORS (``torch.Tensor([2..7])``,
    ``torch.floor(wl2 + torch.Tensor(range(n)) * b)``); the same tensor arithmetic here, on the host, so that every window
    starts where the reference's does: along the longer side the number of extra windows is the one of 1..6 whose overlap
    ``(w^2 - w b) / w^2`` is closest to 0.4; level l = 1..L has windows of ``floor(2 w / (l + 1))`` (skipped when 0)."""
    steps = torch.tensor([2, 3, 4, 5, 6, 7], dtype=torch.float32)
    w = min(W, H)
    b = (max(H, W) - w) / (steps - 1)
    idx = int(torch.min(torch.abs(((w ** 2 - w * b) / w ** 2) - 0.4), 0)[1])
    extra_w, extra_h = (idx + 1 if H < W else 0), (idx + 1 if H > W else 0)
    regions = [(0, 0, H, W)]
    for l in range(1, L + 1):
        wl = math.floor(2 * w / (l + 1))
        if wl == 0:
            continue
        wl2 = math.floor(wl / 2 - 1)

        def starts(size, count):
            step = 0 if count == 1 else (size - wl) / (count - 1)
            return (torch.floor(wl2 + torch.arange(count, dtype=torch.float32) * step) - wl2).tolist()
        for i0 in starts(H, l + extra_h):
            for j0 in starts(W, l + extra_w):
                regions.append((int(i0), int(j0), wl, wl))
    return tuple(regions)


def rmac(x, L=3, eps=1e-6):
    """R-MAC (functional.py:26-72): ``[B,C,H,W] -> [B,C,1,1]``, NOT normalised as a whole (``self.norm`` follows in the network)."""
    return ops.rmac(x.contiguous(), rmac_regions(int(x.shape[2]), int(x.shape[3]), int(L)), eps).reshape(x.shape[0], x.shape[1], 1, 1)


def l2n(x, eps=1e-6):
    """``x / (||x||_2 over dim 1 + eps)``; any trailing singleton dims are kept."""
    shape = x.shape
    flat = x.reshape(shape[0], -1).clone() if x.dim() > 1 else x.reshape(1, -1).clone()
    if x.dim() > 2 and any(s != 1 for s in shape[2:]):
        raise ValueError("l2n on the MI355X path expects [B,D] or [B,D,1,1] (global descriptors)")
    return ops.l2n_rows_(flat.contiguous(), eps=eps).reshape(shape)


class MAC(nn.Module):
    def forward(self, x):
        return mac(x)

    def __repr__(self):
        return self.__class__.__name__ + "()"


class SPoC(nn.Module):
    def forward(self, x):
        return spoc(x)

    def __repr__(self):
        return self.__class__.__name__ + "()"


class GeM(nn.Module):
    """Generalised-mean pooling; ``p`` is a learnable ``Parameter`` of shape [1] under
    the state-dict key ``p`` (``pool.p`` inside ImageRetrievalNet), as upstream."""

    def __init__(self, p=3, eps=1e-6):
        super().__init__()
        self.p = Parameter(torch.ones(1) * p)
        self.eps = eps
        self._p_cache = (None, None)

    def p_value(self):
        """Python float of ``p`` without a device sync per image: cached per parameter version."""
        key = (self.p.data_ptr(), self.p._version)
        if self._p_cache[0] != key:
            self._p_cache = (key, float(self.p.detach().cpu()[0]))
        return self._p_cache[1]

    def forward(self, x):
        return _pool(x, "gem", self.p_value(), self.eps)

    def __repr__(self):
        return self.__class__.__name__ + "(p={:.4f}, eps={})".format(self.p.data.tolist()[0], self.eps)


class RMAC(nn.Module):
    """pooling.py:50-60."""

    def __init__(self, L=3, eps=1e-6):
        super().__init__()
        self.L = L
        self.eps = eps

    def forward(self, x):
        return rmac(x, L=self.L, eps=self.eps)

    def __repr__(self):
        return self.__class__.__name__ + "(L={})".format(self.L)


def roipool(x, rpool, L=3, eps=1e-6):
    """``[B,C,H,W] -> [B,R,C,1,1]``: ``rpool`` of the whole map and of every R-MAC region (functional.py:75-121).  One launch
    for the poolings the library knows (``mdx_roipool``); a foreign module is called region by region, as the reference does."""
    regions = rmac_regions(int(x.shape[2]), int(x.shape[3]), int(L))
    kind = pool_kind(rpool)
    if kind is not None:
        out = ops.roipool(x.contiguous(), regions, kind[0], kind[1], kind[2])
    else:
        out = torch.cat([rpool(x[:, :, i:i + h, j:j + w].contiguous()).reshape(x.shape[0], 1, -1) for i, j, h, w in regions], dim=1)
    return out.reshape(out.shape[0], out.shape[1], out.shape[2], 1, 1)


class Rpool(nn.Module):
    """Regional pooling (pooling.py:62-95): region vectors -> L2N -> (whiten -> L2N) -> sum over the regions -> L2N.
    State-dict keys as upstream: ``rpool.*`` (e.g. ``rpool.p``), ``whiten.weight`` / ``whiten.bias``."""

    def __init__(self, rpool, whiten=None, L=3, eps=1e-6):
        super().__init__()
        self.rpool = rpool
        self.L = L
        self.whiten = whiten
        self.norm = L2N()
        self.eps = eps
        self._whiten_index = (None, None)

    def _whiten_rows(self, rows):
        """``norm(whiten(rows))`` for ``[n, D]`` rows: mdx_scores on the resident weight shard + mdx_l2n_rows with the bias."""
        w = self.whiten.weight
        key = (w.data_ptr(), w._version, str(w.device))
        if self._whiten_index[0] != key:
            self._whiten_index = (key, ops.DescriptorIndex(w.detach().contiguous(), "ND"))
        y = self._whiten_index[1].scores(rows.contiguous(), "ND")              # [n, D_out]: row i = W rows_i
        bias = self.whiten.bias.detach() if self.whiten.bias is not None else None
        return ops.l2n_rows_(y, bias=bias, eps=self.norm.eps)

    def forward(self, x, aggregate=True):
        o = roipool(x, self.rpool, self.L, self.eps)
        b, r, d = o.shape[:3]
        rows = ops.l2n_rows_(o.reshape(b * r, d).contiguous(), eps=self.norm.eps)
        if self.whiten is not None:
            rows = self._whiten_rows(rows)
        o = rows.reshape(b, r, -1, 1, 1)
        if aggregate:
            total = ops.region_sum(o.reshape(b, r, -1).contiguous())           # [B, D]: sum over the regions
            o = ops.l2n_rows_(total, eps=self.norm.eps).reshape(b, -1, 1, 1)
        return o

    def __repr__(self):
        return super().__repr__() + "(L={})".format(self.L)


class L2N(nn.Module):
    def __init__(self, eps=1e-6):
        super().__init__()
        self.eps = eps

    def forward(self, x):
        return l2n(x, eps=self.eps)

    def __repr__(self):
        return self.__class__.__name__ + "(eps=" + str(self.eps) + ")"


POOLING = {"mac": MAC, "spoc": SPoC, "gem": GeM, "rmac": RMAC}   # networks/imageretrievalnet.py:32-37


def pool_kind(pool):
    """("gem"|"mac"|"spoc", p, eps) for a pooling module, or None if it is foreign."""
    if isinstance(pool, GeM):
        return "gem", pool.p_value(), pool.eps
    if isinstance(pool, MAC):
        return "mac", 1.0, 1e-6
    if isinstance(pool, SPoC):
        return "spoc", 1.0, 1e-6
    return None
