"""Pooling and normalisation layers with the cirtorch names and state-dict keys.

Drop-in for ``mdir/external/cirtorch/layers/{functional,pooling,normalization}.py``
(the global poolings ``mac``/``spoc``/``gem`` :11-22, ``l2n`` :130-131; modules
``MAC``/``SPoC``/``GeM`` pooling.py:14-47, ``L2N`` normalization.py:10-20).  The
arithmetic runs in the HIP library (``mdx_pool_l2n`` / ``mdx_l2n_rows``); there is
no torch-op or CPU fallback.  ``rmac`` / ``Rpool`` / losses are out of scope
(SURVEY.md section 2 row 4).
"""
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


class L2N(nn.Module):
    def __init__(self, eps=1e-6):
        super().__init__()
        self.eps = eps

    def forward(self, x):
        return l2n(x, eps=self.eps)

    def __repr__(self):
        return self.__class__.__name__ + "(eps=" + str(self.eps) + ")"


POOLING = {"mac": MAC, "spoc": SPoC, "gem": GeM}   # networks/imageretrievalnet.py:32-37 minus rmac


def pool_kind(pool):
    """("gem"|"mac"|"spoc", p, eps) for a pooling module, or None if it is foreign."""
    if isinstance(pool, GeM):
        return "gem", pool.p_value(), pool.eps
    if isinstance(pool, MAC):
        return "mac", 1.0, 1e-6
    if isinstance(pool, SPoC):
        return "spoc", 1.0, 1e-6
    return None
