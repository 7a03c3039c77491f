"""Batched whitening -- drop-in for ``whitenapply`` of
``mdir/external/cirtorch/utils/whiten.py:4-12`` (called from
``cirtorch/examples/test.py:246-249`` on the whole ``[D,N]`` matrix).

The projection is one ``mdx_scores`` call per 128 descriptors against a resident
shard of ``P[:d]`` (fp32 MFMA, database = rows of P) followed by ``mdx_l2n_rows``.
Compute is fp32 on the GPU; float64 inputs are accepted and the result is returned
in the dtype numpy would have produced, but carries fp32 accuracy (the reference
computes float64 on the CPU when handed float64 ``P``)."""
import numpy as np
import torch

from . import ops


def whitenapply(X, m, P, dimensions=None, device="cuda"):
    """``X [D,N]``, ``m [D,1]``, ``P [D,D]`` -> ``[d,N]`` (numpy in, numpy out)."""
    if not dimensions:
        dimensions = P.shape[0]
    out_dtype = np.result_type(np.asarray(X).dtype, np.asarray(m).dtype, np.asarray(P).dtype)
    dev = torch.device(device)
    Xd = torch.as_tensor(np.ascontiguousarray(X, dtype=np.float32), device=dev)
    Pd = torch.as_tensor(np.ascontiguousarray(np.asarray(P)[:dimensions], dtype=np.float32), device=dev)
    md = torch.as_tensor(np.ascontiguousarray(np.asarray(m).reshape(-1), dtype=np.float32), device=dev)
    shard = ops.DescriptorIndex(Pd, "ND")
    y = ops.l2n_rows_(shard.scores(Xd, "DN", center=md), eps=1e-6)      # [N, d]
    return y.t().contiguous().cpu().numpy().astype(out_dtype, copy=False)


# ---------------------------------------------------------------------------
# learning.  The reference does all of it in float64 (whiten.py:37-53 on float64
# descriptors): the low-variance directions of a 2048-d covariance sit below fp32 noise, so the D x D Gram
# matrices and the projection are float64 here too -- libmdx's own f64 matrix-core kernels (mdx_gram_f64,
# SYRK-shaped; mdx_project_f64 with the centring fused into the operand load; csrc/mdx_gram.hip), not the fp32 chain
# kernel of the hot path.  The small dense factorisations stay on the host, as in the reference.
# ---------------------------------------------------------------------------

def _as_f64(a, device):
    """``device`` must be a ROCm device: the float64 products below run on libmdx's f64 matrix-core kernels and there is no
    CPU route in this package (``ops.gram_f64`` / ``ops.project_f64`` raise on CPU tensors) -- an offline float64 step on
    the host is the reference itself (``cirtorch/utils/whiten.py``)."""
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=torch.device(device))


def gram(A, device="cuda", center=None):
    """``(A - center) @ (A - center).T`` for ``A [D,n]`` in float64: the ``np.dot(df, df.T)`` of ``whiten.py:42,46``
    and, with ``center = m``, the ``np.dot(Xc, Xc.T)`` of ``:21-22`` (``mdx_gram_f64``)."""
    c = None if center is None else _as_f64(np.asarray(center).reshape(-1), device)
    return ops.gram_f64(_as_f64(A, device), c).cpu().numpy()


def project(P, X, m, device="cuda"):
    """``np.dot(P, X - m)`` for ``X [D,N]`` in float64: returns ``[D_out, N]`` (``whiten.py:45``; ``mdx_project_f64``)."""
    return ops.project_f64(_as_f64(P, device), _as_f64(X, device), _as_f64(np.asarray(m).reshape(-1), device)).cpu().numpy()


def cholesky(S):
    """Cholesky factor, adding 1e-10, 1e-9, ... to the diagonal until S is positive definite
    (``whiten.py:55-70``)."""
    alpha = 0
    while True:
        try:
            return np.linalg.cholesky(S + alpha * np.eye(*S.shape))
        except np.linalg.LinAlgError:
            alpha = 1e-10 if alpha == 0 else alpha * 10
            print(">>>> whiten.py::cholesky: Matrix is not positive definite, adding {:.0e} on the diagonal".format(alpha))


def pcawhitenlearn(X, shrink=None, device="cuda"):
    """PCA whitening without annotations (``whiten.py:14-35``): returns ``(m, P)``."""
    N = X.shape[1]
    m = X.mean(axis=1, keepdims=True)
    Xcov = gram(X, device, center=m)
    Xcov = (Xcov + Xcov.T) / (2 * N)
    eigval, eigvec = np.linalg.eig(Xcov)
    order = eigval.argsort()[::-1]
    eigval, eigvec = eigval[order], eigvec[:, order]
    if shrink:
        b = eigval[shrink - 1]
        eigval = (1 - b) * eigval + b
    P = np.dot(np.linalg.inv(np.sqrt(np.diag(eigval))), eigvec.T)
    return m, P


def whitenlearn(X, qidxs, pidxs, device="cuda"):
    """Learned whitening from matching pairs (``whiten.py:37-53``): returns ``(m, P)``."""
    m = X[:, qidxs].mean(axis=1, keepdims=True)
    df = X[:, qidxs] - X[:, pidxs]
    S = gram(df, device) / df.shape[1]
    P = np.linalg.inv(cholesky(S))
    df = project(P, X, m, device)
    D = gram(df, device)
    eigval, eigvec = np.linalg.eig(D)
    order = eigval.argsort()[::-1]
    eigvec = eigvec[:, order]
    P = np.dot(eigvec.T, P)
    return m, P
