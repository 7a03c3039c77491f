"""Batched whitening -- drop-in for ``whitenapply`` of
``mdir/external/cirtorch/utils/whiten.py:4-12`` (called from
``cirtorch/examples/test.py:246-249`` on the whole ``[D,N]`` matrix).

fp32 inputs: one ``mdx_scores`` call against a resident shard of ``P[:d]`` (fp32 MFMA, database = rows of P) followed by
``mdx_l2n_rows``.  When any input is float64 the reference computes in float64 (numpy promotes; the pickled ``Lw`` / ``P`` of
``whitenlearn`` ARE float64): so does this -- ``mdx_project_f64`` (centring fused, f64 MFMA) + ``mdx_l2n_cols_f64`` -- and agrees
with numpy to 1e-12 instead of carrying fp32 accuracy into a float64 result (round 5; a differential run against the reference
showed 1e-7)."""
import numpy as np
import torch

from . import ops


def whitenapply(X, m, P, dimensions=None, device="cuda"):
    """``X [D,N]``, ``m [D,1]``, ``P [D,D]`` -> ``[d,N]`` (numpy in, numpy out)."""
    if not dimensions:
        dimensions = P.shape[0]
    out_dtype = np.result_type(np.asarray(X).dtype, np.asarray(m).dtype, np.asarray(P).dtype)
    dev = torch.device(device)
    if out_dtype == np.float64:
        y = ops.project_f64(_as_f64(np.asarray(P)[:dimensions], dev), _as_f64(X, dev), _as_f64(np.asarray(m).reshape(-1), dev))   # [d, N]
        return ops.l2n_cols_f64_(y, eps=1e-6).cpu().numpy()
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
# kernel of the hot path.  Of the small dense factorisations only the Cholesky (a yes / no decision of the reference's LAPACK
# call) stays on the host; the symmetric eigen-decomposition runs next to the data (rocSOLVER through torch.linalg).
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


def _eigh_descending(M):
    """Eigenvalues (descending) and eigenvectors (columns) of a symmetric float64 matrix ON ITS DEVICE.  The reference calls
    ``np.linalg.eig`` and sorts (whiten.py:23-26, :47-49): on a symmetric matrix the same decomposition, but LAPACK's
    general solver takes 10.3 s for 2048 x 2048 on the host (89 % of one whitening learning, tools/whitenlearn_bench.py)
    where the symmetric solver next to the data takes 0.06 s.  Eigenvectors are defined up to sign either way (the tests
    compare rows of P up to sign; whitened dot products do not see it).  With REPEATED eigenvalues the two solvers may return
    different (equally valid) bases of the shared eigenspace, so P can then differ from the reference's by more than row
    signs; P.T @ P -- all that distances between whitened descriptors depend on -- is the same.  A failure of the solver is
    raised as the reference's callers expect it (mdir/stages/whiten.py catches ``np.linalg.LinAlgError``)."""
    try:
        w, v = torch.linalg.eigh(M)
    except RuntimeError as e:               # torch's _LinAlgError is a RuntimeError
        raise np.linalg.LinAlgError(str(e)) from e
    return torch.flip(w, [0]), torch.flip(v, [1])


def _inverse_lower(L, device):
    """``np.linalg.inv`` of a Cholesky factor (whiten.py:43), as a triangular solve on the device."""
    Ld = _as_f64(L, device)
    return torch.linalg.solve_triangular(Ld, torch.eye(Ld.shape[0], dtype=torch.float64, device=Ld.device), upper=False)


def _like_input(t, X):
    """The float64 result in the dtype numpy would have given the reference: its statements run in the dtype of ``X`` (float32
    descriptors -> float32 ``m`` / ``P``, and with them a float32 ``.lw.pkl`` and float32 whitened descriptors downstream:
    cirtorch_format/test.py:241-268).  Computed in float64 here either way; float64 in, float64 out."""
    a = t.cpu().numpy()
    dt = np.asarray(X).dtype
    return a.astype(dt, copy=False) if dt in (np.float32, np.float64) else a


def pcawhitenlearn(X, shrink=None, device="cuda"):
    """PCA whitening without annotations (``whiten.py:14-35``): returns ``(m, P)``.  The descriptors go to the device once;
    mean, covariance (``mdx_gram_f64`` with the centring fused), eigen-decomposition and scaling happen there."""
    N = X.shape[1]
    Xd = _as_f64(X, device)
    m = Xd.mean(dim=1, keepdim=True)
    Xcov = ops.gram_f64(Xd, m.reshape(-1).contiguous())
    Xcov = (Xcov + Xcov.t()) / (2 * N)
    eigval, eigvec = _eigh_descending(Xcov)
    if shrink:
        b = eigval[shrink - 1]
        eigval = (1 - b) * eigval + b
    P = (1.0 / torch.sqrt(eigval))[:, None] * eigvec.t()       # inv(sqrt(diag(eigval))) @ eigvec.T
    return _like_input(m, X), _like_input(P, X)


def whitenlearn(X, qidxs, pidxs, device="cuda"):
    """Learned whitening from matching pairs (``whiten.py:37-53``): returns ``(m, P)``.  One copy of the descriptors to the
    device; pair differences, the two Gram matrices (``mdx_gram_f64``), the projection (``mdx_project_f64``), the
    triangular inverse, the eigen-decomposition and the final product happen there.  The Cholesky factorisation -- with
    the reference's retry rule on a matrix that is not positive definite, a yes / no decision of ITS LAPACK call -- stays
    on the host.  D = 2048, 20 000 pairs of 40 000 descriptors: 11.6 s -> 0.6 s (tools/whitenlearn_bench.py)."""
    Xd = _as_f64(X, device)
    qi = torch.as_tensor(np.asarray(qidxs, dtype=np.int64), device=Xd.device)
    pi = torch.as_tensor(np.asarray(pidxs, dtype=np.int64), device=Xd.device)
    Xq = Xd[:, qi]
    m = Xq.mean(dim=1, keepdim=True)
    df = (Xq - Xd[:, pi]).contiguous()
    del Xq
    S = ops.gram_f64(df) / df.shape[1]
    del df
    P = _inverse_lower(cholesky(S.cpu().numpy()), device)
    df = ops.project_f64(P.contiguous(), Xd, m.reshape(-1).contiguous())
    D = ops.gram_f64(df)
    del df
    _, eigvec = _eigh_descending(D)
    P = ops.project_f64(eigvec.t().contiguous(), P.contiguous())     # np.dot(eigvec.T, P)
    return _like_input(m, X), _like_input(P, X)
