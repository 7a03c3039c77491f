"""CPU oracle for the mdir / cirtorch descriptor-extraction-and-ranking hot path.

TEST INFRASTRUCTURE ONLY.  This module is a plain numpy restatement of the
reference's arithmetic, written from the reference's behaviour (file:line cited
per function, paths relative to the upstream repo).  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it -- never the product package ``mdir_amd``: the product path must fail
loudly when the HIP library is missing, it has no CPU fallback.

Parity pin: the reference ships NO tests or golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself, generated in the build container by ``tests/golden/make_golden.py``
(which imports the reference with throw-away stubs for torchvision/cv2/h5py)
and committed as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks
every function below against them.

All arithmetic is float32 unless stated (the reference computes in fp32 on the
device and in float64 only inside compute_map / whitenapply-with-f64-P).
"""
import math

import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------
# a4 / a5: pooling and normalisation
# --------------------------------------------------------------------------

def gem(x, p=3.0, eps=1e-6):
    """Generalised-mean pooling of a feature map batch ``[B,C,H,W] -> [B,C]``.

    Reference: ``mdir/external/cirtorch/layers/functional.py:21-22`` --
    clamp to ``eps`` from below, raise to ``p``, average over the whole H x W
    window, raise to ``1/p`` (``1/p`` is itself rounded to fp32 because ``p`` is
    an fp32 Parameter, ``layers/pooling.py:36-44``).
    """
    x = np.asarray(x, dtype=F32)
    p32 = F32(p)
    powed = np.power(np.maximum(x, F32(eps)), p32, dtype=F32)
    mean = powed.reshape(x.shape[0], x.shape[1], -1).mean(axis=2, dtype=F32)
    return np.power(mean, F32(1.0) / p32, dtype=F32)


def mac(x):
    """Global max pooling ``[B,C,H,W] -> [B,C]`` (``layers/functional.py:11-12``)."""
    x = np.asarray(x, dtype=F32)
    return x.reshape(x.shape[0], x.shape[1], -1).max(axis=2)


def spoc(x):
    """Global average pooling ``[B,C,H,W] -> [B,C]`` (``layers/functional.py:16-17``)."""
    x = np.asarray(x, dtype=F32)
    return x.reshape(x.shape[0], x.shape[1], -1).mean(axis=2, dtype=F32)


def rmac_regions(H, W, L=3):
    """The square regions of R-MAC on an ``H x W`` map as ``(i0, j0, size)`` rows, whole-map term NOT included.

    Reference: ``layers/functional.py:26-72`` (the same grid in ``roipool`` :75-121).  The reference computes the grid in
    float32 tensors (``torch.Tensor([...])``, ``torch.floor``); so does this, with numpy float32: the number of regions along
    the longer side is the one of 2..7 whose overlap is closest to 0.4, level ``l`` has windows of ``floor(2 w / (l + 1))``
    spaced evenly with float32 steps.
    """
    ovr = 0.4
    steps = np.array([2, 3, 4, 5, 6, 7], dtype=F32)
    w = min(W, H)
    b = (F32(max(H, W) - w) / (steps - F32(1))).astype(F32)
    idx = int(np.argmin(np.abs(((F32(w * w) - F32(w) * b) / F32(w * w)) - F32(ovr))))        # first minimum, as torch.min
    Wd = idx + 1 if H < W else 0
    Hd = idx + 1 if H > W else 0
    out = []
    for l in range(1, L + 1):
        wl = math.floor(2 * w / (l + 1))
        wl2 = math.floor(wl / 2 - 1)
        bw = 0 if l + Wd == 1 else (W - wl) / (l + Wd - 1)
        cen_w = np.floor(F32(wl2) + np.arange(l - 1 + Wd + 1, dtype=F32) * F32(bw)) - F32(wl2)
        bh = 0 if l + Hd == 1 else (H - wl) / (l + Hd - 1)
        cen_h = np.floor(F32(wl2) + np.arange(l - 1 + Hd + 1, dtype=F32) * F32(bh)) - F32(wl2)
        for i_ in cen_h.tolist():
            for j_ in cen_w.tolist():
                if wl == 0:
                    continue
                out.append((int(i_), int(j_), wl))
    return out


def rmac(x, L=3, eps=1e-6):
    """R-MAC pooling ``[B,C,H,W] -> [B,C]``: the L2-normalised (eps added to the norm) global maximum plus the
    L2-normalised maxima of every region, summed in region order (``layers/functional.py:26-72``)."""
    x = np.asarray(x, dtype=F32)
    B, C, H, W = x.shape
    v = l2n(x.reshape(B, C, -1).max(axis=2), eps)
    for i0, j0, wl in rmac_regions(H, W, L):
        v = (v + l2n(x[:, :, i0:i0 + wl, j0:j0 + wl].reshape(B, C, -1).max(axis=2), eps)).astype(F32)
    return v


def roipool(x, pool, L=3):
    """``[B,C,H,W] -> [B,R,C]``: ``pool`` (a function ``[B,C,h,w] -> [B,C]``) of the whole map and of every R-MAC region, in
    the reference's order (``layers/functional.py:75-121``)."""
    x = np.asarray(x, dtype=F32)
    out = [pool(x)]
    for i0, j0, wl in rmac_regions(x.shape[2], x.shape[3], L):
        out.append(pool(np.ascontiguousarray(x[:, :, i0:i0 + wl, j0:j0 + wl])))
    return np.stack(out, axis=1).astype(F32)


def rpool(x, pool, weight=None, bias=None, L=3, eps=1e-6, aggregate=True):
    """Regional pooling (``layers/pooling.py:62-95``): every region pooled and L2-normalised, optionally whitened
    (``W r + b``) and normalised again, then summed over the regions and normalised -- ``[B,C]`` (``aggregate``) or the
    regional vectors ``[B,R,C]``."""
    o = roipool(x, pool, L)
    B, R, C = o.shape
    o = l2n(o.reshape(B * R, C), eps)
    if weight is not None:
        o = (o @ np.asarray(weight, dtype=F32).T).astype(F32)
        if bias is not None:
            o = (o + np.asarray(bias, dtype=F32)).astype(F32)
        o = l2n(o, eps)
    o = o.reshape(B, R, -1)
    if aggregate:
        return l2n(o.sum(axis=1, dtype=F32), eps)
    return o


def l2n(x, eps=1e-6):
    """L2-normalise over axis 1 with eps ADDED TO THE NORM.

    Reference: ``layers/functional.py:130-131``.  A zero vector maps to zero
    (never NaN) because the divisor is ``0 + eps``.
    """
    x = np.asarray(x, dtype=F32)
    nrm = np.sqrt(np.sum(x * x, axis=1, keepdims=True, dtype=F32), dtype=F32)
    return (x / (nrm + F32(eps))).astype(F32)


def forward_tail(feat, p=3.0, eps=1e-6, whiten_w=None, whiten_b=None, pooling="gem"):
    """Everything ``ImageRetrievalNet.forward`` does after ``features``.

    Reference: ``networks/imageretrievalnet.py:107-115``: pool -> L2N ->
    (optional ``nn.Linear`` whitening -> L2N) -> permute to ``[D,B]``.
    """
    pooled = {"gem": lambda t: gem(t, p, eps), "mac": mac, "spoc": spoc}[pooling](feat)
    o = l2n(pooled)
    if whiten_w is not None:
        w = np.asarray(whiten_w, dtype=F32)
        o = o @ w.T
        if whiten_b is not None:
            o = o + np.asarray(whiten_b, dtype=F32)[None, :]
        o = l2n(o.astype(F32))
    return np.ascontiguousarray(o.T)


# --------------------------------------------------------------------------
# a7: multi-scale aggregation
# --------------------------------------------------------------------------

MS_SCALES = (1.0, 1.0 / np.sqrt(2.0), 0.5)  # mdir/components/data/wrapper.py:93


def bn_act(x, mean, var, weight=None, bias=None, eps=1e-5, residual=None, relu=True):
    """Inference batch-norm + residual add + ReLU of a residual block on ``x [N,C,H,W]``:
    ``relu(bn(x) + identity)`` -- torch.nn.BatchNorm2d in eval mode,
    ``(x - running_mean) / sqrt(running_var + eps) * weight + bias``, as used by the torchvision
    ResNet blocks the reference keeps as ``features`` (cirtorch/networks/imageretrievalnet.py:172-173).
    float64 inside, so it is a reference for fp32 implementations in either operation order."""
    shape = (1, -1, 1, 1)
    y = (x.astype(np.float64) - mean.astype(np.float64).reshape(shape)) / np.sqrt(var.astype(np.float64).reshape(shape) + eps)
    if weight is not None:
        y = y * weight.astype(np.float64).reshape(shape)
    if bias is not None:
        y = y + bias.astype(np.float64).reshape(shape)
    if residual is not None:
        y = y + residual.astype(np.float64)
    return (np.maximum(y, 0.0) if relu else y).astype(F32)


def ms_aggregate(vecs, msp=1.0):
    """Aggregate ``S`` per-scale descriptors ``[S,D] -> [D]``.

    Reference: ``mdir/components/data/wrapper.py:109-119`` (mdir path) and
    ``networks/imageretrievalnet.py:309-324`` (cirtorch path): power-mean with
    exponent ``msp`` then divide by the plain L2 norm -- NO eps here.
    """
    vecs = np.asarray(vecs, dtype=F32)
    msp32 = F32(msp)
    acc = np.zeros(vecs.shape[1], dtype=F32)
    for v in vecs:  # sequential, in scale order, like the reference's += loop
        acc = acc + np.power(v, msp32, dtype=F32)
    v = np.power(acc / F32(vecs.shape[0]), F32(1.0) / msp32, dtype=F32)
    return (v / np.sqrt(np.sum(v * v, dtype=F32), dtype=F32)).astype(F32)


def ms_power(model_meta, nscales, pool_p):
    """Which exponent the mdir wrapper aggregates with.

    Reference: ``wrapper.py:121-124``: ``pool.p`` iff more than one scale, GeM
    pooling, not regional, and the MODEL has no in-network whitening; else 1.
    """
    if nscales > 1 and model_meta["pooling"] == "gem" and not model_meta["regional"] \
            and not model_meta["whitening"]:
        return float(pool_p)
    return 1.0


# --------------------------------------------------------------------------
# a8: whitening projection
# --------------------------------------------------------------------------

def whiten_wrapper(v, m, P, dimensions=None):
    """``CirtorchWhiten.postprocess`` for one descriptor ``[D] -> [d]`` in fp32.

    Reference: ``mdir/components/data/wrapper.py:186-195``: P and m are cast to
    fp32, ``X = P[:d] @ (v - m)``, then ``X / (||X|| + 1e-6)``.
    """
    P = np.asarray(P, dtype=F32)
    m = np.asarray(m, dtype=F32).reshape(-1)
    v = np.asarray(v, dtype=F32).reshape(-1)
    d = dimensions or P.shape[0]
    X = (P[:d, :] @ (v - m)).astype(F32)
    return (X / (np.sqrt(np.sum(X * X, dtype=F32), dtype=F32) + F32(1e-6))).astype(F32)


def whitenapply(X, m, P, dimensions=None):
    """Batched whitening ``[D,N] -> [d,N]`` in the dtype numpy promotes to.

    Reference: ``mdir/external/cirtorch/utils/whiten.py:4-12``.
    """
    d = dimensions or P.shape[0]
    Y = np.dot(P[:d, :], X - m)
    return Y / (np.linalg.norm(Y, ord=2, axis=0, keepdims=True) + 1e-6)


# --------------------------------------------------------------------------
# a11 / a12: similarity and ranking
# --------------------------------------------------------------------------

def scores(vecs, qvecs):
    """``[D,N]`` database x ``[D,Q]`` queries -> fp32 ``[N,Q]`` similarity.

    Reference: ``mdir/components/optim/score/cirscore.py:69``
    (same statement at ``cirtorch/examples/test.py:240``).
    """
    return np.dot(np.asarray(vecs, dtype=F32).T, np.asarray(qvecs, dtype=F32))


def bf16_round(x):
    """fp32 -> the nearest bfloat16 (ties to even), returned as fp32: what ``v_cvt_pk_bf16_f32`` does to a finite value."""
    u = np.ascontiguousarray(x, dtype=F32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(F32).reshape(np.shape(x))


def split3_bf16(x):
    """``x = h + m + l + e`` with bf16 pieces ``h = bf16(x)``, ``m = bf16(x - h)``, ``l = bf16(x - h - m)`` (the
    residuals are exact in fp32) -- the operand split of the library's labelled split-precision similarity
    (``MDX_F32_SPLIT3``, include/mdx.h; mdir_amd/csrc/mdx_scores_split_kernel.h).  Test infrastructure."""
    x = np.asarray(x, dtype=F32)
    h = bf16_round(x)
    r = x - h
    m = bf16_round(r)
    l = bf16_round(r - m)
    return h, m, l


def scores_split3(vecs, qvecs):
    """The split-precision form of ``np.dot(vecs.T, qvecs)`` (cirscore.py:69) as the library computes it, up to the
    order of the fp32 accumulation (here: every product exact, summed in float64): the six piece products
    ``hh + hm + mh + hl + lh + mm``.  ``[D,N]``, ``[D,Q]`` -> fp32 ``[N,Q]``."""
    dh, dm, dl = (p.astype(np.float64) for p in split3_bf16(vecs))
    qh, qm, ql = (p.astype(np.float64) for p in split3_bf16(qvecs))
    s = dh.T @ qh + (dh.T @ qm + dm.T @ qh) + (dh.T @ ql + dl.T @ qh + dm.T @ qm)
    return s.astype(F32)


def _fp16_toward_zero(x):
    """float64 -> the fp16 value of no larger magnitude (what ``v_cvt_pkrtz_f16_f32`` gives a finite in-range value),
    as float64: normals keep 11 significant bits, values under 2^-14 sit on the 2^-24 grid."""
    x = np.asarray(x, dtype=np.float64)
    a = np.abs(x)
    e = np.floor(np.log2(np.where(a > 0, a, 1.0)))
    q = np.exp2(np.maximum(e, -14.0) - 10.0)
    return np.sign(x) * np.floor(a / q) * q


def split2_scale(a):
    """The power of two that brings the matrix' largest finite magnitude into [2^13, 2^14) (1 for an all-zero matrix):
    the block exponent of ``MDX_F32_SPLIT2`` (mdir_amd/csrc/mdx_scores_split_kernel.h ``split2_scale``)."""
    m = np.abs(np.asarray(a, dtype=F32))
    m = m[np.isfinite(m)]
    top = float(m.max()) if m.size else 0.0
    if top == 0.0 or top < np.finfo(np.float32).tiny:
        return 1.0
    return float(np.exp2(np.clip(13 - np.floor(np.log2(top)), -126, 126)))


def scores_split2(vecs, qvecs):
    """The two-piece block-floating form of ``np.dot(vecs.T, qvecs)`` (cirscore.py:69) as the library computes it
    (``MDX_F32_SPLIT2``), up to the order of the fp32 accumulation: each matrix scaled by its ``split2_scale``, every operand
    ``X = h + m / 2^11`` in fp16 (round toward zero, residual exact), products ``hh + (hm + mh) / 2^11`` summed in float64,
    unscaled.  ``[D,N]``, ``[D,Q]`` -> fp32 ``[N,Q]``.  Test infrastructure."""
    sd, sq = split2_scale(vecs), split2_scale(qvecs)

    def pieces(a, s):
        X = np.asarray(a, dtype=F32).astype(np.float64) * s
        h = _fp16_toward_zero(X)
        return h, _fp16_toward_zero((X - h) * 2048.0)
    dh, dm = pieces(vecs, sd)
    qh, qm = pieces(qvecs, sq)
    s = dh.T @ qh + (dh.T @ qm + dm.T @ qh) / 2048.0
    return (s / (sd * sq)).astype(F32)


def ranks(sc):
    """Per-query descending ranking ``[N,Q] -> int64 [N,Q]``.

    Reference: ``cirscore.py:70``.  numpy's default sort is not stable, so tie
    order is implementation-defined there; the build's rule (and this
    oracle's) is descending score, ascending database id -- one of the orders
    the reference may legally produce.
    """
    return np.argsort(-np.asarray(sc), axis=0, kind="stable")


def rank_of(sc, ids):
    """Zero-based rank position of each database id in ``ids`` for ONE query.

    ``sc`` is that query's score column ``[N]``.  Position = number of items
    strictly better, plus equal-scored items with a smaller id (the tie rule of
    :func:`ranks`).  Equals ``np.nonzero(ranks(sc)[:, None] == ids)`` without
    materialising the ranking; feeds :func:`compute_ap` exactly like
    ``evaluate.py:80-81`` does through ``np.in1d``.
    """
    sc = np.asarray(sc)
    ids = np.asarray(ids, dtype=np.int64)
    key = -sc
    out = np.empty(len(ids), dtype=np.int64)
    for t, i in enumerate(ids):
        better = np.count_nonzero(key < key[i])
        tied_before = np.count_nonzero(key[:i] == key[i])
        out[t] = better + tied_before
    return out


def topk(sc, k):
    """First ``k`` rows of :func:`ranks` plus their scores (``[k,Q]`` each)."""
    r = ranks(sc)[:k]
    return r, np.take_along_axis(np.asarray(sc), r, axis=0)


# --------------------------------------------------------------------------
# a13: mean average precision
# --------------------------------------------------------------------------

def compute_ap(pos, nres):
    """Average precision from zero-based ranks of the positives.

    Reference: ``mdir/external/cirtorch/utils/evaluate.py:3-37``: area under
    the precision/recall polyline, one trapezoid per positive, accumulated in
    order in float64.
    """
    ap = 0.0
    step = 1.0 / nres
    for j, rank in enumerate(pos):
        rank = int(rank)
        before = 1.0 if rank == 0 else float(j) / rank
        after = float(j + 1) / (rank + 1)
        ap += (before + after) * step / 2.0
    return ap


def _junk_shift(pos, junk):
    """Move each positive up by the number of junk items ranked before it
    (``evaluate.py:85-94``).  Both inputs ascending."""
    return pos - np.searchsorted(junk, pos, side="left")


def ap_from_positions(pos, junk, nok, kappas=()):
    """AP and precision@kappas for one query from rank POSITIONS.

    ``pos`` / ``junk``: ascending zero-based positions in the ranking of the
    positive / junk database ids; ``nok`` = number of positives.  This is the
    body of the per-query loop of ``evaluate.py:79-106``.
    """
    pos = _junk_shift(np.asarray(pos, dtype=np.int64), np.asarray(junk, dtype=np.int64))
    ap = compute_ap(pos, nok)
    pos1 = pos + 1
    prs = np.zeros(len(kappas))
    for j, kappa in enumerate(kappas):
        kq = min(int(pos1.max()), kappa)
        prs[j] = np.count_nonzero(pos1 <= kq) / kq
    return ap, prs


def compute_map(rk, gnd, kappas=()):
    """mAP over queries: ``(map, aps[Q], pr[K], prs[Q,K])``.

    Reference: ``evaluate.py:39-111``.  ``rk`` is ``[N,Q]`` (column q = database
    ids best to worst), ``gnd[q]`` has ``ok`` and optionally ``junk`` id lists.
    Queries without positives get NaN and are left out of the mean (:68-72,108).
    """
    nq = len(gnd)
    aps = np.zeros(nq)
    prs = np.zeros((nq, len(kappas)))
    pr = np.zeros(len(kappas))
    total = 0.0
    nempty = 0
    where = np.arange(rk.shape[0])
    for q in range(nq):
        ok = np.array(gnd[q]["ok"])
        if ok.shape[0] == 0:
            aps[q] = np.nan
            prs[q, :] = np.nan
            nempty += 1
            continue
        junk_ids = np.array(gnd[q]["junk"]) if "junk" in gnd[q] else np.empty(0)
        column = rk[:, q]
        pos = where[np.isin(column, ok)]
        junk = where[np.isin(column, junk_ids)]
        ap, prs[q, :] = ap_from_positions(pos, junk, len(ok), kappas)
        aps[q] = ap
        total += ap
        pr = pr + prs[q, :]
    return total / (nq - nempty), aps, pr / (nq - nempty), prs


def protocol_gnd(gnd, level):
    """Revisited-Oxford/Paris regrouping into ok/junk for easy|medium|hard
    (``evaluate.py:125-147``)."""
    ok_keys, junk_keys = {"easy": (("easy",), ("junk", "hard")),
                          "medium": (("easy", "hard"), ("junk",)),
                          "hard": (("hard",), ("junk", "easy"))}[level]
    out = []
    for g in gnd:
        out.append({"ok": np.concatenate([g[k] for k in ok_keys]),
                    "junk": np.concatenate([g[k] for k in junk_keys])})
    return out


def compute_map_and_print(dataset, rk, gnd, kappas=(1, 5, 10)):
    """``(averages, per_query)`` dictionaries, reference key names.

    Reference: ``evaluate.py:114-152`` (the mdir-patched variant that RETURNS
    dictionaries): old protocol when the first gnd entry has ``ok``; revisited
    protocol for ``roxford5k*`` / ``rparis6k*``; anything else returns None.
    """
    if "ok" in gnd[0]:
        m, aps, _, _ = compute_map(rk, gnd)
        return {"map": m}, {"ap": aps}
    if dataset.startswith("roxford5k") or dataset.startswith("rparis6k"):
        avg, per = {}, {}
        for level in ("easy", "medium", "hard"):
            m, aps, _, _ = compute_map(rk, protocol_gnd(gnd, level), list(kappas))
            avg["map_" + level] = m
            per["ap_" + level] = aps
        return avg, per
    return None


# ---------------------------------------------------------------------------------------------------------
# Rows SURVEY.md section 8 marks "next" (f1-f3) and the image loader (a1); pinned by tests/golden g12-g15
# ---------------------------------------------------------------------------------------------------------

def cholesky_bumped(S):
    """``cholesky`` of cirtorch/utils/whiten.py:55-70: retry with 1e-10, 1e-9, ... added to the diagonal until S is
    positive definite (the reference also prints a line per retry)."""
    alpha = 0
    while True:
        try:
            return np.linalg.cholesky(S + alpha * np.eye(*S.shape))
        except np.linalg.LinAlgError:
            alpha = 1e-10 if alpha == 0 else alpha * 10


def whitenlearn(X, qidxs, pidxs):
    """cirtorch/utils/whiten.py:37-53: ``(m [D,1], P [D,D])`` from matching pairs, float64 throughout."""
    m = X[:, qidxs].mean(axis=1, keepdims=True)
    df = X[:, qidxs] - X[:, pidxs]
    S = np.dot(df, df.T) / df.shape[1]
    P = np.linalg.inv(cholesky_bumped(S))
    df = np.dot(P, X - m)
    D = np.dot(df, df.T)
    eigval, eigvec = np.linalg.eig(D)
    eigvec = eigvec[:, eigval.argsort()[::-1]]
    return m, np.dot(eigvec.T, P)


def pcawhitenlearn(X, shrink=None):
    """cirtorch/utils/whiten.py:14-35: PCA whitening without annotations, optional eigenvalue shrinkage."""
    N = X.shape[1]
    m = X.mean(axis=1, keepdims=True)
    Xc = X - m
    Xcov = np.dot(Xc, Xc.T)
    Xcov = (Xcov + Xcov.T) / (2 * N)
    eigval, eigvec = np.linalg.eig(Xcov)
    order = eigval.argsort()[::-1]
    eigval, eigvec = eigval[order], eigvec[:, order]
    if shrink:
        b = eigval[shrink - 1]
        eigval = (1 - b) * eigval + b
    return m, np.dot(np.linalg.inv(np.sqrt(np.diag(eigval))), eigvec.T)


def hard_negatives(qvecs, poolvecs, idxs2images, clusters, qidxs, nnum):
    """The selection of cirtorch/datasets/traindataset.py:242-270: scores = poolvecs^T qvecs, sorted descending per
    query; walk the ranking taking pool images of clusters not seen yet (the query's own cluster counts as seen)
    until ``nnum`` are found; the l2 distance of each pick is ``sqrt(sum((q - p + 1e-6)^2))``."""
    sc = np.dot(poolvecs.T, qvecs)                             # [P,Q]
    order = np.argsort(-sc, axis=0, kind="stable")
    nidxs, ndist = [], []
    for q in range(len(qidxs)):
        seen, picks, r = [clusters[qidxs[q]]], [], 0
        while len(picks) < nnum:
            col = order[r, q]
            img = int(idxs2images[col])
            if clusters[img] not in seen:
                picks.append(img)
                seen.append(clusters[img])
                ndist.append(float(np.sqrt(np.sum((qvecs[:, q] - poolvecs[:, col] + np.float32(1e-6)) ** 2, dtype=np.float32))))
            r += 1
        nidxs.append(picks)
    return nidxs, ndist


def embedding_output(nimages, rows):
    """``EmbeddingOutput`` of mdir/components/data/output.py:117-139: ``rows`` = per image a 1-D descriptor or None
    (unreadable); float64 ``[N,D]`` with NaN rows; ``[]`` when nothing was ever added."""
    out = None
    for i, v in enumerate(rows):
        if v is None:
            out[i, :] = np.nan                                 # like the reference: needs a readable image before it
            continue
        if out is None:
            out = np.zeros((nimages, len(v)))
        out[i, :] = np.asarray(v)
    return out if out is not None else []


def load_image(path, imsize=None, bbx=None):
    """``ImagesFromList.__getitem__`` of cirtorch/datasets/genericdataset.py:44-70 without the transform: RGB decode
    (datahelpers.py:24-31), crop to ``bbx = (x1,y1,x2,y2)`` if given, then ``thumbnail((imsize, imsize))`` with the
    filter Pillow called ANTIALIAS until 9.5 and LANCZOS since (datahelpers.py:48-50): aspect-preserving, never
    enlarges.  Returns uint8 ``[H,W,3]``."""
    from PIL import Image
    with open(path, "rb") as f:
        img = Image.open(f).convert("RGB")
    if bbx:
        img = img.crop(bbx)
    if imsize is not None:
        img.thumbnail((imsize, imsize), Image.LANCZOS)
    return np.asarray(img).copy()


# ---- Pillow's thumbnail, restated (third-party arithmetic the reference calls: datahelpers.py:48-50 -> Image.thumbnail,
# Pillow 12.2 here; src/PIL/Image.py thumbnail/resize, src/libImaging/Resample.c).  Pinned against Pillow itself and,
# through load_image, against golden G15 in tests/test_oracle_golden.py. ----------------------------------------------

def thumbnail_size(width, height, imsize):
    """Size ``Image.thumbnail((imsize, imsize))`` gives a ``width x height`` image, or ``None`` when it leaves the
    image alone (it never enlarges): Image.py ``thumbnail.preserve_aspect_ratio``."""
    import math

    def round_aspect(number, key):
        return max(min(math.floor(number), math.ceil(number), key=key), 1)

    x = y = int(math.floor(imsize))
    if x >= width and y >= height:
        return None
    aspect = width / height
    if x / y >= aspect:
        x = round_aspect(y * aspect, key=lambda n: abs(aspect - n / y))
    else:
        y = round_aspect(x / aspect, key=lambda n: 0 if n == 0 else abs(aspect - x / n))
    return x, y


def lanczos_taps(in_size, out_size, in0=0.0, in1=None):
    """Fixed-point taps of Pillow's LANCZOS (support 3) resampling of ``in_size`` samples to ``out_size``:
    ``(bounds int32 [out,2] = (first source index, count), k int32 [out, ksize])``, k = round(w * 2^22).
    Resample.c ``precompute_coeffs`` + ``normalize_coeffs_8bpc``."""
    import math
    in1 = float(in_size) if in1 is None else in1
    scale = (np.float32(in1) - np.float32(in0)).astype(np.float64) / out_size
    filterscale = max(float(scale), 1.0)
    support = 3.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1

    def sinc(x):
        if x == 0.0:
            return 1.0
        x = x * math.pi
        return math.sin(x) / x

    def lanczos(x):
        return sinc(x) * sinc(x / 3) if -3.0 <= x < 3.0 else 0.0

    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * float(scale)
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [lanczos((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << 22)) if v < 0 else int(0.5 + v * (1 << 22))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def resample_u8(img, out_w, out_h):
    """Pillow's two-pass LANCZOS ``resize((out_w, out_h))`` of a uint8 ``[H,W,C]`` image (whole-image box): horizontal
    pass first, uint8 in between, ``clip8((2^21 + sum pixel * k) >> 22)`` (Resample.c ``ImagingResampleHorizontal_8bpc``
    / ``Vertical_8bpc``)."""
    img = np.asarray(img, dtype=np.uint8)
    H, W, _ = img.shape

    def one_pass(src, axis_len, out_len):        # resamples axis 0 of src [axis_len, ...]
        bounds, kk = lanczos_taps(axis_len, out_len)
        out = np.empty((out_len,) + src.shape[1:], dtype=np.uint8)
        s64 = src.astype(np.int64)
        for xx in range(out_len):
            lo, cnt = bounds[xx]
            acc = np.tensordot(kk[xx, :cnt].astype(np.int64), s64[lo:lo + cnt], axes=(0, 0)) + (1 << 21)
            out[xx] = np.clip(acc >> 22, 0, 255)
        return out

    if out_w != W:
        img = one_pass(img.transpose(1, 0, 2), W, out_w).transpose(1, 0, 2)
    if out_h != H:
        img = one_pass(img, H, out_h)
    return np.ascontiguousarray(img)


def thumbnail_u8(img, imsize):
    """``Image.thumbnail((imsize, imsize), LANCZOS)`` of a uint8 ``[H,W,3]`` array for the cases the GPU path takes
    (no integer pre-reduction, i.e. less than 4x down; no 100:1 strips): :func:`thumbnail_size` + :func:`resample_u8`."""
    size = thumbnail_size(img.shape[1], img.shape[0], imsize)
    return np.asarray(img).copy() if size is None else resample_u8(img, size[0], size[1])


# ---- libjpeg's decompression after the entropy decoder, restated (third-party arithmetic the reference calls through
# Pillow: datahelpers.py:24-31 -> Image.open().convert('RGB') -> libjpeg-turbo with its defaults; jidctint.c
# jpeg_idct_islow, jdsample.c h2v1 / h2v2 fancy upsampling, jdcolor.c ycc_rgb_convert).  Pinned against Pillow's own
# decode of the same files in tests/test_oracle_golden.py. ------------------------------------------------------------------

def _jpeg_idct8(v, shift):
    """One 1-D pass of jpeg_idct_islow over axis 0 of an int32 array ``[8, ...]``, descaled by ``shift``."""
    I = np.int32
    z2, z3 = v[2], v[6]
    z1 = (z2 + z3) * I(4433)
    tmp2 = z1 + z3 * I(-15137)
    tmp3 = z1 + z2 * I(6270)
    z2, z3 = v[0], v[4]
    tmp0 = (z2 + z3) << 13
    tmp1 = (z2 - z3) << 13
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    tmp0, tmp1, tmp2, tmp3 = v[7], v[5], v[3], v[1]
    z1, z2, z3, z4 = tmp0 + tmp3, tmp1 + tmp2, tmp0 + tmp2, tmp1 + tmp3
    z5 = (z3 + z4) * I(9633)
    tmp0, tmp1, tmp2, tmp3 = tmp0 * I(2446), tmp1 * I(16819), tmp2 * I(25172), tmp3 * I(12299)
    z1, z2, z3, z4 = z1 * I(-7373), z2 * I(-20995), z3 * I(-16069) + z5, z4 * I(-3196) + z5
    tmp0, tmp1, tmp2, tmp3 = tmp0 + z1 + z3, tmp1 + z2 + z4, tmp2 + z2 + z3, tmp3 + z1 + z4
    half = I(1 << (shift - 1))
    out = [tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3]
    return np.stack([(o + half) >> shift for o in out])


def jpeg_pixels(coef, quant, info):
    """Quantised coefficients (``[nblocks,64]`` int16, natural order, component after component, blocks row by row over
    whole MCUs) + quantisation tables (``[3,64]``) -> uint8 ``[H,W,3]``, as libjpeg with its defaults.  ``info``: dict
    with width, height, ncomp, hsamp, vsamp, blocks_w, blocks_h, block_offset."""
    W, H, nc = info["width"], info["height"], info["ncomp"]
    planes = []
    for c in range(nc):
        bw, bh, off = info["blocks_w"][c], info["blocks_h"][c], info["block_offset"][c]
        blk = coef[off:off + bw * bh].astype(np.int32) * quant[c].astype(np.int32)[None, :]       # dequantise
        blk = blk.reshape(-1, 8, 8)                                                               # [n, row, col]
        ws = _jpeg_idct8(np.moveaxis(blk, 1, 0), 13 - 2)                                          # columns: over the rows
        px = _jpeg_idct8(np.moveaxis(ws, 2, 0), 13 + 2 + 3)                                       # rows: over the columns
        px = np.moveaxis(px, 0, 2)                                                                # [row, n, col] -> [row, n, col]
        # + 128 and limited to 0..255, saturating as libjpeg-turbo's SIMD IDCT does (the C one looks (x & 1023) up in a table
        # that wraps beyond +-512: the same for every sample a sound file can produce)
        val = np.clip(px + 128, 0, 255).astype(np.int32)
        val = np.moveaxis(val, 0, 1).reshape(bh, bw, 8, 8).transpose(0, 2, 1, 3).reshape(bh * 8, bw * 8)
        planes.append(val)
    lum = planes[0][:H, :W]
    if nc == 1:
        return np.repeat(lum[:, :, None], 3, axis=2).astype(np.uint8)
    hs, vs = info["hsamp"][0], info["vsamp"][0]
    dw, dh = -(-W // hs), -(-H // vs)
    chroma = []
    for pl in planes[1:]:
        pl = pl[:dh, :dw]
        if hs == 2 and vs == 2:         # h2v2 fancy: 3 * nearer row + farther row, then 3 * this column + the neighbouring one
            up = np.concatenate([pl[:1], pl[:-1]]), np.concatenate([pl[1:], pl[-1:]])
            rows = np.empty((2 * dh, dw), dtype=np.int32)
            rows[0::2], rows[1::2] = 3 * pl + up[0], 3 * pl + up[1]
            out = np.empty((2 * dh, 2 * dw), dtype=np.int32)
            left = np.concatenate([rows[:, :1], rows[:, :-1]], axis=1)
            right = np.concatenate([rows[:, 1:], rows[:, -1:]], axis=1)
            out[:, 0::2] = (3 * rows + left + 8) >> 4
            out[:, 1::2] = (3 * rows + right + 7) >> 4
            out[:, 0] = (4 * rows[:, 0] + 8) >> 4
            out[:, -1] = (4 * rows[:, -1] + 7) >> 4
            pl = out
        elif hs == 2:                   # h2v1 fancy
            out = np.empty((dh, 2 * dw), dtype=np.int32)
            left = np.concatenate([pl[:, :1], pl[:, :-1]], axis=1)
            right = np.concatenate([pl[:, 1:], pl[:, -1:]], axis=1)
            out[:, 0::2] = (3 * pl + left + 1) >> 2
            out[:, 1::2] = (3 * pl + right + 2) >> 2
            out[:, 0], out[:, -1] = pl[:, 0], pl[:, -1]
            pl = out
        chroma.append(pl[:H, :W] - 128)
    cb, cr = chroma
    r = lum + ((91881 * cr + 32768) >> 16)
    g = lum + ((-22554 * cb + 32768 - 46802 * cr) >> 16)
    b = lum + ((116130 * cb + 32768) >> 16)
    return np.clip(np.stack([r, g, b], axis=2), 0, 255).astype(np.uint8)


# ---- the paper's CLAHE pre-processing, restated.  PARITY UNPINNED: the reference calls OpenCV (transform/functional.py:24-48,
# 106-129: cv2.cvtColor(.., COLOR_RGB2LAB) on float32 RGB in [0,1], cv2.createCLAHE(clipLimit, tileGridSize).apply on the
# uint8 lightness, cv2.cvtColor(.., COLOR_LAB2RGB)); OpenCV is not in the build image and not under /root/reference, so there is
# nothing to generate golden vectors with.  What follows restates OpenCV 4's published algorithms (modules/imgproc/src/clahe.cpp:
# CLAHE_CalcLut_Body, CLAHE_Interpolation_Body, the padding rule of CLAHE_Impl::apply; color_lab.cpp: RGB2Lab_f / Lab2RGB_f with
# sRGB gamma, D65 white point) with exact transfer functions where OpenCV interpolates 1024-entry spline tables (difference
# ~1e-6 in L, far below the uint8 quantisation that follows).  It pins the DEVICE implementation to this restatement only. ----

_XYZ_FROM_RGB = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]], dtype=np.float64)
_RGB_FROM_XYZ = np.array([[3.240479, -1.53715, -0.498535], [-0.969256, 1.875991, 0.041556], [0.055648, -0.204043, 1.057311]], dtype=np.float64)
_D65 = np.array([0.950456, 1.0, 1.088754], dtype=np.float64)


def rgb_to_lab(rgb):
    """float32 RGB in [0,1] ``[H,W,3]`` -> float32 (L 0..100, a, b): cv2.cvtColor(.., COLOR_RGB2LAB) for CV_32F input
    (RGB2Lab_f: clip, sRGB linearisation, XYZ / white point, f(t) = cbrt(t) above 0.008856 else 7.787 t + 16/116)."""
    c = np.clip(rgb.astype(F32), 0, 1)
    lin = np.where(c <= F32(0.04045), c / F32(12.92), np.power((c + F32(0.055)) / F32(1.055), F32(2.4))).astype(F32)
    m = (_XYZ_FROM_RGB / _D65[:, None]).astype(F32)
    xyz = [(lin[..., 0] * m[i, 0] + lin[..., 1] * m[i, 1] + lin[..., 2] * m[i, 2]).astype(F32) for i in range(3)]
    f = [np.where(v > F32(0.008856), np.cbrt(v).astype(F32), (F32(7.787) * v + F32(16.0 / 116.0)).astype(F32)).astype(F32) for v in xyz]
    L = np.where(xyz[1] > F32(0.008856), F32(116.0) * f[1] - F32(16.0), F32(903.3) * xyz[1]).astype(F32)
    return np.stack([L, (F32(500.0) * (f[0] - f[1])).astype(F32), (F32(200.0) * (f[1] - f[2])).astype(F32)], axis=-1)


def lab_to_rgb(lab):
    """float32 Lab -> float32 RGB in [0,1]: cv2.cvtColor(.., COLOR_LAB2RGB) (Lab2RGB_f: inverse f, XYZ * white point -> linear RGB,
    clip, sRGB gamma)."""
    L, a, b = (lab[..., i].astype(F32) for i in range(3))
    lthresh, fthresh = F32(0.008856 * 903.3), F32(7.787 * 0.008856 + 16.0 / 116.0)
    fy_hi = ((L + F32(16.0)) / F32(116.0)).astype(F32)
    y = np.where(L <= lthresh, L / F32(903.3), fy_hi * fy_hi * fy_hi).astype(F32)
    fy = np.where(L <= lthresh, F32(7.787) * y + F32(16.0 / 116.0), fy_hi).astype(F32)
    fx, fz = (a / F32(500.0) + fy).astype(F32), (fy - b / F32(200.0)).astype(F32)
    x = np.where(fx <= fthresh, (fx - F32(16.0 / 116.0)) / F32(7.787), fx * fx * fx).astype(F32)
    z = np.where(fz <= fthresh, (fz - F32(16.0 / 116.0)) / F32(7.787), fz * fz * fz).astype(F32)
    m = (_RGB_FROM_XYZ * _D65[None, :]).astype(F32)
    out = []
    for i in range(3):
        lin = np.clip((x * m[i, 0] + y * m[i, 1] + z * m[i, 2]).astype(F32), 0, 1)
        out.append(np.where(lin <= F32(0.0031308), lin * F32(12.92),
                            F32(1.055) * np.power(lin, F32(1.0 / 2.4)) - F32(0.055)).astype(F32))
    return np.stack(out, axis=-1)


def _reflect101(i, n):
    if n == 1:
        return np.zeros_like(i)
    period = 2 * (n - 1)
    i = np.abs(i) % period
    return np.where(i >= n, period - i, i)


def clahe_luts(l8, clip_limit=4, grid=(8, 8)):
    """Per-tile look-up tables of cv2.createCLAHE(clip_limit, grid).apply for a uint8 plane ``[H,W]``: returns
    ``(luts uint8 [tiles_y, tiles_x, 256], (tile_h, tile_w))``.  clahe.cpp: when either side is not a multiple of the grid
    the plane is padded on the bottom by ``ty - H % ty`` and on the right by ``tx - W % tx`` (BORDER_REFLECT_101; a side that
    IS a multiple still gets a whole extra ``t`` -- OpenCV's rule); histogram per tile, clip at
    ``max(int(clip * area / 256), 1)``, the excess spread evenly with the remainder at stride ``256 / residual``, LUT =
    saturate_cast<uchar>(cumsum * 255 / area)."""
    tx, ty = int(grid[0]), int(grid[1])
    h, w = l8.shape
    if w % tx == 0 and h % ty == 0:
        ext = l8
    else:
        pb, pr = ty - h % ty, tx - w % tx
        rows = _reflect101(np.arange(h + pb), h)
        cols = _reflect101(np.arange(w + pr), w)
        ext = l8[rows][:, cols]
    th, tw = ext.shape[0] // ty, ext.shape[1] // tx
    area = th * tw
    lut_scale = F32(255.0) / F32(area)
    clip = max(int(float(clip_limit) * area / 256), 1) if clip_limit > 0 else 0
    luts = np.empty((ty, tx, 256), dtype=np.uint8)
    for j in range(ty):
        for i in range(tx):
            hist = np.bincount(ext[j * th:(j + 1) * th, i * tw:(i + 1) * tw].reshape(-1), minlength=256).astype(np.int64)
            if clip > 0:
                clipped = int(np.maximum(hist - clip, 0).sum())
                hist = np.minimum(hist, clip)
                batch, residual = clipped // 256, clipped % 256
                hist += batch
                if residual:
                    step = max(256 // residual, 1)
                    k = 0
                    while k < 256 and residual > 0:
                        hist[k] += 1
                        k += step
                        residual -= 1
            cum = np.cumsum(hist).astype(F32) * lut_scale
            luts[j, i] = np.clip(np.rint(cum), 0, 255).astype(np.uint8)
    return luts, (th, tw)


def clahe_apply(l8, luts, tile):
    """CLAHE_Interpolation_Body: bilinear blend of the four neighbouring tiles' LUTs at every pixel (tile centres at
    ``(t + 0.5) * tile``; clamped at the border), ``saturate_cast<uchar>`` of the float32 result."""
    ty, tx = luts.shape[:2]
    th, tw = tile
    h, w = l8.shape
    xf = np.arange(w, dtype=F32) * (F32(1.0) / F32(tw)) - F32(0.5)
    yf = np.arange(h, dtype=F32) * (F32(1.0) / F32(th)) - F32(0.5)
    x1, y1 = np.floor(xf).astype(np.int64), np.floor(yf).astype(np.int64)
    xa, ya = (xf - x1.astype(F32)).astype(F32), (yf - y1.astype(F32)).astype(F32)
    x2, y2 = np.minimum(x1 + 1, tx - 1), np.minimum(y1 + 1, ty - 1)
    x1, y1 = np.maximum(x1, 0), np.maximum(y1, 0)
    v = l8.astype(np.int64)
    l11 = luts[y1[:, None], x1[None, :], v].astype(F32)
    l12 = luts[y1[:, None], x2[None, :], v].astype(F32)
    l21 = luts[y2[:, None], x1[None, :], v].astype(F32)
    l22 = luts[y2[:, None], x2[None, :], v].astype(F32)
    xa1, ya1 = (F32(1.0) - xa)[None, :], (F32(1.0) - ya)[:, None]
    res = (l11 * xa1 + l12 * xa[None, :]) * ya1 + (l21 * xa1 + l22 * xa[None, :]) * ya[:, None]
    return np.clip(np.rint(res.astype(F32)), 0, 255).astype(np.uint8)


def apply_clahe_rgb(rgb_u8, clip_limit=4, grid=8):
    """``ApplyClahe`` of the scenarios (photometric_transforms.py:28-36 -> functional.ImageClahe.apply, colorspace "lab") on what
    ``pil2np`` makes of a uint8 RGB image: returns ``(float32 RGB [H,W,3] in [0,1], lightness uint8 [H,W] before CLAHE)``."""
    grid = (int(grid), int(grid)) if not isinstance(grid, tuple) else grid
    img = rgb_u8.astype(F32) / F32(255.0)
    lab = rgb_to_lab(img)
    spc = ((lab + np.array([0, 128, 128], dtype=F32)) / np.array([100.0, 255.0, 255.0], dtype=F32)).astype(F32)
    l8 = (spc[..., 0] * F32(255.0)).astype(np.uint8)                       # (chan*255).astype(np.uint8): truncation
    luts, tile = clahe_luts(l8, int(clip_limit), grid)
    spc[..., 0] = clahe_apply(l8, luts, tile).astype(F32) / F32(255.0)
    back = (spc * np.array([100.0, 255.0, 255.0], dtype=F32)).astype(F32) - np.array([0, 128, 128], dtype=F32)
    return lab_to_rgb(back.astype(F32)), l8


def nanmean_metric(per_query):
    """The number eval.py prints: nan-filtered mean of the per-query rows
    (``mdir/tools/eventprocessor.py:101-115``)."""
    v = np.asarray(per_query, dtype=np.float64)
    return float(np.mean(v[~np.isnan(v)]))


# --------------------------------------------------------------------------
# synthetic workloads shared by tests and bench
# --------------------------------------------------------------------------

def synth_ranking_problem(n, q=70, d=2048, seed=0, noise=0.05, dtype=F32):
    """Database ``[D,N]`` + queries ``[D,Q]`` in the REFERENCE layout.

    Rows i.i.d. N(0,1), L2-normalised; queries = ``q`` distinct database rows
    plus ``noise`` * N(0,1), re-normalised.  Generated in row blocks so the 1 M
    case never holds more than one float64 block at a time.
    """
    rng = np.random.default_rng(seed)
    vecs = np.empty((d, n), dtype=dtype)
    block = 65536
    for s in range(0, n, block):
        e = min(n, s + block)
        blk = rng.standard_normal((e - s, d), dtype=np.float32)
        blk /= np.linalg.norm(blk, axis=1, keepdims=True)
        vecs[:, s:e] = blk.T
    qid = rng.choice(n, size=q, replace=False)
    qv = vecs[:, qid].T.astype(np.float32) + noise * rng.standard_normal((q, d), dtype=np.float32)
    qv /= np.linalg.norm(qv, axis=1, keepdims=True)
    return vecs, np.ascontiguousarray(qv.T.astype(dtype)), qid


def synth_gnd(nq, n_labelled, seed=1, easy=5, hard=10, junk=5):
    """rOxford-shaped ground truth: disjoint random ids from the first
    ``n_labelled`` database rows per query."""
    rng = np.random.default_rng(seed)
    gnd = []
    for _ in range(nq):
        ids = rng.choice(n_labelled, size=easy + hard + junk, replace=False)
        gnd.append({"easy": np.sort(ids[:easy]), "hard": np.sort(ids[easy:easy + hard]),
                    "junk": np.sort(ids[easy + hard:]), "bbx": None})
    return gnd
