"""Whitening stages of the pipeline -- drop-ins for
``mdir/stages/whiten.py``: ``whiten`` (:10-24), ``learn_lw_whitening`` (:27-69),
``learn_pca_whitening`` (:72-87), ``paste_pca_normalize`` (:90-118).

Same ``(params, data) -> (metadata, ...)`` protocol, same metadata keys.  The matrix products run on
the GPU through ``mdir_amd.whiten`` (``mdx_scores``); the small dense factorisations stay on the
host as in the reference.
"""
import sys
import time

import numpy as np

from . import whiten as W


def resource_usage():
    """``stats.ResourceUsage().take_current_stats().get_resources()`` (mdir/tools/stats.py:82-96):
    virtual memory of this process and the torch allocator's device memory.  The per-process
    figure the reference reads from nvidia-smi has no counterpart here and is reported as None."""
    import torch
    out = {}
    try:
        import psutil
        out["ram_memory_gib"] = round(psutil.Process().memory_info().vms / 2 ** 30, 3)
    except ImportError:
        out["ram_memory_gib"] = None
    if torch.cuda.is_available():
        out["gpu"] = {"memory_nvidia_gib": None,
                      "memory_torch_gib": round(torch.cuda.memory_allocated() / 2 ** 30, 3)}
    return out


def _metadata(label, started, **sections):
    """``{"timings": {label: seconds}, "resource_usage": ..., **sections}`` as every stage reports it."""
    out = dict(sections)
    out["timings"] = {label: round(time.time() - started, 2)}
    out["resource_usage"] = resource_usage()
    return out


def whiten(params, data, device="cuda"):
    """Apply a pre-computed whitening to ``values [N,D]``: returns ``(metadata, names, [N,d])``."""
    dimensions = params.pop("dimensions", None) or None
    assert not params, params.keys()
    whitening, names, values = data
    assert len(names) == len(values)
    started = time.time()
    projected = W.whitenapply(values.T, whitening["m"], whitening["P"], dimensions, device=device)
    return _metadata("whitening_apply", started), names, projected.T


def _pair_subset(qidxs, pidxs, trial, max_trials, max_excluded):
    """Trial 0 uses every (query, positive) pair; trial t keeps a random ``1 - t/max_trials * max_excluded``
    share of them (the reference's answer to a covariance that is not positive definite)."""
    if trial == 0:
        return qidxs, pidxs
    keep = int(len(qidxs) * (1 - trial / max_trials * max_excluded))
    chosen = np.random.permutation(len(qidxs))[:keep]
    print("Using subset of queries (%s/%s) trial %s" % (len(chosen), len(qidxs), trial), file=sys.stderr)
    return qidxs[chosen], pidxs[chosen]


def learn_lw_whitening(params, data, device="cuda"):
    """Learned whitening from (query, positive) name pairs.  ``whitenlearn``'s own ``cholesky``
    already regularises the diagonal, so like in the reference the shrinking-subset retry (up to 100
    trials, at most 95 % of the pairs excluded) only triggers on a ``LinAlgError`` that escapes it."""
    assert not params
    names, values, queries, positives = data
    assert len(names) == len(values)
    assert len(queries) == len(positives)
    row_of = {name: row for row, name in enumerate(names)}
    qidxs = np.array([row_of[name] for name in queries])
    pidxs = np.array([row_of[name] for name in positives])
    columns = values.astype(np.float64).T

    started, max_trials = time.time(), 100
    for trial in range(max_trials):
        qsel, psel = _pair_subset(qidxs, pidxs, trial, max_trials, max_excluded=0.95)
        try:
            mean, proj = W.whitenlearn(columns, qsel, psel, device=device)
            break
        except np.linalg.LinAlgError as err:
            if str(err) != "Matrix is not positive definite" or trial == max_trials - 1:
                raise
    stats = {"failed_times": trial, "vectors_used": round(len(qsel) / float(len(qidxs)), 2),
             "vectors_total": len(qidxs)}
    return _metadata("whitening_learn", started, stats=stats), {"m": mean, "P": proj}


def learn_pca_whitening(params, data, device="cuda"):
    shrink = params.pop("shrink", None) or None
    assert not params
    values, = data
    started = time.time()
    mean, proj = W.pcawhitenlearn(values.astype(np.float64).T, shrink, device=device)
    return _metadata("whitening_learn", started), {"m": mean, "P": proj}


def paste_pca_normalize(params, data, device="cuda"):
    """Concatenate descriptor matrices ``[N,D_i]`` side by side, optionally keep the subspace of
    the ``dimensions`` largest principal directions (projected back to the full width, exactly the
    reference's ``value.dot(vecs.dot(vecs.T))``; NB it subtracts the scalar mean of ALL entries),
    then L2-normalise every row."""
    dimensions = params.pop("dimensions") or None
    assert not params
    assert len(set(len(x) for x in data)) == 1
    if data[0].shape == (0,):
        return {}, data[0]
    pasted = np.concatenate(data, axis=1)
    metadata = {}
    if dimensions:
        started = time.time()
        pasted = pasted - np.mean(pasted)
        eigval, eigvec = np.linalg.eig(W.gram(pasted.T, device).astype(pasted.dtype))     # pasted.T @ pasted, [D,D]
        top = eigvec[:, np.argsort(eigval)[-dimensions:]]
        pasted = pasted.dot(top.dot(top.T))
        metadata = _metadata("pca_compute", started)
    return metadata, pasted / np.linalg.norm(pasted, axis=1, keepdims=True)
