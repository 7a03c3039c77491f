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


def whiten(params, data, device="cuda"):
    """Apply a pre-computed whitening to ``values [N,D]``: returns ``(metadata, names, [N,d])``."""
    dimensions = params.pop("dimensions", None) or None
    assert not params, params.keys()
    whitening, names, values = data
    assert len(names) == len(values)
    time0 = time.time()
    whitened = W.whitenapply(values.T, whitening["m"], whitening["P"], dimensions, device=device)
    timing = time.time() - time0
    metadata = {"timings": {"whitening_apply": round(timing, 2)}, "resource_usage": resource_usage()}
    return metadata, names, whitened.T


def learn_lw_whitening(params, data, device="cuda"):
    """Learned whitening from (query, positive) name pairs.  If the pair covariance is not positive
    definite the reference retries on a shrinking random subset of the pairs (up to 100 trials, at
    most 95 % excluded); ``whitenlearn``'s own ``cholesky`` already regularises the diagonal, so
    like there the retry loop only triggers on a ``LinAlgError`` that escapes it."""
    assert not params
    names, values, queries, positives = data
    assert len(names) == len(values)
    assert len(queries) == len(positives)
    values = values.astype(np.float64).T
    name_index = {x: i for i, x in enumerate(names)}
    qidxs = np.array([name_index[x] for x in queries])
    pidxs = np.array([name_index[x] for x in positives])

    time0 = time.time()
    max_trials, max_excluded, trial = 100, 0.95, 0
    while True:
        try:
            if trial == 0:
                qwhit, pwhit = qidxs, pidxs
            else:
                keep = int(len(qidxs) * (1 - trial / max_trials * max_excluded))
                idxs = np.random.permutation(len(qidxs))[:keep]
                print("Using subset of queries (%s/%s) trial %s" % (len(idxs), len(qidxs), trial), file=sys.stderr)
                qwhit, pwhit = qidxs[idxs], pidxs[idxs]
            whit_m, whit_p = W.whitenlearn(values, qwhit, pwhit, device=device)
            break
        except np.linalg.LinAlgError as err:
            if str(err) != "Matrix is not positive definite" or trial >= max_trials - 1:
                raise
            trial += 1
    timing = time.time() - time0
    metadata = {"stats": {"failed_times": trial, "vectors_used": round(len(qwhit) / float(len(qidxs)), 2),
                          "vectors_total": len(qidxs)},
                "timings": {"whitening_learn": round(timing, 2)}, "resource_usage": resource_usage()}
    return metadata, {"m": whit_m, "P": whit_p}


def learn_pca_whitening(params, data, device="cuda"):
    shrink = params.pop("shrink", None) or None
    assert not params
    values, = data
    values = values.astype(np.float64).T
    time0 = time.time()
    whit_m, whit_p = W.pcawhitenlearn(values, shrink, device=device)
    timing = time.time() - time0
    metadata = {"timings": {"whitening_learn": round(timing, 2)}, "resource_usage": resource_usage()}
    return metadata, {"m": whit_m, "P": whit_p}


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
    value = np.concatenate(data, axis=1)
    if dimensions:
        time0 = time.time()
        value = value - np.mean(value)
        eigval, eigvec = np.linalg.eig(W.gram(value.T, device).astype(value.dtype))     # value.T @ value, [D,D]
        vecs = eigvec[:, np.argsort(eigval)[-dimensions:]]
        value = value.dot(vecs.dot(vecs.T))
        timing = time.time() - time0
        metadata = {"timings": {"pca_compute": round(timing, 2)}, "resource_usage": resource_usage()}
    else:
        metadata = {}
    value = value / np.expand_dims(np.linalg.norm(value, axis=1), axis=1)
    return metadata, value
