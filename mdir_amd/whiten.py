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
