"""Hard-negative mining search.

The device-side part of ``TuplesDataset.create_epoch_tuples``
(``mdir/external/cirtorch/datasets/traindataset.py:239-270``):

    scores = torch.mm(poolvecs.t(), qvecs)
    scores, ranks = torch.sort(scores, dim=0, descending=True)
    for q: walk ranks[:, q] until `nnum` pool images from distinct clusters (none from the
           query's own cluster) are found; record their l2 distance to the query

is the same similarity + ranking pair as evaluation, so it reuses ``mdx_scores`` and
``mdx_topk`` unchanged.  The reference reads ``ranks[r, q]`` one element at a time from the
GPU; here only a short prefix of every ranking is materialised (``mdx_topk``) and copied
once; the prefix is doubled until every query has its negatives.
"""
import numpy as np
import torch

from . import ops


def search_hard_negatives(qvecs, poolvecs, idxs2images, clusters, qidxs, nnum, prefix=None):
    """Select ``nnum`` hard negatives per query.

    qvecs ``[D,Q]`` / poolvecs ``[D,P]``: device descriptors in the reference layout.
    idxs2images ``[P]``: image id of every pool column; ``clusters[image]``: cluster id;
    ``qidxs[q]``: image id of query q.  Returns ``(nidxs, ndist)``: per-query lists of image
    ids (as the reference's ``self.nidxs``) and the flat list of l2 distances
    (``sqrt(sum((q - p + 1e-6)^2))``, the reference's statistic) in selection order.
    """
    idxs2images = np.asarray(idxs2images).reshape(-1)
    P, Q = poolvecs.shape[1], qvecs.shape[1]
    if nnum == 0:
        return [[] for _ in range(Q)], []
    index = ops.DescriptorIndex(poolvecs, "DN")
    scores = index.scores(qvecs, "DN")                      # [Q,P] = (poolvecs.t() @ qvecs).t()
    k = min(P, prefix or max(64, 8 * nnum))
    while True:
        ids, _ = ops.topk(scores, k)                        # first k rows of the descending sort
        ranks = ids.cpu().numpy()
        nidxs, picked, done = [], [], True
        for q in range(Q):
            seen = [clusters[qidxs[q]]]
            sel, cols = [], []
            for r in range(k):
                potential = int(idxs2images[ranks[q, r]])
                if clusters[potential] not in seen:
                    sel.append(potential)
                    cols.append(int(ranks[q, r]))
                    seen.append(clusters[potential])
                    if len(sel) == nnum:
                        break
            if len(sel) < nnum and k < P:
                done = False
                break
            if len(sel) < nnum:
                raise IndexError("pool too small: query %d found %d of %d negatives" % (q, len(sel), nnum))
            nidxs.append(sel)
            picked.append(cols)
        if done:
            break
        k = min(P, 2 * k)
    index.close()
    cols = torch.as_tensor(np.asarray(picked, dtype=np.int64), device=poolvecs.device)      # [Q,nnum]
    diff = qvecs.t().unsqueeze(1) - poolvecs.t()[cols] + 1e-6                               # [Q,nnum,D]
    ndist = diff.pow(2).sum(dim=2).sqrt().reshape(-1).cpu().tolist()
    return nidxs, ndist
