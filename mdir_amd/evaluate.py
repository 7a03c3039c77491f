"""Mean average precision -- the cirtorch evaluation API, position-based.

Drop-in for ``mdir/external/cirtorch/utils/evaluate.py`` (``compute_ap`` :3-37,
``compute_map`` :39-111, ``compute_map_and_print`` :114-152, the mdir-patched
variant that returns dictionaries).  Same signatures, same return values, same
float64 accumulation order, so results are bit-identical to the reference's.

The per-query work is expressed on rank POSITIONS of the labelled ids.  Positions
come either from a materialised ranking (``ranks[:, q]``, numpy or torch, any
device -- the reference call style) or straight from the scores through the HIP
counting kernel (``positions_from_scores`` -> ``mdx_rank_of``), which yields the
same numbers without sorting a million-row database.
"""
import numpy as np

try:  # torch is only needed when rankings live on the GPU
    import torch
except ImportError:  # pragma: no cover
    torch = None


def compute_ap(ranks, nres):
    """AP from ascending zero-based ranks of the positives (evaluate.py:3-37)."""
    ap = 0.0
    recall_step = 1.0 / nres
    for j, rank in enumerate(ranks):
        rank = int(rank)
        p0 = 1.0 if rank == 0 else float(j) / rank
        p1 = float(j + 1) / (rank + 1)
        ap += (p0 + p1) * recall_step / 2.0
    return ap


def _ap_and_precisions(pos, junk, nok, kappas):
    """One query: positions (ascending) of positives / junk -> (ap, P@kappas).
    evaluate.py:85-106: positives move up by the junk ranked before them."""
    pos = np.asarray(pos, dtype=np.int64)
    junk = np.asarray(junk, dtype=np.int64)
    if len(junk):
        pos = pos - np.searchsorted(junk, pos, side="left")
    ap = compute_ap(pos, nok)
    prs = np.zeros(len(kappas))
    if len(kappas):
        pos1 = pos + 1
        top = int(pos1.max())
        for j, kappa in enumerate(kappas):
            kq = min(top, kappa)
            prs[j] = (pos1 <= kq).sum() / kq
    return ap, prs


def _column_positions(ranks, q, ids):
    """Ascending positions at which ``ids`` occur in column q of ``ranks`` [N,Q]."""
    if len(ids) == 0:
        return np.empty(0, dtype=np.int64)
    if torch is not None and isinstance(ranks, torch.Tensor):
        col = ranks[:, q]
        want = torch.as_tensor(np.asarray(ids, dtype=np.int64), device=col.device)
        return torch.nonzero(torch.isin(col, want)).reshape(-1).cpu().numpy()
    return np.nonzero(np.isin(ranks[:, q], np.asarray(ids)))[0]


def map_from_positions(pos_lists, junk_lists, nok, kappas=()):
    """mAP from per-query position arrays (unsorted ok; ``nok[q]`` positives).

    Same outputs as :func:`compute_map`: ``(map, aps, pr, prs)``."""
    nq = len(pos_lists)
    aps = np.zeros(nq)
    pr = np.zeros(len(kappas))
    prs = np.zeros((nq, len(kappas)))
    total, nempty = 0.0, 0
    for q in range(nq):
        if nok[q] == 0:
            aps[q] = float("nan")
            prs[q, :] = float("nan")
            nempty += 1
            continue
        ap, prs[q, :] = _ap_and_precisions(np.sort(pos_lists[q]), np.sort(junk_lists[q]), nok[q], kappas)
        aps[q] = ap
        total += ap
        pr = pr + prs[q, :]
    return total / (nq - nempty), aps, pr / (nq - nempty), prs


class _Positions:
    """Rank positions of every labelled id of every query, fetched ONCE (one kernel pass + one copy to the host) and then
    looked up per protocol level and list.  ``fetch(id_lists)`` -> one int64 array per query aligned with the sorted unique
    non-negative ids it was given; -1 = the id has no position (not a database row / not in the ranking)."""

    def __init__(self, gnd, fetch):
        self.ids = []
        for g in gnd:
            parts = [np.asarray(g[k], dtype=np.int64).reshape(-1) for k in ("ok", "easy", "hard", "junk") if k in g]
            ids = np.unique(np.concatenate(parts)) if parts else np.empty(0, dtype=np.int64)
            self.ids.append(ids[ids >= 0])
        self.pos = fetch(self.ids)

    def of(self, q, ids):
        """Ascending positions of the SET ``ids`` in query q (what ``np.arange(N)[np.in1d(ranks[:, q], ids)]`` yields)."""
        ids = np.unique(np.asarray(ids, dtype=np.int64).reshape(-1))
        have = self.ids[q]
        if len(ids) == 0 or len(have) == 0:
            return np.empty(0, dtype=np.int64)
        at = np.minimum(np.searchsorted(have, ids), len(have) - 1)
        found = self.pos[q][at[have[at] == ids]]
        return np.sort(found[found >= 0])

    def _flat(self):
        """All queries' (id, position) pairs as flat arrays, ordered by (query, position) -- built on first use."""
        if getattr(self, "_f", None) is None:
            nq = len(self.ids)
            lens = np.array([len(x) for x in self.ids], dtype=np.int64)
            ids = np.concatenate(self.ids) if nq else np.empty(0, dtype=np.int64)
            pos = np.concatenate(self.pos) if nq else np.empty(0, dtype=np.int64)
            qq = np.repeat(np.arange(nq, dtype=np.int64), lens)
            m = int(ids.max()) + 1 if len(ids) else 1
            key = qq * m + ids                              # ascending: queries ascending, ids sorted inside a query
            order = np.lexsort((pos, qq))
            q_s = qq[order]
            starts = np.searchsorted(q_s, np.arange(nq + 1))
            self._f = (m, key, order, q_s, pos[order], pos[order] >= 0, starts)
        return self._f

    def _mark(self, lists):
        """Boolean membership of the flat pairs in the per-query id SETS ``lists`` (ids that were never fetched mark nothing)."""
        m, key = self._flat()[:2]
        lens = np.array([len(x) for x in lists], dtype=np.int64)
        flat = np.concatenate(lists) if len(lists) else np.empty(0, dtype=np.int64)
        qq = np.repeat(np.arange(len(lists), dtype=np.int64), lens)
        keep = (flat >= 0) & (flat < m)
        want = qq[keep] * m + flat[keep]
        mask = np.zeros(len(key), dtype=bool)
        if len(key) and len(want):
            at = np.minimum(np.searchsorted(key, want), len(key) - 1)
            mask[at[key[at] == want]] = True
        return mask

    def map(self, gnd, kappas=()):
        """:func:`map_from_positions` for ``gnd`` (``ok`` / optional ``junk`` per query) on the fetched positions, with the
        set lookups, the junk shift and the AP terms computed for all queries at once.  Every float64 operation of
        ``compute_ap`` / ``compute_map`` (evaluate.py:3-37, :85-106) is kept, in its order: the terms are the same
        expressions elementwise, each query's AP is the left-to-right sum of its terms (``np.cumsum``), the means add up
        in query order -- bit-identical to the per-query statement (tests/test_evaluate.py)."""
        empty = np.empty(0, dtype=np.int64)
        oks = [np.asarray(g["ok"], dtype=np.int64).reshape(-1) for g in gnd]
        nok = [len(o) for o in oks]
        junks = [np.asarray(g["junk"], dtype=np.int64).reshape(-1) if ("junk" in g and n) else empty for g, n in zip(gnd, nok)]
        order, valid_s = self._flat()[2], self._flat()[5]
        return self._map_of_masks(self._mark(oks)[order] & valid_s, self._mark(junks)[order] & valid_s, nok, kappas)

    def map_levels(self, gnd, ok_keys, junk_keys, kappas=()):
        """:meth:`map` for the revisited protocol's levels (evaluate.py:123-147: ``ok`` = the concatenation of ``gnd[q][k]``
        for k in ``ok_keys``, ``junk`` likewise): the membership of every fetched id in ``easy`` / ``hard`` / ``junk`` is
        looked up once and the three levels combine the masks."""
        if getattr(self, "_key_masks", None) is None:
            self._key_masks = {}
        order, valid_s = self._flat()[2], self._flat()[5]
        for k in tuple(ok_keys) + tuple(junk_keys):
            if k not in self._key_masks:
                self._key_masks[k] = self._mark([np.asarray(g[k], dtype=np.int64).reshape(-1) for g in gnd])[order] & valid_s
        nok = [sum(len(np.asarray(g[k]).reshape(-1)) for k in ok_keys) for g in gnd]
        ok_s = np.logical_or.reduce([self._key_masks[k] for k in ok_keys])
        junk_s = np.logical_or.reduce([self._key_masks[k] for k in junk_keys])
        return self._map_of_masks(ok_s, junk_s, nok, kappas)

    def _map_of_masks(self, ok_s, junk_s, nok, kappas):
        """The arithmetic of :meth:`map` on membership masks over the (query, position)-ordered pairs (junk of a query
        without positives is never looked at: the query is skipped)."""
        nq = len(nok)
        _, _, order, q_s, pos_s, valid_s, starts = self._flat()
        jcum = np.cumsum(junk_s)
        before = jcum - junk_s                                                  # junk strictly earlier in the whole array
        base = np.where(starts[:-1] > 0, jcum[np.maximum(starts[:-1], 1) - 1], 0) if len(jcum) else np.zeros(nq, dtype=np.int64)
        sel = np.nonzero(ok_s)[0]
        q_o = q_s[sel]
        adj = pos_s[sel] - (before[sel] - base[q_o])                            # positives move up by the junk ranked before them
        counts = np.bincount(q_o, minlength=nq) if len(q_o) else np.zeros(nq, dtype=np.int64)
        first = np.concatenate([[0], np.cumsum(counts)])
        j = np.arange(len(sel), dtype=np.int64) - first[q_o]
        jf, af = j.astype(np.float64), adj.astype(np.float64)
        p0 = np.ones(len(sel))
        nz = adj != 0
        p0[nz] = jf[nz] / af[nz]
        p1 = (jf + 1.0) / (af + 1.0)
        with np.errstate(divide="ignore"):
            step = 1.0 / np.asarray(nok, dtype=np.float64)
        terms = (p0 + p1) * step[q_o] / 2.0
        aps = np.zeros(nq)
        pr = np.zeros(len(kappas))
        prs = np.zeros((nq, len(kappas)))
        live = [q for q in range(nq) if nok[q]]
        if len(kappas) and live:
            # P@k for all queries at once (evaluate.py:102-106): kq = min(max(pos) + 1, kappa), (pos + 1 <= kq).sum() / kq
            if any(counts[q] == 0 for q in live):
                raise ValueError("zero-size array to reduction operation maximum which has no identity")   # max(pos) of no positive
            pos1 = adj + 1
            seg = first[:-1][counts > 0]                                       # reduceat: only non-empty segments
            top = np.zeros(nq, dtype=np.int64)
            top[counts > 0] = np.maximum.reduceat(pos1, seg)
            for i, kappa in enumerate(kappas):
                kq = np.minimum(top, kappa)
                hits = np.zeros(nq, dtype=np.int64)
                hits[counts > 0] = np.add.reduceat((pos1 <= kq[q_o]).astype(np.int64), seg)
                with np.errstate(divide="ignore", invalid="ignore"):
                    prs[:, i] = hits / kq
        total, nempty = 0.0, 0
        term_list = terms.tolist()
        for q in range(nq):
            if nok[q] == 0:
                aps[q] = float("nan")
                prs[q, :] = float("nan")
                nempty += 1
                continue
            ap = 0.0
            for t in term_list[first[q]:first[q + 1]]:                          # left to right, as compute_ap adds them
                ap += t
            aps[q] = ap
            total += ap
            pr = pr + prs[q, :]
        return total / (nq - nempty), aps, pr / (nq - nempty), prs


def _is_device_tensor(x):
    return torch is not None and isinstance(x, torch.Tensor) and x.is_cuda


def positions_in_ranking(ranks, id_lists):
    """Positions of ids inside a DEVICE ranking ``[N,Q]`` (column q = query q; the transposed view of the ``[Q,N]`` matrix
    mdx_rank_full writes costs nothing) through ``mdx_rank_positions``: one pass over the ranking for all queries and
    lists.  One int64 array per query aligned with ``id_lists[q]`` (non-negative, unique), -1 where absent."""
    from . import ops
    rows = ranks.t()
    if rows.shape[1] != 1 and rows.stride(1) != 1:      # (a one-row database: [Q,1], whatever the stride of its only column)
        rows = rows.contiguous()
    pos, off = ops.rank_positions(rows, id_lists)
    pos = pos.cpu().numpy()
    return [pos[off[q]:off[q + 1]] for q in range(len(id_lists))]


def compute_map(ranks, gnd, kappas=[], _positions=None):
    """mAP of a ranking (evaluate.py:39-111).

    ``ranks``: ``[N,Q]`` ids best-to-worst per column -- numpy array or torch tensor
    (CPU or GPU; a transposed view of the ``[Q,N]`` matrix mdx_rank_full writes is
    fine).  ``gnd[q]``: ``ok`` ids, optional ``junk`` ids.  Queries with no
    positives are NaN and excluded from the mean.  A ranking on the GPU is searched by
    ``mdx_rank_positions`` (all queries and lists in one pass; each id is taken to occur once per
    column, as in any ranking), a host ranking with ``np.isin`` column by column.
    Returns ``(map, aps, pr, prs)``.
    """
    nq = len(gnd)
    if _positions is None and _is_device_tensor(ranks):
        _positions = _Positions(gnd, lambda lists: positions_in_ranking(ranks, lists))
    if _positions is not None:
        return _positions.map(gnd, kappas)
    pos_lists, junk_lists, nok = [], [], []
    for q in range(nq):
        ok = np.asarray(gnd[q]["ok"])
        nok.append(ok.shape[0])
        if ok.shape[0] == 0:
            pos_lists.append(np.empty(0, dtype=np.int64))
            junk_lists.append(np.empty(0, dtype=np.int64))
            continue
        junk = np.asarray(gnd[q]["junk"]) if "junk" in gnd[q] else np.empty(0)
        pos_lists.append(_column_positions(ranks, q, ok))
        junk_lists.append(_column_positions(ranks, q, junk))
    return map_from_positions(pos_lists, junk_lists, nok, kappas)


def positions_from_scores(scores, id_lists):
    """Rank positions of labelled ids from a device score matrix ``[Q,N]`` through
    the HIP counting kernel -- no ranking is materialised.  Returns a list of numpy
    arrays aligned with ``id_lists``."""
    from . import ops
    pos, _, off = ops.rank_of(scores, id_lists)
    pos = pos.cpu().numpy()
    return [pos[off[q]:off[q + 1]] for q in range(len(id_lists))]


def _positions_of_rows(n, gnd, positions_of):
    """:class:`_Positions` from a function that ranks database rows: ``positions_of(id_lists)`` -> one array of positions
    per query for ids in ``[0, n)``; ids outside ``[0, n)`` are not database rows and have no position (``np.in1d`` finds
    nothing for them, evaluate.py:80-81)."""
    def fetch(lists):
        inside = [ids[ids < n] for ids in lists]
        got = positions_of(inside)
        out = []
        for ids, sub, p in zip(lists, inside, got):
            full = np.full(len(ids), -1, dtype=np.int64)
            full[:len(sub)] = p                      # ids are sorted: those < n come first
            out.append(full)
        return out
    return _Positions(gnd, fetch)


def _score_positions(scores, gnd):
    """:class:`_Positions` from device scores ``[Q,N]`` (the counting kernel, no ranking)."""
    return _positions_of_rows(scores.shape[1], gnd, lambda lists: positions_from_scores(scores, lists))


def labelled_lists(gnd, n):
    """Per query ``(ok ids, junk ids, nok)`` for the sort-free route, with the meaning ``np.in1d`` gives them in
    the reference (evaluate.py:80-81): an id listed twice occupies ONE rank position and an id that is not a
    database row occupies none -- so the lists are made unique and restricted to ``[0, n)`` -- while the
    normaliser stays ``len(ok)`` AS GIVEN (evaluate.py:101 passes ``len(qgnd)``).  Queries without positives
    carry no junk either (they are skipped before junk is looked at, evaluate.py:68-72)."""
    oks, junks, nok = [], [], []
    for g in gnd:
        ok = np.asarray(g["ok"], dtype=np.int64).reshape(-1)
        junk = np.asarray(g["junk"], dtype=np.int64).reshape(-1) if "junk" in g and len(ok) else np.empty(0, dtype=np.int64)
        nok.append(len(ok))
        oks.append(np.unique(ok[(ok >= 0) & (ok < n)]))
        junks.append(np.unique(junk[(junk >= 0) & (junk < n)]))
    return oks, junks, nok


def compute_map_from_scores(scores, gnd, kappas=[], _positions=None):
    """:func:`compute_map` on scores ``[Q,N]`` (device) instead of a ranking (ids as ``labelled_lists`` reads them)."""
    if _positions is None:
        _positions = _score_positions(scores, gnd)
    return _positions.map(gnd, kappas)


def _protocol_gnd(gnd, ok_keys, junk_keys):
    return [{"ok": np.concatenate([g[k] for k in ok_keys]),
             "junk": np.concatenate([g[k] for k in junk_keys])} for g in gnd]


_LEVELS = (("easy", ("easy",), ("junk", "hard")),
           ("medium", ("easy", "hard"), ("junk",)),
           ("hard", ("hard",), ("junk", "easy")))


def _evaluate(dataset, gnd, kappas, one_map, positions=None):
    """Protocol dispatch shared by the ranking- and the scores-based entry points (``positions``: a :class:`_Positions` of
    this ``gnd`` -- the revisited protocol's levels then share its membership masks)."""
    if "ok" in gnd[0]:  # old protocol (evaluate.py:117-120)
        m, aps, _, _ = one_map(gnd, [])
        print(">> {}: mAP {:.2f}".format(dataset, np.around(m * 100, decimals=2)))
        return {"map": m}, {"ap": aps}
    if dataset.startswith("roxford5k") or dataset.startswith("rparis6k"):  # evaluate.py:123-152
        avg, per, mpr = {}, {}, {}
        for level, ok_keys, junk_keys in _LEVELS:
            if positions is not None:
                m, aps, pr, _ = positions.map_levels(gnd, ok_keys, junk_keys, list(kappas))
            else:
                m, aps, pr, _ = one_map(_protocol_gnd(gnd, ok_keys, junk_keys), list(kappas))
            avg["map_" + level], per["ap_" + level], mpr[level] = m, aps, pr
        r = lambda v: np.around(v * 100, decimals=2)
        print(">> {}: mAP E: {}, M: {}, H: {}".format(dataset, r(avg["map_easy"]), r(avg["map_medium"]),
                                                      r(avg["map_hard"])))
        print(">> {}: mP@k{} E: {}, M: {}, H: {}".format(dataset, list(kappas), r(mpr["easy"]), r(mpr["medium"]),
                                                         r(mpr["hard"])))
        return avg, per
    return None  # the reference falls off the end for other datasets


def compute_map_and_print(dataset, ranks, gnd, kappas=[1, 5, 10]):
    """``(averages, per_query)`` dicts with the reference's keys (evaluate.py:114-152).  A ranking on the GPU is searched
    once for every labelled id (all protocol levels share the positions)."""
    positions = _Positions(gnd, lambda lists: positions_in_ranking(ranks, lists)) if _is_device_tensor(ranks) else None
    return _evaluate(dataset, gnd, kappas, lambda g, k: compute_map(ranks, g, k, _positions=positions), positions)


def compute_map_and_print_from_scores(dataset, scores, gnd, kappas=[1, 5, 10]):
    """Same as :func:`compute_map_and_print`, from device scores ``[Q,N]`` (one counting pass for all protocol levels)."""
    positions = _score_positions(scores, gnd)
    return _evaluate(dataset, gnd, kappas, lambda g, k: compute_map_from_scores(scores, g, k, _positions=positions), positions)
