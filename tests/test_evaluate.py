"""mdir_amd.evaluate (the shipped host code) against the reference's outputs (golden) and the oracle."""
import json

import numpy as np
import pytest
import torch

from mdir_amd import evaluate as E
from oracle import oracle as O


def _gnd(g):
    return json.loads(bytes(g["gnd_json"]).decode())


def test_compute_map_matches_reference_outputs(golden):
    g = golden("g8_map.npz")
    rk, gnd = g["ranks"].astype(np.int64), _gnd(g)
    avg, per = E.compute_map_and_print("roxford5k", rk, gnd)
    for lvl in ("easy", "medium", "hard"):
        assert avg["map_" + lvl] == float(g["rox_map_" + lvl])
        np.testing.assert_array_equal(per["ap_" + lvl], g["rox_ap_" + lvl])
    m, aps, pr, prs = E.compute_map(rk, O.protocol_gnd(gnd, "medium"), [1, 5, 10])
    assert m == float(g["medium_map"])
    np.testing.assert_array_equal(aps, g["medium_aps"])
    np.testing.assert_array_equal(pr, g["medium_pr"])
    np.testing.assert_array_equal(prs, g["medium_prs"])
    old = [{"ok": x["easy"] + x["hard"], "junk": x["junk"]} for x in gnd]
    avg, per = E.compute_map_and_print("247tokyo1k", rk, old)
    assert avg["map"] == float(g["old_map"])
    np.testing.assert_array_equal(per["ap"], g["old_ap"])
    m, aps, _, _ = E.compute_map(rk, [{"ok": x["ok"]} for x in old])
    assert m == float(g["nojunkkey_map"])
    assert E.compute_map_and_print("oxford5k", rk, gnd) is None
    assert E.compute_ap([0], 1) == 1.0 and E.compute_ap([1], 1) == 0.25
    assert E.compute_ap([0, 2], 2) == float(g["ap_r02_n2"]) and E.compute_ap([], 3) == 0


def test_torch_and_transposed_inputs(golden):
    g = golden("g8_map.npz")
    rk, gnd = g["ranks"].astype(np.int64), _gnd(g)
    want = E.compute_map(rk, O.protocol_gnd(gnd, "medium"), [1, 5, 10])
    qn = torch.from_numpy(np.ascontiguousarray(rk.T))          # [Q,N] as mdx_rank_full writes it
    got = E.compute_map(qn.t(), O.protocol_gnd(gnd, "medium"), [1, 5, 10])
    for a, b in zip(want, got):
        np.testing.assert_array_equal(np.asarray(a), np.asarray(b))


def test_positions_route_equals_ranking_route():
    rng = np.random.default_rng(2)
    n, nq = 300, 9
    sc = rng.standard_normal((nq, n)).astype(np.float32)
    rk = O.ranks(sc.T)
    gnd = [{"ok": rng.choice(n, rng.integers(0, 6), replace=False), "junk": rng.choice(n, 4, replace=False)}
           for _ in range(nq)]
    for g_ in gnd:
        g_["junk"] = np.setdiff1d(g_["junk"], g_["ok"])
    want = E.compute_map(rk, gnd, [1, 5])
    pos = [O.rank_of(sc[q], gnd[q]["ok"]) for q in range(nq)]
    junk = [O.rank_of(sc[q], gnd[q]["junk"]) if len(gnd[q]["ok"]) else np.empty(0, np.int64) for q in range(nq)]
    got = E.map_from_positions(pos, junk, [len(g_["ok"]) for g_ in gnd], [1, 5])
    for a, b in zip(want, got):
        np.testing.assert_array_equal(np.asarray(a), np.asarray(b))
    ref = O.compute_map(rk, gnd, [1, 5])
    for a, b in zip(want, ref):
        np.testing.assert_array_equal(np.asarray(a), np.asarray(b))


def _brute_force_ap(ranking, ok, junk):
    """AP exactly as the retrieval benchmarks define it: drop junk from the ranking, then the
    trapezoidal area under precision/recall at every positive (evaluate.py:3-37 restated from the
    definition, without the position bookkeeping of the reference)."""
    ok, junk = set(int(x) for x in ok), set(int(x) for x in junk)
    if not ok:
        return float("nan")
    cleaned = [int(r) for r in ranking if int(r) not in junk]
    ap, hits = 0.0, 0
    for j, r in enumerate(cleaned):
        if r in ok:
            prec_before = hits / j if j else 1.0
            hits += 1
            ap += (prec_before + hits / (j + 1)) / 2.0 / len(ok)
    return ap


def test_compute_map_property_random_rankings():
    """hypothesis: shipped compute_map == oracle compute_map == the brute-force definition, on random
    permutations with random positive / junk sets (incl. empty positives and junk overlapping nothing)."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=60, deadline=None)
    @given(st.integers(2, 60), st.integers(1, 6), st.integers(0, 2 ** 31 - 1))
    def check(n, nq, seed):
        rng = np.random.default_rng(seed)
        rk = np.stack([rng.permutation(n) for _ in range(nq)], axis=1)          # [N,Q] as the reference
        gnd = []
        for _ in range(nq):
            ok = rng.choice(n, int(rng.integers(0, min(n, 6) + 1)), replace=False)
            rest = np.setdiff1d(np.arange(n), ok)
            junk = rng.choice(rest, int(rng.integers(0, min(len(rest), 5) + 1)), replace=False) if len(rest) else rest
            gnd.append({"ok": ok, "junk": junk})
        if all(len(g["ok"]) == 0 for g in gnd):
            # no query with positives: the reference divides by nq - nempty = 0 (evaluate.py:108) and raises
            import pytest
            with pytest.raises(ZeroDivisionError):
                E.compute_map(rk, gnd, [1, 5])
            return
        m, aps, pr, prs = E.compute_map(rk, gnd, [1, 5])
        mo, apso, pro, prso = O.compute_map(rk, gnd, [1, 5])
        np.testing.assert_array_equal(aps, apso)
        np.testing.assert_array_equal(prs, prso)
        assert (np.isnan(m) and np.isnan(mo)) or m == mo
        want = np.array([_brute_force_ap(rk[:, q], gnd[q]["ok"], gnd[q]["junk"]) for q in range(nq)])
        np.testing.assert_allclose(aps, want, rtol=0, atol=1e-12, equal_nan=True)

    check()


def test_sort_free_route_with_repeated_and_foreign_ids(monkeypatch):
    """A ground-truth list may repeat an id or name an id that is not a database row (user-supplied TSV
    datasets).  The reference's np.in1d gives every rank position once and ignores foreign ids, while its
    normaliser stays len(ok) as listed (evaluate.py:80-81,101): the sort-free route (positions from the
    scores, CirDatasetAp's default) must return the same APs as compute_map on the ranking."""
    import torch
    import fake_ops
    fake_ops.install(monkeypatch)
    rng = np.random.default_rng(3)
    n, nq = 40, 4
    sc = rng.standard_normal((nq, n)).astype(np.float32)
    rk = np.argsort(-sc, axis=1, kind="stable").T                                   # [N,Q]
    gnd = [{"ok": [3, 3, 7, 9], "junk": [1, 2]},
           {"ok": [4, 5], "junk": [5, 6, 6]},
           {"ok": [8, n + 5, 11], "junk": [n, 0, -1]},                               # foreign ids on both lists
           {"ok": [], "junk": [1]}]
    want = O.compute_map(rk, gnd, [1, 5, 10])
    got_ranking = E.compute_map(rk, gnd, [1, 5, 10])
    got_scores = E.compute_map_from_scores(torch.from_numpy(sc), gnd, [1, 5, 10])
    for a, b, c in zip(want, got_ranking, got_scores):
        np.testing.assert_array_equal(np.asarray(a), np.asarray(b))
        np.testing.assert_array_equal(np.asarray(a), np.asarray(c))
    oks, junks, nok = E.labelled_lists(gnd, n)
    assert [o.tolist() for o in oks] == [[3, 7, 9], [4, 5], [8, 11], []] and nok == [4, 2, 3, 0]
    assert [j.tolist() for j in junks] == [[1, 2], [5, 6], [0], []]


def test_vectorised_map_on_fetched_positions_is_bit_identical_to_the_per_query_statement():
    """`_Positions.map` (what compute_map runs on a GPU ranking and compute_map_from_scores on GPU scores: one lookup pass,
    junk shift and AP terms for all queries at once) against `compute_map` on the host ranking (np.isin column by column +
    the reference's loops): same floats bit for bit, same NaNs, same exceptions -- ids in two lists, listed twice, negative,
    beyond the database, queries without positives, positives that are all missing, no `junk` key, with and without kappas."""
    from mdir_amd import evaluate as E
    rng = np.random.default_rng(1)
    for trial in range(150):
        n, nq = int(rng.integers(5, 300)), int(rng.integers(1, 9))
        ranks = np.stack([rng.permutation(n) for _ in range(nq)], axis=1)
        gnd = []
        for q in range(nq):
            k_ok, k_j = int(rng.integers(0, min(n, 12))), int(rng.integers(0, min(n, 8)))
            ok, junk = rng.integers(-2, n + 3, size=k_ok), rng.integers(-2, n + 3, size=k_j)
            if k_ok and rng.random() < 0.3:
                junk = np.concatenate([junk, ok[:2]])
            if k_ok and rng.random() < 0.3:
                ok = np.concatenate([ok, ok[:1]])
            g = {"ok": ok.astype(np.int64)}
            if rng.random() < 0.85:
                g["junk"] = junk.astype(np.int64)
            gnd.append(g)
        kappas = [1, 5, 10] if rng.random() < 0.7 else []

        def fetch(lists):
            out = []
            for q, ids in enumerate(lists):
                inv = {int(v): i for i, v in enumerate(ranks[:, q])}
                out.append(np.array([inv.get(int(x), -1) for x in ids], dtype=np.int64))
            return out

        def run(f):
            try:
                return f()
            except (ValueError, ZeroDivisionError) as exc:
                return type(exc).__name__

        want = run(lambda: E.compute_map(ranks, gnd, kappas))
        got = run(lambda: E._Positions(gnd, fetch).map(gnd, kappas))
        if isinstance(want, str) or isinstance(got, str):
            assert want == got
            continue
        for a, b in zip(want, got):
            np.testing.assert_array_equal(np.asarray(a), np.asarray(b))


def test_protocol_levels_on_shared_masks_equal_the_concatenated_lists():
    """`_Positions.map_levels` (easy / hard / junk memberships looked up once, the three levels of evaluate.py:123-147
    combine the masks) against `map` on the concatenated lists and against the host statement -- ids shared between keys,
    duplicates, ids without a position, queries without positives."""
    from mdir_amd import evaluate as E
    rng = np.random.default_rng(7)
    for trial in range(60):
        n, nq = int(rng.integers(20, 400)), int(rng.integers(1, 8))
        ranks = np.stack([rng.permutation(n) for _ in range(nq)], axis=1)
        gnd = []
        for q in range(nq):
            g = {k: rng.integers(-1, n + 2, size=int(rng.integers(0, 9))).astype(np.int64) for k in ("easy", "hard", "junk")}
            if rng.random() < 0.3 and len(g["easy"]):
                g["junk"] = np.concatenate([g["junk"], g["easy"][:1]])
            if rng.random() < 0.2:
                g["easy"], g["hard"] = g["easy"][:0], g["hard"][:0]
            gnd.append(g)

        def fetch(lists):
            out = []
            for q, ids in enumerate(lists):
                inv = {int(v): i for i, v in enumerate(ranks[:, q])}
                out.append(np.array([inv.get(int(x), -1) for x in ids], dtype=np.int64))
            return out

        for _, ok_keys, junk_keys in E._LEVELS:
            level = E._protocol_gnd(gnd, ok_keys, junk_keys)

            def run(f):
                try:
                    return f()
                except (ValueError, ZeroDivisionError) as exc:
                    return type(exc).__name__
            want = run(lambda: E.compute_map(ranks, level, [1, 5, 10]))
            a = run(lambda: E._Positions(gnd, fetch).map(level, [1, 5, 10]))
            b = run(lambda: E._Positions(gnd, fetch).map_levels(gnd, ok_keys, junk_keys, [1, 5, 10]))
            if isinstance(want, str):
                assert a == want and b == want
                continue
            for x, y, z in zip(want, a, b):
                np.testing.assert_array_equal(np.asarray(x), np.asarray(y))
                np.testing.assert_array_equal(np.asarray(x), np.asarray(z))


def test_random_map_problems_match_the_reference():
    """G19: 300 random compute_map problems and 100 random revisited-protocol problems (evaluate.py:39-152) whose inputs are
    regenerated from seeds (tests/golden/make_golden.py: fuzz_map_case / fuzz_revisited_case) against the reference's stored
    outputs -- lists and arrays of ids, empty and overlapping ok / junk, a missing junk key, kappas beyond N, queries without
    positives (NaN, excluded), and the reference's own ZeroDivisionError when NO query has positives."""
    import contextlib
    import copy
    import io
    import os
    import sys
    from conftest import GOLDEN
    sys.path.insert(0, GOLDEN)
    try:
        from make_golden import fuzz_map_case, fuzz_revisited_case
    finally:
        sys.path.remove(GOLDEN)
    g = np.load(os.path.join(GOLDEN, "g19_map_fuzz.npz"))
    errors = 0
    for seed in range(300):
        ranks, gnd, kappas = fuzz_map_case(seed)
        if "error_%d" % seed in g:
            errors += 1
            with pytest.raises(BaseException) as caught:
                E.compute_map(ranks.copy(), copy.deepcopy(gnd), list(kappas))
            assert type(caught.value).__name__ == str(g["error_%d" % seed][0]), seed
            continue
        mAP, aps, pr, prs = E.compute_map(ranks.copy(), copy.deepcopy(gnd), list(kappas))
        np.testing.assert_allclose(mAP, g["map_%d" % seed][0], rtol=0, atol=1e-12, err_msg=str(seed))
        np.testing.assert_allclose(np.asarray(aps, dtype=np.float64), g["aps_%d" % seed], rtol=0, atol=1e-12, err_msg=str(seed))
        np.testing.assert_allclose(np.asarray(pr, dtype=np.float64), g["pr_%d" % seed], rtol=0, atol=1e-12, err_msg=str(seed))
        np.testing.assert_allclose(np.asarray(prs, dtype=np.float64), g["prs_%d" % seed], rtol=0, atol=1e-12, err_msg=str(seed))
    assert 0 < errors < 100
    for seed in range(100):
        ranks, gnd = fuzz_revisited_case(1000 + seed)
        name = "roxford5k" if seed % 2 else "rparis6k"
        if "rev_error_%d" % seed in g:
            with pytest.raises(BaseException) as caught, contextlib.redirect_stdout(io.StringIO()):
                E.compute_map_and_print(name, ranks.copy(), copy.deepcopy(gnd))
            assert type(caught.value).__name__ == str(g["rev_error_%d" % seed][0]), seed
            continue
        with contextlib.redirect_stdout(io.StringIO()):
            avg, per = E.compute_map_and_print(name, ranks.copy(), copy.deepcopy(gnd))
        for k in ("map_easy", "map_medium", "map_hard"):
            np.testing.assert_allclose(avg[k], g["rev_%s_%d" % (k, seed)][0], rtol=0, atol=1e-12, err_msg="%s %d" % (k, seed))
        for k in ("ap_easy", "ap_medium", "ap_hard"):
            np.testing.assert_allclose(np.asarray(per[k], dtype=np.float64), g["rev_%s_%d" % (k, seed)], rtol=0, atol=1e-12, err_msg="%s %d" % (k, seed))
