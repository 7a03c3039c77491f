"""N>1 path on CPU: 2 (and 3) processes over gloo run mdir_amd.sharded.ShardedIndex with the
compute backend replaced by the oracle (tests may; the product default is the HIP library).
Checks that shard -> all-to-all -> query-split ranking and the sort-free positions route
give exactly the single-process oracle result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


class OracleBackend:
    """Same four calls as mdir_amd.sharded.HipBackend, computed by the CPU oracle."""

    class _Index:
        def __init__(self, vecs, layout, row_offset):
            v = vecs.numpy()
            self.dn = np.ascontiguousarray(v if layout == "DN" else v.T)
            self.row_offset = row_offset

        def scores(self, queries, qlayout="DN"):
            from oracle import chain as OC
            q = queries.numpy()
            return torch.from_numpy(OC.scores_chain(self.dn, np.ascontiguousarray(q if qlayout == "DN" else q.T)))

    def make_index(self, vecs, layout, row_offset):
        return self._Index(vecs, layout, row_offset)

    def rank_full(self, scores, id_offset=0):
        from oracle import chain as OC
        return torch.from_numpy(OC.rank_full(scores.numpy()) + id_offset)

    def rank_full_segments(self, blocks, id_offset=0):
        assert len(blocks) > 1 and all(b.shape[0] == blocks[0].shape[0] for b in blocks)
        return self.rank_full(torch.cat(blocks, dim=1), id_offset)

    def topk(self, scores, k, id_offset=0):
        from oracle import chain as OC
        rk = OC.rank_full(scores.numpy())[:, :k]
        return torch.from_numpy(rk + id_offset), torch.from_numpy(np.take_along_axis(scores.numpy(), rk, axis=1))

    def gather_scores(self, scores, ids, offsets):
        s, off = scores.numpy(), offsets.numpy()
        out = np.empty(len(ids), dtype=np.float32)
        for q in range(len(off) - 1):
            out[off[q]:off[q + 1]] = s[q][ids.numpy()[off[q]:off[q + 1]]]
        return torch.from_numpy(out)

    def rank_count_(self, cnt, scores, id_offset, ref_scores, ref_ids, offsets):
        from oracle import chain as OC
        s, off = scores.numpy(), offsets.numpy()
        for q in range(len(off) - 1):
            keys = np.array([OC.desc_key(x) for x in s[q]], dtype=np.uint64)
            gid = np.arange(s.shape[1]) + id_offset
            for t in range(off[q], off[q + 1]):
                rk, ri = OC.desc_key(float(ref_scores[t])), int(ref_ids[t])
                cnt[t] += int(np.count_nonzero((keys < rk) | ((keys == rk) & (gid < ri))))
        return cnt


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, nq, d, out_dir, chunks, exchange=None):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if chunks:
        os.environ["MDIR_AMD_EXCHANGE_CHUNKS"] = str(chunks)
    if exchange:
        os.environ["MDIR_AMD_EXCHANGE"] = exchange
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mdir_amd.sharded import ShardedIndex, shard_bounds
    from oracle import oracle as O
    from test_sharded_gloo import OracleBackend
    vecs, qvecs, _ = O.synth_ranking_problem(n, nq, d, seed=4)
    vecs[:, 7] = vecs[:, 3]; vecs[:, n - 1] = vecs[:, 3]       # exact ties across shards
    lo, hi = shard_bounds(n, world, rank)
    sh = ShardedIndex(torch.from_numpy(np.ascontiguousarray(vecs[:, lo:hi])), "DN", n, backend=OracleBackend())
    assert sh.chunks == (chunks or 1) and len(sh.parts) == sh.chunks
    rk, sc, (qlo, qhi) = sh.rank_queries(torch.from_numpy(qvecs), "DN")
    # a second exchange of the same sizes must not touch what the first one returned (`sc` is saved below)
    _, sc2, _ = sh.rank_queries(torch.from_numpy(np.ascontiguousarray(-qvecs)), "DN")
    assert qhi == qlo or not np.array_equal(sc2.dense().numpy(), sc.dense().numpy())
    gnd = O.synth_gnd(nq, n, seed=1, easy=3, hard=4, junk=2)
    lists = [np.concatenate([g["easy"], g["hard"], g["junk"]]) for g in gnd]
    lists[0] = np.array([3, 7, n - 1])
    pos, off = sh.positions(sh.local_scores(torch.from_numpy(qvecs), "DN"), lists)
    assert np.array_equal(sh.all_scores(torch.from_numpy(qvecs), "DN").numpy(),
                          __import__("oracle.chain", fromlist=["x"]).scores_chain(vecs, qvecs))      # all-gather form
    # the exchange as its own steps (sharded.py `exchange`, `exchanged_scores`): [Q, n_local] on every rank -> [Q_mine, N]; the
    # BlockScores container answers like the dense matrix
    want_all = __import__("oracle.chain", fromlist=["x"]).scores_chain(vecs, qvecs)
    dense, (elo, ehi) = sh.exchange(sh.local_scores(torch.from_numpy(qvecs), "DN"))
    assert (elo, ehi) == (qlo, qhi) and np.array_equal(dense.numpy(), want_all[qlo:qhi])
    again, bounds = sh.exchanged_scores(torch.from_numpy(qvecs), "DN")
    assert bounds == (qlo, qhi) and np.array_equal(again.numpy(), want_all[qlo:qhi])
    assert len(sc) == qhi - qlo and sc.shape == (qhi - qlo, n) and np.array_equal(sc.cpu().numpy(), want_all[qlo:qhi])
    if qhi > qlo:
        assert np.array_equal(sc[0].numpy(), want_all[qlo])
    from mdir_amd.sharded import gather_query_vectors
    qrows = torch.from_numpy(np.ascontiguousarray(qvecs.T))
    a_lo, a_hi = shard_bounds(nq, world, rank)
    assert np.array_equal(gather_query_vectors(qrows[a_lo:a_hi], nq).numpy(), qrows.numpy())          # even and uneven slices
    tk_ids, tk_vals = sh.topk_queries(torch.from_numpy(qvecs), 9, "DN")
    big_ids, _ = sh.topk_queries(torch.from_numpy(qvecs), n + 5, "DN")          # k beyond every shard and beyond N
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), ranks=rk.numpy(), scores=sc.dense().numpy(), q=np.array([qlo, qhi]),
             pos=pos.numpy(), off=np.array(off), tk_ids=tk_ids.numpy(), tk_vals=tk_vals.numpy(), big_ids=big_ids.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n,nq,chunks,exchange", [(2, 301, 7, 0, None), (3, 100, 2, 0, None), (2, 301, 7, 3, None),
                                                        (3, 101, 5, 2, None), (3, 101, 5, 2, "allgather"),
                                                        # the node's shapes: 70 queries over 4 and 8 ranks (uneven: 8.75 per rank),
                                                        # shards of 8-9 rows against k = 9, fewer queries than ranks (idle sorters)
                                                        (4, 203, 70, 0, None), (8, 71, 70, 0, None), (8, 203, 70, 2, None),
                                                        (4, 101, 3, 0, "allgather"),
                                                        (2, 100, 4, 0, None)])           # queries divide evenly (plain all_gather of the query slices)
def test_sharded_equals_single_process(tmp_path, world, n, nq, chunks, exchange):
    """chunks > 0: every shard is cut into row chunks whose all-to-alls are in flight together
    (the overlap pipeline used for big shards); exchange="allgather": the fallback taken when the
    backend refuses an uneven all-to-all."""
    from oracle import chain as OC
    from oracle import oracle as O
    d = 32
    mp.spawn(_worker, args=(world, _free_port(), n, nq, d, str(tmp_path), chunks, exchange), nprocs=world, join=True)
    vecs, qvecs, _ = O.synth_ranking_problem(n, nq, d, seed=4)
    vecs[:, 7] = vecs[:, 3]; vecs[:, n - 1] = vecs[:, 3]
    want_sc = OC.scores_chain(vecs, qvecs)
    want_rk = OC.rank_full(want_sc)
    gnd = O.synth_gnd(nq, n, seed=1, easy=3, hard=4, junk=2)
    lists = [np.concatenate([g["easy"], g["hard"], g["junk"]]) for g in gnd]
    lists[0] = np.array([3, 7, n - 1])
    covered = 0
    for r in range(world):
        g = np.load(tmp_path / ("r%d.npz" % r))
        qlo, qhi = g["q"]
        np.testing.assert_array_equal(g["scores"], want_sc[qlo:qhi])
        np.testing.assert_array_equal(g["ranks"], want_rk[qlo:qhi])
        covered += qhi - qlo
        # global top-k from per-shard candidates == prefix of the full ranking, on every rank (ties included)
        np.testing.assert_array_equal(g["tk_ids"], want_rk[:, :9])
        np.testing.assert_array_equal(g["tk_vals"], np.take_along_axis(want_sc, want_rk[:, :9], axis=1))
        np.testing.assert_array_equal(g["big_ids"], want_rk)
        for q in range(nq):
            np.testing.assert_array_equal(g["pos"][g["off"][q]:g["off"][q + 1]], OC.rank_of(want_sc[q], lists[q]))
    assert covered == nq


def test_exchange_chunk_policy():
    from mdir_amd.sharded import chunk_bounds, exchange_chunks
    assert exchange_chunks(1004993, 1) == 1
    assert exchange_chunks(1004993, 2) == 3          # 502 k-row shards: 4/7, 2/7, 1/7 -- only the last transfer is exposed
    assert exchange_chunks(1004993, 4) == 2 and exchange_chunks(1004993, 8) == 2 and exchange_chunks(1004993, 16) == 1
    assert exchange_chunks(4993, 8) == 1
    # 125 k-row shards (G = 8): the first chunk is exactly one round of the chip's 512 workgroup slots (65 536 rows), on every
    # rank (shards differ by a row); shards outside (1.5, 2] rounds are cut in equal halves; bigger shards halve
    assert [y - x for x, y in chunk_bounds(0, 125625, 2)] == [65536, 60089]
    assert [y - x for x, y in chunk_bounds(7, 7 + 125624, 2)] == [65536, 60088]
    assert [y - x for x, y in chunk_bounds(0, 90000, 2)] == [45000, 45000]
    assert [y - x for x, y in chunk_bounds(0, 251249, 2)] == [167499, 83750]
    b = chunk_bounds(10, 21, 3)
    assert b[0][0] == 10 and b[-1][1] == 21 and all(x[1] == y[0] for x, y in zip(b, b[1:]))
    sizes = [y - x for x, y in chunk_bounds(0, 700000, 3)]
    assert sizes == [400000, 200000, 100000]
    assert chunk_bounds(5, 9, 1) == [(5, 9)] and all(y >= x for x, y in chunk_bounds(0, 2, 3))


def test_shard_bounds_cover_everything():
    from mdir_amd.sharded import shard_bounds
    for n in (1, 7, 70, 1004993):
        for w in (1, 2, 3, 8):
            edges = [shard_bounds(n, w, r) for r in range(w)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1


def test_default_backend_is_hip_and_refuses_cpu():
    from mdir_amd.sharded import ShardedIndex
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ShardedIndex(torch.zeros(8, 16), "DN", 16)


def _map_worker(rank, world, port, root, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MDIR_AMD_WORKERS="0", CIRTORCH_ROOT=root)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import fake_ops
    fake_ops.install_globally()
    import pickle
    from mdir_amd.networks import init_network
    from mdir_amd.score import initialize_score
    torch.manual_seed(0)
    net = init_network({"architecture": "alexnet", "pooling": "gem", "whitening": False, "pretrained": False}).eval()
    net.meta["out_channels"] = 256
    res = {}
    for ds in ("roxford5k", "247tokyo1k"):
        score = initialize_score({"type": "cirdatasetap", "image_size": 224, "dataset": ds,
                                  "transforms": "pil2np | totensor | normalize",
                                  "mean_std": [net.meta["mean"], net.meta["std"]]})
        rows = []
        with torch.no_grad():       # the score object takes the sharded route by itself (world size 2)
            score(net, "cpu", lambda it, size, label, value, dtype: rows.append((label, value)))
        assert rows[0][0] == "dataset" and set(rows[0][1]) == {"extract_descriptors", "compute_score", "total_s"}
        per = {k: np.array([r[1][k] for r in rows[2:]]) for k in rows[2][1]}
        res[ds] = (rows[1][1], per)
    with open(os.path.join(out_dir, "map%d.pkl" % rank), "wb") as f:
        pickle.dump(res, f)
    dist.destroy_process_group()


def test_sharded_extraction_and_map_equal_single_process(tmp_path, monkeypatch):
    """Each rank extracts its slice of the database (= its shard) and of the queries; the sort-free
    distributed mAP equals the single-process CirDatasetAp on the same data (both protocols,
    incl. the query == database shortcut of 247tokyo1k)."""
    import pickle
    import subprocess
    import fake_ops
    root = str(tmp_path / "synth")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_synthetic_eval.py"), root],
                          stdout=subprocess.DEVNULL)
    mp.spawn(_map_worker, args=(2, _free_port(), root, str(tmp_path)), nprocs=2, join=True)
    got = [pickle.load(open(tmp_path / ("map%d.pkl" % r), "rb")) for r in range(2)]
    # single process, full ranking route
    fake_ops.install(monkeypatch)
    monkeypatch.setenv("MDIR_AMD_WORKERS", "0")
    monkeypatch.setenv("CIRTORCH_ROOT", root)
    from mdir_amd.networks import init_network
    from mdir_amd.score import initialize_score
    torch.manual_seed(0)
    net = init_network({"architecture": "alexnet", "pooling": "gem", "whitening": False, "pretrained": False}).eval()
    net.meta["out_channels"] = 256
    for ds in ("roxford5k", "247tokyo1k"):
        score = initialize_score({"type": "cirdatasetap", "image_size": 224, "dataset": ds,
                                  "transforms": "pil2np | totensor | normalize",
                                  "mean_std": [net.meta["mean"], net.meta["std"]]})
        rows = []
        with torch.no_grad():
            score(net, "cpu", lambda it, size, label, value, dtype: rows.append((label, value)))
        want_avg = rows[1][1]
        want_per = {k: np.array([r[1][k] for r in rows[2:]]) for k in rows[2][1]}
        for r in range(2):
            avg, per = got[r][ds]
            assert avg.keys() == want_avg.keys()
            for k in avg:
                np.testing.assert_allclose(avg[k], want_avg[k], rtol=0, atol=1e-12)
            for k in want_per:
                np.testing.assert_allclose(per[k], want_per[k], rtol=0, atol=1e-12, equal_nan=True)
        assert got[0][ds][0] == got[1][ds][0]


def test_single_process_forms_and_argument_errors():
    """world == 1 (no process group): the sharded calls answer without any collective -- `exchange` hands the scores back,
    `all_scores` are the local ones, `gather_query_vectors` the slice itself, `phase_ms` has nothing to report; wrong shard
    sizes and compute modes are refused."""
    from mdir_amd.sharded import ShardedIndex, gather_query_vectors
    from oracle import chain as OC
    from oracle import oracle as O
    vecs, qvecs, _ = O.synth_ranking_problem(50, 4, 16, seed=2)
    sh = ShardedIndex(torch.from_numpy(vecs), "DN", 50, backend=OracleBackend())
    want = OC.scores_chain(vecs, qvecs)
    local = sh.local_scores(torch.from_numpy(qvecs), "DN")
    same, bounds = sh.exchange(local)
    assert same is local and bounds == (0, 4)
    assert np.array_equal(sh.all_scores(torch.from_numpy(qvecs), "DN").numpy(), want)
    rk, sc, (qlo, qhi) = sh.rank_queries(torch.from_numpy(qvecs), "DN")
    assert (qlo, qhi) == (0, 4) and np.array_equal(rk.numpy(), OC.rank_full(want)) and sh.phase_ms() is None
    ids, vals = sh.topk_queries(torch.from_numpy(qvecs), 5, "DN")
    assert np.array_equal(ids.numpy(), OC.rank_full(want)[:, :5])
    q = torch.rand(3, 16)
    assert gather_query_vectors(q, 3) is q
    with pytest.raises(ValueError, match="holds 50 rows, expected 49"):
        ShardedIndex(torch.from_numpy(vecs), "DN", 49, backend=OracleBackend())
    with pytest.raises(ValueError, match="compute"):
        ShardedIndex(torch.from_numpy(vecs), "DN", 50, backend=OracleBackend(), compute="bf16x9")
    with pytest.raises(ValueError, match="multiplies an fp32 shard"):
        ShardedIndex(torch.from_numpy(vecs), "DN", 50, backend=OracleBackend(), storage="f16", compute="split3")


def test_preflight_vocabulary_and_the_order_forms_are_tried_in(monkeypatch):
    """tools/preflight_ranks.py (round 6): the environment each exchange form stands for, the fall-back order from each
    requested form, what bench.py asks for by default and under --comm / MDIR_AMD_EXCHANGE; a preflight whose children cannot
    start reports every form with its reason instead of raising (bench.py then keeps the requested form)."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    sys.path.insert(0, root)
    import bench
    import preflight_ranks as P
    assert P.form_env("p2p") == {"MDIR_AMD_COMM": "p2p", "MDIR_AMD_EXCHANGE": ""}
    assert P.form_env("mdx")["MDIR_AMD_COMM"] == "mdx" and P.form_env("allgather")["MDIR_AMD_EXCHANGE"] == "allgather"
    assert P.form_env("torch") == {"MDIR_AMD_COMM": "", "MDIR_AMD_EXCHANGE": ""}
    with pytest.raises(ValueError):
        P.form_env("ring")
    assert P.fallbacks("p2p") == ["p2p", "mdx", "torch", "allgather"] and P.fallbacks("torch") == ["torch", "allgather"]
    assert P.fallbacks("allgather") == ["allgather"]
    for comm, exchange, want in (("", "", "torch"), ("mdx", "", "mdx"), ("p2p", "allgather", "p2p"), ("", "allgather", "allgather")):
        monkeypatch.setenv("MDIR_AMD_COMM", comm)
        monkeypatch.setenv("MDIR_AMD_EXCHANGE", exchange)
        assert bench.requested_form() == want
    # children that cannot even start (a python that does not exist): every form tried, none passed, reasons kept, nothing raised
    pre = None
    try:
        pre = P.run(2, "torch", dryrun=True, python="/nonexistent/python", budget_s=60)
    except FileNotFoundError:
        pass                                    # (subprocess raises before a child exists: bench.preflight turns that into a verdict)
    if pre is not None:
        assert pre["form"] is None and all(not t["ok"] for t in pre["tried"])
    monkeypatch.setattr(P, "run", lambda *a, **k: (_ for _ in ()).throw(OSError("no children here")))
    verdict = bench.preflight(2)
    assert verdict["form"] is None and "no children here" in verdict["tried"][0]["reason"]
