"""Round-5 GPU tests: the TSV/CSV dataset branch of CirDatasetAp on the real library, the ADVICE round-4 cases, the
8-rank dry run of bench.py on one GPU."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT
from oracle import chain as OC
from oracle import oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

sys.path.insert(0, os.path.join(ROOT, "tests"))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def test_dict_dataset_equals_official_branch_on_gpu(tmp_path, monkeypatch, capsys):
    """cirscore.py:24-38 (db / queries tables, here .csv + .tsv.gz) against cirscore.py:39-45 (gnd pickle) on the same
    synthetic old-protocol set, through the real extraction + mdx_scores_rowmajor + mdx_rank_of: same lists, same mAP and
    per-query APs; and the mAP is the oracle's on descriptors extracted with the public extract_vectors."""
    from mdir_amd.networks import extract_vectors
    from test_host_api import run_both_dataset_branches
    monkeypatch.setenv("MDIR_AMD_WORKERS", "0")
    ((official, rows_a), (tables, rows_b)), gnd, net = run_both_dataset_branches(tmp_path, monkeypatch, DEV)
    assert official.images == tables.images and official.qimages == tables.qimages and official.bbxs == tables.bbxs
    assert tables.gnd == [{k: g[k] for k in ("ok", "junk")} for g in gnd]
    assert [r[2] for r in rows_b] == ["dataset", "score_avg", "score", "score", "score"]
    assert [r[3] for r in rows_a[1:]] == [r[3] for r in rows_b[1:]]
    assert capsys.readouterr().out.count(">> oxford5k: mAP") == 2
    assert 0.0 < rows_b[1][3]["map"] <= 1.0
    # the mAP is the oracle's on the descriptors the public extract_vectors returns for the table branch's lists
    with torch.no_grad():
        vecs = extract_vectors(net, tables.images, 224, tables.transforms, device=DEV)
        qvecs = extract_vectors(net, tables.qimages, 224, tables.transforms, device=DEV, bbxs=tables.bbxs)
    assert tuple(vecs.shape) == (256, 9) and not vecs.is_cuda
    avg, per = O.compute_map_and_print("oxford5k", O.ranks(O.scores(vecs.numpy(), qvecs.numpy())), tables.gnd)
    np.testing.assert_allclose(rows_b[1][3]["map"], avg["map"], rtol=0, atol=1e-12)
    for i in range(3):
        np.testing.assert_allclose(rows_b[2 + i][3]["ap"], per["ap"][i], rtol=0, atol=1e-12)


def test_rowmajor_product_with_infinities_in_a_padded_chunk():
    """d % 64 != 0 with +-Inf among a row's last four values.  The in-place row-major route read those
    values again for the padded k and multiplied them by zero query tiles (0 * Inf = NaN); np.dot and the index route give
    +-Inf.  All three routes now equal the chain oracle bit for bit."""
    from mdir_amd import ops
    rng = np.random.default_rng(3)
    for n, d, nq in ((40_000, 100, 21), (300, 68, 5), (33_000, 2044, 70)):
        db = (rng.standard_normal((n, d)) / 8).astype(np.float32)
        qv = (rng.standard_normal((nq, d)) / 8).astype(np.float32)
        db[5, d - 1] = np.inf
        db[6, d - 3] = -np.inf
        db[7, d - 4], db[7, d - 2] = np.inf, np.inf
        db[n - 1, d - 1] = np.inf                     # the row that rows >= n of the last tile re-read
        db[9, d - 2] = np.nan
        want = OC.scores_chain(np.ascontiguousarray(db.T), np.ascontiguousarray(qv.T))
        assert np.isinf(want[:, 5]).all() and np.isinf(want[:, 6]).all() and np.isnan(want[:, 9]).all()
        assert np.isfinite(want[:, 4]).all() and np.isfinite(want[:, n - 2]).all()
        direct = ops.scores_rowmajor(dev(db), dev(qv), "ND").cpu().numpy()
        index = ops.DescriptorIndex(dev(db), "ND").scores(dev(qv), "ND").cpu().numpy()
        np.testing.assert_array_equal(direct, want)
        np.testing.assert_array_equal(index, want)


def test_shared_pool_graphs_replayed_in_reverse_capture_order():
    """all hipGraphs of a ShapeGraphs share one memory pool and are replayed in any order.  Capture shapes
    A, B, C (growing, then shrinking intermediates), replay C, B, A, B, C, A ...: every output equals the eager function's."""
    from mdir_amd.graphs import ShapeGraphs
    torch.manual_seed(0)
    w1 = torch.randn(64, 3, 3, 3, device=DEV)
    w2 = torch.randn(32, 64, 3, 3, device=DEV)

    def fn(x):
        y = torch.nn.functional.conv2d(x, w1, padding=1).relu_()
        z = torch.nn.functional.conv2d(y, w2, padding=1)
        return z.mean(dim=(2, 3)), y.amax(dim=(1, 2, 3))

    sg = ShapeGraphs(fn, warmup=0)
    shapes = [(1, 3, 96, 128), (2, 3, 200, 260), (1, 3, 40, 56), (4, 3, 64, 64)]
    xs = [torch.randn(s, device=DEV) for s in shapes]
    for x in xs:                                                         # capture order A, B, C, D
        sg(x)
    assert sg.captures == len(shapes) and sg.pool is not None
    order = [3, 2, 1, 0, 1, 3, 0, 2, 2, 0]
    kept = []
    for i in order:
        x = torch.randn(shapes[i], device=DEV)
        kept.append((x, sg(x)))
    assert sg.replays == len(shapes) + len(order) and sg.captures == len(shapes)      # a capture is followed by its first replay
    torch.cuda.synchronize()
    for x, (a, b) in kept:                                               # outputs survive later replays (they are clones)
        ea, eb = fn(x)
        np.testing.assert_allclose(a.cpu().numpy(), ea.cpu().numpy(), rtol=0, atol=1e-5)
        np.testing.assert_array_equal(b.cpu().numpy(), eb.cpu().numpy())


def test_positions_in_a_one_row_database():
    """a database of ONE row: the [N,Q] = [1,Q] ranking's transpose has strides (1, Q); the device route
    of compute_map takes it like any other."""
    from mdir_amd.evaluate import compute_map, positions_in_ranking
    ranks = torch.zeros((1, 5), dtype=torch.int64, device=DEV)
    gnd = [{"ok": [0], "junk": []}, {"ok": [], "junk": [0]}, {"ok": [0], "junk": []}, {"ok": [0], "junk": [0]}, {"ok": [0]}]
    pos = positions_in_ranking(ranks, [[0], [], [0], [0], [0]])
    assert [p.tolist() for p in pos] == [[0], [], [0], [0], [0]]
    got = compute_map(ranks, gnd)
    want = O.compute_map(np.zeros((1, 5), dtype=np.int64), gnd)
    np.testing.assert_allclose(got[0], want[0])
    np.testing.assert_array_equal(np.isnan(got[1]), np.isnan(want[1]))
    # the same ranking as a column slice of a wider matrix (row stride > 1)
    wide = torch.zeros((5, 4), dtype=torch.int64, device=DEV)
    assert [p.tolist() for p in positions_in_ranking(wide[:, :1].t(), [[0]] * 5)] == [[0]] * 5


def test_eigh_failure_is_numpy_linalgerror():
    """mdir/stages/whiten.py catches np.linalg.LinAlgError around whitening learning; a NaN covariance on
    the device solver raises that type."""
    from mdir_amd import whiten
    bad = torch.full((8, 8), float("nan"), dtype=torch.float64, device=DEV)
    try:
        w, v = whiten._eigh_descending(bad)
    except np.linalg.LinAlgError:
        return
    assert torch.isnan(w).all()         # some solver builds return NaNs instead of raising: nothing to convert then


# ---------------------------------------------------------------- configs[3]: eight ranks of the real kernels (dry run on one GPU)

def _bench(args, env_extra):
    env = dict(os.environ, **env_extra)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, text=True,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert proc.returncode == 0, (proc.stdout[-2000:], proc.stderr[-4000:])
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout[-2000:]
    return json.loads(lines[0])


@pytest.fixture(scope="module")
def one_rank_line():
    return _bench(["--rows", "200000", "--steps", "3", "--warmup", "1", "--no-secondary", "--no-pipelined", "--extract-images", "0"], {})


@pytest.mark.parametrize("form", ["all_to_all", "all_gather", "all_to_all_chunks2"])
def test_bench_with_eight_ranks_on_one_gpu(form, one_rank_line):
    """`bench.py --gpus 8` as the driver will launch it, here with all eight rank processes on this GPU over gloo
    (MDIR_AMD_DRYRUN_ONE_GPU: functional, never a measurement): the real kernels on eight shards, the query-split exchange
    (default), the literal all-gather of partial scores, and the chunked exchange -- each gives the single-process mAP, names
    configs[3], carries roofline + cpu_baseline + the per-step spread, and every rank's rows of the ranking verify on the device."""
    env = {"MDIR_AMD_DRYRUN_ONE_GPU": "1"}
    if form == "all_gather":
        env["MDIR_AMD_EXCHANGE"] = "allgather"
    if form == "all_to_all_chunks2":
        env["MDIR_AMD_EXCHANGE_CHUNKS"] = "2"
    line = _bench(["--gpus", "8", "--rows", "200000", "--steps", "3", "--warmup", "1", "--no-secondary", "--extract-images", "0"], env)
    one = one_rank_line
    assert line["n_gpus"] == 8 and line["nranks_seen"] == 8 and "DRY RUN" in line["data"]
    assert line["config"]["workload"].startswith("configs[3]") and "x8" in line["config"]["workload"]
    assert one["config"]["workload"].startswith("configs[2]")
    assert line["config"]["db_rows_per_gpu"] == 25000
    assert line["map_medium"] == one["map_medium"]
    assert line["phases_ms_per_rank"]["exchange"] == ("all_gather" if form == "all_gather" else "all_to_all")
    assert line["phases_ms_per_rank"]["chunks"] == (2 if form == "all_to_all_chunks2" else 1)
    assert len(line["phases_ms_per_rank"]["scores"]) == 8 and "ranking_verified_on_device" in line
    for ln in (line, one):
        assert ln["roofline"]["bound"] == "mfma" and ln["roofline"]["frac"] > 0 and "traffic" in ln["roofline"]
        assert ln["cpu_baseline"]["value"] > 0 and ln["cpu_baseline"]["kind"] == "port" and ln["cpu_baseline"]["cores"] >= 1
        assert abs(ln["map_medium_cpu"] - ln["map_medium"]) <= 1e-5
        sp = ln["spread_over_timed_steps"]
        assert sp["steps"] == 3 and sp["step_ms"]["min"] <= sp["step_ms"]["median"] <= sp["step_ms"]["max"]
        assert sp["value"]["min"] <= sp["value"]["median"] <= sp["value"]["max"]
    assert set(one["spread_over_timed_steps"]) >= {"kernel_ms", "rank_ms", "step_ms", "value"}
    assert line["roofline"]["algorithmic_flops_per_rank"] == 2.0 * 70 * 25000 * 2048
