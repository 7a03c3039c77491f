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


@pytest.mark.parametrize("form", ["all_to_all", "all_gather", "all_to_all_chunks2", "p2p"])
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
    more = ["--comm", "p2p"] if form == "p2p" else []        # round 6: the direct-store exchange (same-device hipIpc here)
    line = _bench(["--gpus", "8", "--rows", "200000", "--steps", "3", "--warmup", "1", "--no-secondary", "--no-preflight", "--extract-images", "0"] + more, env)
    one = one_rank_line
    assert line["n_gpus"] == 8 and line["nranks_seen"] == 8 and "DRY RUN" in line["data"]
    assert line["config"]["workload"].startswith("configs[3]") and "x8" in line["config"]["workload"]
    assert one["config"]["workload"].startswith("configs[2]")
    assert line["config"]["db_rows_per_gpu"] == 25000
    assert line["map_medium"] == one["map_medium"]
    assert line["phases_ms_per_rank"]["exchange"] == {"all_gather": "all_gather", "p2p": "direct_store"}.get(form, "all_to_all")
    if form == "p2p":
        assert line["comm"].startswith("p2p") and line["p2p_late_peers"] == 0
    assert line["phases_ms_per_rank"]["chunks"] == (2 if form == "all_to_all_chunks2" else 1)
    assert len(line["phases_ms_per_rank"]["scores"]) == 8 and "ranking_verified_on_device" in line
    for ln in (line, one):
        assert ln["roofline"]["bound"] == "mfma" and ln["roofline"]["frac"] > 0 and "traffic" in ln["roofline"]
        assert ln["cpu_baseline"]["value"] > 0 and ln["cpu_baseline"]["kind"] == "port" and ln["cpu_baseline"]["cores"] >= 1
        assert abs(ln["map_medium_cpu"] - ln["map_medium"]) <= 1e-5 and ln.get("map_within_1e-5_of_cpu_path", True)
        sp = ln["spread_over_timed_steps"]
        assert sp["steps"] == 3 and sp["step_ms"]["min"] <= sp["step_ms"]["median"] <= sp["step_ms"]["max"]
        assert sp["value"]["min"] <= sp["value"]["median"] <= sp["value"]["max"]
    assert set(one["spread_over_timed_steps"]) >= {"kernel_ms", "rank_ms", "step_ms", "value"}
    assert line["roofline"]["algorithmic_flops_per_rank"] == 2.0 * 70 * 25000 * 2048


# ---------------------------------------------------------------- host-surface branches that need the device (profiles/r05_host_branches.md)

def test_mac_spoc_functions_modules_and_their_place_in_the_network(golden):
    """cirtorch/layers/functional.py:11-16 (mac, spoc), pooling.py:14-33 (MAC / SPoC modules and their repr), through
    mdx_pool_l2n; golden G1 holds the reference's values.  `pool_kind` names them for the fused network tail."""
    from mdir_amd import layers
    from mdir_amd.networks import init_network
    from conftest import sparse_map
    g = golden("g1_pool.npz")
    for c, h, w in [(2048, 17, 23), (256, 7, 5)]:
        x = dev(sparse_map(int(g[f"seed_c{c}_h{h}_w{w}"]), (1, c, h, w)))
        for fn, mod, key in ((layers.mac, layers.MAC(), "mac"), (layers.spoc, layers.SPoC(), "spoc")):
            want = g[f"{key}_c{c}_h{h}_w{w}"]
            for got in (fn(x), mod(x)):
                assert tuple(got.shape) == (1, c, 1, 1)
                np.testing.assert_allclose(got.cpu().numpy().reshape(-1), want.reshape(-1), rtol=5e-6, atol=1e-7)
        np.testing.assert_allclose(layers.gem(x, p=torch.tensor([2.2]), eps=1e-6).cpu().numpy().reshape(-1),
                                   g[f"gem_c{c}_h{h}_w{w}_p2.2"].reshape(-1), rtol=2e-5, atol=1e-7)
    assert repr(layers.MAC()) == "MAC()" and repr(layers.SPoC()) == "SPoC()" and repr(layers.L2N()) == "L2N(eps=1e-06)"
    assert repr(layers.GeM(p=2.5)) == "GeM(p=2.5000, eps=1e-06)"
    assert layers.pool_kind(layers.MAC()) == ("mac", 1.0, 1e-6) and layers.pool_kind(layers.SPoC()) == ("spoc", 1.0, 1e-6)
    assert layers.pool_kind(torch.nn.AdaptiveAvgPool2d(1)) is None
    with pytest.raises(ValueError, match="global descriptors"):
        layers.l2n(torch.rand(2, 3, 4, 4, device=DEV))
    y = layers.L2N()(dev(golden("g2_l2n.npz")["x"]))
    np.testing.assert_allclose(y.cpu().numpy(), golden("g2_l2n.npz")["y"], rtol=1e-6, atol=1e-9)
    # a MAC network end to end == its statements: L2N(MAC(features(x)))
    torch.manual_seed(0)
    net = init_network({"architecture": "alexnet", "pooling": "mac", "whitening": False, "pretrained": False}).to(DEV).eval()
    xin = torch.rand(2, 3, 97, 130, device=DEV)
    with torch.no_grad():
        got = net(xin)
        feat = net.features(xin)
        want = torch.nn.functional.normalize(feat.amax(dim=(2, 3)), dim=1, eps=0) if False else feat.amax(dim=(2, 3))
        want = want / (want.norm(dim=1, keepdim=True) + 1e-6)
    np.testing.assert_allclose(got.t().cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-7)


def test_general_multiscale_routes_when_the_tail_cannot_be_fused(monkeypatch):
    """imageretrievalnet.py:309-324 (extract_ms) and wrapper.py:84-136 with a network whose tail is NOT pooling + L2N only (an
    in-network whitening follows): every scale goes through the whole network, descriptors are aggregated by mdx_ms_aggregate --
    for one image and for a batch (mdx_ms_aggregate_batch) -- and equal the statements written in torch; msp = 1 because the
    network whitens (quirk Q3)."""
    from mdir_amd.networks import extract_ms, init_network
    from mdir_amd.wrapper import CirMultiscaleAggregation, Compose
    torch.manual_seed(2)
    net = init_network({"architecture": "alexnet", "pooling": "gem", "whitening": True, "pretrained": False}).to(DEV).eval()
    net.meta["in_channels"], net.meta["out_channels"] = 3, net.meta["outputdim"]
    assert net.fusable_tail() is None and net.whiten is not None
    ms = [1, 2 ** -0.5, 0.5]

    def reference(x):                                   # per image: sum of the scales' descriptors / S, renormalised (msp = 1)
        acc = 0
        for s in ms:
            xs = x if s == 1 else torch.nn.functional.interpolate(x, scale_factor=s, mode="bilinear", align_corners=False)
            acc = acc + net(xs).t()
        acc = acc / len(ms)
        return acc / acc.norm(dim=1, keepdim=True)
    with torch.no_grad():
        one, four = torch.rand(1, 3, 160, 120, device=DEV), torch.rand(4, 3, 96, 128, device=DEV)
        got1, got4 = extract_ms(net, one, ms, 1), extract_ms(net, four, ms, 1)
        assert tuple(got1.shape) == (net.meta["outputdim"],) and tuple(got4.shape) == (4, net.meta["outputdim"])
        np.testing.assert_allclose(got1.cpu().numpy(), reference(one)[0].cpu().numpy(), rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(got4.cpu().numpy(), reference(four).cpu().numpy(), rtol=2e-5, atol=2e-6)
        # the wrapper chain on the same network: fused_tail declines (None), the general route aggregates the batch in one launch
        wrap = CirMultiscaleAggregation(True, DEV)
        assert wrap._msp(net) == 1
        chain = Compose([wrap], DEV)
        pyramid, waslist = wrap.preprocess(four, net)
        assert wrap.fused_tail(pyramid, net, net, waslist, DEV) is None
        np.testing.assert_allclose(chain(four, net).cpu().numpy(), reference(four).cpu().numpy(), rtol=2e-5, atol=2e-6)
        monkeypatch.setenv("MDIR_AMD_FUSED_TAIL", "0")          # the same general route on a network whose tail COULD be fused
        plain = init_network({"architecture": "alexnet", "pooling": "gem", "whitening": False, "pretrained": False}).to(DEV).eval()
        plain.meta["in_channels"], plain.meta["out_channels"] = 3, plain.meta["outputdim"]
        general = extract_ms(plain, four, ms, plain.pool.p_value())
        monkeypatch.delenv("MDIR_AMD_FUSED_TAIL")
        fused = extract_ms(plain, four, ms, plain.pool.p_value())
        np.testing.assert_array_equal(general.cpu().numpy(), fused.cpu().numpy())          # "bit-identical to the general route"


def test_compute_map_on_a_non_contiguous_device_ranking(golden):
    """evaluate.py:39-111 on a device ranking handed over as [N,Q] with its OWN memory layout (a copy, not the transposed view of
    the [Q,N] matrix the sort writes): rows are made contiguous once; same result as the host route."""
    from mdir_amd.evaluate import compute_map
    rng = np.random.default_rng(4)
    n, nq = 3000, 6
    ranks = np.stack([rng.permutation(n) for _ in range(nq)], axis=1).astype(np.int64)          # [N,Q], C-contiguous
    gnd = [{"ok": rng.choice(n, 7, replace=False), "junk": rng.choice(n, 3, replace=False)} for _ in range(nq)]
    want = O.compute_map(ranks, gnd, kappas=[1, 5, 10])
    got = compute_map(dev(ranks), gnd, kappas=[1, 5, 10])
    np.testing.assert_allclose(got[0], want[0], rtol=0, atol=1e-12)
    np.testing.assert_allclose(got[1], want[1], rtol=0, atol=1e-12)
    np.testing.assert_allclose(got[3], want[3], rtol=0, atol=1e-12)


_REFUSAL_SCRIPT = r"""
import sys, warnings
import torch
sys.path.insert(0, %(root)r)
from mdir_amd import graphs
from mdir_amd.graphs import ShapeGraphs
dev = "cuda:0"
ok = ShapeGraphs(lambda t: t * 2 + 1, warmup=0)
a = torch.rand(3, 5, device=dev)
assert torch.equal(ok(a), a * 2 + 1) and ok.captures == 1

def syncing(x):
    return x * float(x.sum().item() > -1e30)               # .item(): a host read inside the function
sg = ShapeGraphs(syncing, warmup=0)
x = torch.rand(4, 8, device=dev)
with warnings.catch_warnings(record=True) as seen:
    warnings.simplefilter("always")
    y = sg(x)                                               # the capture is refused, the SAME call answers eagerly
assert any("capture refused" in str(w.message) for w in seen), [str(w.message) for w in seen]
assert torch.equal(y, x) and len(sg.refused) == 1 and sg.captures == 0 and graphs._captures_off
assert torch.equal(sg(x), x) and sg.replays == 0            # stays eager, silently
z = torch.rand(4, 8, device=dev)                            # PyTorch's generator works again (its capture flag was left set)
assert torch.equal(sg(z), z)
b = torch.rand(3, 5, device=dev)
assert torch.equal(ok(b), b * 2 + 1) and ok.replays == 2    # a graph captured BEFORE the refusal keeps replaying
later = ShapeGraphs(lambda t: t - 1, warmup=0)
c = torch.rand(6, 7, device=dev)
assert torch.equal(later(c), c - 1) and later.captures == 0  # no further capture is attempted in this process
torch.cuda.synchronize()
print("REFUSAL_OK")
"""


def test_graph_bookkeeping_refusal_and_eviction():
    """mdir_amd/graphs.py: beyond `max_graphs` shapes the least recently used graph goes; few images left of a shape
    (`upcoming` < PAYOFF_IMAGES) are not worth a capture; a function that cannot be captured (it synchronises with the host) is
    refused with a warning, the same call and all later ones answer eagerly with the eager result, earlier graphs keep
    replaying and no further capture is attempted (own process: PyTorch cannot capture again after a failed capture)."""
    from mdir_amd.graphs import ShapeGraphs
    small = ShapeGraphs(lambda t: t * 2 + 1, warmup=0, max_graphs=2)
    shapes = [(2, 3), (4, 5), (6, 7)]
    for s in shapes:
        t = torch.rand(s, device=DEV)
        assert torch.equal(small(t), t * 2 + 1)
    assert small.captures == 3 and len(small.graphs) == 2 and (tuple(shapes[0]), torch.float32, 0) not in small.graphs
    t = torch.rand(shapes[0], device=DEV)
    assert torch.equal(small(t), t * 2 + 1) and small.captures == 4           # captured again after its eviction
    few = ShapeGraphs(lambda t: t + 1, warmup=0)
    few.upcoming = 3
    assert torch.equal(few(t), t + 1) and few.captures == 0                  # too few images left: eager
    proc = subprocess.run([sys.executable, "-c", _REFUSAL_SCRIPT % {"root": ROOT}], capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0 and "REFUSAL_OK" in proc.stdout, (proc.stdout[-2000:], proc.stderr[-4000:])


def test_loader_routes_what_the_device_decoder_does_not_take(tmp_path, monkeypatch):
    """genericdataset.py:44-70 with the device JPEG route on: a PNG, a CMYK JPEG and a progressive JPEG are not
    candidates for mdx_jpeg_* (`_coefficients` answers None, nothing is raised) and come out of the loader exactly as Pillow
    decodes them; a baseline JPEG goes through the device and equals Pillow too.  jpeg.box_on_device / resample.on_device:
    the cases the device declines."""
    from PIL import Image
    from mdir_amd import jpeg, resample
    from mdir_amd.datasets import ImagesFromList, ToUint8HWC
    rng = np.random.default_rng(7)
    arr = rng.integers(0, 255, (90, 130, 3), dtype=np.uint8)
    Image.fromarray(arr).save(str(tmp_path / "a.png"))
    Image.fromarray(arr).save(str(tmp_path / "b.jpg"), quality=92)
    Image.fromarray(arr).save(str(tmp_path / "c.jpg"), quality=92, progressive=True)
    Image.fromarray(arr).convert("CMYK").save(str(tmp_path / "d.jpg"), quality=92)
    (tmp_path / "e.jpg").write_bytes(b"")
    names = [str(tmp_path / n) for n in ("a.png", "b.jpg", "c.jpg", "d.jpg")]
    ds = ImagesFromList("", names + [str(tmp_path / "e.jpg")], imsize=64, transform=ToUint8HWC())
    took = [ds._coefficients(i) for i in range(5)]
    assert took[0] is None and took[1] is not None and took[3] is None and took[4] is None
    assert jpeg.entropy_decode(b"") is None
    assert jpeg.box_on_device(None, 10, 10) and jpeg.box_on_device((0, 0, 10, 10), 10, 10)
    assert not jpeg.box_on_device((0, 0, 11, 10), 10, 10) and not jpeg.box_on_device((5, 5, 5, 9), 10, 10) and not jpeg.box_on_device((-1, 0, 4, 4), 10, 10)
    assert resample.on_device(100, 80, 200) is None and resample.on_device(100, 80, 100) is None      # nothing to shrink
    assert resample.on_device(4000, 3000, 64) is None                                                 # Pillow reduces by an integer factor first
    assert resample.on_device(100, 80, 50) == (50, 40)
    assert resample.on_device(3, 1000, 200) is None or isinstance(resample.on_device(3, 1000, 200), tuple)
    img = rng.integers(0, 256, (47, 61, 3), dtype=np.uint8)
    assert np.array_equal(resample._host_resample(img, (61, 47)), img)                               # both axes already at size
    for i, name in enumerate(names):
        with Image.open(name) as im:
            want = im.convert("RGB")
            want.thumbnail((64, 64), Image.LANCZOS)
            want = np.asarray(want)
        got = ds[i]
        got = got.numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
        assert got.shape == want.shape, name
        np.testing.assert_array_equal(got, want, err_msg=name)


def test_argument_checks_of_the_python_shim():
    """SURVEY section 8(b) "Errors": the C ABI returns a status, the Python shim raises the matching exception BEFORE any launch --
    RuntimeError for host tensors (no CPU fallback), TypeError for a wrong dtype, ValueError for shapes the kernels do not take."""
    from mdir_amd import ops
    f = lambda *s: torch.rand(*s, device=DEV)
    u8 = torch.zeros((1, 8, 8, 3), dtype=torch.uint8, device=DEV)
    ix = ops.DescriptorIndex(f(32, 16), "ND")
    closed = ops.DescriptorIndex(f(32, 16), "ND")
    closed.close()
    i32 = lambda *s: torch.zeros(s, dtype=torch.int32, device=DEV)
    cases = [
        (RuntimeError, lambda: ops.pool_l2n(torch.rand(1, 4, 3, 3))),
        (TypeError, lambda: ops.pool_l2n(f(1, 4, 3, 3).double())),
        (ValueError, lambda: ops.pool_l2n(f(1, 4, 3, 6)[:, :, :, ::2])),
        (ValueError, lambda: ops.pool_l2n(f(4, 3, 3))),
        (ValueError, lambda: ops.l2n_rows_(f(4))),
        (ValueError, lambda: ops.l2n_rows_(f(2, 4), bias=f(3))),
        (ValueError, lambda: ops.ms_aggregate([])),
        (ValueError, lambda: ops.ms_aggregate([f(4), f(5)])),
        (ValueError, lambda: ops.ms_aggregate_batch([f(2, 4)] * 9)),
        (ValueError, lambda: ops.ms_aggregate_batch([f(2, 4), f(3, 4)])),
        (ValueError, lambda: ops.pool_multi([])),
        (ValueError, lambda: ops.pool_multi([f(4, 3, 3)])),
        (ValueError, lambda: ops.pool_multi([f(1, 4, 3, 3), f(1, 5, 2, 2)])),
        (ValueError, lambda: ops.l2n_aggregate(f(2, 4))),
        (ValueError, lambda: ops.u8_to_chw(u8.cpu(), [0.5] * 3, [0.2] * 3)),
        (ValueError, lambda: ops.u8_to_chw(u8, [0.5] * 2, [0.2] * 3)),
        (ValueError, lambda: ops.clahe_u8_to_chw(u8[..., :2].contiguous(), 4, 8, [0.5] * 3, [0.2] * 3)),
        (ValueError, lambda: ops.clahe_u8_to_chw(u8, 4, 8, [0.5] * 2, [0.2] * 3)),
        (ValueError, lambda: ops.bilinear_pyramid(f(3, 8, 8), [0.5])),
        (ValueError, lambda: ops.bilinear_pyramid(f(1, 3, 8, 8), [0.9, 0.8, 0.7, 0.6, 0.5, 0.4, 0.3, 0.2, 0.15])),
        (ValueError, lambda: ops.resample_u8(u8.cpu(), 0, i32(4, 2), i32(4, 3))),
        (ValueError, lambda: ops.resample_u8(u8, 2, i32(4, 2), i32(4, 3))),
        (ValueError, lambda: ops.resample_u8(u8, 0, i32(4, 3), i32(4, 3))),
        (ValueError, lambda: ops.bn_act_(f(4, 3, 3), f(4), f(4))),
        (ValueError, lambda: ops.bn_act_(f(1, 4, 3, 3), f(4), None)),
        (ValueError, lambda: ops.bn_act_(f(1, 4, 3, 3), f(4), f(4), residual=f(1, 4, 3, 2))),
        (ValueError, lambda: ops.conv1x1_bn_act(f(4, 3, 3), f(4, 8), f(8), f(8), f(8), f(8), 1e-5, None, True)),
        (ValueError, lambda: ops.conv1x1_bn_act(f(1, 4, 3, 3), f(5, 8), f(8), f(8), f(8), f(8), 1e-5, None, True)),
        (ValueError, lambda: ops.conv1x1_bn_act(f(1, 4, 3, 3), f(4, 8), f(7), f(8), f(8), f(8), 1e-5, None, True)),
        (ValueError, lambda: ops.conv1x1_bn_act(f(1, 4, 3, 3), f(4, 8), f(8), f(8), f(8), f(8), 1e-5, f(1, 8, 3, 2), True)),
        (RuntimeError, lambda: closed.scores(f(2, 16), "ND")),
        (ValueError, lambda: ix.scores(f(2, 16), "ND", center=f(15))),
        (ValueError, lambda: ix.scores(f(2, 16), "ND", out=f(3, 32))),
        (ValueError, lambda: ix.scores(f(2, 16), "sideways")),
        (ValueError, lambda: ix.scores(f(2, 2, 16), "ND")),
        (ValueError, lambda: ops.rank_full_segments([f(2, 5), f(3, 5)])),
        (ValueError, lambda: ops.scores_rowmajor(f(2, 3, 16), f(2, 16), "ND")),
        (ValueError, lambda: ops.scores_rowmajor(f(32, 16), f(2, 12), "ND")),
        (ValueError, lambda: ops.scores_rowmajor(f(32, 16), f(2, 16), "ND", center=f(3))),
        (ValueError, lambda: ops.scores_rowmajor(f(32, 16), f(2, 16), "ND", out=f(3, 32))),
        (ValueError, lambda: ops.rank_positions(torch.zeros((2, 4), dtype=torch.int32, device=DEV), [[0], [1]])),
        (ValueError, lambda: ops.rank_positions(torch.zeros((2, 4), dtype=torch.int64, device=DEV), [[0]])),
        (ValueError, lambda: ops.gram_f64(f(4).double())),
        (ValueError, lambda: ops.gram_f64(f(4, 9).double(), f(3).double())),
        (ValueError, lambda: ops.project_f64(f(4, 5).double(), f(4, 9).double())),
        (ValueError, lambda: ops.project_f64(f(4, 5).double(), f(5, 9).double(), f(4).double())),
    ]
    for i, (exc, call) in enumerate(cases):
        with pytest.raises(exc):
            call()
        assert True, i
    ix.close()
    # what is legal at the edges: empty batches come back empty, a pyramid of the identity scale is the input itself
    assert ops.u8_to_chw(u8[:0], [0.5] * 3, [0.2] * 3).shape[0] == 0
    assert ops.bilinear_pyramid(f(1, 3, 8, 8), [1])[0].shape == (1, 3, 8, 8)
    e = f(0, 4, 3, 3)
    assert ops.bn_act_(e, f(4), f(4)) is e


def test_remaining_device_branches(tmp_path, monkeypatch):
    """What was left of profiles/r05_host_branches.md on the device side: the GeM MODULE called on its own (pooling.py:36-47),
    `embed` without a device argument (cirtorch_format/test.py:40-41: `net.cuda()`), more than 32 peer blocks into the ranking
    (concatenated, same result), an architecture the backbones do not know, the communicator's argument checks."""
    from PIL import Image
    from mdir_amd import cirtorch_format as C
    from mdir_amd import layers, ops
    from mdir_amd.backbones import build_features
    from mdir_amd.networks import init_network
    from mdir_amd.sharded import HipBackend
    x = torch.rand(2, 16, 5, 7, device=DEV) + 0.1
    gem = layers.GeM(p=2.5).to(DEV)
    want = x.clamp(min=1e-6).pow(2.5).mean(dim=(2, 3)).pow(1 / 2.5)
    np.testing.assert_allclose(gem(x).cpu().numpy().reshape(2, 16), want.cpu().numpy(), rtol=2e-5)
    # embed(device=None) moves the network to the GPU itself
    monkeypatch.setenv("MDIR_AMD_WORKERS", "0")
    rng = np.random.default_rng(5)
    (tmp_path / "ims").mkdir()
    imgs = ["a%d.jpg" % i for i in range(3)]
    for name in imgs:
        Image.fromarray(rng.integers(0, 255, (120, 160, 3), dtype=np.uint8)).save(tmp_path / "ims" / name, format="JPEG")
    torch.manual_seed(2)
    net = init_network({"architecture": "alexnet", "pooling": "gem", "whitening": False, "pretrained": False})
    meta = {"architecture": "alexnet", "pooling": "gem", "whitening": False, "mean": net.meta["mean"], "std": net.meta["std"],
            "outputdim": 256, "local_whitening": False, "regional": False}
    torch.save({"meta": meta, "state_dict": net.state_dict()}, str(tmp_path / "up.pth"))
    res = C.embed({"net": str(tmp_path / "up.pth"), "imgdir": str(tmp_path / "ims"), "image_size": 128, "multiscale": True}, (imgs,))
    on_dev = C.embed({"net": str(tmp_path / "up.pth"), "imgdir": str(tmp_path / "ims"), "image_size": 128, "multiscale": True}, (imgs,), device=DEV)
    assert res[1] == imgs and res[2].shape == (3, 256)
    np.testing.assert_allclose(res[2], on_dev[2], rtol=0, atol=1e-6)
    np.testing.assert_allclose(np.linalg.norm(res[2], axis=1), 1.0, atol=1e-4)
    # 40 peer blocks: beyond the segment table (32) the blocks are concatenated; same ranking as the dense call
    sc = (rng.standard_normal((3, 4000)) * 0.03).astype(np.float32)
    sc[:, 100] = sc[:, 7]
    blocks = [dev(sc[:, 100 * g:100 * (g + 1)]) for g in range(40)]
    np.testing.assert_array_equal(HipBackend().rank_full_segments(blocks, 5).cpu().numpy(), OC.rank_full(sc) + 5)
    np.testing.assert_array_equal(HipBackend().rank_full_segments(blocks[:32], 0).cpu().numpy(), OC.rank_full(sc[:, :3200]))
    with pytest.raises(ValueError, match="1..32 blocks"):
        ops.rank_full_segments(blocks)
    with pytest.raises(ValueError, match="Unsupported or unknown architecture"):
        build_features("resnet7")
    with pytest.raises(ValueError, match="need one id list per query"):
        ops.rank_of(dev(sc), [[1], [2]])
    # the communicator (one rank: all a one-GPU box can hold)
    with pytest.raises(ValueError, match="128 bytes"):
        ops.Comm(b"short", 1, 0, DEV)
    comm = ops.Comm(ops.Comm.unique_id(), 1, 0, DEV)
    with pytest.raises(ValueError, match="one width per rank"):
        comm.allgather_scores(dev(sc), [4000, 10])
    with pytest.raises(ValueError, match="local block is"):
        comm.allgather_scores(dev(sc), [3999])
    with pytest.raises(ValueError, match="local block is"):
        comm.exchange_scores(dev(sc), [3999])
    comm.close()
    comm.close()                                            # closing twice is harmless
    assert ops.conv1x1_bn_act(torch.rand(0, 16, 3, 3, device=DEV), torch.rand(16, 32, device=DEV), torch.rand(32, device=DEV),
                              torch.rand(32, device=DEV)).shape == (0, 32, 3, 3)


def test_rmac_on_the_device(golden):
    """mdx_rmac (LF.rmac, functional.py:26-72) against golden G17 at the path's map sizes, against the oracle on odd shapes, and
    inside a network: `pooling: rmac` end to end == l2n(rmac(features))."""
    from conftest import sparse_map
    from mdir_amd import layers, ops
    from mdir_amd.networks import init_network
    g = golden("g17_rmac.npz")
    for c, h, w, b in [(2048, 24, 32, 1), (512, 48, 64, 1), (64, 17, 23, 2), (256, 7, 5, 1), (16, 3, 40, 2), (8, 12, 12, 1), (4, 2, 2, 1)]:
        x = sparse_map(int(g["seed_c%d_h%d_w%d_b%d" % (c, h, w, b)]), (b, c, h, w))
        for L in (3, 2):
            got = layers.rmac(dev(x), L=L).cpu().numpy().reshape(b, c)
            np.testing.assert_allclose(got, g["rmac_c%d_h%d_w%d_b%d_L%d" % (c, h, w, b, L)], rtol=2e-6, atol=2e-6)
    rng = np.random.default_rng(0)
    for b, c, h, w in [(3, 70, 33, 9), (1, 1, 1, 1), (2, 300, 5, 64), (1, 2048, 12, 16)]:
        x = (rng.standard_normal((b, c, h, w)) * (rng.random((b, c, h, w)) > 0.5)).astype(np.float32)      # negative values too
        np.testing.assert_allclose(layers.RMAC()(dev(x)).cpu().numpy().reshape(b, c), O.rmac(x, 3, 1e-6), rtol=2e-6, atol=2e-6)
    zero = layers.rmac(torch.zeros(1, 8, 6, 6, device=DEV))
    assert not torch.isnan(zero).any() and float(zero.abs().max()) == 0.0        # eps is added to every norm
    with pytest.raises(ValueError, match="regions"):
        ops.rmac(torch.rand(1, 4, 3, 3, device=DEV), [])
    with pytest.raises(Exception, match="outside"):
        ops.rmac(torch.rand(1, 4, 3, 3, device=DEV), [(0, 0, 3, 3), (1, 1, 3, 3)])
    torch.manual_seed(0)
    net = init_network({"architecture": "resnet18", "pooling": "rmac", "whitening": False, "pretrained": False}).to(DEV).eval()
    x = torch.rand(2, 3, 224, 160, device=DEV)
    with torch.no_grad():
        got = net(x)
        want = O.l2n(O.rmac(net.features(x).cpu().numpy(), 3, 1e-6), 1e-6)
    np.testing.assert_allclose(got.t().cpu().numpy(), want, rtol=1e-5, atol=1e-6)


def test_regional_pooling_on_the_device(golden):
    """mdx_roipool + mdx_l2n_rows + the regional whitening on the weight shard + mdx_region_sum = Rpool.forward (pooling.py:62-95):
    golden G18 (GeM / MAC / SPoC regions, with and without whitening, aggregated and per region) and a `regional: True` network."""
    from conftest import sparse_map
    from mdir_amd import layers
    from mdir_amd.networks import init_network
    SHAPES, TOL, DEVICE = [(64, 24, 32), (32, 17, 23), (16, 7, 5), (8, 12, 12)], 2e-5, DEV
    to_dev = dev

    g = golden("g18_rpool.npz")
    for c, h, w in SHAPES:
        x = to_dev(sparse_map(int(g["seed_c%d_h%d_w%d" % (c, h, w)]), (2, c, h, w)))
        for name, mod in (("gem", layers.GeM(p=2.5)), ("mac", layers.MAC()), ("spoc", layers.SPoC())):
            for tag in ("plain", "whiten"):
                lin = None
                if tag == "whiten":
                    lin = torch.nn.Linear(c, c)
                    lin.load_state_dict({"weight": torch.from_numpy(g["weight_c%d" % c]), "bias": torch.from_numpy(g["bias_c%d" % c])})
                rp = layers.Rpool(mod, lin).to(x.device)
                with torch.no_grad():
                    agg, reg = rp(x), rp(x, aggregate=False)
                assert tuple(agg.shape) == (2, c, 1, 1) and tuple(reg.shape[:1]) == (2,) and tuple(reg.shape[2:]) == (c, 1, 1)
                np.testing.assert_allclose(agg.cpu().numpy().reshape(2, c), g["agg_%s_%s_c%d_h%d_w%d" % (name, tag, c, h, w)], rtol=TOL, atol=2e-6)
                np.testing.assert_allclose(reg.cpu().numpy()[..., 0, 0], g["reg_%s_%s_c%d_h%d_w%d" % (name, tag, c, h, w)], rtol=TOL, atol=2e-6)
    assert repr(layers.Rpool(layers.MAC())).endswith("(L=3)")
    torch.manual_seed(0)
    net = init_network({"architecture": "alexnet", "pooling": "gem", "regional": True, "whitening": False, "pretrained": False}).to(DEVICE).eval()
    assert isinstance(net.pool, layers.Rpool) and set(k for k in net.state_dict() if k.startswith("pool.")) == {"pool.rpool.p", "pool.whiten.weight", "pool.whiten.bias"}
    assert net.meta["regional"] is True and net.fusable_tail() is None
    xin = torch.rand(2, 3, 130, 97, device=DEVICE)
    with torch.no_grad():
        got = net(xin)
        feat = net.features(xin).cpu().numpy()
    want = O.l2n(O.rpool(feat, lambda a: O.gem(a, 3.0, 1e-6), net.pool.whiten.weight.detach().cpu().numpy(), net.pool.whiten.bias.detach().cpu().numpy()), 1e-6)
    np.testing.assert_allclose(got.t().cpu().numpy(), want, rtol=1e-4, atol=2e-6)


def test_whitenapply_in_float64_and_random_map_problems_on_the_device():
    """whitenapply (whiten.py:4-12) with the float64 `P` / `m` that whitenlearn pickles: float64 on the f64 matrix cores, equal to
    numpy to 1e-12 (fp32 inputs keep the fp32 route).  compute_map on a DEVICE ranking (mdx_rank_positions) over the 300 random
    problems of golden G19: the reference's values."""
    import copy
    from conftest import GOLDEN
    from mdir_amd.evaluate import compute_map
    from mdir_amd.whiten import whitenapply
    rng = np.random.default_rng(8)
    for d, n, dims in ((64, 500, None), (300, 1025, 128), (2048, 300, None), (17, 3, 5)):
        X, m, P = rng.standard_normal((d, n)), rng.standard_normal((d, 1)) * 0.01, rng.standard_normal((d, d)) / np.sqrt(d)
        kept = P[:dims] if dims else P
        want = kept @ (X - m)
        want = want / (np.linalg.norm(want, ord=2, axis=0, keepdims=True) + 1e-6)
        got = whitenapply(X.copy(), m, P, dims)
        assert got.dtype == np.float64 and got.shape == want.shape
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)
        got32 = whitenapply(X.astype(np.float32), m.astype(np.float32), P.astype(np.float32), dims)
        assert got32.dtype == np.float32
        np.testing.assert_allclose(got32, want, rtol=0, atol=5e-6)
        mixed = whitenapply(X.astype(np.float32), m, P, dims)          # fp32 descriptors, float64 Lw (examples/test.py:246-249): float64
        assert mixed.dtype == np.float64
        np.testing.assert_allclose(mixed, (lambda y: y / (np.linalg.norm(y, ord=2, axis=0, keepdims=True) + 1e-6))(kept @ (X.astype(np.float32) - m)),
                                   rtol=0, atol=1e-12)
    sys.path.insert(0, GOLDEN)
    try:
        from make_golden import fuzz_map_case
    finally:
        sys.path.remove(GOLDEN)
    g = np.load(os.path.join(GOLDEN, "g19_map_fuzz.npz"))
    checked = 0
    for seed in range(300):
        ranks, gnd, kappas = fuzz_map_case(seed)
        if "error_%d" % seed in g:
            continue
        mAP, aps, pr, prs = compute_map(dev(ranks), copy.deepcopy(gnd), list(kappas))
        np.testing.assert_allclose(mAP, g["map_%d" % seed][0], rtol=0, atol=1e-12, err_msg=str(seed))
        np.testing.assert_allclose(np.asarray(aps, dtype=np.float64), g["aps_%d" % seed], rtol=0, atol=1e-12, err_msg=str(seed))
        np.testing.assert_allclose(np.asarray(prs, dtype=np.float64), g["prs_%d" % seed], rtol=0, atol=1e-12, err_msg=str(seed))
        checked += 1
    assert checked > 250


def test_regional_and_local_vectors_on_the_device(tmp_path, monkeypatch):
    """extract_regional_vectors / extract_local_vectors (imageretrievalnet.py:325-384) with device=None (the network moves to the GPU,
    as upstream): per image the oracle's regional vectors of the device's own feature map and its per-location L2N."""
    from PIL import Image
    from mdir_amd.datasets import Compose, ImagesFromList, Normalize, ToTensor
    from mdir_amd.networks import extract_local_vectors, extract_regional_vectors, init_network
    monkeypatch.setenv("MDIR_AMD_WORKERS", "0")
    torch.manual_seed(0)
    net = init_network({"architecture": "alexnet", "pooling": "gem", "regional": True, "whitening": False, "pretrained": False})
    net.meta["out_channels"] = 256
    rng = np.random.default_rng(1)
    paths = []
    for i, (w, h) in enumerate(((150, 110), (97, 160), (64, 64))):
        paths.append(str(tmp_path / ("i%d.png" % i)))
        Image.fromarray(rng.integers(0, 255, (h, w, 3), dtype=np.uint8)).save(paths[-1])
    tr = Compose([ToTensor(), Normalize(net.meta["mean"], net.meta["std"])])
    reg = extract_regional_vectors(net, paths, 128, tr)
    loc = extract_local_vectors(net, paths, 128, tr)
    assert next(net.parameters()).is_cuda
    for i, p in enumerate(paths):
        x = ImagesFromList("", [p], imsize=128, transform=tr)[0][None].to(DEV)
        with torch.no_grad():
            feat = net.features(x).cpu().numpy()
        want = O.rpool(feat, lambda a: O.gem(a, 3.0, 1e-6), net.pool.whiten.weight.detach().cpu().numpy(), net.pool.whiten.bias.detach().cpu().numpy(),
                       aggregate=False)[0]
        np.testing.assert_allclose(reg[i].numpy(), want.T, rtol=1e-4, atol=2e-6)
        rows = feat[0].reshape(256, -1)
        np.testing.assert_allclose(loc[i].numpy(), rows / (np.linalg.norm(rows, axis=0, keepdims=True) + 1e-6), rtol=1e-5, atol=1e-7)


def test_whitening_stages_on_the_device(golden):
    """mdir/stages/whiten.py on the default device: `learn_lw_whitening` on golden G12's descriptors and pairs gives the reference's
    (m, P) (rows up to sign) through mdx_gram_f64 / mdx_project_f64; `whiten` with that float64 whitening equals numpy to 1e-12."""
    from mdir_amd import stages
    g = golden("g12_whitenlearn.npz")
    X, q, p = g["X"], g["qidxs"], g["pidxs"]                  # [D, N] float32 descriptors, pair indices
    names = ["v%d" % i for i in range(X.shape[1])]
    meta, Lw = stages.learn_lw_whitening({}, (names, X.T.copy(), [names[i] for i in q], [names[i] for i in p]))
    assert meta["stats"]["failed_times"] == 0 and "gpu" in meta["resource_usage"]
    np.testing.assert_allclose(Lw["m"], g["m_lw"], rtol=1e-9, atol=1e-12)
    sign = np.sign(np.sum(Lw["P"] * g["P_lw"], axis=1, keepdims=True))
    assert np.abs(Lw["P"] * sign - g["P_lw"]).max() <= 1e-7 * np.abs(g["P_lw"]).max()
    _, _, white = stages.whiten({"dimensions": None}, (Lw, names, X.T.copy()))
    want = Lw["P"] @ (X.astype(np.float64) - Lw["m"])
    want = (want / (np.linalg.norm(want, ord=2, axis=0, keepdims=True) + 1e-6)).T
    np.testing.assert_allclose(white, want, rtol=0, atol=1e-12)
    _, pca = stages.learn_pca_whitening({}, (X.T.copy(),))
    sign = np.sign(np.sum(pca["P"] * g["P_pca"], axis=1, keepdims=True))
    assert np.abs(pca["P"] * sign - g["P_pca"]).max() <= 1e-7 * np.abs(g["P_pca"]).max()
    # paste_pca_normalize (stages/whiten.py:90-118) on the device: the numpy statements on the same matrices
    rng = np.random.default_rng(6)
    a, b = rng.standard_normal((500, 40)), rng.standard_normal((500, 24))
    _, out = stages.paste_pca_normalize({"dimensions": 16}, [a.copy(), b.copy()])
    v = np.concatenate([a, b], axis=1)
    v = v - np.mean(v)
    w, vec = np.linalg.eigh(v.T @ v)
    vecs = vec[:, np.argsort(w)[-16:]]
    proj = v @ (vecs @ vecs.T)
    np.testing.assert_allclose(out, proj / np.linalg.norm(proj, axis=1, keepdims=True), rtol=1e-8, atol=1e-10)


def test_learn_whitening_moves_the_network_itself(tmp_path, monkeypatch):
    """cirtorch_format/test.py:92-152 with no device argument (`net.cuda()`, :131): a training set on disk -> the supervised whitening,
    equal to the run with an explicit device."""
    from PIL import Image
    from mdir_amd import cirtorch_format as C
    from mdir_amd.networks import init_network
    monkeypatch.setenv("MDIR_AMD_WORKERS", "0")
    monkeypatch.setenv("CIRTORCH_ROOT", str(tmp_path))
    rng = np.random.default_rng(3)
    root = tmp_path / "data" / "train" / "toyset"
    cids = ["%032x" % i for i in range(24)]
    for cid in cids:
        path = C.cid2filename(cid, str(root / "ims"))
        os.makedirs(os.path.dirname(path), exist_ok=True)
        Image.fromarray(rng.integers(0, 255, (200, 260, 3), dtype=np.uint8)).save(path, format="JPEG")
    import pickle
    with open(root / "toyset-whiten.pkl", "wb") as f:
        pickle.dump({"cids": cids, "qidxs": list(range(8)), "pidxs": list(range(8, 16))}, f)
    torch.manual_seed(2)
    net = init_network({"architecture": "alexnet", "pooling": "gem", "whitening": False, "pretrained": False})
    meta = {"architecture": "alexnet", "pooling": "gem", "whitening": False, "mean": net.meta["mean"], "std": net.meta["std"], "outputdim": 256,
            "local_whitening": False, "regional": False}
    torch.save({"meta": meta, "state_dict": net.state_dict()}, str(tmp_path / "up.pth"))
    args = {"net": str(tmp_path / "up.pth"), "whitening": "toyset", "image_size": 224, "multiscale": True}
    _, a = C.learn_whitening(dict(args), ())
    _, b = C.learn_whitening(dict(args), (), device=DEV)
    # float32 descriptors in -> float32 (m, P) out, as numpy gives the reference (test.py:241-268 hands `wvecs.numpy()` to whitenlearn:
    # ADVICE round 5); computed in float64 on the device either way
    assert a["P"].shape == (256, 256) and a["P"].dtype == np.float32 and a["m"].dtype == np.float32
    np.testing.assert_allclose(a["m"], b["m"], rtol=0, atol=1e-9)
    sign = np.sign(np.sum(a["P"] * b["P"], axis=1, keepdims=True))
    assert np.abs(a["P"] - b["P"] * sign).max() <= 1e-6 * np.abs(a["P"]).max()
