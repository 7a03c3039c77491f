"""The host API against the REAL library on an MI355X, plus full-size property checks.

Small cases compare with outputs of the reference (tests/golden) and the oracle; the
BASELINE.json-size cases (N = 1 004 993, Q = 70, D = 2048) use size-independent
properties and oracle checks on samples."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import ROOT
from oracle import chain as OC
from oracle import oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _toy_net(g):
    from mdir_amd.layers import GeM
    from mdir_amd.networks import ImageRetrievalNet
    conv = nn.Conv2d(3, g["conv_w"].shape[0], 3, stride=2, padding=1)
    conv.load_state_dict({"weight": torch.from_numpy(g["conv_w"]), "bias": torch.from_numpy(g["conv_b"])})
    c = g["conv_w"].shape[0]
    meta = {"architecture": "toy", "local_whitening": False, "pooling": "gem", "regional": False,
            "whitening": False, "mean": [0, 0, 0], "std": [1, 1, 1], "outputdim": c, "in_channels": 3, "out_channels": c}
    return ImageRetrievalNet([conv, nn.ReLU(inplace=True)], None, GeM(p=float(g["gem_p"])), None, meta).eval().to(DEV)


def test_wrapper_chain_on_gpu_matches_reference(golden, tmp_path):
    from mdir_amd.networks import extract_ms
    from mdir_amd.wrapper import initialize_wrappers
    g = golden("g6_chain.npz")
    net = _toy_net(g)
    pkl = str(tmp_path / "whiten.pkl")
    with open(pkl, "wb") as f:
        pickle.dump({"P": g["P"], "m": g["m"]}, f)
    img = torch.from_numpy(g["img"])
    with torch.no_grad():
        chain = initialize_wrappers({"0_cirwhiten": {"whitening": pkl, "dimensions": None},
                                     "1_cirmultiscale": {"scales": True}}, DEV)
        out = chain(img.clone(), net)
        assert out.is_cuda
        np.testing.assert_allclose(out.cpu().numpy(), g["chain_out"], rtol=2e-5, atol=1e-6)
        chain32 = initialize_wrappers({"0_cirwhiten": {"whitening": pkl, "dimensions": 32},
                                       "1_cirmultiscale": {"scales": True}}, DEV)
        np.testing.assert_allclose(chain32(img.clone(), net).cpu().numpy(), g["chain_out_dims32"], rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(net(img.to(DEV)).cpu().numpy(), g["single_scale_out"], rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(extract_ms(net, img.to(DEV), [1, 1. / np.sqrt(2), 1. / 2], 2.5).cpu().numpy(),
                                   g["extract_ms_out"], rtol=2e-5, atol=1e-6)


def test_forward_tail_and_whitenapply_on_gpu(golden):
    from mdir_amd.layers import GeM, gem, l2n
    from mdir_amd.networks import ImageRetrievalNet
    from mdir_amd.whiten import whitenapply
    g = golden("g3_tail.npz")
    C = g["w"].shape[0]
    meta = {"architecture": "toy", "local_whitening": False, "pooling": "gem", "regional": False,
            "whitening": True, "mean": [0, 0, 0], "std": [1, 1, 1], "outputdim": C}
    lin = nn.Linear(C, C)
    lin.load_state_dict({"weight": torch.from_numpy(g["w"]), "bias": torch.from_numpy(g["b"])})
    for p in (3.0, 2.92):
        net = ImageRetrievalNet([nn.Identity()], None, GeM(p=p), lin, dict(meta)).eval().to(DEV)
        with torch.no_grad():
            np.testing.assert_allclose(net(dev(g["feat"])).cpu().numpy(), g[f"out_whiten_p{p}"], rtol=1e-5, atol=2e-7)
    x = dev(g["feat"])
    np.testing.assert_allclose(l2n(gem(x, p=torch.ones(1) * 3)).cpu().numpy().reshape(2, C).T, g["out_plain_p3.0"],
                               rtol=1e-5, atol=1e-7)
    w = golden("g5_whiten.npz")
    for dims in (None, 48):
        got = whitenapply(w["X"], w["m"].astype(np.float32), w["P"].astype(np.float32), dims)
        np.testing.assert_allclose(got, w[f"whitenapply_f32_dims{dims}"], rtol=1e-5, atol=2e-7)


def test_eval_py_end_to_end(tmp_path):
    """./eval.py on a generated roxford5k + 247tokyo1k set-up; the printed numbers equal an
    independent oracle pipeline (torch-CPU backbone + numpy tail + reference statements)."""
    root = str(tmp_path / "synth")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_synthetic_eval.py"), root])
    env = dict(os.environ, CIRTORCH_ROOT=root, MDIR_AMD_WORKERS="2")
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "eval.py"), "eval.yml", os.path.join(root, "eval_synth.yml")],
                          env=env, text=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    out = proc.stdout
    assert proc.returncode == 0, out[-3000:]
    printed = {}
    for line in out.splitlines():
        for label in ("roxford.5k medium", "247tokyo.1k"):
            if line.strip().startswith(label):
                printed[label] = float(line.split()[-1])
    assert set(printed) == {"roxford.5k medium", "247tokyo.1k"}, out

    # independent pipeline on the CPU
    from mdir_amd.datasets import ImagesFromList, configdataset, initialize_transforms
    from mdir_amd.network import load_checkpoint
    from mdir_amd.networks import init_network
    import torch.nn.functional as F
    state = load_checkpoint(os.path.join(root, "net.pth"))["net"]
    model = init_network({"architecture": "alexnet", "pretrained": False})
    model.load_state_dict(state["model_state"])
    model.eval()
    p = float(model.pool.p)
    wh = pickle.load(open(os.path.join(root, "whiten.pkl"), "rb"))
    tr = initialize_transforms("pil2np | totensor | normalize", [model.meta["mean"], model.meta["std"]])

    def describe(paths, bbxs):
        vecs = []
        for img in ImagesFromList("", paths, imsize=320, bbxs=bbxs, transform=tr):
            per = []
            for s in O.MS_SCALES:
                x = img[None] if s == 1 else F.interpolate(img[None], scale_factor=float(s), mode="bilinear",
                                                          align_corners=False)
                with torch.no_grad():
                    per.append(O.l2n(O.gem(model.features(x).numpy(), p))[0])
            vecs.append(O.whiten_wrapper(O.ms_aggregate(np.stack(per), p), wh["m"], wh["P"]))
        return np.stack(vecs, axis=1)

    want = {}
    for ds, label, key in (("roxford5k", "roxford.5k medium", "ap_medium"), ("247tokyo1k", "247tokyo.1k", "ap")):
        cfg = configdataset(ds, os.path.join(root, "data", "test"))
        ims = [cfg["im_fname"](cfg, i) for i in range(cfg["n"])]
        qims = [cfg["qim_fname"](cfg, i) for i in range(cfg["nq"])]
        bbxs = [tuple(g["bbx"]) if g.get("bbx") else None for g in cfg["gnd"]]
        vecs = describe(ims, None)
        qvecs = vecs.copy() if ims == qims and set(bbxs) == {None} else describe(qims, bbxs)
        _, per = O.compute_map_and_print(ds, O.ranks(O.scores(vecs, qvecs)), cfg["gnd"])
        want[label] = round(100 * O.nanmean_metric(per[key]), 2)
    assert printed == want, (printed, want, out)


# --------------------------------------------------------------- full-size properties

@pytest.fixture(scope="module")
def big():
    """rOxford5k+1M shaped problem resident on the GPU (BASELINE.json configs[2])."""
    from mdir_amd import ops
    n, nq, d = 1_004_993, 70, 2048
    g = torch.Generator(device=DEV)
    g.manual_seed(0)
    rows = torch.empty((n, d), dtype=torch.float32, device=DEV)
    for s in range(0, n, 65536):
        e = min(n, s + 65536)
        blk = torch.randn((e - s, d), generator=g, device=DEV)
        rows[s:e] = blk / blk.norm(dim=1, keepdim=True)
    qid = torch.randperm(n, generator=torch.Generator().manual_seed(1))[:nq]
    q = rows[qid.to(DEV)] + 0.05 * torch.randn((nq, d), generator=g, device=DEV)
    q /= q.norm(dim=1, keepdim=True)
    rows[123] = rows[77]          # exact duplicates -> tied scores
    rows[900_000] = rows[77]
    rows[5] = 0.0                 # zero vector -> score exactly 0
    ix = ops.DescriptorIndex(rows, "ND")
    sc = ix.scores(q.contiguous(), "ND")
    rk = ops.rank_full(sc)
    torch.cuda.synchronize()
    return {"rows": rows, "q": q, "qid": qid, "ix": ix, "sc": sc, "rk": rk, "n": n, "nq": nq}


def test_full_size_scores_bit_exact_on_samples(big):
    """Sampled database rows: GPU scores == fmaf-chain oracle, bit for bit."""
    rng = np.random.default_rng(0)
    ids = np.unique(np.concatenate([rng.choice(big["n"], 3000, replace=False), [0, 5, 77, 123, 900_000, big["n"] - 1],
                                    big["qid"].numpy()]))
    sub = big["rows"][torch.from_numpy(ids).to(DEV)].cpu().numpy()
    want = OC.scores_chain(np.ascontiguousarray(sub.T), np.ascontiguousarray(big["q"].cpu().numpy().T))
    got = big["sc"][:, torch.from_numpy(ids).to(DEV)].cpu().numpy()
    np.testing.assert_array_equal(got, want)
    assert np.all(got[:, list(ids).index(5)] == 0.0)
    ref = O.scores(np.ascontiguousarray(sub.T), np.ascontiguousarray(big["q"].cpu().numpy().T)).T
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-5)        # vs the reference's BLAS: north-star tolerance


def test_full_size_in_place_product_and_device_ranking_map(big):
    """At the headline size: mdx_scores_rowmajor (the row-major matrix read where it lies) gives the index route's 70 M scores
    bit for bit (those are pinned to the chain oracle above), and compute_map on the device ranking (mdx_rank_positions)
    equals compute_map from the scores (mdx_rank_of) for random labelled sets."""
    from mdir_amd import ops
    from mdir_amd.evaluate import compute_map, compute_map_from_scores
    got = ops.scores_rowmajor(big["rows"], big["q"].contiguous(), "ND")
    assert torch.equal(got, big["sc"])
    del got
    rng = np.random.default_rng(4)
    gnd = [{"ok": rng.choice(big["n"], 25, replace=False), "junk": rng.choice(big["n"], 10, replace=False)} for _ in range(big["nq"])]
    gnd[3]["ok"] = np.array([77, 123, 900_000, 5], dtype=np.int64)          # tied rows and the zero vector
    a = compute_map(big["rk"].t(), gnd, [1, 5, 10])
    b = compute_map_from_scores(big["sc"], gnd, [1, 5, 10])
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("mode", ["split3", "split2"])
def test_full_size_split_modes_on_the_same_shard(big, mode):
    """MDX_F32_SPLIT3 / MDX_F32_SPLIT2 at BASELINE's full size, on the index the exact tests use (no second copy): every score within the
    summation-order bound 2e-6 of the exact chain (all 70 M of them, on the device), sampled rows as close to the float64 dot
    product as the chain is, exact zeros stay zeros, duplicated rows still tie exactly, and the ranking of the split scores
    differs from the exact ranking only between rows whose EXACT scores are closer than the bound."""
    from mdir_amd import ops
    sc, rk, n, nq = big["sc"], big["rk"], big["n"], big["nq"]
    s3 = big["ix"].scores(big["q"].contiguous(), "ND", compute=mode)
    assert float((s3 - sc).abs().max()) <= 2e-6
    assert bool((s3[:, 5] == 0).all()) and bool((s3[:, 123] == s3[:, 77]).all()) and bool((s3[:, 900_000] == s3[:, 77]).all())
    rng = np.random.default_rng(3)
    ids = np.unique(np.concatenate([rng.choice(n, 2000, replace=False), big["qid"].numpy()]))
    sub = big["rows"][torch.from_numpy(ids).to(DEV)].cpu().numpy().astype(np.float64)
    exact = big["q"].cpu().numpy().astype(np.float64) @ sub.T
    err3 = np.abs(s3[:, torch.from_numpy(ids).to(DEV)].cpu().numpy() - exact).max()
    errc = np.abs(sc[:, torch.from_numpy(ids).to(DEV)].cpu().numpy() - exact).max()
    assert err3 <= max(2.0 * errc, 3e-7), (err3, errc)
    k = 1000
    ids3, _ = ops.topk(s3, k)
    differ = torch.nonzero(ids3 != rk[:, :k])
    if len(differ):
        qq = differ[:, 0]
        gap = (sc[qq, ids3[qq, differ[:, 1]]] - sc[qq, rk[qq, differ[:, 1]]]).abs().max()
        assert float(gap) <= 2e-6
    assert len(differ) < 0.01 * ids3.numel() and bool((ids3[:, 0].cpu() == big["qid"]).all())


def test_full_size_ranking_properties(big):
    sc, rk, n, nq = big["sc"], big["rk"], big["n"], big["nq"]
    # a permutation of 0..n-1 per query
    srt, _ = torch.sort(rk, dim=1)
    assert bool((srt == torch.arange(n, device=DEV)[None, :]).all())
    del srt
    # non-increasing scores; inside a run of equal scores ids ascend (the documented tie rule)
    g = torch.gather(sc, 1, rk)
    assert bool((g[:, 1:] <= g[:, :-1]).all())
    tie = g[:, 1:] == g[:, :-1]
    assert int(tie.sum()) >= nq                                     # rows 77 / 123 / 900000 tie for every query
    assert bool((rk[:, 1:][tie] > rk[:, :-1][tie]).all())
    # every query retrieves its (noisy) source row first
    assert bool((rk[:, 0].cpu() == big["qid"]).all())
    # ALL 70 rankings against the C oracle, bit-exact (it sorts a million-row column in ~50 ms)
    for q0 in range(0, nq, 14):
        np.testing.assert_array_equal(rk[q0:q0 + 14].cpu().numpy(), OC.rank_full(sc[q0:q0 + 14].cpu().numpy()))


def test_full_size_topk_rank_of_and_map(big):
    from mdir_amd import ops
    from mdir_amd.evaluate import compute_map, compute_map_from_scores
    sc, rk, n, nq = big["sc"], big["rk"], big["n"], big["nq"]
    ids, vals = ops.topk(sc, 1000)
    assert bool((ids == rk[:, :1000]).all())
    assert bool((vals == torch.gather(sc, 1, rk[:, :1000])).all())
    gnd = O.synth_gnd(nq, 4993, seed=1)
    gnd[3]["easy"], gnd[3]["hard"] = np.array([77, 123]), np.array([900_000 % 4993])   # tied items among the labels
    lists = [np.concatenate([g["easy"], g["hard"], g["junk"]]) for g in gnd]
    pos, _, off = ops.rank_of(sc, lists)
    inv = torch.empty_like(rk)
    inv.scatter_(1, rk, torch.arange(n, device=DEV)[None, :].expand(nq, n))
    for qi in range(nq):
        want = inv[qi][torch.from_numpy(lists[qi]).to(DEV)]
        assert bool((pos[off[qi]:off[qi + 1]] == want).all())
    gm = O.protocol_gnd(gnd, "medium")
    a = compute_map(rk.t(), gm, [1, 5, 10])
    b = compute_map_from_scores(sc, gm, [1, 5, 10])
    for x, y in zip(a, b):
        np.testing.assert_array_equal(np.asarray(x), np.asarray(y))


def test_full_size_shards_equal_whole(big):
    """Row-sharding is exact: per-shard scores concatenate to the full matrix and per-shard
    counts add up to the global positions (what ShardedIndex does across GPUs)."""
    from mdir_amd import ops
    from mdir_amd.sharded import shard_bounds
    sc, rk, n, nq = big["sc"], big["rk"], big["n"], big["nq"]
    G = 3
    lists = [np.array([77, 123, 900_000, 5, int(big["qid"][qi])]) for qi in range(nq)]
    ids_t, off_t, off = ops._csr(lists, torch.device(DEV))
    ref = ops.gather_scores(sc, ids_t, off_t)
    cnt = torch.zeros(ids_t.numel(), dtype=torch.int64, device=DEV)
    for r in range(G):
        lo, hi = shard_bounds(n, G, r)
        shard = ops.DescriptorIndex(big["rows"][lo:hi], "ND", row_offset=lo)
        s_local = shard.scores(big["q"].contiguous(), "ND")
        assert bool((s_local == sc[:, lo:hi]).all())
        ops.rank_count_(cnt, s_local, lo, ref, ids_t, off_t)
        shard.close()
    inv = torch.empty_like(rk)
    inv.scatter_(1, rk, torch.arange(n, device=DEV)[None, :].expand(nq, n))
    for qi in range(nq):
        assert bool((cnt[off[qi]:off[qi + 1]] == inv[qi][ids_t[off[qi]:off[qi + 1]]]).all())
    # what a rank of an 8-GPU job sorts: its 9 queries against the 8 peer blocks, read in place as segments
    blocks = [sc[:9, lo:hi].contiguous() for lo, hi in (shard_bounds(n, 8, r) for r in range(8))]
    assert bool((ops.rank_full_segments(blocks) == rk[:9]).all())


def test_extraction_with_device_thumbnail_equals_host_thumbnail(tmp_path, monkeypatch):
    """The loader's LANCZOS down-scale on the device (default) against the same list shrunk by Pillow in the workers
    (MDIR_AMD_GPU_RESIZE=0): same pixels, hence the same descriptors; JPEG files of two orientations with a crop box,
    batches and single images, graphs and eager."""
    from PIL import Image
    from mdir_amd import ops
    from mdir_amd.datasets import initialize_transforms
    from mdir_amd.graphs import ShapeGraphs
    from mdir_amd.networks import extract_vectors_device, init_network
    rng = np.random.default_rng(12)
    paths, bbxs = [], []
    for i in range(11):
        w, h = (400, 300) if i % 3 else (300, 400)
        low = rng.integers(0, 255, (h // 20 + 1, w // 20 + 1, 3)).astype(np.float32)
        img = np.clip(np.kron(low, np.ones((20, 20, 1), dtype=np.float32))[:h, :w] + rng.normal(0, 10, (h, w, 3)), 0, 255)
        p = str(tmp_path / ("im%d.jpg" % i))
        Image.fromarray(img.astype(np.uint8)).save(p, quality=90)
        paths.append(p)
        bbxs.append((10.5, 19.6, 390.4, 280.5) if i == 4 else None)       # fractional corners, as rOxford's query boxes have
    torch.manual_seed(2)
    net = init_network({"architecture": "resnet18", "pooling": "gem", "whitening": False, "pretrained": False}).to(DEV).eval()
    tr = initialize_transforms("pil2np | totensor | normalize", [net.meta["mean"], net.meta["std"]])
    monkeypatch.setattr(ShapeGraphs, "PAYOFF_IMAGES", 0)
    monkeypatch.setenv("MDIR_AMD_WORKERS", "2")
    for name, value in (("MDIR_AMD_LOADER", "threads"), ("MDIR_AMD_GPU_JPEG", "1"), ("MDIR_AMD_GPU_RESIZE", "1")):
        monkeypatch.setenv(name, value)                                  # the defaults, whatever the caller's environment says
    calls, decoded = [], []
    real = ops.resample_u8
    monkeypatch.setattr(ops, "resample_u8", lambda *a: (calls.append(a[1]), real(*a))[1])
    from mdir_amd import networks
    real_px = networks.pixels_of
    monkeypatch.setattr(networks, "pixels_of", lambda item, dev: (decoded.append(item.size), real_px(item, dev))[1])
    on_dev = extract_vectors_device(net, paths, 224, tr, bbxs=bbxs, ms=[1, 0.5], msp=1.0, device=DEV)
    assert calls.count(0) >= 3 and calls.count(1) >= 3                   # width and height passes ran (eager + captured)
    assert len(decoded) == len(paths)                                    # every JPEG went through the device half of the decoder
    ncalls = len(calls)
    monkeypatch.setenv("MDIR_AMD_GPU_RESIZE", "0")
    on_host = extract_vectors_device(net, paths, 224, tr, bbxs=bbxs, ms=[1, 0.5], msp=1.0, device=DEV)
    assert len(calls) == ncalls and len(decoded) == len(paths)            # Pillow decoded and shrank these
    np.testing.assert_allclose(on_dev.cpu().numpy(), on_host.cpu().numpy(), rtol=0, atol=2e-6)    # MIOpen is not run-to-run bit-stable
    monkeypatch.setenv("MDIR_AMD_GPU_RESIZE", "1")
    monkeypatch.setenv("MDIR_AMD_GRAPHS", "0")
    eager = extract_vectors_device(net, paths, 224, tr, bbxs=bbxs, ms=[1, 0.5], msp=1.0, device=DEV)
    np.testing.assert_allclose(eager.cpu().numpy(), on_host.cpu().numpy(), rtol=0, atol=2e-6)


def test_hard_negative_mining_on_gpu(golden):
    """f1 on the device (mdx_scores + mdx_topk + the walk): golden G13 = the reference's create_epoch_tuples on a toy
    pool, then a mining-sized problem against the oracle restatement that G13 pins."""
    from mdir_amd.mining import search_hard_negatives
    g = golden("g13_mining.npz")
    for prefix in (None, 3):
        nidxs, ndist = search_hard_negatives(dev(g["qvecs"]), dev(g["poolvecs"]), g["idxs2images"], g["clusters"].tolist(),
                                             g["qidxs"].tolist(), int(g["nnum"]), prefix=prefix)
        assert nidxs == g["nidxs"].tolist()
        np.testing.assert_allclose(ndist, g["ndist"], rtol=1e-5)
    rng = np.random.default_rng(9)
    D, P, Q, nimg = 512, 20000, 300, 60000
    pool = rng.standard_normal((P, D)).astype(np.float32)
    pool /= np.linalg.norm(pool, axis=1, keepdims=True)
    qv = pool[rng.choice(P, Q, replace=False)] + 0.05 * rng.standard_normal((Q, D)).astype(np.float32)
    qv /= np.linalg.norm(qv, axis=1, keepdims=True)
    idxs2images = rng.permutation(nimg)[:P]
    clusters = rng.integers(0, 700, nimg).tolist()
    qidxs = rng.choice(nimg, Q, replace=False).tolist()
    qvecs, poolvecs = np.ascontiguousarray(qv.T), np.ascontiguousarray(pool.T)
    got, gd = search_hard_negatives(dev(qvecs), dev(poolvecs), idxs2images, clusters, qidxs, 5)
    want, wd = O.hard_negatives(qvecs, poolvecs, idxs2images, clusters, qidxs, 5)
    assert got == want
    np.testing.assert_allclose(gd, wd, rtol=1e-4)


def test_infer_and_whitening_learning_on_gpu(tmp_path, golden):
    """f2 + f3 on the device.  infer stage -> float64 embeddings, NaN row for a missing file, every other row equal to
    extract_vectors of the same network (EmbeddingOutput itself is pinned by golden G14 in the CPU suite);
    whitenlearn / pcawhitenlearn on the device reproduce the reference's (m, P) of golden G12."""
    from mdir_amd import stages
    from mdir_amd.datasets import initialize_transforms
    from mdir_amd.network import CirNetwork, SingleNetwork
    from mdir_amd.networks import extract_vectors, init_network
    from mdir_amd.stages import EmbeddingOutput
    from mdir_amd.whiten import pcawhitenlearn, whitenlearn
    from test_host_api import _write_images
    rng = np.random.default_rng(3)
    names = ["im%02d" % i for i in range(12)]
    _write_images(str(tmp_path / "imgs"), names, rng, size=(224, 160))
    torch.manual_seed(0)
    model_params = {"architecture": "cirnet", "cir_architecture": "alexnet", "local_whitening": False,
                    "pooling": "gem", "regional": False, "whitening": False, "pretrained": True}
    model = init_network({"architecture": "alexnet", "pretrained": False})
    model.meta["in_channels"], model.meta["out_channels"] = 3, 256
    runtime = {"wrappers": "cirmultiscale:True", "data": {"transforms": "pil2np | totensor | normalize"}}
    net = CirNetwork(model, SingleNetwork.NetworkParams(model_params, runtime), "cpu", frozen=True)
    ckpt = str(tmp_path / "net.pth")
    torch.save(net.state_dict()["net"], ckpt)
    images = [n + ".jpg" for n in names[:5]] + ["missing.jpg"] + [n + ".jpg" for n in names[5:]]
    params = {"network": {"path": ckpt, "runtime": {}},
              "data": {"test": {"dataset": {"name": "CirImageList", "image_dir": str(tmp_path / "imgs"),
                                            "image_size": 224, "ignore_errors": True}}},
              "output": {"inference": {"name": "embedding"}}}
    import os
    os.environ["MDIR_AMD_WORKERS"] = "2"
    meta, imgs_out, vecs = stages.infer(params, (images,))
    assert imgs_out == images and vecs.shape == (13, 256) and vecs.dtype == np.float64 and np.isnan(vecs[5]).all()
    good = np.delete(vecs, 5, axis=0)
    np.testing.assert_allclose(np.linalg.norm(good, axis=1), 1.0, atol=1e-5)
    tr = initialize_transforms("pil2np | totensor | normalize", net.network_params.runtime["data"]["mean_std"])
    gpu_net = stages.load_network(params["network"], DEV).eval()
    with torch.no_grad():
        want = extract_vectors(gpu_net, [str(tmp_path / "imgs" / (n + ".jpg")) for n in names], 224, tr, device=DEV).numpy()
    np.testing.assert_allclose(good, want.T, rtol=0, atol=2e-6)
    # EmbeddingOutput fed with DEVICE tensors gives the golden matrix
    g14 = golden("g14_embedding_output.npz")
    out = EmbeddingOutput(([str(x) for x in g14["names"]],), {})
    for i in (0, 2, 3):
        out.add(i, True, dev(g14["vec"][i]))
    out.add(1, None, None)
    np.testing.assert_array_equal(out.postprocess()[1], g14["result"])

    g = golden("g12_whitenlearn.npz")
    up = lambda a, b: a * np.sign(np.sum(a * b, axis=1, keepdims=True))
    m, P = whitenlearn(g["X"], g["qidxs"], g["pidxs"])                   # device = "cuda": float64 GEMMs on the GPU
    np.testing.assert_allclose(m, g["m_lw"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(up(P, g["P_lw"]), g["P_lw"], rtol=1e-7, atol=1e-9)
    m2, P2 = pcawhitenlearn(g["X"], shrink=8)
    np.testing.assert_allclose(up(np.real(P2), g["P_pca_shrink8"]), g["P_pca_shrink8"], rtol=1e-7, atol=1e-9)
    # a retrieval-sized problem: the learned projection whitens the matching-pair differences
    rng = np.random.default_rng(0)
    D, N, npairs = 256, 6000, 2500
    basis = np.linalg.qr(rng.standard_normal((D, D)))[0] * np.geomspace(3.0, 0.05, D)
    X = basis @ rng.standard_normal((D, N))
    X /= np.linalg.norm(X, axis=0, keepdims=True)
    qidxs, pidxs = rng.choice(N, npairs, replace=False), rng.choice(N, npairs, replace=False)
    m, P = whitenlearn(X, qidxs, pidxs)
    mo, Po = O.whitenlearn(X, qidxs, pidxs)                              # pinned by G12
    np.testing.assert_allclose(up(P, Po), Po, rtol=1e-5, atol=1e-7)
    dfw = P @ (X[:, qidxs] - X[:, pidxs])
    np.testing.assert_allclose(dfw @ dfw.T / npairs, np.eye(D), atol=1e-8)


def test_graph_replay_extraction_equals_eager(tmp_path, monkeypatch):
    """extract_vectors_device replays one hipGraph per input shape from the third image of that shape
    on; descriptors equal the eager run of the same images (two shapes, multi-scale + whitening chain
    and plain cirtorch multi-scale)."""
    from PIL import Image
    from mdir_amd import networks
    from mdir_amd.datasets import initialize_transforms
    from mdir_amd.graphs import ShapeGraphs
    from mdir_amd.networks import extract_vectors_device, init_network
    rng = np.random.default_rng(3)
    paths = []
    for i in range(16):
        size = (160, 120) if i % 5 else (120, 160)
        p = str(tmp_path / ("im%d.png" % i))
        Image.fromarray(rng.integers(0, 255, (size[1], size[0], 3), dtype=np.uint8)).save(p)
        paths.append(p)
    torch.manual_seed(1)
    net = init_network({"architecture": "resnet18", "pooling": "gem", "whitening": False, "pretrained": False}).to(DEV).eval()
    tr = initialize_transforms("pil2np | totensor | normalize", [net.meta["mean"], net.meta["std"]])
    ms = [1, 2 ** -0.5, 0.5]
    made = []
    orig = ShapeGraphs.__init__

    def spy(self, *a, **k):
        orig(self, *a, **k)
        made.append(self)
    monkeypatch.setattr(ShapeGraphs, "__init__", spy)
    monkeypatch.setattr(ShapeGraphs, "PAYOFF_IMAGES", 0)      # capture even for a dozen images (default: only when >= 32 follow)
    monkeypatch.setenv("MDIR_AMD_WORKERS", "0")
    monkeypatch.setenv("MDIR_AMD_BATCH", "4")                 # the counts below are those of batches of four (default: 8, then 4)
    from mdir_amd.networks import _same_shape_order
    assert _same_shape_order(paths, None) == [0, 5, 10, 15, 1, 2, 3, 4, 6, 7, 8, 9, 11, 12, 13, 14]      # equal sizes consecutive
    graphed = extract_vectors_device(net, paths, 160, tr, ms=ms, msp=net.pool.p_value(), device=DEV)
    # 4 images of one size = one batch of 4 (eager); 12 of the other = three batches (eager, replay, replay)
    assert len(made) == 1 and made[0].replays == 2 and len(made[0].graphs) == 1 and not made[0].refused
    monkeypatch.setenv("MDIR_AMD_BATCH", "1")
    single = extract_vectors_device(net, paths, 160, tr, ms=ms, msp=net.pool.p_value(), device=DEV)
    assert len(made) == 2 and made[1].replays == 16 - 2 and len(made[1].graphs) == 2 and not made[1].refused
    np.testing.assert_allclose(graphed.cpu().numpy(), single.cpu().numpy(), rtol=0, atol=2e-6)
    # the default: batches of eight, what is left of a size as one batch of four, then one by one (12 = 8 + 4, 4 = 4)
    monkeypatch.delenv("MDIR_AMD_BATCH")
    by8 = extract_vectors_device(net, paths, 160, tr, ms=ms, msp=net.pool.p_value(), device=DEV)
    np.testing.assert_allclose(by8.cpu().numpy(), single.cpu().numpy(), rtol=0, atol=2e-6)
    n_made = len(made)
    monkeypatch.setenv("MDIR_AMD_GRAPHS", "0")
    eager = extract_vectors_device(net, paths, 160, tr, ms=ms, msp=net.pool.p_value(), device=DEV)
    assert len(made) == n_made
    np.testing.assert_allclose(graphed.cpu().numpy(), eager.cpu().numpy(), rtol=0, atol=2e-6)
    # GPU-side `pil2np | totensor | normalize` (uint8 through the loader) == the host transform chain
    assert tr.device_tail() is not None
    monkeypatch.setenv("MDIR_AMD_GPU_PREPROCESS", "0")
    host = extract_vectors_device(net, paths, 160, tr, ms=ms, msp=net.pool.p_value(), device=DEV)
    np.testing.assert_allclose(host.cpu().numpy(), eager.cpu().numpy(), rtol=0, atol=1e-6)   # MIOpen is not run-to-run bit-stable
    from mdir_amd import ops
    from mdir_amd.datasets import ToUint8HWC
    pic = Image.open(paths[0])
    got = ops.u8_to_chw(ToUint8HWC()(pic)[None].to(DEV), *tr.device_tail())
    np.testing.assert_array_equal(got.cpu().numpy()[0], tr(pic).numpy())                 # bit-identical arithmetic


def test_eval_py_two_processes_print_the_same_numbers(tmp_path):
    """`torchrun --nproc-per-node 2 eval.py ...` (both ranks on this one GPU, collectives through
    gloo): sharded extraction + sort-free distributed mAP print what the single process prints."""
    import socket
    root = str(tmp_path / "synth")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_synthetic_eval.py"), root])
    env = dict(os.environ, CIRTORCH_ROOT=root, MDIR_AMD_WORKERS="0")
    args = [os.path.join(ROOT, "eval.py"), "eval.yml", os.path.join(root, "eval_synth.yml")]

    def printed(cmd, extra):
        proc = subprocess.run(cmd, env=dict(env, **extra), text=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert proc.returncode == 0, proc.stdout[-3000:]
        out = {}
        for line in proc.stdout.splitlines():
            for label in ("roxford.5k medium", "247tokyo.1k"):
                if line.strip().startswith(label):
                    out[label] = float(line.split()[-1])
        assert set(out) == {"roxford.5k medium", "247tokyo.1k"}, proc.stdout[-3000:]
        return out

    single = printed([sys.executable] + args, {})
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    double = printed([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                      "--master-addr", "127.0.0.1", "--master-port", str(port)] + args, {"MDIR_AMD_DRYRUN_ONE_GPU": "1"})
    assert single == double, (single, double)


