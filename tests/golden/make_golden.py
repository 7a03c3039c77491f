#!/usr/bin/env python3
"""Generate tests/golden/*.npz|json by RUNNING THE REFERENCE in the build container.

The reference (jenicek/mdir + its vendored cirtorch) is imported from
/root/reference with throw-away stub modules for the three third-party packages
this image lacks (torchvision, cv2, h5py); none of its source is copied.  Only
inputs (or the seeds that regenerate them) and the reference's outputs are
stored.  Re-run:  python tests/golden/make_golden.py

The fixtures are DATA; the GPU box never sees /root/reference.
"""
import io
import json
import os
import pickle
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Anything:
    def __init__(self, *a, **k):
        pass


def import_reference():
    tv = _stub("torchvision", get_image_backend=lambda: "PIL")
    tv.models = _stub("torchvision.models")
    tr = _stub("torchvision.transforms", Compose=_Anything, ToTensor=_Anything, RandomCrop=_Anything,
               RandomHorizontalFlip=_Anything, CenterCrop=_Anything)
    tr.functional = _stub("torchvision.transforms.functional")
    tv.transforms = tr
    _stub("cv2", setNumThreads=lambda n: None)
    _stub("h5py")
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "mdir", "external"))
    import cirtorch  # noqa: F401
    import mdir  # noqa: F401


def sparse_map(seed, shape):
    """ReLU-like feature map: U(0,1) with about half the entries zeroed."""
    rng = np.random.default_rng(seed)
    x = rng.random(shape, dtype=np.float32)
    x *= (rng.random(shape, dtype=np.float32) > 0.5)
    return x


def unit_rows(rng, n, d):
    v = rng.standard_normal((n, d)).astype(np.float32)
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def main():
    import_reference()
    import cirtorch.layers.functional as LF
    from cirtorch.layers.pooling import GeM
    from cirtorch.networks.imageretrievalnet import ImageRetrievalNet, extract_ms
    from cirtorch.utils.evaluate import compute_ap, compute_map, compute_map_and_print
    from cirtorch.utils.whiten import whitenapply
    from mdir.components.data import wrapper as W
    from daan.core.experiments import dict_deep_overlay
    import torch.nn as nn
    import torch.nn.functional as F

    torch.manual_seed(0)
    torch.set_num_threads(1)
    out = {}

    # ---- G1 gem / mac / spoc ------------------------------------------------
    g1 = {}
    cases = [(2048, 24, 32), (2048, 17, 23), (512, 48, 64), (256, 7, 5)]
    for ci, (c, h, w) in enumerate(cases):
        seed = 100 + ci
        x = sparse_map(seed, (1, c, h, w))
        for p in (3.0, 2.2, 1.0):
            y = LF.gem(torch.from_numpy(x), p=torch.ones(1) * p, eps=1e-6)
            g1[f"gem_c{c}_h{h}_w{w}_p{p}"] = y.squeeze().numpy()
        g1[f"mac_c{c}_h{h}_w{w}"] = LF.mac(torch.from_numpy(x)).squeeze().numpy()
        g1[f"spoc_c{c}_h{h}_w{w}"] = LF.spoc(torch.from_numpy(x)).squeeze().numpy()
        g1[f"seed_c{c}_h{h}_w{w}"] = np.int64(seed)
    g1["x_c256_h7_w5"] = sparse_map(103, (1, 256, 7, 5))
    np.savez_compressed(os.path.join(HERE, "g1_pool.npz"), **g1)

    # ---- G2 l2n ---------------------------------------------------------------
    rng = np.random.default_rng(7)
    x = rng.standard_normal((4, 512)).astype(np.float32)
    x[2] = 0.0
    x[3] *= 1e-7
    np.savez_compressed(os.path.join(HERE, "g2_l2n.npz"), x=x,
                        y=LF.l2n(torch.from_numpy(x)).numpy())

    # ---- G3 forward tail with in-network whitening -------------------------
    rng = np.random.default_rng(8)
    C = 128
    feat = sparse_map(9, (2, C, 9, 11))
    lin = nn.Linear(C, C, bias=True)
    with torch.no_grad():
        lin.weight.copy_(torch.from_numpy(rng.standard_normal((C, C)).astype(np.float32) / 16))
        lin.bias.copy_(torch.from_numpy(rng.standard_normal(C).astype(np.float32) / 16))
    meta = {"architecture": "toy", "local_whitening": False, "pooling": "gem", "regional": False,
            "whitening": True, "mean": [0, 0, 0], "std": [1, 1, 1], "outputdim": C, "out_channels": C}
    g3 = {"feat": feat, "w": lin.weight.detach().numpy(), "b": lin.bias.detach().numpy()}
    for p in (3.0, 2.92):
        net = ImageRetrievalNet([nn.Identity()], None, GeM(p=p), lin, dict(meta)).eval()
        with torch.no_grad():
            g3[f"out_whiten_p{p}"] = net(torch.from_numpy(feat)).numpy()
        net = ImageRetrievalNet([nn.Identity()], None, GeM(p=p), None, dict(meta)).eval()
        with torch.no_grad():
            g3[f"out_plain_p{p}"] = net(torch.from_numpy(feat)).numpy()
    np.savez_compressed(os.path.join(HERE, "g3_tail.npz"), **g3)

    # ---- G4 multi-scale aggregation ---------------------------------------
    rng = np.random.default_rng(10)
    vs = np.abs(unit_rows(rng, 3, 2048))
    vs /= np.linalg.norm(vs, axis=1, keepdims=True)
    g4 = {"vecs": vs}
    for msp in (1.0, 3.0, 2.92):
        t = [torch.from_numpy(v.copy()).unsqueeze(1) for v in vs]
        g4[f"agg_msp{msp}"] = W.CirMultiscaleAggregation.aggregate_tensor(t, 3, 2048, msp).numpy()
    np.savez_compressed(os.path.join(HERE, "g4_aggregate.npz"), **g4)

    # ---- G5 whitening --------------------------------------------------------
    rng = np.random.default_rng(2)
    D = 128
    qmat, _ = np.linalg.qr(rng.standard_normal((D, D)))
    P64 = (qmat * rng.uniform(0.5, 2.0, size=(1, D))).T.copy()
    m64 = rng.normal(0, 0.01, size=(D, 1))
    X = unit_rows(rng, 40, D).T.copy()  # [D,N]
    g5 = {"P": P64, "m": m64, "X": X}
    with tempfile.TemporaryDirectory() as tmp:
        pkl = os.path.join(tmp, "whiten.pkl")
        with open(pkl, "wb") as f:
            pickle.dump({"P": P64, "m": m64}, f)
        for dims in (None, 48):
            wr = W.CirtorchWhiten(pkl, dims, device="cpu")
            cols = [wr.postprocess(torch.from_numpy(X[:, i].copy()), None, None).numpy()
                    for i in range(X.shape[1])]
            g5[f"wrapper_dims{dims}"] = np.stack(cols, axis=1)
            g5[f"whitenapply_f64_dims{dims}"] = whitenapply(X.astype(np.float64), m64, P64, dims)
            g5[f"whitenapply_f32_dims{dims}"] = whitenapply(X, m64.astype(np.float32),
                                                            P64.astype(np.float32), dims)
    np.savez_compressed(os.path.join(HERE, "g5_whiten.npz"), **g5)

    # ---- G6 whole wrapper chain on a toy network (+ G10 interpolate) -------
    torch.manual_seed(11)
    Cout = 64
    conv = nn.Conv2d(3, Cout, 3, stride=2, padding=1)
    feats = [conv, nn.ReLU(inplace=True)]
    meta6 = {"architecture": "toy", "local_whitening": False, "pooling": "gem", "regional": False,
             "whitening": False, "mean": [0, 0, 0], "std": [1, 1, 1], "outputdim": Cout,
             "in_channels": 3, "out_channels": Cout}
    net6 = ImageRetrievalNet(feats, None, GeM(p=2.5), None, meta6).eval()
    rng = np.random.default_rng(12)
    img = rng.standard_normal((1, 3, 75, 107)).astype(np.float32)
    q6, _ = np.linalg.qr(rng.standard_normal((Cout, Cout)))
    P6 = (q6 * rng.uniform(0.5, 2.0, size=(1, Cout))).T.copy()
    m6 = rng.normal(0, 0.01, size=(Cout, 1))
    g6 = {"conv_w": conv.weight.detach().numpy(), "conv_b": conv.bias.detach().numpy(),
          "img": img, "P": P6, "m": m6, "gem_p": np.float32(2.5)}
    with tempfile.TemporaryDirectory() as tmp, torch.no_grad():
        pkl = os.path.join(tmp, "whiten.pkl")
        with open(pkl, "wb") as f:
            pickle.dump({"P": P6, "m": m6}, f)
        chain = W.initialize_wrappers({"0_cirwhiten": {"whitening": pkl, "dimensions": None},
                                       "1_cirmultiscale": {"scales": True}}, "cpu")
        g6["chain_out"] = chain(torch.from_numpy(img.copy()), net6).numpy()
        chain32 = W.initialize_wrappers({"0_cirwhiten": {"whitening": pkl, "dimensions": 32},
                                         "1_cirmultiscale": {"scales": True}}, "cpu")
        g6["chain_out_dims32"] = chain32(torch.from_numpy(img.copy()), net6).numpy()
        ms_only = W.initialize_wrappers("cirmultiscale:True", "cpu")
        g6["ms_only_out"] = ms_only(torch.from_numpy(img.copy()), net6).numpy()
        g6["single_scale_out"] = net6(torch.from_numpy(img.copy())).numpy()
        # upstream cirtorch multi-scale path (extract_ms) with the same scales
        scales = [1, 1. / np.sqrt(2), 1. / 2]
        g6["extract_ms_out"] = extract_ms(net6, torch.from_numpy(img.copy()), scales, 2.5).numpy()
        for si, s in enumerate(scales[1:], start=1):
            g6[f"interp_s{si}"] = F.interpolate(torch.from_numpy(img), scale_factor=s, mode="bilinear",
                                                align_corners=False).numpy()
    g6["interp_size_1024x768"] = np.array(
        [list(F.interpolate(torch.zeros(1, 1, 768, 1024), scale_factor=s, mode="bilinear",
                            align_corners=False).shape[2:]) for s in scales], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "g6_chain.npz"), **g6)

    # ---- G7 ranking -----------------------------------------------------------
    # Small cases stored whole; they are built so that neighbouring scores of every
    # query differ by > 1e-6, i.e. tie-free under any fp32 summation order.
    g7 = {}
    for name, (d, n, q) in {"a": (32, 300, 7), "b": (256, 120, 8), "c": (64, 100, 100)}.items():
        seed = 20
        while True:
            rng = np.random.default_rng(seed)
            db = unit_rows(rng, n, d)
            if name == "c":
                qv = db.copy()  # query == database shortcut (cirscore.py:56-57)
            else:
                qv = db[rng.choice(n, q, replace=False)] + 0.05 * rng.standard_normal((q, d)).astype(np.float32)
                qv /= np.linalg.norm(qv, axis=1, keepdims=True)
            vecs, qvecs = np.ascontiguousarray(db.T), np.ascontiguousarray(qv.T)
            sc = np.dot(vecs.T, qvecs)
            gaps = np.diff(np.sort(sc.astype(np.float64), axis=0), axis=0)
            if gaps.min() > 1e-6:
                break
            seed += 1
        rk = np.argsort(-sc, axis=0)
        g7[f"{name}_vecs"], g7[f"{name}_qvecs"] = vecs, qvecs
        g7[f"{name}_scores"], g7[f"{name}_ranks"] = sc, rk.astype(np.int32)
        g7[f"{name}_mingap"] = np.float64(gaps.min())
        g7[f"{name}_seed"] = np.int64(seed)
    # tie fixture: duplicated and zero rows -> runs of exactly equal scores
    rng = np.random.default_rng(33)
    db = unit_rows(rng, 64, 32)
    db[10] = db[3]; db[40] = db[3]; db[41] = db[3]; db[20] = 0; db[21] = 0
    db = np.round(db * 64) / 64  # exactly representable -> order-independent sums
    qv = db[[3, 20, 5]]
    sc = np.dot(db, qv.T).astype(np.float32)
    g7["tie_vecs"], g7["tie_qvecs"] = np.ascontiguousarray(db.T), np.ascontiguousarray(qv.T)
    g7["tie_scores"] = sc
    g7["tie_ranks_numpy_default"] = np.argsort(-sc, axis=0).astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "g7_ranking.npz"), **g7)

    # rOxford-shaped case (D=2048, N=4993, Q=70): inputs regenerated from the seed by
    # oracle.synth_ranking_problem; stored: top-100 ids, sampled scores.
    sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
    from oracle import oracle as O
    vecs, qvecs, qid = O.synth_ranking_problem(4993, 70, 2048, seed=0)
    sc = np.dot(vecs.T, qvecs)
    rk = np.argsort(-sc, axis=0)
    top = rk[:100]
    topsc = np.take_along_axis(sc, top, axis=0).astype(np.float64)
    g7b = {"qid": qid.astype(np.int64), "top100": top.astype(np.int32),
           "top100_scores": np.take_along_axis(sc, top, axis=0),
           "top100_mingap": np.float64(np.min(-np.diff(topsc, axis=0))),
           "scores_rows_0_4992_step_97": sc[::97].copy(),
           "vecs_checksum": np.float64(vecs.astype(np.float64).sum()),
           "qvecs_checksum": np.float64(qvecs.astype(np.float64).sum())}
    # mAP of the reference on the synthetic rOxford-shaped gnd
    gnd = O.synth_gnd(70, 4993, seed=1)
    avg, per = compute_map_and_print("roxford5k", rk, gnd)
    for k, v in avg.items():
        g7b[k] = np.float64(v)
    for k, v in per.items():
        g7b[k] = v
    np.savez_compressed(os.path.join(HERE, "g7_roxford_shape.npz"), **g7b)

    # ---- G8 / G9 compute_map -------------------------------------------------
    rng = np.random.default_rng(40)
    n, q = 500, 12
    rk = np.stack([rng.permutation(n) for _ in range(q)], axis=1).astype(np.int64)
    gnd = []
    for i in range(q):
        ids = rng.choice(n, 30, replace=False)
        ne, nh, nj = rng.integers(0, 8), rng.integers(0, 8), rng.integers(0, 8)
        gnd.append({"easy": ids[:ne].tolist(), "hard": ids[ne:ne + nh].tolist(),
                    "junk": ids[ne + nh:ne + nh + nj].tolist(), "bbx": None})
    gnd[4]["easy"], gnd[4]["hard"] = [], []                 # no positives at any level
    gnd[7]["easy"] = []                                      # empty for 'easy' only
    gnd[8]["junk"] = []                                      # no junk
    g8 = {"ranks": rk.astype(np.int32),
          "gnd_json": np.frombuffer(json.dumps(gnd).encode(), dtype=np.uint8)}
    avg, per = compute_map_and_print("roxford5k", rk, gnd)
    for k, v in avg.items():
        g8["rox_" + k] = np.float64(v)
    for k, v in per.items():
        g8["rox_" + k] = v
    gnd_m = [{"ok": np.concatenate([g["easy"], g["hard"]]), "junk": np.array(g["junk"])} for g in gnd]
    m, aps, pr, prs = compute_map(rk, gnd_m, [1, 5, 10])
    g8["medium_map"], g8["medium_aps"], g8["medium_pr"], g8["medium_prs"] = np.float64(m), aps, pr, prs
    old = [{"ok": g["easy"] + g["hard"], "junk": g["junk"]} for g in gnd]
    avg, per = compute_map_and_print("247tokyo1k", rk, old)
    g8["old_map"], g8["old_ap"] = np.float64(avg["map"]), per["ap"]
    nojunk = [{"ok": g["ok"]} for g in old]
    m, aps, _, _ = compute_map(rk, nojunk)
    g8["nojunkkey_map"], g8["nojunkkey_aps"] = np.float64(m), aps
    g8["other_dataset_returns_none"] = np.bool_(compute_map_and_print("oxford5k", rk, gnd) is None)
    ka = {"r0_n1": compute_ap(np.array([0]), 1), "r1_n1": compute_ap(np.array([1]), 1),
          "r02_n2": compute_ap(np.array([0, 2]), 2), "empty_n3": compute_ap(np.array([]), 3),
          "r0_4_9_n5": compute_ap(np.array([0, 4, 9]), 5)}
    for k, v in ka.items():
        g8["ap_" + k] = np.float64(v)
    np.savez_compressed(os.path.join(HERE, "g8_map.npz"), **g8)

    # ---- G11 scenario overlay + metadata keys --------------------------------
    cases = [
        ({"a": {"b": 1, "c": {"d": 2}}, "l": [1, 2]}, {"a": {"c": {"e": 3}}, "l*": [9]}),
        ({"a": {"b": 1}, "l": [1, 2]}, {"a": None, "l+": [3]}),
        ({"net": {"path": None, "runtime": {"wrappers": {"eval": {"0_w": {"k": None}}}}}},
         {"net": {"path": "x.pth", "runtime": {"wrappers": {"eval": {"0_w": {"k": "f.pkl"}}}}}}),
        ({"a": 1}, {"a": {"b": 2}}),
        ({"x": [{"k": 1}, {"k": 2}]}, {"x": {1: {"k": 5}}}),
    ]
    import copy
    overlay = []
    for a, b in cases:
        res = dict_deep_overlay(copy.deepcopy(a), copy.deepcopy(b))
        overlay.append({"base": a, "over": b, "result": res})
    try:
        dict_deep_overlay({"l": [1]}, {"l": [2]})
        list_err = False
    except ValueError:
        list_err = True
    three = dict_deep_overlay({"a": 1, "b": {"c": 1}}, {"b": {"d": 2}}, {"a": 3, "b": {"c": 7}})

    meta_out = None
    try:
        from mdir.tools.eventprocessor import initialize_processor
        events = initialize_processor({"progress": {"print_each": 100, "key_suffix": "validation/loss:total"}},
                                      dataroot=None)
        aps = [0.5, float("nan"), 0.25, 1.0]
        lg = lambda it, size, label, value, dtype: events.register_data(
            0, it, size, "roxford5k/validation/%s" % label, value, dtype)
        lg(None, 4, "dataset", {"extract_descriptors": 1.0, "compute_score": 2.0, "total_s": 3.0}, "scalar/time")
        lg(None, 4, "score_avg", {"map_medium": 0.58333}, "scalar/score")
        for i, a in enumerate(aps):
            lg(i, 4, "score", {"ap_medium": a, "ap_easy": a / 2}, "scalar/score")
        events.close_epoch()
        meta_out = {k: [float(x) for x in v] for k, v in events.metadata.metadata().items()}
    except Exception as e:  # matplotlib etc. missing: record that
        meta_out = {"error": repr(e)}
    with open(os.path.join(HERE, "g11_scenario.json"), "w") as f:
        json.dump({"overlay": [{"base": c["base"], "over": {str(k): v for k, v in c["over"].items()}
                                if False else _jsonable(c["over"]), "result": _jsonable(c["result"])}
                               for c in overlay],
                   "list_merge_raises": list_err, "three_way": three, "metadata": meta_out}, f, indent=1)
    make_next_rows()
    make_tables()
    make_rmac()
    make_rpool()
    make_map_fuzz()
    print("golden fixtures written to", HERE)
    for fn in sorted(os.listdir(HERE)):
        print("  %-28s %8d B" % (fn, os.path.getsize(os.path.join(HERE, fn))))


def make_next_rows():
    """G12-G15: the rows SURVEY.md section 8 marks "next" (f1 mining, f2 embedding output, f3 whitening learning) and the
    image loader (a1).  Same rule: the reference is imported and RUN; only inputs and its outputs are stored."""
    import io
    from PIL import Image
    # ---- G12 whitening learning (cirtorch/utils/whiten.py:14-70) ------------------------------------------------
    from cirtorch.utils.whiten import cholesky, pcawhitenlearn, whitenlearn
    rng = np.random.default_rng(12)
    D, N, npairs = 24, 300, 140
    basis = np.linalg.qr(rng.standard_normal((D, D)))[0] * np.geomspace(2.0, 0.2, D)
    X = basis @ rng.standard_normal((D, N))
    X /= np.linalg.norm(X, axis=0, keepdims=True)                     # [D,N] float64 unit columns
    qidxs = rng.choice(N, npairs, replace=False)
    pidxs = rng.choice(N, npairs, replace=False)
    X[:, pidxs] = X[:, qidxs] + 0.15 * rng.standard_normal((D, npairs))      # matching pairs are close
    X /= np.linalg.norm(X, axis=0, keepdims=True)
    m_lw, P_lw = whitenlearn(X.copy(), qidxs, pidxs)
    m_pca, P_pca = pcawhitenlearn(X.copy())
    _, P_shr = pcawhitenlearn(X.copy(), shrink=8)
    S_sing = np.ones((3, 3))
    with open(os.devnull, "w") as sink:
        old, sys.stdout = sys.stdout, sink
        try:
            L_sing = cholesky(S_sing)
        finally:
            sys.stdout = old
    S_pd = np.cov(rng.standard_normal((5, 40)))
    np.savez_compressed(os.path.join(HERE, "g12_whitenlearn.npz"), X=X, qidxs=qidxs, pidxs=pidxs, m_lw=m_lw, P_lw=P_lw,
                        m_pca=m_pca, P_pca=np.real(P_pca), P_pca_shrink8=np.real(P_shr), S_singular=S_sing, L_singular=L_sing,
                        S_pd=S_pd, L_pd=cholesky(S_pd))

    # ---- G13 hard-negative selection (cirtorch/datasets/traindataset.py:178-271) -----------------------------------
    import torch.nn as nn
    from cirtorch.datasets.traindataset import TuplesDataset
    rng = np.random.default_rng(13)
    nimg, nclusters, dim = 90, 12, 16
    tmp = tempfile.mkdtemp()
    proto = rng.integers(0, 255, (nclusters, 12, 16, 3))
    clusters = rng.integers(0, nclusters, nimg).tolist()
    images = []
    for i in range(nimg):
        arr = np.clip(0.6 * proto[clusters[i]] + 0.4 * rng.integers(0, 255, (12, 16, 3)), 0, 255).astype(np.uint8)
        path = os.path.join(tmp, "im%03d.png" % i)
        Image.fromarray(np.kron(arr, np.ones((4, 4, 1), dtype=np.uint8))).save(path)
        images.append(path)

    class ToyNet(nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(5)
            self.conv = nn.Conv2d(3, dim // 4, 5, stride=3)
            self.meta = {"out_channels": dim}

        def forward(self, x):
            v = torch.relu(self.conv(x - 0.5))
            v = nn.functional.adaptive_avg_pool2d(v, 2).flatten(1)          # [1, dim]: 4 channels x 2 x 2 cells
            v = v - v.mean(dim=1, keepdim=True)
            return (v / v.norm(dim=1, keepdim=True)).t()

    to_tensor = lambda img: torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1).float() / 255.0
    ds = object.__new__(TuplesDataset)
    pairs = [(a, b) for a in range(nimg) for b in range(nimg) if a < b and clusters[a] == clusters[b]][:60]
    ds.name, ds.mode, ds.imsize, ds.transform, ds.print_freq = "toy", "train", None, to_tensor, 1000
    ds.images, ds.clusters = images, clusters
    ds.qpool, ds.ppool = [a for a, _ in pairs], [b for _, b in pairs]
    ds.qsize, ds.poolsize, ds.nnum = 14, 50, 4
    drawn = []
    real_randperm = torch.randperm
    torch.randperm = lambda n: drawn.append(real_randperm(n)) or drawn[-1]
    torch.manual_seed(77)
    net = ToyNet()
    torch.manual_seed(77)
    with open(os.devnull, "w") as sink:
        old, sys.stdout = sys.stdout, sink
        try:
            stats = ds.create_epoch_tuples(net, device=torch.device("cpu"))
        finally:
            sys.stdout = old
            torch.randperm = real_randperm
    idxs2images = drawn[1][:ds.poolsize].numpy()
    with torch.no_grad():
        describe = lambda ids: torch.cat([net(to_tensor(Image.open(images[i]).convert("RGB"))[None]) for i in ids], dim=1)
        qvecs, poolvecs = describe(ds.qidxs), describe(idxs2images.tolist())
    np.savez_compressed(os.path.join(HERE, "g13_mining.npz"), qvecs=qvecs.numpy(), poolvecs=poolvecs.numpy(),
                        idxs2images=idxs2images, clusters=np.array(clusters), qidxs=np.array(ds.qidxs), pidxs=np.array(ds.pidxs),
                        nnum=np.int64(ds.nnum), nidxs=np.array([[int(x) for x in row] for row in ds.nidxs]),
                        ndist=np.array(stats["average_negative_distance"]))

    # ---- G14 EmbeddingOutput (mdir/components/data/output.py:117-139) --------------------------------------------------
    from mdir.components.data.output import EmbeddingOutput
    rng = np.random.default_rng(14)
    names = ["a.jpg", "b.jpg", "c.jpg", "d.jpg"]
    vec = rng.standard_normal((4, 6)).astype(np.float32)
    out = EmbeddingOutput((names,), {})
    out.add(0, object(), torch.from_numpy(vec[0]))
    out.add(1, None, None)                              # unreadable image
    out.add(2, object(), torch.from_numpy(vec[2]))
    out.add(3, object(), torch.from_numpy(vec[3]))
    res_names, res = out.postprocess()
    boxes = [None, (1, 2, 3, 4), None, None]
    out_b = EmbeddingOutput((names, boxes), {}, bbxs=True)
    empty = EmbeddingOutput((names,), {}).postprocess()
    np.savez_compressed(os.path.join(HERE, "g14_embedding_output.npz"), vec=vec, result=res, result_dtype=str(res.dtype),
                        names=np.array(res_names), preprocess_bbxs=np.array([repr(out_b.preprocess())]),
                        empty_second=np.array([repr(empty[1])]))

    # ---- G15 image loader (cirtorch/datasets/genericdataset.py:44-70, datahelpers.py:24-50) ---------------------------
    # Pillow >= 10 has no Image.ANTIALIAS; it was an alias of LANCZOS (Pillow 2.7 - 9.5), restored here for the reference
    Image.ANTIALIAS = Image.LANCZOS
    from cirtorch.datasets.genericdataset import ImagesFromList
    rng = np.random.default_rng(15)
    g15 = {}
    files = {}
    for name, (h, w) in (("landscape", (150, 221)), ("portrait", (203, 97)), ("small", (40, 30))):
        yy, xx = np.mgrid[0:h, 0:w]
        arr = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx * 7 + yy * 13) % 256)], axis=2)
        arr = np.clip(arr + rng.integers(-20, 20, arr.shape), 0, 255).astype(np.uint8)
        buf = io.BytesIO()
        Image.fromarray(arr).save(buf, format="PNG")
        files[name] = buf.getvalue()
        g15["file_" + name] = np.frombuffer(files[name], dtype=np.uint8)
        with open(os.path.join(tmp, name + ".png"), "wb") as f:
            f.write(files[name])
    cases = [("landscape", 64, None), ("landscape", 64, (10, 20, 200, 140)), ("portrait", 64, None), ("portrait", 100, (5, 5, 90, 60)),
             ("small", 64, None), ("landscape", None, (0, 0, 17, 9))]
    for ci, (name, imsize, bbx) in enumerate(cases):
        dsl = ImagesFromList(root=tmp, images=[name + ".png"], imsize=imsize, bbxs=[bbx], transform=lambda im: np.asarray(im).copy())
        g15["case%d_out" % ci] = dsl[0]
        g15["case%d_spec" % ci] = np.array([repr((name, imsize, bbx))])
    np.savez_compressed(os.path.join(HERE, "g15_loader.npz"), **g15)


def make_tables():
    """G16: the TSV/CSV dataset branch of CirDatasetAp (mdir/components/optim/score/cirscore.py:24-38) -- what
    daan/data/file_readers.py's TsvReader returns for small tables (empty cells, bracketed and half-bracketed cells, "\r",
    quotes, .gz/.xz, csv), and the lists CirDatasetAp.__init__ builds from a db/queries pair."""
    import base64
    import gzip
    import lzma
    from daan.data.file_readers import initialize_file_reader
    from mdir.components.optim.score.cirscore import CirDatasetAp
    queries = ("query\tbbx\tok\tjunk\tnote\n"
               "q/a.jpg\t[1, 2, 30, 40]\t[\"x.jpg\", \"y.jpg\"]\t[]\tfirst\n"
               "q/b.jpg\t\t[\"z.jpg\"]\t[\"x.jpg\"]\t \n"
               "q/c.jpg\t[]\t[\"w.jpg\",\"x.jpg\"]\t[\"y.jpg\",\"z.jpg\"]\t\"quoted, cell\"\n")
    db = "identifier,width\nx.jpg,640\ny.jpg,\nz.jpg,480\nw.jpg,[1\n"
    odd = (" name\tvalue\tlast \n"                      # header stripped on both sides
           "a\t{\"k\": [1, 2]}\t{}\n"
           "b\t[1, 2\t2]\n"                             # bracketed on one side only: stays a string
           "c\t{\"k\": 1}]\tplain\r\n"                  # mismatched pair; "\r" survives strip("\n")
           "\t3.5\ttrue\n")                             # empty first cell; numbers and literals stay strings
    files = {"queries.tsv": queries.encode(), "db.csv": db.encode(), "odd.tsv": odd.encode(),
             "queries.tsv.gz": gzip.compress(queries.encode(), mtime=0), "db.csv.xz": lzma.compress(db.encode()),
             "tsv.v2.csv": b"a,b\tc\n1,2\t3\n"}          # separator rule: "tsv" among the last two dot pieces -> tab
    reads = [("queries.tsv", ["query", "bbx", "ok", "junk"]), ("queries.tsv", None), ("db.csv", ["identifier"]),
             ("db.csv", None), ("odd.tsv", None), ("odd.tsv", ["last", "name"]), ("queries.tsv.gz", ["junk", "query"]),
             ("db.csv.xz", ["width", "identifier"]), ("tsv.v2.csv", None)]
    g16 = {"files": {k: base64.b64encode(v).decode() for k, v in files.items()}, "reads": [], "datasets": []}
    with tempfile.TemporaryDirectory() as tmp:
        for name, data in files.items():
            with open(os.path.join(tmp, name), "wb") as f:
                f.write(data)
        for name, keys in reads:
            with initialize_file_reader(os.path.join(tmp, name), keys=keys) as reader:
                got = reader.get()
            g16["reads"].append({"file": name, "keys": keys, "columns": list(got.keys()), "out": list(got.values())})
        for qname, dbname in (("queries.tsv", "db.csv"), ("queries.tsv.gz", "db.csv.xz")):
            score = CirDatasetAp({"image_size": 64, "transforms": "pil2np | totensor | normalize",
                                  "mean_std": [[0.4, 0.4, 0.4], [0.2, 0.2, 0.2]],
                                  "dataset": {"name": "toy", "imgdir": "/img", "queries": os.path.join(tmp, qname),
                                              "db": os.path.join(tmp, dbname)}})
            g16["datasets"].append({"queries": qname, "db": dbname, "name": score.dataset, "images": score.images,
                                    "qimages": score.qimages, "bbxs": [list(b) if b else None for b in score.bbxs],
                                    "gnd": score.gnd})
    with open(os.path.join(HERE, "g16_tables.json"), "w") as f:
        json.dump(g16, f, indent=1)


def make_rmac():
    """G17: R-MAC pooling (cirtorch/layers/functional.py:26-72, pooling.py:50-60) on ReLU-like maps regenerated from seeds;
    only the reference's outputs are stored."""
    import cirtorch.layers.functional as LF
    g17 = {}
    for c, h, w, b in [(2048, 24, 32, 1), (512, 48, 64, 1), (64, 17, 23, 2), (256, 7, 5, 1), (16, 3, 40, 2), (8, 12, 12, 1), (4, 2, 2, 1)]:
        seed = 1700 + h * 100 + w
        x = torch.from_numpy(sparse_map(seed, (b, c, h, w)))
        g17["seed_c%d_h%d_w%d_b%d" % (c, h, w, b)] = seed
        for L in (3, 2):
            g17["rmac_c%d_h%d_w%d_b%d_L%d" % (c, h, w, b, L)] = LF.rmac(x.clone(), L=L, eps=1e-6).numpy().reshape(b, c)
    np.savez_compressed(os.path.join(HERE, "g17_rmac.npz"), **g17)


def make_rpool():
    """G18: regional pooling (cirtorch/layers/functional.py:75-121 roipool, layers/pooling.py:62-95 Rpool) with GeM / MAC / SPoC
    regions, with and without the regional whitening, aggregated and per region."""
    from cirtorch.layers.pooling import GeM, MAC, Rpool, SPoC
    torch.manual_seed(18)
    g18 = {}
    for c, h, w in [(64, 24, 32), (32, 17, 23), (16, 7, 5), (8, 12, 12)]:
        seed = 1800 + h * 100 + w
        x = torch.from_numpy(sparse_map(seed, (2, c, h, w)))
        lin = torch.nn.Linear(c, c)
        g18["seed_c%d_h%d_w%d" % (c, h, w)] = seed
        g18["weight_c%d" % c], g18["bias_c%d" % c] = lin.weight.detach().numpy(), lin.bias.detach().numpy()
        for name, mod in (("gem", GeM(p=2.5)), ("mac", MAC()), ("spoc", SPoC())):
            for tag, wh in (("plain", None), ("whiten", lin)):
                rp = Rpool(mod, wh)
                with torch.no_grad():
                    g18["agg_%s_%s_c%d_h%d_w%d" % (name, tag, c, h, w)] = rp(x).numpy().reshape(2, -1)
                    g18["reg_%s_%s_c%d_h%d_w%d" % (name, tag, c, h, w)] = rp(x, aggregate=False).numpy()[..., 0, 0]
    np.savez_compressed(os.path.join(HERE, "g18_rpool.npz"), **g18)


def fuzz_map_case(seed):
    """One random compute_map problem (shared with tests/test_evaluate.py through this generator): permutation rankings, id lists
    as lists or arrays, empty / overlapping ok and junk, a missing junk key, kappas beyond N."""
    rng = np.random.default_rng(seed)
    n, nq = int(rng.integers(1, 60)), int(rng.integers(1, 6))
    ranks = np.stack([rng.permutation(n) for _ in range(nq)], axis=1)
    gnd = []
    for _ in range(nq):
        ok = rng.choice(n, int(rng.integers(0, min(n, 6) + 1)), replace=False)
        junk = rng.choice(n, int(rng.integers(0, min(n, 4) + 1)), replace=False)
        g = {"ok": ok.tolist() if rng.random() < 0.5 else ok, "junk": junk.tolist() if rng.random() < 0.5 else junk}
        if rng.random() < 0.2:
            del g["junk"]
        gnd.append(g)
    kappas = sorted(set(int(v) for v in rng.integers(1, n + 5, size=int(rng.integers(0, 4)))))
    return ranks, gnd, kappas


def fuzz_revisited_case(seed):
    rng = np.random.default_rng(seed)
    n, nq = int(rng.integers(5, 80)), int(rng.integers(1, 5))
    ranks = np.stack([rng.permutation(n) for _ in range(nq)], axis=1)
    gnd = []
    for _ in range(nq):
        ids = rng.permutation(n)[:int(rng.integers(0, min(n, 12)))]
        cut = sorted(rng.integers(0, len(ids) + 1, size=2))
        gnd.append({"easy": ids[:cut[0]], "hard": ids[cut[0]:cut[1]], "junk": ids[cut[1]:], "bbx": None})
    return ranks, gnd


def make_map_fuzz():
    """G19: 300 random compute_map problems and 100 random revisited-protocol problems (cirtorch/utils/evaluate.py:39-152): the
    inputs come back from the seeds (fuzz_map_case / fuzz_revisited_case above), only the reference's outputs are stored."""
    import contextlib
    import copy
    from cirtorch.utils.evaluate import compute_map, compute_map_and_print
    g19 = {}
    for seed in range(300):
        ranks, gnd, kappas = fuzz_map_case(seed)
        try:
            mAP, aps, pr, prs = compute_map(ranks.copy(), copy.deepcopy(gnd), list(kappas))
        except Exception as err:          # e.g. no query with positives: the reference divides by zero (evaluate.py:108)
            g19["error_%d" % seed] = np.array([type(err).__name__])
            continue
        g19["map_%d" % seed] = np.array([mAP])
        g19["aps_%d" % seed], g19["pr_%d" % seed], g19["prs_%d" % seed] = np.asarray(aps, dtype=np.float64), np.asarray(pr, dtype=np.float64), np.asarray(prs, dtype=np.float64)
    for seed in range(100):
        ranks, gnd = fuzz_revisited_case(1000 + seed)
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                avg, per = compute_map_and_print("roxford5k" if seed % 2 else "rparis6k", ranks.copy(), copy.deepcopy(gnd))
        except Exception as err:
            g19["rev_error_%d" % seed] = np.array([type(err).__name__])
            continue
        for k in ("map_easy", "map_medium", "map_hard"):
            g19["rev_%s_%d" % (k, seed)] = np.array([avg[k]])
        for k in ("ap_easy", "ap_medium", "ap_hard"):
            g19["rev_%s_%d" % (k, seed)] = np.asarray(per[k], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "g19_map_fuzz.npz"), **g19)


def _jsonable(o):
    if isinstance(o, dict):
        return {("int:%d" % k if isinstance(k, int) else k): _jsonable(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_jsonable(v) for v in o]
    return o


if __name__ == "__main__":
    main()
