#!/usr/bin/env python3
"""Generate tests/golden/*.npz|json by RUNNING THE REFERENCE in the build container.

The reference (jenicek/mdir + its vendored cirtorch) is imported from
/root/reference with throw-away stub modules for the three third-party packages
this image lacks (torchvision, cv2, h5py); none of its source is copied.  Only
inputs (or the seeds that regenerate them) and the reference's outputs are
stored.  Re-run:  python tests/golden/make_golden.py

The fixtures are DATA; the GPU box never sees /root/reference.
"""
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
    print("golden fixtures written to", HERE)
    for fn in sorted(os.listdir(HERE)):
        print("  %-28s %8d B" % (fn, os.path.getsize(os.path.join(HERE, fn))))


def _jsonable(o):
    if isinstance(o, dict):
        return {("int:%d" % k if isinstance(k, int) else k): _jsonable(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_jsonable(v) for v in o]
    return o


if __name__ == "__main__":
    main()
