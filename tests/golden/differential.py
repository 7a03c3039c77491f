#!/usr/bin/env python3
"""Differential runs of the host surface against THE REFERENCE ITSELF -- build container only (needs /root/reference; nothing of
it is stored).  Random problems go through the reference's function and through the drop-in; values AND raised error types must
agree.  What these runs found in round 5 became fixtures (G16, G19) or fixes (`whitenapply` in float64).

    python tests/golden/differential.py            # prints one "... mismatches: N" line per family; N must be 0

Families: compute_map / compute_map_and_print (cirtorch/utils/evaluate.py), whitenapply (utils/whiten.py:4-12), dict_deep_overlay
(daan/core/experiments.py), the table reader (daan/data/file_readers.py:101-135), the image loader (datasets/genericdataset.py:
44-70, host route), extract_vectors on toy networks with every pooling incl. rmac / regional, local and in-network whitening,
multi-scale with and without msp, boxes (networks/imageretrievalnet.py:277-324; kernels replaced by the oracle, tests/fake_ops.py).
"""
import sys, copy, io, contextlib, os, tempfile, json, gzip
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [HERE, ROOT, os.path.join(ROOT, 'tests')]
import make_golden as mg
mg.import_reference()
import numpy as np
from cirtorch.utils import evaluate as R
from mdir_amd import evaluate as M
rng = np.random.default_rng(0)
bad = 0
def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)
for trial in range(400):
    n = int(rng.integers(1, 60)); nq = int(rng.integers(1, 6))
    ranks = np.stack([rng.permutation(n) for _ in range(nq)], axis=1)
    gnd = []
    for q in range(nq):
        k_ok = int(rng.integers(0, min(n, 6) + 1)); k_j = int(rng.integers(0, min(n, 4) + 1))
        ok = rng.choice(n, k_ok, replace=False); junk = rng.choice(n, k_j, replace=False)
        g = {"ok": ok.tolist() if rng.random() < 0.5 else ok, "junk": junk.tolist() if rng.random() < 0.5 else junk}
        if rng.random() < 0.2: del g["junk"]
        gnd.append(g)
    kappas = sorted(set(int(v) for v in rng.integers(1, n + 5, size=int(rng.integers(0, 4)))))
    try:
        want = R.compute_map(ranks.copy(), copy.deepcopy(gnd), list(kappas))
        werr = None
    except Exception as e:
        want, werr = None, type(e).__name__
    try:
        got = M.compute_map(ranks.copy(), copy.deepcopy(gnd), list(kappas))
        gerr = None
    except Exception as e:
        got, gerr = None, type(e).__name__
    if werr or gerr:
        if werr != gerr:
            bad += 1; print("compute_map error mismatch", trial, werr, gerr, n, nq, kappas)
        continue
    for a, b in zip(want, got):
        if not np.allclose(np.asarray(a, dtype=float), np.asarray(b, dtype=float), rtol=0, atol=1e-12, equal_nan=True):
            bad += 1; print("compute_map value mismatch", trial, n, nq, kappas, a, b); break
# revisited protocol
for trial in range(200):
    n = int(rng.integers(5, 80)); nq = int(rng.integers(1, 5))
    ranks = np.stack([rng.permutation(n) for _ in range(nq)], axis=1)
    gnd = []
    for q in range(nq):
        ids = rng.permutation(n)[:int(rng.integers(0, min(n, 12)))]
        cut = sorted(rng.integers(0, len(ids) + 1, size=2))
        gnd.append({"easy": ids[:cut[0]], "hard": ids[cut[0]:cut[1]], "junk": ids[cut[1]:], "bbx": None})
    for name in ("roxford5k", "rparis6k"):
        try:
            want = quiet(R.compute_map_and_print, name, ranks.copy(), copy.deepcopy(gnd)); werr = None
        except Exception as e:
            want, werr = None, type(e).__name__
        try:
            got = quiet(M.compute_map_and_print, name, ranks.copy(), copy.deepcopy(gnd)); gerr = None
        except Exception as e:
            got, gerr = None, type(e).__name__
        if werr != gerr:
            bad += 1; print("map_and_print error mismatch", trial, werr, gerr); continue
        if werr: continue
        for dw, dg in zip(want, got):
            if set(dw) != set(dg): bad += 1; print("keys", dw.keys(), dg.keys()); break
            for k in dw:
                if not np.allclose(np.asarray(dw[k], dtype=float), np.asarray(dg[k], dtype=float), atol=1e-12, equal_nan=True):
                    bad += 1; print("map_and_print mismatch", trial, k, dw[k], dg[k])
print("evaluate mismatches:", bad)
# whiten
from cirtorch.utils import whiten as RW
from mdir_amd import whiten as MW
import tests.fake_ops as fake
class MP:
    def setattr(self, obj, name, val): setattr(obj, name, val)
fake.install(MP())
bad = 0
for trial in range(30):
    d = int(rng.integers(2, 12)); n = int(rng.integers(d + 5, 60))
    X = rng.standard_normal((d, n))
    m = rng.standard_normal((d, 1)); P = rng.standard_normal((d, d))
    dims = None if rng.random() < 0.5 else int(rng.integers(1, d + 1))
    a = RW.whitenapply(X.copy(), m, P, dims); b = MW.whitenapply(X.copy(), m, P, dims, device="cpu") if 'device' in MW.whitenapply.__code__.co_varnames else MW.whitenapply(X.copy(), m, P, dims)
    if not np.allclose(a, b, rtol=1e-6, atol=1e-9): bad += 1; print("whitenapply", trial, np.abs(a-b).max())
print("whiten mismatches:", bad)
# scenario utils
from daan.core.experiments import dict_deep_overlay as RD
from mdir_amd.scenario import dict_deep_overlay as MD
bad = 0
def rand_dict(depth=0):
    out = {}
    for k in rng.choice(list("abcde"), size=int(rng.integers(0, 4)), replace=False):
        r = rng.random()
        if r < 0.35 and depth < 3: out[str(k)] = rand_dict(depth + 1)
        elif r < 0.5: out[str(k)] = [int(v) for v in rng.integers(0, 5, size=int(rng.integers(0, 3)))]
        elif r < 0.6: out[str(k)] = None
        else: out[str(k)] = int(rng.integers(0, 9))
    return out
for trial in range(500):
    ds = [rand_dict() for _ in range(int(rng.integers(1, 4)))]
    lr = bool(rng.random() < 0.5)
    try: want = RD(*copy.deepcopy(ds), list_replace=lr); we = None
    except Exception as e: want, we = None, type(e).__name__
    try: got = MD(*copy.deepcopy(ds), list_replace=lr); ge = None
    except Exception as e: got, ge = None, type(e).__name__
    if we != ge or want != got:
        bad += 1; print("overlay mismatch", ds, lr, want, got, we, ge)
        if bad > 5: break
print("overlay mismatches:", bad)

# ---------------------------------------------------------------- tables and the host image loader
import lzma
import torch
from PIL import Image
Image.ANTIALIAS = Image.LANCZOS
rng = np.random.default_rng(1)
# ---- 5. table reader
from daan.data.file_readers import initialize_file_reader
from mdir_amd.score import _read_table
bad = 0
alphabet = ['', 'a', 'b c', '[1,2]', '[]', '{}', '{"k": [1]}', '[1,2', '1,2]', '{"a":1}]', ' ', '"q"', '3.5', 'null', '[null]', 'x\r', "[\"a\", \"b\"]"]
tmp = tempfile.mkdtemp()
for trial in range(300):
    ext = rng.choice(['.tsv', '.csv', '.tsv.gz', '.csv.xz', '.csv.gz', '.tsv.xz'])
    sep = '\t' if 'tsv' in ext else ','
    ncol = int(rng.integers(1, 5)); nrow = int(rng.integers(0, 6))
    header = ['c%d' % i for i in range(ncol)]
    cells = [[str(rng.choice(alphabet)) for _ in range(ncol)] for _ in range(nrow)]
    cells = [[c if sep not in c else c.replace(sep, ';') for c in row] for row in cells]
    text = sep.join(header) + '\n' + ''.join(sep.join(r) + '\n' for r in cells)
    path = os.path.join(tmp, 't%d%s' % (trial, ext))
    op = gzip.open if ext.endswith('.gz') else lzma.open if ext.endswith('.xz') else open
    with op(path, 'wb') as f: f.write(text.encode())
    keys = None if rng.random() < 0.5 else [str(k) for k in rng.permutation(header)[:int(rng.integers(1, ncol + 1))]]
    try:
        with initialize_file_reader(path, keys=keys) as r: want = r.get(); we = None
    except Exception as e: want, we = None, type(e).__name__
    try: got = _read_table(path, keys); ge = None
    except Exception as e: got, ge = None, type(e).__name__
    if we != ge or (want is not None and (list(want.keys()) != list(got.keys()) or list(want.values()) != list(got.values()))):
        bad += 1; print('table mismatch', trial, ext, keys, we, ge, repr(text)[:200]); 
        if bad > 5: break
print('table mismatches', bad)
# ---- 1. host loader vs reference loader
from cirtorch.datasets.genericdataset import ImagesFromList as RI
from mdir_amd.datasets import ImagesFromList as MI
bad = 0
for trial in range(150):
    w, h = int(rng.integers(8, 300)), int(rng.integers(8, 300))
    arr = rng.integers(0, 255, (h, w, 3), dtype=np.uint8)
    mode = rng.choice(['RGB', 'L', 'RGBA', 'P'])
    img = Image.fromarray(arr).convert(mode)
    fmt = rng.choice(['png', 'jpg']) if mode in ('RGB', 'L') else 'png'
    path = os.path.join(tmp, 'i%d.%s' % (trial, fmt)); img.save(path)
    imsize = None if rng.random() < 0.15 else int(rng.integers(4, 400))
    if rng.random() < 0.5:
        bbx = None
    else:
        x1, y1 = float(rng.uniform(-5, w * 0.7)), float(rng.uniform(-5, h * 0.7))
        bbx = (x1, y1, x1 + float(rng.uniform(1, w)), y1 + float(rng.uniform(1, h)))
        if rng.random() < 0.5: bbx = tuple(int(round(v)) for v in bbx)
    tr = lambda im: np.asarray(im).copy()
    try: want = RI(root='', images=[path], imsize=imsize, bbxs=[bbx], transform=tr)[0]; we = None
    except Exception as e: want, we = None, type(e).__name__
    try: got = MI('', [path], imsize=imsize, bbxs=[bbx], transform=tr)[0]; ge = None
    except Exception as e: got, ge = None, type(e).__name__
    if we != ge or (want is not None and (want.shape != got.shape or not np.array_equal(want, got))):
        bad += 1; print('loader mismatch', trial, (w, h), mode, fmt, imsize, bbx, we, ge, None if want is None else want.shape, None if got is None else got.shape)
        if bad > 5: break
print('loader mismatches', bad)

# ---------------------------------------------------------------- extract_vectors on toy networks
import torch.nn as nn
os.environ['MDIR_AMD_WORKERS'] = '0'
from cirtorch.networks import imageretrievalnet as RN
from cirtorch.layers import pooling as RP
from mdir_amd import networks as MN, layers as ML
from mdir_amd.datasets import Compose, ToTensor, Normalize
rng = np.random.default_rng(3)
tmp = tempfile.mkdtemp()
bad = 0
for trial in range(40):
    torch.manual_seed(trial)
    c = int(rng.choice([8, 16]))
    feats = [nn.Conv2d(3, c, 3, stride=2, padding=1), nn.ReLU(inplace=True), nn.Conv2d(c, c, 3, stride=2, padding=1), nn.ReLU(inplace=True)]
    pooling = str(rng.choice(["gem", "mac", "spoc", "rmac"]))
    regional = bool(rng.random() < 0.25) and pooling != "rmac"
    whitening = bool(rng.random() < 0.5)
    lw = bool(rng.random() < 0.2)
    meta = {"architecture": "toy", "local_whitening": lw, "pooling": pooling, "regional": regional, "whitening": whitening,
            "mean": [0.485, 0.456, 0.406], "std": [0.229, 0.224, 0.225], "outputdim": c}
    rpool = {"gem": RP.GeM, "mac": RP.MAC, "spoc": RP.SPoC, "rmac": RP.RMAC}[pooling]()
    if regional: rpool = RP.Rpool(rpool, nn.Linear(c, c))
    rnet = RN.ImageRetrievalNet(copy.deepcopy(feats), nn.Linear(c, c) if lw else None, rpool, nn.Linear(c, c) if whitening else None, dict(meta)).eval()
    mpool = ML.POOLING[pooling]()
    if regional: mpool = ML.Rpool(mpool, nn.Linear(c, c))
    mnet = MN.ImageRetrievalNet(copy.deepcopy(feats), nn.Linear(c, c) if lw else None, mpool, nn.Linear(c, c) if whitening else None, dict(meta)).eval()
    missing = mnet.load_state_dict(rnet.state_dict(), strict=True)
    mnet.meta["out_channels"] = c; rnet.meta["out_channels"] = c
    n = int(rng.integers(1, 5))
    paths, bbxs = [], []
    for i in range(n):
        w, h = int(rng.integers(40, 200)), int(rng.integers(40, 200))
        pth = os.path.join(tmp, "t%d_%d.png" % (trial, i))
        Image.fromarray(rng.integers(0, 255, (h, w, 3), dtype=np.uint8)).save(pth); paths.append(pth)
        bbxs.append(None if rng.random() < 0.5 else (5, 5, w - 3, h - 7))
    use_bbx = rng.random() < 0.5
    ms = [[1], [1, 2 ** -0.5, 0.5], [1, 0.5]][int(rng.integers(0, 3))]
    msp = 1 if (len(ms) == 1 or rng.random() < 0.5) else 2.5
    imsize = int(rng.integers(48, 160))
    tr = Compose([ToTensor(), Normalize(meta["mean"], meta["std"])])
    kw = dict(bbxs=[b for b in bbxs] if use_bbx and all(b is not None for b in bbxs) else None, ms=ms, msp=msp)
    try:
        with contextlib.redirect_stdout(io.StringIO()), torch.no_grad():
            want = RN.extract_vectors(rnet, paths, imsize, tr, device="cpu", **kw); we = None
    except Exception as e:
        want, we = None, type(e).__name__ + ": " + str(e)[:80]
    try:
        with contextlib.redirect_stdout(io.StringIO()), torch.no_grad():
            got = MN.extract_vectors(mnet, paths, imsize, tr, device="cpu", **kw); ge = None
    except Exception as e:
        got, ge = None, type(e).__name__ + ": " + str(e)[:80]
    if (we is None) != (ge is None):
        bad += 1; print("error mismatch", trial, pooling, regional, whitening, lw, ms, msp, we, ge); continue
    if we: continue
    if tuple(want.shape) != tuple(got.shape) or not np.allclose(want.numpy(), got.numpy(), rtol=1e-4, atol=2e-6, equal_nan=True):
        bad += 1; print("value mismatch", trial, pooling, regional, whitening, lw, ms, msp, tuple(want.shape), tuple(got.shape), float((want - got).abs().max()) if tuple(want.shape) == tuple(got.shape) else None)
print("extract_vectors mismatches:", bad)
# ---------------------------------------------------------------- regional / local descriptors (imageretrievalnet.py:325-384)
bad = 0
for trial in range(12):
    torch.manual_seed(100 + trial)
    c = 8
    feats = [nn.Conv2d(3, c, 3, stride=2, padding=1), nn.ReLU(inplace=True), nn.Conv2d(c, c, 3, stride=2, padding=1), nn.ReLU(inplace=True)]
    pooling = str(rng.choice(["gem", "mac", "spoc"]))
    meta = {"architecture": "toy", "local_whitening": False, "pooling": pooling, "regional": True, "whitening": False,
            "mean": [0.485, 0.456, 0.406], "std": [0.229, 0.224, 0.225], "outputdim": c, "out_channels": c}
    rnet = RN.ImageRetrievalNet(copy.deepcopy(feats), None, RP.Rpool({"gem": RP.GeM, "mac": RP.MAC, "spoc": RP.SPoC}[pooling](), nn.Linear(c, c)), None, dict(meta)).eval()
    mnet = MN.ImageRetrievalNet(copy.deepcopy(feats), None, ML.Rpool(ML.POOLING[pooling](), nn.Linear(c, c)), None, dict(meta)).eval()
    mnet.load_state_dict(rnet.state_dict(), strict=True)
    paths = []
    for i in range(2):
        w, h = int(rng.integers(40, 160)), int(rng.integers(40, 160))
        pth = os.path.join(tmp, "r%d_%d.png" % (trial, i))
        Image.fromarray(rng.integers(0, 255, (h, w, 3), dtype=np.uint8)).save(pth); paths.append(pth)
    tr = Compose([ToTensor(), Normalize(meta["mean"], meta["std"])])
    with torch.no_grad():
        # the reference's functions call .cuda(): their one-image forms on the same loader items instead
        from mdir_amd.datasets import ImagesFromList
        items = [ImagesFromList("", [pth], imsize=96, transform=tr)[0][None] for pth in paths]
        want_r = [RN.extract_ssr(rnet, x) for x in items]
        want_l = [RN.extract_ssl(rnet, x) for x in items]
        with contextlib.redirect_stdout(io.StringIO()):
            got_r = MN.extract_regional_vectors(mnet, paths, 96, tr, device="cpu")
            got_l = MN.extract_local_vectors(mnet, paths, 96, tr, device="cpu")
    for a_, b_ in list(zip(want_r, got_r)) + list(zip(want_l, got_l)):
        if tuple(a_.shape) != tuple(b_.shape) or not np.allclose(a_.numpy(), b_.numpy(), rtol=1e-4, atol=2e-6):
            bad += 1; print("regional/local mismatch", trial, pooling, tuple(a_.shape), tuple(b_.shape))
print("regional / local mismatches:", bad)
# ---------------------------------------------------------------- whitening stages (mdir/stages/whiten.py)
from mdir.stages import whiten as RS
from mdir_amd import stages as MS
bad = 0
for trial in range(10):
    n, d = int(rng.integers(40, 90)), int(rng.integers(4, 14))
    vals = rng.standard_normal((n, d)).astype(np.float32)
    names = ["n%d" % i for i in range(n)]
    k = int(rng.integers(8, n // 2))
    queries, positives = names[:k], names[k:2 * k]
    _, want = RS.learn_lw_whitening({}, (names, vals.copy(), queries, positives))
    _, got = MS.learn_lw_whitening({}, (names, vals.copy(), queries, positives), device="cpu")
    # eigenvectors are defined up to sign: compare P up to the sign of its rows, and what whitening does to the data
    sign = np.sign(np.sum(want["P"] * got["P"], axis=1, keepdims=True))
    if not (np.allclose(want["m"], got["m"]) and np.abs(want["P"] - got["P"] * sign).max() <= 1e-6 * np.abs(want["P"]).max()):       # relative to the largest entry: small pair sets are ill-conditioned
        bad += 1; print("learn_lw_whitening mismatch", trial, np.abs(want["P"] - got["P"] * sign).max(), np.abs(want["P"]).max())
    _, wpca = RS.learn_pca_whitening({"shrink": None}, (vals.copy(),))
    _, gpca = MS.learn_pca_whitening({"shrink": None}, (vals.copy(),), device="cpu")
    sign = np.sign(np.sum(wpca["P"] * gpca["P"], axis=1, keepdims=True))
    if not np.allclose(wpca["P"], gpca["P"] * sign, rtol=1e-6, atol=1e-8):
        bad += 1; print("learn_pca_whitening mismatch", trial)
    dims = None if trial % 2 else int(rng.integers(1, d + 1))
    _, _, ww = RS.whiten({"dimensions": dims}, (want, names, vals.copy()))
    _, _, gw = MS.whiten({"dimensions": dims}, (want, names, vals.copy()), device="cpu")
    if ww.shape != gw.shape or not np.allclose(ww, gw, rtol=0, atol=1e-12):
        bad += 1; print("whiten stage mismatch", trial, np.abs(ww - gw).max())
    parts = [rng.standard_normal((n, int(rng.integers(2, 6)))) for _ in range(int(rng.integers(1, 4)))]
    total = sum(x.shape[1] for x in parts)
    dims = None if trial % 3 == 0 else int(rng.integers(1, total + 1))
    _, wp = RS.paste_pca_normalize({"dimensions": dims}, [x.copy() for x in parts])
    _, gp = MS.paste_pca_normalize({"dimensions": dims}, [x.copy() for x in parts], device="cpu")
    if wp.shape != gp.shape or not np.allclose(np.real(wp), gp, rtol=1e-7, atol=1e-9):
        bad += 1; print("paste_pca_normalize mismatch", trial, dims, np.abs(np.real(wp) - gp).max())
print("whitening stage mismatches:", bad)
