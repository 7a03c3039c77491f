"""``embed`` for upstream-format cirtorch checkpoints (a directory of images ->
a descriptor matrix).  Same stage protocol as ``mdir/stages/cirtorch_format/test.py:17-89`` --
``embed(params, data) -> (metadata, names, vecs [N,D][, whitened [N,d]])`` -- built from this package's own
pieces: ``init_network`` + ``extract_vectors`` (device buffer, one hipGraph per image shape) and ``whitenapply``
(``mdx_scores`` + ``mdx_l2n_rows``).  ``learn_whitening`` (test.py:92-152 + ``_compute_whitening`` :241-268; round 5) learns the
supervised whitening of a training set that is on disk (nothing is downloaded); ``convert_contained_net`` (:156-201) rewrites an
upstream checkpoint as the ``CirNetwork`` checkpoint ``load_network`` reads, ``load_whitening`` (:204-238) takes the whitening an
upstream checkpoint carries in ``meta['Lw']``.

An upstream checkpoint is ``{"meta": {architecture, pooling, whitening, mean, std, ...}, "state_dict"}``; a
whitening is a pickled ``{'m': [D,1], 'P': [D,D]}`` stored as ``<whitening>_None_<image_size>_<multiscale>.lw.pkl``
in ``whitening_dir``.
"""
import os
import pickle

import torch

from .datasets import Compose, Normalize, ToTensor
from .networks import extract_vectors, init_network
from .scenario import path_join
from .whiten import whitenapply

MS_SCALES = [1, 2 ** -0.5, 0.5]


def load_upstream(path):
    """The ``ImageRetrievalNet`` an upstream checkpoint describes, weights loaded, nothing downloaded."""
    state = torch.load(path, map_location="cpu", weights_only=False)
    meta = state["meta"]
    net = init_network({key: meta[key] for key in ("architecture", "pooling", "whitening", "mean", "std")} | {"pretrained": False})
    net.load_state_dict(state["state_dict"])
    return net


def embed(params, data, device=None):
    params = dict(params)
    checkpoint, imgdir = params.pop("net"), params.pop("imgdir")
    whitening, whitening_dir = params.pop("whitening", None), params.pop("whitening_dir", None)
    image_size, multiscale = params.pop("image_size", 1024), params.pop("multiscale", True)
    assert not params, params.keys()
    names, bbxs = (data[0], None) if len(data) == 1 else data
    if not names:
        return ({"status": "skipped"}, [], []) + (([],) if whitening_dir else ())

    net = load_upstream(checkpoint).eval()
    if device is None:
        net.cuda()                      # cirtorch_format/test.py:55
    else:
        net.to(device)                  # `device` is this build's extension (CPU tests, cuda:N): the network follows it
    scales = (MS_SCALES if multiscale else [1]) if isinstance(multiscale, bool) else multiscale
    # GeM exponent as the power of the multi-scale mean only when nothing follows the pooling in the network
    msp = float(net.pool.p) if net.meta["pooling"] == "gem" and net.whiten is None and len(scales) > 1 else 1
    transform = Compose([ToTensor(), Normalize(net.meta["mean"], net.meta["std"])])
    vecs = extract_vectors(net, [path_join(imgdir, x) for x in names], image_size, transform, bbxs=bbxs, ms=scales, msp=msp,
                           device=device).numpy()
    if not whitening_dir:
        return {}, names, vecs.T
    with open(os.path.join(whitening_dir, "%s_%s_%s_%s.lw.pkl" % (whitening, None, image_size, multiscale)), "rb") as handle:
        lw = pickle.load(handle)
    return {}, names, vecs.T, whitenapply(vecs, lw["m"], lw["P"], device=device or "cuda").T


def cid2filename(cid, prefix):
    """``cirtorch/datasets/datahelpers.py:9-22``: ``<prefix>/<last 2>/<2 before>/<2 before>/<cid>``."""
    return os.path.join(prefix, cid[-2:], cid[-4:-2], cid[-6:-4], cid)


def _compute_whitening(whitening, net, image_size, transform, ms, msp, device=None):
    """``test.py:241-268``: descriptors of the training set ``<data root>/train/<whitening>`` (``<whitening>-whiten.pkl`` with
    ``cids`` / ``qidxs`` / ``pidxs``, images under ``ims/``), then ``whitenlearn`` on the device."""
    import time
    from .datasets import get_data_root
    from .whiten import whitenlearn
    start = time.time()
    print(">> {}: Learning whitening...".format(whitening))
    db_root = os.path.join(get_data_root(), "train", whitening)
    with open(os.path.join(db_root, "{}-whiten.pkl".format(whitening)), "rb") as f:
        db = pickle.load(f)
    images = [cid2filename(db["cids"][i], os.path.join(db_root, "ims")) for i in range(len(db["cids"]))]
    print(">> {}: Extracting...".format(whitening))
    wvecs = extract_vectors(net, images, image_size, transform, ms=ms, msp=msp, device=device)
    print(">> {}: Learning...".format(whitening))
    m, P = whitenlearn(wvecs.numpy(), db["qidxs"], db["pidxs"], device=device or "cuda")
    elapsed = time.time() - start
    print(">> {}: elapsed time: {:.0f}s".format(whitening, elapsed))
    return {"m": m, "P": P}, elapsed


def learn_whitening(params, data, device=None):
    """``test.py:92-152``: ``({"whitening_learn": seconds}, Lw)``, or ``({"whitening_learn": seconds},)`` with the whitening
    pickled as ``<whitening_dir>/<whitening>_None_<image_size>_<multiscale>.lw.pkl`` (the name ``embed`` looks for)."""
    params = dict(params)
    checkpoint, whitening = params.pop("net"), params.pop("whitening")
    whitening_dir = params.pop("whitening_dir", None)
    image_size, multiscale = params.pop("image_size", 1024), params.pop("multiscale", True)
    params.pop("imgdir", None)
    assert not params
    assert not data
    assert os.path.exists(checkpoint), checkpoint
    whitening = {"sfm30k": "retrieval-SfM-30k", "sfm120k": "retrieval-SfM-120k"}.get(whitening, whitening)
    print(">> Loading network:\n>>>> '{}'".format(checkpoint))
    net = load_upstream(checkpoint).eval()
    print(">>>> loaded network: ")
    print(net.meta_repr())
    if device is None:
        net.cuda()
    else:
        net.to(device)
    scales = (MS_SCALES if multiscale else [1]) if isinstance(multiscale, bool) else multiscale
    msp = float(net.pool.p) if net.meta["pooling"] == "gem" and net.whiten is None and len(scales) > 1 else 1
    transform = Compose([ToTensor(), Normalize(net.meta["mean"], net.meta["std"])])
    Lw, elapsed = _compute_whitening(whitening, net, image_size, transform, scales, msp, device)
    if whitening_dir:
        os.makedirs(whitening_dir, exist_ok=True)
        with open(os.path.join(whitening_dir, "%s_%s_%s_%s.lw.pkl" % (whitening, None, image_size, multiscale)), "wb") as handle:
            pickle.dump(Lw, handle)
        return {"whitening_learn": int(elapsed)},
    return {"whitening_learn": int(elapsed)}, Lw


def convert_contained_net(params, data):
    """``test.py:156-201``: upstream ``{"meta", "state_dict"}`` at ``source`` -> this path's checkpoint ``{"type": "CirNetwork",
    "network_params": {"model", "runtime"}, "model_state"}`` at ``net`` (learning/network.py:142-150).  Every key of the upstream
    ``meta`` must be accounted for (``outputdim`` and ``Lw`` are dropped), as upstream's integrity check demands."""
    params = dict(params)
    source, target = params.pop("source"), params.pop("net")
    assert not params
    assert not data
    assert os.path.exists(source), source
    print(">> Loading network:\n>>>> '{}'".format(source))
    official = torch.load(source, map_location="cpu", weights_only=False)
    meta = dict(official.pop("meta"))
    model = {"architecture": "cirnet", "cir_architecture": meta.pop("architecture"), "local_whitening": meta.pop("local_whitening", False),
             "pooling": meta.pop("pooling"), "regional": meta.pop("regional", False), "whitening": meta.pop("whitening"), "pretrained": True}
    runtime = {"wrappers": "", "data": {"mean_std": [meta.pop("mean"), meta.pop("std")], "transforms": "pil2np | totensor | normalize"}}
    # "frozen": the one key added to what upstream writes (test.py:169-189) -- upstream's own loader demands it
    # (learning/network.py:156 asserts the key set {"type", "frozen", "network_params", "model_state"}) and would refuse upstream's file
    net_state = {"type": "CirNetwork", "frozen": True, "network_params": {"model": model, "runtime": runtime},
                 "model_state": official.pop("state_dict")}
    del meta["outputdim"]
    del meta["Lw"]
    assert not meta, meta
    assert not official, official
    if os.path.dirname(target):
        os.makedirs(os.path.dirname(target), exist_ok=True)
    torch.save(net_state, target)
    return {},


def load_whitening(params, data):
    """``test.py:204-238``: the whitening ``meta['Lw'][<set>]['ms' | 'ss']`` of an upstream checkpoint -- returned, or pickled
    under the name ``embed`` looks for when ``whitening_dir`` is given."""
    params = dict(params)
    checkpoint, whitening = params.pop("net"), params.pop("whitening")
    whitening_dir = params.pop("whitening_dir", None)
    image_size, multiscale = params.pop("image_size", 1024), params.pop("multiscale", True)
    params.pop("imgdir", None)
    assert not params
    assert not data
    assert os.path.exists(checkpoint), checkpoint
    whitening = {"sfm30k": "retrieval-SfM-30k", "sfm120k": "retrieval-SfM-120k"}.get(whitening, whitening)
    print(">> Loading network:\n>>>> '{}'".format(checkpoint))
    state = torch.load(checkpoint, map_location="cpu", weights_only=False)
    assert isinstance(multiscale, bool)
    Lw = state["meta"]["Lw"][whitening]["ms" if multiscale else "ss"]
    if whitening_dir:
        os.makedirs(whitening_dir, exist_ok=True)
        with open(os.path.join(whitening_dir, "%s_%s_%s_%s.lw.pkl" % (whitening, None, image_size, multiscale)), "wb") as handle:
            pickle.dump(Lw, handle)
        return {},
    return {}, Lw
