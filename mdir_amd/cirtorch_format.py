"""Stages that speak the upstream cirtorch checkpoint format -- drop-ins for
``mdir/stages/cirtorch_format/test.py``: ``embed`` (:17-89), ``learn_whitening`` (:92-153),
``convert_contained_net`` (:156-201), ``load_whitening`` (:204-238), ``_compute_whitening``
(:241-268).  SURVEY.md section 8 row f2: directory of images -> descriptor matrix.

An upstream checkpoint is ``{"meta": {architecture, pooling, whitening, mean, std, outputdim,
[local_whitening, regional, Lw]}, "state_dict"}``; whitening files are pickled ``{'m','P'}`` named
``<whitening>_None_<image_size>_<multiscale>.lw.pkl`` inside ``whitening_dir``.
"""
import math
import os
import pickle
import time

import torch

from .datasets import Compose, Normalize, ToTensor, get_data_root
from .networks import extract_vectors, init_network
from .scenario import path_join
from .whiten import whitenapply, whitenlearn

WHITENING_ALIASES = {"sfm30k": "retrieval-SfM-30k", "sfm120k": "retrieval-SfM-120k"}


def cid2filename(cid, prefix):
    """``cirtorch/datasets/datahelpers.py:9-21``: ``prefix/c[-2:]/c[-4:-2]/c[-6:-4]/cid``."""
    return os.path.join(prefix, cid[-2:], cid[-4:-2], cid[-6:-4], cid)


def htime(c):
    c = round(c)
    days, rest = divmod(c, 86400)
    hours, rest = divmod(rest, 3600)
    minutes, seconds = divmod(rest, 60)
    if days > 0:
        return "{:d}d {:d}h {:d}m {:d}s".format(days, hours, minutes, seconds)
    if hours > 0:
        return "{:d}h {:d}m {:d}s".format(hours, minutes, seconds)
    if minutes > 0:
        return "{:d}m {:d}s".format(minutes, seconds)
    return "{:d}s".format(seconds)


def _whitening_file(whitening_dir, whitening, image_size, multiscale):
    return os.path.join(whitening_dir, "%s_%s_%s_%s.lw.pkl" % (whitening, None, image_size, multiscale))


def _load_upstream(path):
    """Network + multi-scale set-up shared by ``embed`` and ``learn_whitening`` (test.py:31-64)."""
    assert os.path.exists(path), path
    print(">> Loading network:\n>>>> '{}'".format(path))
    state = torch.load(path, map_location="cpu", weights_only=False)
    meta = state["meta"]
    net = init_network({"architecture": meta["architecture"], "pooling": meta["pooling"],
                        "whitening": meta["whitening"], "mean": meta["mean"], "std": meta["std"],
                        "pretrained": False})
    net.load_state_dict(state["state_dict"])
    print(">>>> loaded network: ")
    print(net.meta_repr())
    return net


def _scales(net, multiscale):
    ms = multiscale if not isinstance(multiscale, bool) else [1, 1. / math.sqrt(2), 1. / 2] if multiscale else [1]
    # the reference leaves `msp` undefined (NameError) unless pooling is GeM without in-network
    # whitening; every other pooling aggregates with the plain mean, which is msp = 1
    msp = net.pool.p.data.tolist()[0] if net.meta["pooling"] == "gem" and net.whiten is None else 1
    return ms, msp


def _transform(net):
    return Compose([ToTensor(), Normalize(net.meta["mean"], net.meta["std"])])


def embed(params, data, device=None):
    """Images of a directory -> ``({}, names, vecs [N,D])`` (+ whitened ``[N,D]`` when a whitening
    file is given)."""
    net = params.pop("net")
    imgdir = params.pop("imgdir")
    whitening = params.pop("whitening", None)
    whitening_dir = params.pop("whitening_dir", None)
    image_size = params.pop("image_size", 1024)
    multiscale = params.pop("multiscale", True)
    assert not params, params.keys()
    input_images, bbxs = (data[0], None) if len(data) == 1 else data
    impaths = [path_join(imgdir, x) for x in input_images]
    if not data[0]:
        return ({"status": "skipped"}, [], []) + (([],) if whitening_dir else tuple())

    net = _load_upstream(net)
    ms, msp = _scales(net, multiscale)
    if device is None:
        net.cuda()
    net.eval()
    Lw = None
    if whitening_dir:
        print(">> {}: Loading whitening...".format(whitening))
        with open(_whitening_file(whitening_dir, whitening, image_size, multiscale), "rb") as handle:
            Lw = pickle.load(handle)

    print(">> Images descriptors...")
    vecs = extract_vectors(net, impaths, image_size, _transform(net), bbxs=bbxs, ms=ms, msp=msp, device=device)
    print(">> Evaluating...")
    vecs = vecs.numpy()
    if Lw is not None:
        vecs_lw = whitenapply(vecs, Lw["m"], Lw["P"], device=device or "cuda")
        return {}, input_images, vecs.T, vecs_lw.T
    return {}, input_images, vecs.T


def _compute_whitening(whitening, net, image_size, transform, ms, msp, device=None):
    start = time.time()
    print(">> {}: Learning whitening...".format(whitening))
    db_root = os.path.join(get_data_root(), "train", whitening)
    ims_root = os.path.join(db_root, "ims")
    with open(os.path.join(db_root, "{}-whiten.pkl".format(whitening)), "rb") as f:
        db = pickle.load(f)
    images = [cid2filename(db["cids"][i], ims_root) for i in range(len(db["cids"]))]
    print(">> {}: Extracting...".format(whitening))
    wvecs = extract_vectors(net, images, image_size, transform, ms=ms, msp=msp, device=device)
    print(">> {}: Learning...".format(whitening))
    m, P = whitenlearn(wvecs.numpy(), db["qidxs"], db["pidxs"], device=device or "cuda")
    elapsed = time.time() - start
    print(">> {}: elapsed time: {}".format(whitening, htime(elapsed)))
    return {"m": m, "P": P}, elapsed


def learn_whitening(params, data, device=None):
    net = params.pop("net")
    whitening = params.pop("whitening")
    whitening_dir = params.pop("whitening_dir", None)
    image_size = params.pop("image_size", 1024)
    multiscale = params.pop("multiscale", True)
    params.pop("imgdir", None)
    assert not params
    assert not data
    whitening = WHITENING_ALIASES.get(whitening, whitening)
    net = _load_upstream(net)
    ms, msp = _scales(net, multiscale)
    if device is None:
        net.cuda()
    net.eval()
    Lw, elapsed = _compute_whitening(whitening, net, image_size, _transform(net), ms, msp, device)
    if whitening_dir:       # back-compatible option: store beside, return only the metadata
        os.makedirs(whitening_dir, exist_ok=True)
        with open(_whitening_file(whitening_dir, whitening, image_size, multiscale), "wb") as handle:
            pickle.dump(Lw, handle)
        return {"whitening_learn": int(elapsed)},
    return {"whitening_learn": int(elapsed)}, Lw


def convert_contained_net(params, data):
    """Upstream checkpoint -> the ``CirNetwork`` checkpoint ``mdir_amd.network.load_network`` reads."""
    source = params.pop("source")
    net = params.pop("net")
    assert not params
    assert not data
    assert os.path.exists(source), source
    print(">> Loading network:\n>>>> '{}'".format(source))
    official = torch.load(source, map_location="cpu", weights_only=False)
    meta = official.pop("meta")
    net_state = {
        "type": "CirNetwork",
        # not written by the reference's converter, but required by its own (and this) loader
        # (learning/network.py:156): without it the converted file cannot be read back
        "frozen": False,
        "network_params": {
            "model": {"architecture": "cirnet", "cir_architecture": meta.pop("architecture"),
                      "local_whitening": meta.pop("local_whitening", False), "pooling": meta.pop("pooling"),
                      "regional": meta.pop("regional", False), "whitening": meta.pop("whitening"),
                      "pretrained": True},
            "runtime": {"wrappers": "",
                        "data": {"mean_std": [meta.pop("mean"), meta.pop("std")],
                                 "transforms": "pil2np | totensor | normalize"}},
        },
        "model_state": official.pop("state_dict"),
    }
    del meta["outputdim"]
    del meta["Lw"]
    assert not meta, meta           # integrity: nothing of the upstream file is dropped silently
    assert not official, official
    if os.path.dirname(net) and not os.path.exists(os.path.dirname(net)):
        os.makedirs(os.path.dirname(net))
    torch.save(net_state, net)
    return {},


def load_whitening(params, data):
    """Whitening stored inside an upstream checkpoint (``meta['Lw'][name]['ms'|'ss']``)."""
    net = params.pop("net")
    whitening = params.pop("whitening")
    whitening_dir = params.pop("whitening_dir", None)
    image_size = params.pop("image_size", 1024)
    multiscale = params.pop("multiscale", True)
    params.pop("imgdir", None)
    assert not params
    assert not data
    assert os.path.exists(net), net
    whitening = WHITENING_ALIASES.get(whitening, whitening)
    print(">> Loading network:\n>>>> '{}'".format(net))
    state = torch.load(net, map_location="cpu", weights_only=False)
    assert isinstance(multiscale, bool)
    Lw = state["meta"]["Lw"][whitening]["ms" if multiscale else "ss"]
    if whitening_dir:
        os.makedirs(whitening_dir, exist_ok=True)
        with open(_whitening_file(whitening_dir, whitening, image_size, multiscale), "wb") as handle:
            pickle.dump(Lw, handle)
        return {},
    return {}, Lw
