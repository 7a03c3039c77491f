"""``embed`` for upstream-format cirtorch checkpoints (a directory of images ->
a descriptor matrix).  Same stage protocol as ``mdir/stages/cirtorch_format/test.py:17-89`` --
``embed(params, data) -> (metadata, names, vecs [N,D][, whitened [N,d]])`` -- built from this package's own
pieces: ``init_network`` + ``extract_vectors`` (device buffer, one hipGraph per image shape) and ``whitenapply``
(``mdx_scores`` + ``mdx_l2n_rows``).  The other stages of that reference file (learning / converting / storing
whitenings) are outside the hot path and not provided.

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
