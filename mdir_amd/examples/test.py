"""Testing CLI of cirtorch on the MI355X path -- the second caller of the operator API
(``mdir/external/cirtorch/examples/test.py``).

    python -m mdir_amd.examples.test --network-path NET.pth --datasets roxford5k,rparis6k \\
        --image-size 1024 --multiscale '[1, 1/2**(1/2), 1/2]' [--whitening retrieval-SfM-30k]

Same options and printed lines as upstream.  What runs where: descriptors through
``extract_vectors`` (HIP tail), ``np.dot`` + ``np.argsort`` (test.py:240-241,250-251) through a
resident ``DescriptorIndex`` + ``rank_full``, ``whitenlearn`` / ``whitenapply`` (:197, :246-247)
through ``mdir_amd.whiten``.  Nothing is downloaded: networks are given by path (cirtorch
checkpoint ``{"meta", "state_dict"}``), datasets live under ``$CIRTORCH_ROOT/data``.
"""
import argparse
import os
import pickle
import time

import numpy as np
import torch

from .. import ops
from ..datasets import Compose, Normalize, ToTensor, configdataset, get_data_root
from ..evaluate import compute_map_and_print
from ..networks import extract_vectors, init_network
from ..whiten import whitenapply, whitenlearn

datasets_names = ["oxford5k", "paris6k", "roxford5k", "rparis6k"]
whitening_names = ["retrieval-SfM-30k", "retrieval-SfM-120k", "load:retrieval-SfM-30k", "load:retrieval-SfM-120k"]


def cid2filename(cid, prefix):
    """datahelpers.py:9-22."""
    return cid if cid[0] == "/" else os.path.join(prefix, cid[-2:], cid[-4:-2], cid[-6:-4], cid)


def htime(c):
    """general.py:14-30."""
    c = round(c)
    days, hours, minutes, seconds = c // 86400, c // 3600 % 24, c // 60 % 60, c % 60
    if days > 0:
        return "{:d}d {:d}h {:d}m {:d}s".format(days, hours, minutes, seconds)
    if hours > 0:
        return "{:d}h {:d}m {:d}s".format(hours, minutes, seconds)
    if minutes > 0:
        return "{:d}m {:d}s".format(minutes, seconds)
    return "{:d}s".format(seconds)


def build_parser():
    parser = argparse.ArgumentParser(description="CNN Image Retrieval Testing (MI355X)")
    group = parser.add_mutually_exclusive_group(required=True)
    group.add_argument("--network-path", "-npath", metavar="NETWORK")
    group.add_argument("--network-offtheshelf", "-noff", metavar="NETWORK",
                       help="'ARCHITECTURE-POOLING[-whiten]' with random weights (nothing is downloaded)")
    parser.add_argument("--datasets", "-d", metavar="DATASETS", default="oxford5k,paris6k")
    parser.add_argument("--image-size", "-imsize", default=1024, type=int, metavar="N")
    parser.add_argument("--multiscale", "-ms", metavar="MULTISCALE", default="[1]")
    parser.add_argument("--whitening", "-w", metavar="WHITENING", default=None, choices=whitening_names)
    parser.add_argument("--gpu-id", "-g", default="0", metavar="N")
    return parser


def rank(vecs, qvecs, device):
    """``np.argsort(-np.dot(vecs.T, qvecs), axis=0)`` -> device ``[N,Q]`` view (test.py:240-241)."""
    index = ops.DescriptorIndex(torch.as_tensor(vecs, device=device).float().contiguous(), "DN")
    ranks = ops.rank_full(index.scores(torch.as_tensor(qvecs, device=device).float().contiguous(), "DN"))
    index.close()
    return ranks.t()


def main(argv=None):
    args = build_parser().parse_args(argv)
    for dataset in args.datasets.split(","):
        if dataset not in datasets_names:
            raise ValueError("Unsupported or unknown dataset: {}!".format(dataset))
    os.environ["CUDA_VISIBLE_DEVICES"] = args.gpu_id
    device = torch.device("cuda")

    if args.network_path is not None:
        print(">> Loading network:\n>>>> '{}'".format(args.network_path))
        state = torch.load(args.network_path, map_location="cpu", weights_only=False)
        meta = state["meta"]
        net = init_network({"architecture": meta["architecture"], "pooling": meta["pooling"],
                            "local_whitening": meta.get("local_whitening", False),
                            "regional": meta.get("regional", False), "whitening": meta.get("whitening", False),
                            "mean": meta["mean"], "std": meta["std"], "pretrained": False})
        net.load_state_dict(state["state_dict"])
        if "Lw" in meta:
            net.meta["Lw"] = meta["Lw"]
    else:
        offtheshelf = args.network_offtheshelf.split("-")
        print(">> Loading off-the-shelf network:\n>>>> '{}'".format(args.network_offtheshelf))
        net = init_network({"architecture": offtheshelf[0], "pooling": offtheshelf[1],
                            "local_whitening": "lwhiten" in offtheshelf[2:], "regional": "reg" in offtheshelf[2:],
                            "whitening": "whiten" in offtheshelf[2:], "pretrained": True})
    print(">>>> loaded network: ")
    print(net.meta_repr())

    ms = list(eval(args.multiscale))
    if len(ms) > 1 and net.meta["pooling"] == "gem" and not net.meta["regional"] and not net.meta["whitening"]:
        msp = net.pool.p.item()
        print(">> Set-up multiscale:\n>>>> ms: {}\n>>>> msp: {}".format(ms, msp))
    else:
        msp = 1
    net.cuda()
    net.eval()
    transform = Compose([ToTensor(), Normalize(net.meta["mean"], net.meta["std"])])

    Lw = None
    if args.whitening is not None:
        start = time.time()
        if args.whitening.startswith("load"):
            name = args.whitening.split(":", 1)[1]
            assert "Lw" in net.meta and name in net.meta["Lw"]
            print(">> {}: Whitening is precomputed, loading it...".format(name))
            Lw = net.meta["Lw"][name]["ms" if len(ms) > 1 else "ss"]
        else:
            whiten_fn = None
            if args.network_path is not None:
                whiten_fn = args.network_path + "_{}_whiten".format(args.whitening) + ("_ms" if len(ms) > 1 else "") + ".pth"
            if whiten_fn is not None and os.path.isfile(whiten_fn):
                print(">> {}: Whitening is precomputed, loading it...".format(args.whitening))
                Lw = torch.load(whiten_fn, weights_only=False)
            else:
                print(">> {}: Learning whitening...".format(args.whitening))
                db_root = os.path.join(get_data_root(), "train", args.whitening)
                with open(os.path.join(db_root, "{}-whiten.pkl".format(args.whitening)), "rb") as f:
                    db = pickle.load(f)
                images = [cid2filename(db["cids"][i], os.path.join(db_root, "ims")) for i in range(len(db["cids"]))]
                print(">> {}: Extracting...".format(args.whitening))
                wvecs = extract_vectors(net, images, args.image_size, transform, ms=ms, msp=msp).numpy()
                print(">> {}: Learning...".format(args.whitening))
                m, P = whitenlearn(wvecs, db["qidxs"], db["pidxs"])
                Lw = {"m": m, "P": P}
                if whiten_fn is not None:
                    print(">> {}: Saving to {}...".format(args.whitening, whiten_fn))
                    torch.save(Lw, whiten_fn)
        print(">> {}: elapsed time: {}".format(args.whitening, htime(time.time() - start)))

    results = {}
    for dataset in args.datasets.split(","):
        start = time.time()
        print(">> {}: Extracting...".format(dataset))
        cfg = configdataset(dataset, os.path.join(get_data_root(), "test"))
        images = [cfg["im_fname"](cfg, i) for i in range(cfg["n"])]
        qimages = [cfg["qim_fname"](cfg, i) for i in range(cfg["nq"])]
        # upstream assumes every query has a box (test.py:225); a missing one means "whole image" here
        bbxs = [tuple(cfg["gnd"][i]["bbx"]) if cfg["gnd"][i].get("bbx") else None for i in range(cfg["nq"])]
        print(">> {}: database images...".format(dataset))
        vecs = extract_vectors(net, images, args.image_size, transform, ms=ms, msp=msp)
        print(">> {}: query images...".format(dataset))
        qvecs = extract_vectors(net, qimages, args.image_size, transform, bbxs=bbxs, ms=ms, msp=msp)
        print(">> {}: Evaluating...".format(dataset))
        vecs, qvecs = vecs.numpy(), qvecs.numpy()
        results[dataset] = compute_map_and_print(dataset, rank(vecs, qvecs, device), cfg["gnd"])
        if Lw is not None:
            vecs_lw = whitenapply(vecs, Lw["m"], Lw["P"])
            qvecs_lw = whitenapply(qvecs, Lw["m"], Lw["P"])
            results[dataset + " + whiten"] = compute_map_and_print(dataset + " + whiten", rank(vecs_lw, qvecs_lw, device),
                                                                    cfg["gnd"])
        print(">> {}: elapsed time: {}".format(dataset, htime(time.time() - start)))
    return results


if __name__ == "__main__":
    main()
