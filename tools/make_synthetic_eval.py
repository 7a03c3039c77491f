#!/usr/bin/env python3
"""Write a tiny, self-contained evaluation set-up that eval.py can run without any download:

    <root>/data/test/roxford5k/{jpg/*.jpg, gnd_roxford5k.pkl}     revisited protocol (easy/hard/junk, bbx)
    <root>/data/test/247tokyo1k/{jpg/*.jpg, gnd_247tokyo1k.pkl}   old protocol (ok/junk), query == database
    <root>/net.pth          CirNetwork checkpoint (mdir layout), random weights, seed fixed
    <root>/whiten.pkl       {'P','m'} whitening
    <root>/eval_synth.yml   overlay for scenarios/eval.yml

    python tools/make_synthetic_eval.py <root> [arch [tokyo images]] && CIRTORCH_ROOT=<root> ./eval.py eval.yml <root>/eval_synth.yml
"""
import os
import pickle
import sys

import numpy as np
import torch
import yaml
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def write_images(folder, names, rng, base):
    """Images are noisy variations of a few base patterns, so that retrieval is non-trivial."""
    os.makedirs(folder, exist_ok=True)
    for i, name in enumerate(names):
        w, h = 320 + 16 * (i % 3), 240 + 8 * (i % 2)
        # the image's own pattern, blended with a random distractor pattern and heavy noise: retrieval
        # is possible but far from perfect (mAP well below 100)
        pat = 0.55 * base[i % len(base)] + 0.45 * rng.integers(0, 255, base[0].shape)
        img = np.kron(pat, np.ones((h // pat.shape[0] + 1, w // pat.shape[1] + 1, 1)))[:h, :w]
        img = np.clip(img + rng.normal(0, 60, img.shape), 0, 255).astype(np.uint8)
        Image.fromarray(img).save(os.path.join(folder, name + ".jpg"), quality=92)


def main(root, arch="alexnet", tokyo_n=15):
    tokyo_n = int(tokyo_n)
    from mdir_amd.network import CirNetwork, SingleNetwork
    from mdir_amd.networks import init_network
    rng = np.random.default_rng(0)
    base = [rng.integers(0, 255, (6, 8, 3)).astype(np.float64) for _ in range(5)]
    n, nq = 40, 5
    names = ["db%03d" % i for i in range(n)]
    rox = os.path.join(root, "data", "test", "roxford5k")
    write_images(os.path.join(rox, "jpg"), names, rng, base)
    gnd = []
    for q in range(nq):
        same = [i for i in range(n) if i % len(base) == q % len(base) and i != q]
        gnd.append({"bbx": [8.0, 8.0, 300.0, 220.0] if q % 2 == 0 else None, "easy": same[:3], "hard": same[3:6],
                    "junk": [q] + same[6:]})
    with open(os.path.join(rox, "gnd_roxford5k.pkl"), "wb") as f:
        pickle.dump({"imlist": names, "qimlist": names[:nq], "gnd": gnd}, f)
    tok = os.path.join(root, "data", "test", "247tokyo1k")
    tnames = ["tk%03d" % i for i in range(tokyo_n)]
    write_images(os.path.join(tok, "jpg"), tnames, rng, base)
    tgnd = [{"ok": [j for j in range(tokyo_n) if j % len(base) == i % len(base) and j != i], "junk": [i], "bbx": None}
            for i in range(tokyo_n)]
    with open(os.path.join(tok, "gnd_247tokyo1k.pkl"), "wb") as f:
        pickle.dump({"imlist": tnames, "qimlist": tnames, "gnd": tgnd}, f)

    torch.manual_seed(0)
    model_params = {"architecture": "cirnet", "cir_architecture": arch, "local_whitening": False, "pooling": "gem",
                    "regional": False, "whitening": False, "pretrained": True}
    model = init_network({"architecture": arch, "pretrained": False})
    with torch.no_grad():
        model.pool.p.fill_(2.85)
    model.meta["in_channels"], model.meta["out_channels"] = 3, model.meta["outputdim"]
    runtime = {"wrappers": "", "data": {"transforms": "pil2np | totensor | normalize"}}
    net = CirNetwork(model, SingleNetwork.NetworkParams(model_params, runtime), "cpu", frozen=True)
    torch.save(net.state_dict()["net"], os.path.join(root, "net.pth"))
    d = model.meta["outputdim"]
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    with open(os.path.join(root, "whiten.pkl"), "wb") as f:
        pickle.dump({"P": (q * rng.uniform(0.5, 2.0, (1, d))).T.copy(), "m": rng.normal(0, 0.01, (d, 1))}, f)
    overlay = {"network": {"path": os.path.join(root, "net.pth"),
                           "runtime": {"wrappers": {"eval": {"0_cirwhiten": {"whitening": os.path.join(root, "whiten.pkl")}}}}},
               "validation": {"roxford5k": {"criterion": {"image_size": 320}},
                              "247tokyo1k": {"criterion": {"image_size": 320}},
                              "rparis6k*": False}}
    with open(os.path.join(root, "eval_synth.yml"), "w") as f:
        yaml.safe_dump(overlay, f)
    print("synthetic evaluation set-up written to", root)


if __name__ == "__main__":
    main(*sys.argv[1:])
