"""Round-4 GPU tests: the loader's error contract through the whole extraction, the second caller's sequence
(cirtorch/examples/test.py), the split-precision similarity mode."""
import io
import os
import pickle
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn
from PIL import Image

from conftest import ROOT
from oracle import chain as OC
from oracle import oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

sys.path.insert(0, os.path.join(ROOT, "tests"))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _alexnet_ckpt(tmp_path):
    from mdir_amd.network import CirNetwork, SingleNetwork
    from mdir_amd.networks import init_network
    torch.manual_seed(0)
    model_params = {"architecture": "cirnet", "cir_architecture": "alexnet", "local_whitening": False,
                    "pooling": "gem", "regional": False, "whitening": False, "pretrained": True}
    model = init_network({"architecture": "alexnet", "pretrained": False})
    model.meta["in_channels"], model.meta["out_channels"] = 3, 256
    runtime = {"wrappers": "cirmultiscale:True", "data": {"transforms": "pil2np | totensor | normalize"}}
    net = CirNetwork(model, SingleNetwork.NetworkParams(model_params, runtime), "cpu", frozen=True)
    ckpt = str(tmp_path / "net.pth")
    torch.save(net.state_dict()["net"], ckpt)
    return ckpt, net


def test_malformed_jpeg_through_extraction_and_infer(tmp_path, monkeypatch):
    """genericdataset.py:52-59 / stages/infer.py:50-51 on the default route (JPEG coefficients, thread loader, graphs):
    a list holding VERDICT round 3's 224-byte file, a text file and a missing file -- extract_vectors raises the loader's
    OSError (the process lives), the infer stage (ignore_errors) gives NaN rows exactly there and the other rows are the
    descriptors of the clean list; a truncated file is an image (LOAD_TRUNCATED_IMAGES, datahelpers.py:7), as in the reference."""
    import fuzz_jpeg
    from mdir_amd import stages
    from mdir_amd.datasets import initialize_transforms
    from mdir_amd.networks import extract_vectors
    from test_host_api import _write_images
    monkeypatch.setenv("MDIR_AMD_WORKERS", "3")
    rng = np.random.default_rng(5)
    names = ["im%02d" % i for i in range(10)]
    root = tmp_path / "imgs"
    _write_images(str(root), names, rng, size=(224, 160))
    (root / "poc.jpg").write_bytes(b"\xff\xd8" + fuzz_jpeg.dht(0x10, [200] + [0] * 15, [0] * 200) + b"\xff\xd9")
    (root / "text.jpg").write_bytes(b"not a picture\n" * 20)
    whole = (root / "im03.jpg").read_bytes()
    (root / "cut.jpg").write_bytes(whole[:len(whole) * 3 // 5])
    ckpt, net = _alexnet_ckpt(tmp_path)
    good = [n + ".jpg" for n in names]
    images = good[:2] + ["poc.jpg"] + good[2:5] + ["text.jpg", "cut.jpg"] + good[5:] + ["missing.jpg"]
    bad = [2, 6, len(images) - 1]
    params = {"network": {"path": ckpt, "runtime": {}},
              "data": {"test": {"dataset": {"name": "CirImageList", "image_dir": str(root), "image_size": 224, "ignore_errors": True}}},
              "output": {"inference": {"name": "embedding"}}}
    meta, imgs_out, vecs = stages.infer(params, (images,))
    assert imgs_out == images and vecs.shape == (len(images), 256)
    assert [bool(np.isnan(v).all()) for v in vecs] == [i in bad for i in range(len(images))]
    tr = initialize_transforms("pil2np | totensor | normalize", net.network_params.runtime["data"]["mean_std"])
    gpu_net = stages.load_network(params["network"], DEV).eval()
    clean = [str(root / x) for i, x in enumerate(images) if i not in bad]
    with torch.no_grad():
        want = extract_vectors(gpu_net, clean, 224, tr, device=DEV).numpy()
    np.testing.assert_allclose(np.delete(vecs, bad, axis=0), want.T, rtol=0, atol=2e-6)
    # the truncated file is Pillow's padded picture: its descriptor is the one of that picture saved losslessly
    cut = Image.open(io.BytesIO((root / "cut.jpg").read_bytes())).convert("RGB")
    cut.save(root / "cut.png")
    with torch.no_grad():
        v = extract_vectors(gpu_net, [str(root / "cut.png")], 224, tr, device=DEV).numpy()
    np.testing.assert_allclose(vecs[7], v[:, 0], rtol=0, atol=2e-6)
    # without ignore_errors: the reference re-raises the loader's OSError
    for broken in ("poc.jpg", "text.jpg", "missing.jpg"):
        with pytest.raises(OSError):
            with torch.no_grad():
                extract_vectors(gpu_net, [str(root / x) for x in good[:3] + [broken] + good[3:]], 224, tr, device=DEV)
    # and the process still extracts afterwards
    with torch.no_grad():
        again = extract_vectors(gpu_net, clean, 224, tr, device=DEV).numpy()
    np.testing.assert_array_equal(again, want)
