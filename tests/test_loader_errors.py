"""The loader's error contract on malformed images, on the DEFAULT route (JPEG coefficients for the device).

Reference: ``cirtorch/datasets/genericdataset.py:52-59`` -- ``pil_loader`` (datahelpers.py:24-31) returns the OSError, the
dataset writes a warning and re-raises it, or returns ``{}`` under ``ignore_errors`` (consumed at ``mdir/stages/infer.py:50-51``
as a NaN row).  VERDICT round 3: the default route handed raw bytes to the library's own header parser BEFORE Pillow had seen
them, and a 224-byte file killed the process.  Now Pillow parses the header first, the parser is bounds-checked (and fuzzed
under ASan, test_fuzz_asan.py), and every malformed file takes the host route, which reports it as the reference does.
"""
import io
import os
import sys

import numpy as np
import pytest
import torch
from PIL import Image

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_jpeg  # noqa: E402


def _files(tmp_path):
    """name -> (bytes, what the reference's loader does with it)."""
    rng = np.random.default_rng(11)
    arr = np.clip(np.kron(rng.integers(0, 255, (8, 10, 3)), np.ones((16, 16, 1))) + rng.normal(0, 9, (128, 160, 3)), 0, 255).astype(np.uint8)
    b = io.BytesIO()
    Image.fromarray(arr).save(b, format="JPEG", quality=90)
    good = b.getvalue()
    b = io.BytesIO()
    Image.fromarray(arr).save(b, format="JPEG", quality=90, progressive=True)
    prog = b.getvalue()
    sos = good.find(b"\xff\xda")
    bits = [0] * 16
    bits[0] = 200
    files = {
        "good.jpg": good,
        "prog.jpg": prog,
        # VERDICT round 3's proof of concept: SOI + DHT(bits[1] = 200) + EOI.  Pillow: "cannot identify image file"
        "poc.jpg": b"\xff\xd8" + fuzz_jpeg.dht(0x10, [200] + [0] * 15, [0] * 200) + b"\xff\xd9",
        # the same hostile table inside an otherwise sound file: Pillow opens it (it does not read DHT at open); libjpeg then
        # refuses the table, which LOAD_TRUNCATED_IMAGES turns into a (grey) picture
        "badtable.jpg": good[:sos] + fuzz_jpeg.dht(0x10, bits, [0] * 200) + good[sos:],
        # cut in the middle of the scan: ImageFile.LOAD_TRUNCATED_IMAGES (datahelpers.py:7) makes Pillow pad it -- an IMAGE, not an error
        "cut.jpg": good[:sos + (len(good) - sos) // 2],
        "cutprog.jpg": prog[:len(prog) * 2 // 3],
        # headers only, no scan data
        "headers.jpg": good[:sos],
        "text.jpg": b"this is not a picture\n" * 10,
        "empty.jpg": b"",
        # a frame header announcing 65535 x 65535 on a 3 KB file
        "bomb.jpg": None,
    }
    sof = good.find(b"\xff\xc0")
    g = bytearray(good)
    g[sof + 5:sof + 9] = (65535).to_bytes(2, "big") * 2
    files["bomb.jpg"] = bytes(g)
    for name, data in files.items():
        (tmp_path / name).write_bytes(data)
    return files


def _reference_loader(path):
    """``pil_loader`` + the ``__getitem__`` branch of the reference, restated: what the item must be / raise."""
    try:
        with open(path, "rb") as f:
            return Image.open(f).convert("RGB")
    except OSError as e:
        return e


def test_malformed_files_take_the_reference_route(tmp_path, capsys):
    from mdir_amd.datasets import ImagesFromList, ToUint8HWC
    from mdir_amd.jpeg import JpegCoefficients
    files = _files(tmp_path)
    names = sorted(files) + ["missing.jpg"]
    kw = dict(imsize=None, transform=ToUint8HWC(), resize_on_device=True, decode_on_device=True)
    strict = ImagesFromList(str(tmp_path), names, **kw)
    lenient = ImagesFromList(str(tmp_path), names, ignore_errors=True, **kw)
    assert strict.decode_on_device
    for i, name in enumerate(names):
        path = str(tmp_path / name)
        try:
            want = _reference_loader(path)
            raised = None
        except Exception as e:                      # not an OSError: the reference lets it fly out of the worker
            want, raised = None, type(e)
        if raised is not None:
            with pytest.raises(raised):
                strict[i]
            with pytest.raises(raised):
                lenient[i]
        elif isinstance(want, Exception):
            with pytest.raises(type(want)):
                strict[i]
            assert lenient[i] == {}, name
            assert "Warning: Image '%s' was not found" % path in capsys.readouterr().err
        else:
            for ds in (strict, lenient):
                item = ds[i]
                if isinstance(item, JpegCoefficients):
                    assert name in ("good.jpg", "prog.jpg"), name
                else:                               # the host route's pixels are Pillow's (padded truncated files included)
                    np.testing.assert_array_equal(item.numpy(), np.asarray(want), err_msg=name)
    # the sound files DO leave as coefficients; nothing malformed does
    def kind(i):
        try:
            return isinstance(lenient[i], JpegCoefficients)
        except Image.DecompressionBombError:          # bomb.jpg: not an OSError, flies out of the reference's loader too
            return False
    kinds = {n: kind(i) for i, n in enumerate(names)}
    assert kinds["good.jpg"] and kinds["prog.jpg"] and sum(kinds.values()) == 2, kinds
    # what the reference does with the interesting ones, pinned so that the cases above stay meaningful
    assert isinstance(_reference_loader(str(tmp_path / "poc.jpg")), OSError)
    # (badtable.jpg: with LOAD_TRUNCATED_IMAGES Pillow swallows libjpeg's complaint and returns a picture -- so does this loader, above)
    assert isinstance(_reference_loader(str(tmp_path / "cut.jpg")), Image.Image)
    assert isinstance(_reference_loader(str(tmp_path / "missing.jpg")), FileNotFoundError)


def test_threaded_loader_survives_malformed_files(tmp_path):
    """Through the thread-pool loader (the default of extract_vectors): ``{}`` at the malformed items' turns under
    ``ignore_errors``, the reference's OSError at the first of them otherwise -- and a live process either way."""
    from mdir_amd.datasets import ImagesFromList, ThreadedLoader, ToUint8HWC
    from mdir_amd.jpeg import JpegCoefficients
    _files(tmp_path)
    names = ["good.jpg", "poc.jpg", "prog.jpg", "badtable.jpg", "good.jpg", "text.jpg", "missing.jpg", "cut.jpg"]
    kw = dict(imsize=None, transform=ToUint8HWC(), resize_on_device=True, decode_on_device=True)
    items = list(ThreadedLoader(ImagesFromList(str(tmp_path), names, ignore_errors=True, **kw), range(len(names)), workers=3, pin_memory=False))
    unreadable = [isinstance(_reference_loader(str(tmp_path / n)), Exception) for n in names]
    assert unreadable == [False, True, False, False, False, True, True, False]
    assert [isinstance(x, dict) and x == {} for x in items] == unreadable
    assert [isinstance(x, JpegCoefficients) for x in items] == [True, False, True, False, True, False, False, False]
    for i in (3, 7):                                            # hostile table / truncated file: Pillow's picture, from the host route
        np.testing.assert_array_equal(items[i][0].numpy(), np.asarray(_reference_loader(str(tmp_path / names[i]))))
    seen = []
    with pytest.raises(OSError):
        for x in ThreadedLoader(ImagesFromList(str(tmp_path), names, **kw), range(len(names)), workers=3, pin_memory=False):
            seen.append(x)
    assert len(seen) == 1


def test_entropy_decode_never_raises_on_garbage(tmp_path):
    """``jpeg.entropy_decode`` on the hostile set of the fuzz driver, through the shipped library: None or coefficients."""
    from mdir_amd import jpeg
    rng = np.random.default_rng(0)
    seeds = fuzz_jpeg.seeds(rng)
    n = 0
    for data in fuzz_jpeg.hostile(seeds) + [fuzz_jpeg.mutate(rng, seeds[i % len(seeds)], seeds[(7 * i) % len(seeds)]) for i in range(3000)]:
        if len(data) == 0:
            continue
        item = jpeg.entropy_decode(data)
        if item is not None:
            n += 1
            assert item.coef.shape == (item.info.nblocks, 64) and item.info.width * item.info.height // 512 <= len(data)
    assert n > 300
