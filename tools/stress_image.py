"""Soak test of the image front end on the device against Pillow: random JPEG files (size, chroma subsampling, quality,
progressive or not, grey, restart markers) through the split decoder, then a random LANCZOS thumbnail; every pixel must
be Pillow's.  python tools/stress_image.py [seed] [iterations]"""
import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from mdir_amd import jpeg
from mdir_amd.resample import DeviceThumbnail, on_device

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
bad = declined = shrunk = 0
t0 = time.time()
for it in range(iters):
    w, h = int(rng.integers(16, 1800)), int(rng.integers(2, 1400))
    kind = it % 4
    if kind == 0:
        a = rng.integers(0, 256, (h, w, 3))
    elif kind == 1:
        low = rng.integers(0, 255, (h // 16 + 1, w // 16 + 1, 3)).astype(np.float32)
        a = np.clip(np.kron(low, np.ones((16, 16, 1), np.float32))[:h, :w] + rng.normal(0, 8, (h, w, 3)), 0, 255)
    elif kind == 2:
        a = rng.integers(0, 2, (h, w, 3)) * 255
    else:
        yy, xx = np.mgrid[0:h, 0:w]
        a = np.stack([xx * 255 // max(w - 1, 1), yy * 255 // max(h - 1, 1), (xx * 7 + yy * 13) % 256], axis=2)
    im = Image.fromarray(a.astype(np.uint8))
    kw = {"quality": int(rng.integers(1, 101)), "subsampling": int(rng.integers(0, 3))}
    if rng.random() < 0.4:
        kw["progressive"] = True
    elif rng.random() < 0.3:
        kw["restart_marker_blocks"] = int(rng.integers(1, 40))
    if rng.random() < 0.15:
        im, kw = im.convert("L"), {k: v for k, v in kw.items() if k != "subsampling"}
    buf = io.BytesIO()
    try:
        im.save(buf, format="JPEG", **kw)
    except OSError:
        continue                                    # Pillow's encoder buffer: not a decoder matter
    data = buf.getvalue()
    want = Image.open(io.BytesIO(data)).convert("RGB")
    item = jpeg.entropy_decode(data)
    if item is None:
        declined += 1
        continue
    got = jpeg.pixels(item, dev)
    if not np.array_equal(got[0].cpu().numpy(), np.asarray(want)):
        bad += 1
        print("DECODE MISMATCH", (w, h), kw, flush=True)
        continue
    imsize = int(rng.choice([64, 224, 362, 800, 1024]))
    if on_device(w, h, imsize) is not None:
        shrunk += 1
        want.thumbnail((imsize, imsize), Image.LANCZOS)
        if not np.array_equal(DeviceThumbnail(imsize)(got)[0].cpu().numpy(), np.asarray(want)):
            bad += 1
            print("THUMBNAIL MISMATCH", (w, h), imsize, flush=True)
print("image stress done: %d files, %d declined, %d thumbnails, %d mismatches, %.1f s" % (iters, declined, shrunk, bad, time.time() - t0))
