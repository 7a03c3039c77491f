"""Steady-state throughput of the image loader alone (no GPU work): which part bounds it."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from mdir_amd.datasets import ImagesFromList, ToUint8HWC, initialize_transforms

tmp = tempfile.mkdtemp(); rng = np.random.default_rng(0); paths = []
for i in range(64):
    w, h = (1600, 1200)
    img = Image.fromarray(rng.integers(0, 255, (h // 16, w // 16, 3), dtype=np.uint8)).resize((w, h), Image.BICUBIC)
    p = os.path.join(tmp, "im%03d.jpg" % i); img.save(p, quality=90); paths.append(p)
paths = paths * 6
full = initialize_transforms("pil2np | totensor | normalize", [[0.485, 0.456, 0.406], [0.229, 0.224, 0.225]])
for name, tr in (("float chain", full), ("uint8", ToUint8HWC())):
    for wk in (6, 16):
        for pin in (False, True):
            dl = torch.utils.data.DataLoader(ImagesFromList("", paths, imsize=1024, transform=tr), batch_size=1,
                                             num_workers=wk, pin_memory=pin)
            t0 = time.perf_counter(); n = 0
            for i, x in enumerate(dl):
                if i == 2 * wk:
                    t1 = time.perf_counter()
                n += 1
            t2 = time.perf_counter()
            print("%-12s workers %2d pin %d: start-up %.2f s, then %.1f images/s" % (name, wk, pin, t1 - t0, (n - 2 * wk) / (t2 - t1)), flush=True)

# the thread-pool loader (the default of the extraction loops): Pillow does everything / only decodes (thumbnail on the device) /
# only undoes the entropy coding (IDCT, upsampling, colour conversion and thumbnail on the device)
if torch.cuda.is_available():
    from mdir_amd.datasets import ThreadedLoader
    from mdir_amd.jpeg import pixels
    from mdir_amd.resample import DeviceThumbnail
    dev = torch.device("cuda:0")
    shrink = DeviceThumbnail(1024)
    for name, kw in (("Pillow decodes and shrinks", {}), ("Pillow decodes, device shrinks", {"resize_on_device": True}),
                     ("host entropy stage, device decodes and shrinks", {"resize_on_device": True, "decode_on_device": True})):
        for wk in (2, 4, 8, 16):
            dl = ThreadedLoader(ImagesFromList("", paths, imsize=1024, transform=ToUint8HWC(), **kw), range(len(paths)), wk)
            n = 0
            for i, x in enumerate(dl):
                if i == 2 * wk:
                    torch.cuda.synchronize(); t1 = time.perf_counter()
                y = pixels(x, dev) if hasattr(x, "coef") else x.to(dev, non_blocking=True)
                y = shrink(y) if kw else y
                assert tuple(y.shape) == (1, 768, 1024, 3)
                n += 1
            torch.cuda.synchronize(); t2 = time.perf_counter()
            print("%-48s threads %2d: %.1f images/s into device memory (1600x1200 JPEG -> 1024x768 uint8)" % (name, wk, (n - 2 * wk) / (t2 - t1)), flush=True)
