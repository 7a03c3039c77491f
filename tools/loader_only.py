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

# the LANCZOS thumbnail in the workers (Pillow) against the workers decoding only + mdx_resample_u8 on the device
if torch.cuda.is_available():
    from mdir_amd.resample import DeviceThumbnail
    dev = torch.device("cuda:0")
    shrink = DeviceThumbnail(1024)
    for name, deferred in (("thumbnail in the workers", False), ("thumbnail on the device", True)):
        for wk in (8, 16):
            dl = torch.utils.data.DataLoader(ImagesFromList("", paths, imsize=1024, transform=ToUint8HWC(), resize_on_device=deferred),
                                             batch_size=1, num_workers=wk, pin_memory=True)
            n = 0
            for i, x in enumerate(dl):
                if i == 2 * wk:
                    torch.cuda.synchronize(); t1 = time.perf_counter()
                y = shrink(x.to(dev, non_blocking=True)) if deferred else x.to(dev, non_blocking=True)
                assert tuple(y.shape) == (1, 768, 1024, 3)
                n += 1
            torch.cuda.synchronize(); t2 = time.perf_counter()
            print("%-26s workers %2d: %.1f images/s into device memory (1600x1200 JPEG -> 1024x768 uint8)" % (name, wk, (n - 2 * wk) / (t2 - t1)), flush=True)
    x = torch.randint(0, 255, (1, 1200, 1600, 3), dtype=torch.uint8, device=dev)
    shrink(x); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        shrink(x)
    b.record(); torch.cuda.synchronize()
    print("device thumbnail 1600x1200 -> 1024x768: %.1f us per image (two launches)" % (a.elapsed_time(b) / 50 * 1e3))
