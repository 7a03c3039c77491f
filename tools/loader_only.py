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
