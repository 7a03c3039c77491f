"""Start-up cost of DataLoader workers from a GPU-initialised parent: fork vs forkserver."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from mdir_amd.datasets import ImagesFromList, ToUint8HWC

if __name__ == "__main__":
    tmp = tempfile.mkdtemp(); rng = np.random.default_rng(0); paths = []
    for i in range(32):
        img = Image.fromarray(rng.integers(0, 255, (75, 100, 3), dtype=np.uint8)).resize((1600, 1200), Image.BICUBIC)
        p = os.path.join(tmp, "im%03d.jpg" % i); img.save(p, quality=90); paths.append(p)
    paths = paths * 12
    x = torch.randn(1 << 28, device="cuda")            # GPU-initialised parent with a few GB mapped
    torch.cuda.synchronize()
    for ctx in ("fork", "forkserver"):
        for wk in (6, 16):
            dl = torch.utils.data.DataLoader(ImagesFromList("", paths, imsize=1024, transform=ToUint8HWC()), batch_size=1,
                                             num_workers=wk, pin_memory=True, multiprocessing_context=ctx)
            t0 = time.perf_counter(); n = 0
            for i, item in enumerate(dl):
                if i == 0:
                    t1 = time.perf_counter()
                n += 1
            t2 = time.perf_counter()
            del dl
            t3 = time.perf_counter()
            print("%-10s workers %2d: first item after %.2f s, %d items in %.2f s (%.0f/s steady), teardown %.2f s"
                  % (ctx, wk, t1 - t0, n, t2 - t0, (n - 1) / (t2 - t1), t3 - t2), flush=True)
