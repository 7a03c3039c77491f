"""End-to-end descriptors/sec INCLUDING the image loader: JPEG files on disk -> decode, thumbnail,
normalise in DataLoader workers -> H2D -> 3-scale ResNet101-GeM -> device descriptor matrix."""
import argparse, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=96)
    ap.add_argument("--arch", default="resnet101")
    ap.add_argument("--workers", default="6,16,32")
    args = ap.parse_args()
    from mdir_amd.datasets import initialize_transforms
    from mdir_amd.networks import extract_vectors_device, init_network
    tmp = tempfile.mkdtemp()
    rng = np.random.default_rng(0)
    paths = []
    for i in range(args.images):
        w, h = (1600, 1200) if i % 4 else (1200, 1600)          # camera-sized originals, thumbnailed to 1024
        base = rng.integers(0, 255, (h // 16, w // 16, 3), dtype=np.uint8)
        img = Image.fromarray(base).resize((w, h), Image.BICUBIC)
        p = os.path.join(tmp, "im%03d.jpg" % i)
        img.save(p, quality=90)
        paths.append(p)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = init_network({"architecture": args.arch, "pooling": "gem", "whitening": False, "pretrained": False}).to(dev).eval()
    tr = initialize_transforms("pil2np | totensor | normalize", [net.meta["mean"], net.meta["std"]])
    ms = [1, 2 ** -0.5, 0.5]
    os.environ["MDIR_AMD_WORKERS"] = "6"
    extract_vectors_device(net, paths[:24], 1024, tr, ms=ms, msp=net.pool.p_value(), device=dev, print_freq=1000)   # warm-up (MIOpen)
    from mdir_amd.graphs import ShapeGraphs
    made, orig = [], ShapeGraphs.__init__

    def spy(self, *a, **k):
        orig(self, *a, **k)
        made.append(self)
    ShapeGraphs.__init__ = spy
    for wk in [int(x) for x in args.workers.split(",")]:
        os.environ["MDIR_AMD_WORKERS"] = str(wk)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        extract_vectors_device(net, paths, 1024, tr, ms=ms, msp=net.pool.p_value(), device=dev, print_freq=1000)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("workers %2d: %.1f descriptors/s (%.2f ms per image, %d images, loader start-up included); graph replays %s refused %s"
              % (wk, args.images / dt, 1e3 * dt / args.images, args.images, made[-1].replays if made else None,
                 made[-1].refused if made else None), flush=True)


main()
