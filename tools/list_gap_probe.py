"""Where does the per-image time of extract_vectors_device on an image LIST go?  Same network and shapes as
tools/bench_extract.py --list; (a) graph replays of batches of 4 with inputs already on the device, shape after shape;
(b) the same fed from host tensors (pinned H2D + cat + replay); (c) the real loop with the loader."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from bench_extract import LIST_SHAPES, _write_jpegs
from mdir_amd import ops
from mdir_amd.datasets import initialize_transforms
from mdir_amd.graphs import ShapeGraphs
from mdir_amd.network import CirNetwork, SingleNetwork
from mdir_amd.networks import extract_vectors_device, init_network

dev = torch.device("cuda:0")
torch.manual_seed(3)
model = init_network({"architecture": "resnet101", "pooling": "gem", "whitening": False, "pretrained": False})
D = model.meta["outputdim"]
model.meta["in_channels"], model.meta["out_channels"] = 3, D
rng = np.random.default_rng(2)
qm, _ = np.linalg.qr(rng.standard_normal((D, D)))
wh = {"P": (qm * rng.uniform(0.5, 2.0, (1, D))).T.copy(), "m": rng.normal(0, 0.01, (D, 1))}
mp = {"architecture": "cirnet", "cir_architecture": "resnet101", "local_whitening": False, "pooling": "gem", "regional": False, "whitening": False, "pretrained": False}
rt = {"wrappers": {"train": "", "eval": {"0_cirwhiten": {"whitening": wh, "dimensions": None}, "1_cirmultiscale": {"scales": True}}}, "data": {"transforms": "pil2np | totensor | normalize"}}
net = CirNetwork(model.to(dev), SingleNetwork.NetworkParams(mp, rt), dev, frozen=True).eval()
tr = initialize_transforms("pil2np | totensor | normalize", net.network_params.runtime["data"]["mean_std"])
mean, std = tr.device_tail()
chain = net.wrappers["eval"]
held = chain.defer_final_whitening()
describe = ShapeGraphs(lambda u8: net(ops.u8_to_chw(u8, mean, std)), warmup=1)
with torch.no_grad():
    u8 = {s: torch.randint(0, 255, (4, s[1], s[0], 3), dtype=torch.uint8, device=dev) for s in LIST_SHAPES}
    for s in LIST_SHAPES:
        for _ in range(3):
            describe(u8[s])
    torch.cuda.synchronize()
    print("graphs", len(describe.graphs), "replays", describe.replays)
    reps = 4
    t0 = time.perf_counter()
    for _ in range(reps):
        for s in LIST_SHAPES:
            for _ in range(4):
                describe(u8[s])
    torch.cuda.synchronize()
    ta = (time.perf_counter() - t0) / (reps * 16 * 16)
    print("(a) resident inputs, replays shape after shape: %.2f ms per image" % (1e3 * ta))
    host = {s: [torch.randint(0, 255, (1, s[1], s[0], 3), dtype=torch.uint8).pin_memory() for _ in range(4)] for s in LIST_SHAPES}
    t0 = time.perf_counter()
    for _ in range(reps):
        for s in LIST_SHAPES:
            for _ in range(4):
                describe(torch.cat([h.to(dev, non_blocking=True) for h in host[s]], dim=0))
    torch.cuda.synchronize()
    tb = (time.perf_counter() - t0) / (reps * 16 * 16)
    print("(b) + pinned H2D and cat per batch: %.2f ms per image" % (1e3 * tb))
chain.restore_whitening(held)
with tempfile.TemporaryDirectory() as folder:
    files = _write_jpegs(folder, LIST_SHAPES, 8)
    paths = [f for r in range(8) for f in files]          # 64 per size
    for tag in ("first", "second"):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.no_grad():
            extract_vectors_device(net, paths, 1024, tr, device=dev, num_workers=8, print_freq=10 ** 9)
        torch.cuda.synchronize()
        print("(c) %s full list of %d through the loader: %.2f ms per image" % (tag, len(paths), 1e3 * (time.perf_counter() - t0) / len(paths)))

# (d) the loop of extract_vectors_device taken apart: time blocked in the loader's next() against time in the device calls
from mdir_amd.datasets import ImagesFromList, ToUint8HWC
from mdir_amd.networks import ShapeOrder
with tempfile.TemporaryDirectory() as folder:
    files = _write_jpegs(folder, LIST_SHAPES, 8)
    paths = [f for r in range(8) for f in files]
    for mode in ("loader + device work", "graphs kept, loader items ignored (resident inputs)", "graphs kept, ignored, no pin", "graphs kept, ignored, no pin, forkserver",
                 "graphs kept, ignored, no pin, 2 workers"):
        order = ShapeOrder(paths, None)
        loader = torch.utils.data.DataLoader(ImagesFromList(root="", images=paths, imsize=1024, transform=ToUint8HWC(), resize_on_device=True),
                                             batch_size=1, shuffle=False, sampler=order, num_workers=2 if "2 workers" in mode else 8, pin_memory="no pin" not in mode,
                                             multiprocessing_context="forkserver" if "forkserver" in mode else None)
        if not mode.startswith("graphs kept") and mode != "loader + device work, graphs kept":
            describe = ShapeGraphs(lambda u8: net(ops.u8_to_chw(u8, mean, std)), warmup=1)
        held = chain.defer_final_whitening()
        t_wait = t_dev = 0.0
        buf = []
        evs = []
        torch.cuda.synchronize(); t_all = time.perf_counter()
        it = iter(loader)
        with torch.no_grad():
            for k in range(len(paths)):
                t0 = time.perf_counter()
                item = next(it)
                t1 = time.perf_counter()
                t_wait += t1 - t0
                if mode != "loader only":
                    item = u8[(item.shape[2], item.shape[1])][:1] if "ignored" in mode else item.to(dev, non_blocking=True)
                    if buf and buf[0].shape != item.shape:
                        for b in buf: describe(b)
                        buf = []
                    buf.append(item)
                    if len(buf) == 4:
                        describe.upcoming = order.upcoming[k]
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(); describe(torch.cat(buf, dim=0)); e1.record(); evs.append((e0, e1)); buf = []
                    t_dev += time.perf_counter() - t1
        torch.cuda.synchronize()
        chain.restore_whitening(held)
        tot = time.perf_counter() - t_all
        if evs:
            print("     GPU time between the events around the batch calls: %.2f ms per image" % (sum(a.elapsed_time(b) for a, b in evs) / (4 * len(evs))))
        print("(d) %-36s %.2f ms per image; blocked in next(): %.2f, in device calls (host side): %.2f; graphs %d captures %d"
              % (mode, 1e3 * tot / len(paths), 1e3 * t_wait / len(paths), 1e3 * t_dev / len(paths), len(describe.graphs), describe.captures))
