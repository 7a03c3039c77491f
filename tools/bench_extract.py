#!/usr/bin/env python3
"""Descriptors/sec of the extraction path on one GPU (BASELINE.json configs[1]/[4] shape):
ResNet101-GeM (or VGG16-GeM), random weights, synthetic 1024x768 images already resident on the
device, 3-scale pyramid + learned whitening through the mdir wrapper chain
(0_cirwhiten + 1_cirmultiscale), descriptors written to one device [N,D] buffer.

Reports the split the hand-written part is responsible for: backbone (PyTorch-ROCm/MIOpen) vs
descriptor tail (mdx_pool_l2n, mdx_ms_aggregate, mdx_scores(P)+mdx_l2n_rows), measured with HIP
events on the current stream, and the same tail expressed with stock torch ops (the reference's
LF.gem / LF.l2n / aggregate_tensor / CirtorchWhiten.postprocess statements) for comparison.

    python tools/bench_extract.py [--arch resnet101] [--images 30] [--fp16-backbone]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="resnet101")
    ap.add_argument("--images", type=int, default=30)
    ap.add_argument("--channels-last", action="store_true")
    ap.add_argument("--no-graphs", action="store_true", help="eager launches instead of one hipGraph replay per image")
    ap.add_argument("--batch", type=int, default=4, help="equal-sized images per trunk pass (1 = the reference's batch size)")
    ap.add_argument("--miopen-find", action="store_true", help="torch.backends.cudnn.benchmark = True (MIOpen find mode)")
    print(json.dumps(measure(ap.parse_args())))


def measure(args):
    """``args``: namespace with arch / images / channels_last / miopen_find (also called by bench.py)."""
    from mdir_amd import ops
    from mdir_amd.networks import init_network
    from mdir_amd.wrapper import initialize_wrappers
    dev = torch.device("cuda", torch.cuda.current_device())
    torch.backends.cudnn.benchmark = bool(args.miopen_find)
    torch.manual_seed(3)
    net = init_network({"architecture": args.arch, "pooling": "gem", "whitening": False, "pretrained": False})
    net.meta["in_channels"], net.meta["out_channels"] = 3, net.meta["outputdim"]
    net = net.to(dev).eval()
    if args.channels_last:
        net.features = net.features.to(memory_format=torch.channels_last)
    D = net.meta["outputdim"]
    rng = np.random.default_rng(2)
    q, _ = np.linalg.qr(rng.standard_normal((D, D)))
    wh = {"P": (q * rng.uniform(0.5, 2.0, (1, D))).T.copy(), "m": rng.normal(0, 0.01, (D, 1))}
    chain = initialize_wrappers({"0_cirwhiten": {"whitening": wh, "dimensions": None},
                                 "1_cirmultiscale": {"scales": True}}, dev)
    imgs = [torch.randn(1, 3, 768, 1024, device=dev) for _ in range(4)]
    if args.channels_last:
        imgs = [i.contiguous(memory_format=torch.channels_last) for i in imgs]
    vecs = torch.empty(args.images, D, device=dev)
    P32 = torch.tensor(wh["P"], dtype=torch.float32, device=dev)
    m32 = torch.tensor(wh["m"], dtype=torch.float32, device=dev)

    from mdir_amd.graphs import ShapeGraphs, graphs_enabled
    describe = lambda x: chain(x, net)
    if graphs_enabled(dev) and not getattr(args, "no_graphs", False):
        describe = ShapeGraphs(describe)       # as extract_vectors_device does

    bmax = 1 if getattr(args, "no_graphs", False) else max(1, getattr(args, "batch", 4))

    def run(n):                 # as extract_vectors_device: equal-sized images in batches of bmax
        i = 0
        while i < n:
            if bmax > 1 and i + bmax <= n:
                rows = describe(torch.cat([imgs[(i + j) % 4] for j in range(bmax)], dim=0))
                for j in range(bmax):
                    vecs[(i + j) % args.images].copy_(rows[j].reshape(-1))
                i += bmax
            else:
                vecs[i % args.images].copy_(describe(imgs[i % 4]).reshape(-1))
                i += 1

    with torch.no_grad():
        run(3 * bmax)           # eager warm-up of every (shape, batch) + the graph captures
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(args.images)
        torch.cuda.synchronize()
        total = time.perf_counter() - t0

        # split: features only / tail only (on precomputed feature maps)
        pyr = [F.interpolate(imgs[0], scale_factor=s, mode="bilinear", align_corners=False) if s != 1 else imgs[0]
               for s in chain.wrappers[1].scales]
        feats = [net.features(x).contiguous() for x in pyr]
        ev = lambda: torch.cuda.Event(enable_timing=True)
        reps = 20

        def timeit(fn):
            fn(); torch.cuda.synchronize()
            a, b = ev(), ev()
            a.record()
            for _ in range(reps):
                fn()
            b.record(); torch.cuda.synchronize()
            return a.elapsed_time(b) / reps

        t_backbone_eager = timeit(lambda: [net.features(x) for x in pyr])
        trunk = ShapeGraphs(lambda x: [net.features(p) for p in
                                       [x] + [F.interpolate(x, scale_factor=s, mode="bilinear", align_corners=False)
                                              for s in chain.wrappers[1].scales if s != 1]], warmup=1)
        trunk(imgs[0]); trunk(imgs[0])
        t_backbone = timeit(lambda: trunk(imgs[0])) if trunk.graphs else t_backbone_eager
        p = net.pool.p_value()

        def tail_mdx():     # per image, as the wrapper chain runs it: ONE pooling launch over the three maps, ONE launch for the
            return ops.l2n_aggregate(ops.pool_multi(feats, "gem", p, 1e-6), 1e-6, p)   # three L2Ns + the aggregation; the
                                                                                   # whitening goes over the finished [N,D] matrix
        def tail_mdx_r1():  # the seven launches this replaced (pool + L2N per scale, batched aggregation)
            return ops.ms_aggregate_batch([ops.pool_l2n(f, "gem", p, 1e-6, 1e-6) for f in feats], p)

        rows = torch.cat([tail_mdx() for _ in range(args.images)], dim=0)          # [N,D] unwhitened descriptors

        def tail_whiten_all():
            return chain.wrappers[0].whiten_rows(rows)

        def tail_torch():   # the reference's statements with stock torch ops
            per = []
            for f in feats:
                o = F.avg_pool2d(f.clamp(min=1e-6).pow(p), (f.size(-2), f.size(-1))).pow(1. / p)
                per.append((o / (torch.norm(o, p=2, dim=1, keepdim=True) + 1e-6)).squeeze(-1).squeeze(-1).permute(1, 0))
            v = torch.zeros(D, device=dev)
            for s in per:
                v += s.pow(p).squeeze()
            v = (v / len(per)).pow(1. / p)
            v /= v.norm()
            X = P32.mm(v.unsqueeze(1).sub(m32))
            return X.div(torch.norm(X, p=2, dim=0, keepdim=True) + 1e-6).squeeze()

        a, b = tail_whiten_all()[0].reshape(-1), tail_torch().reshape(-1)
        err = float((a - b).abs().max())
        assert torch.equal(tail_mdx(), tail_mdx_r1())
        t_tail_seven = timeit(tail_mdx_r1) + timeit(tail_whiten_all) / args.images
        t_tail_eager = timeit(tail_mdx) + timeit(tail_whiten_all) / args.images
        # as extraction runs it: inside a hipGraph replay (no Python / dispatcher time between the launches)
        tail_graph = ShapeGraphs(lambda x: tail_mdx(), warmup=1)
        dummy = torch.zeros(1, device=dev)
        tail_graph(dummy); tail_graph(dummy)
        t_tail_replay = (timeit(lambda: tail_graph(dummy)) if tail_graph.graphs else t_tail_eager) + timeit(tail_whiten_all) / args.images
        t_tail = t_tail_eager
        t_tail_torch = timeit(tail_torch)
        tail_bytes = sum(4.0 * f.numel() for f in feats) + 4.0 * (len(feats) + 3) * D           # maps read once + the vectors
    return {"metric": "descriptors/sec, %s-GeM, 3 scales of 1024x768 + whitening, 1 GPU" % args.arch,
            "value": round(args.images / total, 2), "unit": "descriptors/s",
            "ms_per_image": round(1e3 * total / args.images, 3),
            "backbone_ms_per_image": round(t_backbone, 3), "backbone_ms_per_image_eager_launches": round(t_backbone_eager, 3),
            "tail_ms_per_image_mdx": round(t_tail, 4), "tail_ms_per_image_mdx_as_its_own_graph_replay": round(t_tail_replay, 4),
            "tail_ms_per_image_torch_ops": round(t_tail_torch, 4), "tail_ms_per_image_seven_launches": round(t_tail_seven, 4),
            "roofline_tail": {"bound": "hbm", "bytes_per_image": tail_bytes, "achieved": round(tail_bytes / (t_tail * 1e-3) / 1e9, 1),
                              "peak": 8000.0, "unit": "GB/s", "frac": round(tail_bytes / (t_tail * 1e-3) / 1e9 / 8000.0, 4),
                              "what": "GeM of the three maps (one launch), their L2Ns + the aggregation (one launch), whitening of the "
                                      "finished [N,D] matrix / N: two short dependent launches per image (or per batch of 8), so launch "
                                      "latency, not bandwidth, bounds it"},
            "tail_max_abs_diff_vs_torch_ops": err, "dtype": "f32", "data": "synthetic",
            "hipgraph_replays": getattr(describe, "replays", 0)}


# rOxford-like sizes (W, H): longer side 1024 after the thumbnail, the usual camera aspect ratios, both orientations
LIST_SHAPES = [(1024, 768), (1024, 683), (1024, 681), (1024, 685), (1024, 576), (1024, 819), (1024, 640), (1024, 724),
               (768, 1024), (683, 1024), (681, 1024), (685, 1024), (576, 1024), (819, 1024), (640, 1024), (724, 1024)]


def _write_jpegs(folder, shapes, per_shape, seed=5):
    """JPEG files of the given sizes (smooth blobs + grain, so that they decode like photographs), shuffled."""
    from PIL import Image
    rng = np.random.default_rng(seed)
    paths = []
    for si, (w, h) in enumerate(shapes):
        for k in range(per_shape):
            low = rng.integers(0, 255, (h // 32 + 1, w // 32 + 1, 3)).astype(np.float32)
            img = np.kron(low, np.ones((32, 32, 1), dtype=np.float32))[:h, :w]
            img = np.clip(img + rng.normal(0, 12, img.shape), 0, 255).astype(np.uint8)
            path = os.path.join(folder, "s%02d_%03d.jpg" % (si, k))
            Image.fromarray(img).save(path, format="JPEG", quality=90)
            paths.append(path)
    return [paths[i] for i in rng.permutation(len(paths))]


def _conv_flops(net, scales, h, w):
    """Multiply-add x 2 of the trunk's convolutions for one image at the given pyramid scales (hooks on an eager pass)."""
    total = [0]

    def hook(mod, inp, out):
        k = mod.kernel_size[0] * mod.kernel_size[1] * (mod.in_channels // mod.groups)
        total[0] += 2 * k * out.numel()

    handles = [m.register_forward_hook(hook) for m in net.features.modules() if isinstance(m, torch.nn.Conv2d)]
    dev = next(net.parameters()).device
    with torch.no_grad():
        for s in scales:
            net.features(torch.zeros(1, 3, int(h * s), int(w * s), device=dev))
    for hd in handles:
        hd.remove()
    return float(total[0])


def measure_list(arch="resnet101", workers=8, short=12, mid=40, long=64):
    """Descriptors/sec of ``extract_vectors_device`` on an image LIST: 16 sizes, JPEG files through the real loader
    (decode + thumbnail in worker processes, uint8 over PCIe, /255-mean-std on the GPU), 3 scales + learned whitening
    through the wrapper chain.  The FIRST list of the process (``short`` images per size: 12 = one batch of eight + one of
    four, the two batch shapes extraction uses) pays for what a new size costs (MIOpen picks and loads its kernels for every
    (size, scale, batch)); two later lists (``mid`` / ``long`` per size, both long enough for a graph per
    size: one eager batch, one capture, replays) differ only in replays, which gives the steady state."""
    import tempfile
    from mdir_amd.datasets import ImagesFromList, initialize_transforms
    from mdir_amd.graphs import ShapeGraphs
    # both warm lists must be long enough per size for the capture to be made (ShapeGraphs.PAYOFF_IMAGES): only then
    # is their difference pure replay time
    mid = max(mid, ShapeGraphs.PAYOFF_IMAGES + 8)
    long = max(long, mid + 24)
    from mdir_amd.network import CirNetwork, SingleNetwork
    from mdir_amd.networks import extract_vectors_device, init_network
    dev = torch.device("cuda", torch.cuda.current_device())
    torch.manual_seed(3)
    model = init_network({"architecture": arch, "pooling": "gem", "whitening": False, "pretrained": False})
    D = model.meta["outputdim"]
    model.meta["in_channels"], model.meta["out_channels"] = 3, D
    rng = np.random.default_rng(2)
    q, _ = np.linalg.qr(rng.standard_normal((D, D)))
    wh = {"P": (q * rng.uniform(0.5, 2.0, (1, D))).T.copy(), "m": rng.normal(0, 0.01, (D, 1))}
    model_params = {"architecture": "cirnet", "cir_architecture": arch, "local_whitening": False, "pooling": "gem",
                    "regional": False, "whitening": False, "pretrained": False}
    runtime = {"wrappers": {"train": "", "eval": {"0_cirwhiten": {"whitening": wh, "dimensions": None}, "1_cirmultiscale": {"scales": True}}},
               "data": {"transforms": "pil2np | totensor | normalize"}}
    net = CirNetwork(model.to(dev), SingleNetwork.NetworkParams(model_params, runtime), dev, frozen=True).eval()
    transform = initialize_transforms("pil2np | totensor | normalize", net.network_params.runtime["data"]["mean_std"])
    out = {}
    with tempfile.TemporaryDirectory() as folder:
        lists = {}
        uniq = 8                                                             # distinct files per size; longer lists repeat them
        files = _write_jpegs(folder, LIST_SHAPES, uniq)                      # shuffled; "sSS_KKK.jpg" = size SS, copy KKK
        for n in (short, mid, long):
            lists[n] = [f for r in range(-(-n // uniq)) for f in files if r * uniq + int(os.path.basename(f)[4:7]) < n]
        times = {}
        # cold: the first list of the process meets every (size, scale, batch) for the first time -- MIOpen picks and
        # loads its kernels there; warm: the same sizes again (new graphs are captured, MIOpen already knows the shapes)
        for tag, paths in (("cold", lists[short]), ("warm_short", lists[short]), ("warm_mid", lists[mid]), ("warm_long", lists[long])):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.no_grad():
                vecs = extract_vectors_device(net, paths, 1024, transform, device=dev, num_workers=workers, print_freq=10 ** 9)
            torch.cuda.synchronize()
            times[tag] = time.perf_counter() - t0
            assert vecs.shape == (len(paths), D) and bool(torch.isfinite(vecs).all())
        # loader alone (decode + thumbnail + uint8 tensor, no GPU work) on the long list
        from mdir_amd.datasets import ToUint8HWC
        from mdir_amd.datasets import make_loader
        loader = make_loader(ImagesFromList("", lists[long], imsize=1024, transform=ToUint8HWC()), range(len(lists[long])), workers, dev)
        t0 = time.perf_counter()
        for _ in loader:
            pass
        t_loader = time.perf_counter() - t0
    ns, nm, nl = short * len(LIST_SHAPES), mid * len(LIST_SHAPES), long * len(LIST_SHAPES)
    steady = (times["warm_long"] - times["warm_mid"]) / (nl - nm)           # s per image once its size has a graph
    first = (times["cold"] - times["warm_short"]) / len(LIST_SHAPES)        # extra s per size the process has never seen
    flops = np.mean([_conv_flops(model, [1, 2 ** -0.5, 0.5], h, w) for w, h in LIST_SHAPES[:1] + LIST_SHAPES[4:5] + LIST_SHAPES[8:9]])
    # the headline figure is a WHOLE list, start to finish, graph captures included (one wall-clock timing); the difference
    # of two such timings -- the steady state between captures -- swings with the box and is a derived estimate only
    return {"value": round(nl / times["warm_long"], 2), "unit": "descriptors/s",
            "what": "extract_vectors_device on a whole list of %d JPEG files of %d sizes (%s, 3 scales + whitening), %d loader "
                    "threads, start to finish, after the process has seen every (size, batch) once (graph captures of this list included)"
                    % (nl, len(LIST_SHAPES), arch, workers),
            "ms_per_image": round(1e3 * times["warm_long"] / nl, 3),
            "steady_state_estimate": {"descriptors_per_s": round(1.0 / steady, 2),
                                      "what": "(t[%d images] - t[%d images]) / %d: a difference of two wall-clock timings, "
                                              "both lists long enough for one graph per size" % (nl, nm, nl - nm)},
            "ms_per_image_steady": round(1e3 * steady, 3),
            "first_occurrence_s_per_size": round(first, 3),
            "whole_list_descriptors_per_s": {"cold_%d_images" % ns: round(ns / times["cold"], 2),
                                             "warm_%d_images_eager_batches" % ns: round(ns / times["warm_short"], 2),
                                             "warm_%d_images" % nm: round(nm / times["warm_mid"], 2),
                                             "warm_%d_images" % nl: round(nl / times["warm_long"], 2)},
            "graph_capture_and_first_batches_s_per_size": round((times["warm_mid"] - nm * steady) / len(LIST_SHAPES), 3),
            "loader_only_images_per_s": round(nl / t_loader, 1),
            "trunk_conv_tflops_at_steady_state": round(flops / steady / 1e12, 2), "trunk_conv_gflop_per_image": round(flops / 1e9, 1),
            # the extraction half's own roofline: the trunk's convolutions are the flops of an image (the hand-written tail is
            # `roofline_tail`, launch-bound).  Two readings: over the whole warm list (what `value` is) and at the steady state.
            "roofline": {"kernel": "trunk convolutions: MIOpen (owner of the kernels) + mdx::conv1x1_bn_act_kernel for the Bottleneck expand 1x1 "
                                   "convolutions, mdx::bn_act_kernel epilogues; fp32",
                         "bound": "mfma", "achieved": round(flops * nl / times["warm_long"] / 1e12, 2), "peak": 157.3, "unit": "TFLOP/s",
                         "frac": round(flops * nl / times["warm_long"] / 1e12 / 157.3, 4),
                         "achieved_at_steady_state": round(flops / steady / 1e12, 2), "frac_at_steady_state": round(flops / steady / 1e12 / 157.3, 4),
                         "algorithmic_flops_per_image": flops, "traffic": None,
                         "what": "2 x multiply-adds of every convolution of the three scales (hooks on an eager pass, mean of three of the 16 sizes) "
                                 "x images / wall time of the whole warm list (loader, graph captures, tail and whitening included in the time)"},
            "miopen_find_mode": os.environ.get("MIOPEN_FIND_MODE"),
            "graph_captures": __import__("mdir_amd.graphs", fromlist=["capture_stats"]).capture_stats()}


def _cpu_loop_leg(arch, images, size, threads, budget_s):
    """One leg of ``cpu_reference_loop`` in THIS process (a CPU-only child of bench.py): ``threads`` torch threads."""
    from mdir_amd.networks import init_network
    torch.set_num_threads(threads)
    torch.manual_seed(3)
    net = init_network({"architecture": arch, "pooling": "gem", "whitening": False, "pretrained": False})
    feats = net.features.eval()
    D = net.meta["outputdim"]
    p = float(net.pool.p_value())
    rng = np.random.default_rng(2)
    q, _ = np.linalg.qr(rng.standard_normal((D, D)))
    P = torch.tensor((q * rng.uniform(0.5, 2.0, (1, D))).T.copy(), dtype=torch.float32)
    m = torch.tensor(rng.normal(0, 0.01, (D, 1)), dtype=torch.float32)
    g = torch.Generator()
    g.manual_seed(11)
    imgs = [torch.randn((1, 3) + tuple(size), generator=g) for _ in range(2)]
    scales = [1.0, 2 ** -0.5, 0.5]

    def one(x):
        v = torch.zeros(D)
        for s in scales:
            xs = x if s == 1.0 else F.interpolate(x, scale_factor=s, mode="bilinear", align_corners=False)
            f = feats(xs)
            o = F.avg_pool2d(f.clamp(min=1e-6).pow(p), (f.size(-2), f.size(-1))).pow(1.0 / p)
            o = o / (torch.norm(o, p=2, dim=1, keepdim=True) + 1e-6)
            v += o.reshape(-1).pow(p)
        v = (v / len(scales)).pow(1.0 / p)
        v /= v.norm()
        X = P.mm(v.unsqueeze(1).sub(m))
        return X.div(torch.norm(X, p=2, dim=0, keepdim=True) + 1e-6).squeeze().cpu()

    with torch.no_grad():
        one(imgs[0])                             # warm-up: oneDNN builds its primitives for the three shapes
        t0, done = time.perf_counter(), 0
        while done < images and (done == 0 or time.perf_counter() - t0 < budget_s):
            one(imgs[done % 2])
            done += 1
        dt = time.perf_counter() - t0
    return {"descriptors_per_s": round(done / dt, 4), "s_per_image": round(dt / done, 3), "images": done, "threads": threads}


def cpu_reference_loop(arch="resnet101", images=8, size=(768, 1024), budget_s=30.0, leg_timeout_s=75):
    """The REFERENCE-STYLE extraction loop on the host cores, for `descriptors_per_s.cpu_baseline`: batch 1, three scales
    (``[1, 1/sqrt(2), 1/2]``, F.interpolate bilinear), one trunk forward per scale, GeM / L2N / multi-scale aggregation /
    whitening as stock torch ops, ``.cpu()`` per image -- the loop of cirtorch/networks/imageretrievalnet.py:284-324
    (``extract_vectors`` -> ``extract_ms``) under mdir's wrapper chain (components/data/wrapper.py:104-119, 193-195), restated
    here (nothing of the reference is imported); the same random-init trunk as the GPU leg, synthetic 1024x768 inputs already
    decoded (the reference overlaps decoding in 6 loader workers).  Two legs, each a CPU-only CHILD process with a time limit:
    3 threads (what the reference asks for, mdir/stages/validate.py:10-12) and 16 threads.  (Not "all cores": on the pool's
    256-core hosts oneDNN with 256 threads took 243 s per image -- measured in round 6, the first default run of this leg --
    against 1.2 s with 3; a leg that cannot finish is ended by its limit and reported as such.)"""
    import subprocess
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    out = {}
    for name, threads in (("threads_3", min(3, cores)), ("threads_16", min(16, cores))):
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-loop-leg", arch, str(images), str(size[0]), str(size[1]), str(threads), str(budget_s)]
        try:
            proc = subprocess.run(cmd, env=env, text=True, capture_output=True, timeout=leg_timeout_s)
            lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
            out[name] = json.loads(lines[-1]) if lines and proc.returncode == 0 else {"error": (proc.stderr or "no output")[-300:], "threads": threads}
        except subprocess.TimeoutExpired:
            out[name] = {"error": "not finished within %d s" % leg_timeout_s, "threads": threads}
    good = [r for r in out.values() if "descriptors_per_s" in r]
    if not good:
        return {"value": None, "unit": "descriptors/s", "kind": "port", "legs": out, "host_cores": cores}
    best = max(good, key=lambda r: r["descriptors_per_s"])
    return {"value": best["descriptors_per_s"], "unit": "descriptors/s", "cores": best["threads"], "kind": "port",
            "sample": "%d images of %dx%d per leg (bounded), batch 1, 3 scales, torch CPU ops (oneDNN convolutions) for the trunk and the "
                      "tail, .cpu() per image: the loop of imageretrievalnet.py:284-324 under the wrapper chain, restated; %s-GeM random init; "
                      "each leg a CPU-only child process" % (images, size[1], size[0], arch),
            "value_3_threads": out["threads_3"].get("descriptors_per_s"), "legs": out, "host_cores": cores}


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-loop-leg":
        a = sys.argv[2:]
        print(json.dumps(_cpu_loop_leg(a[0], int(a[1]), (int(a[2]), int(a[3])), int(a[4]), float(a[5]))))
    elif len(sys.argv) > 1 and sys.argv[1] == "--cpu-loop":
        print(json.dumps(cpu_reference_loop(images=int(sys.argv[2]) if len(sys.argv) > 2 else 2)))
    elif len(sys.argv) > 1 and sys.argv[1] == "--list":
        print(json.dumps(measure_list(*(sys.argv[2:3] or ["resnet101"]))))
    else:
        main()
