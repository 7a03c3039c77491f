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

        def tail_mdx():
            per = [ops.pool_l2n(f, "gem", p, 1e-6, 1e-6).reshape(-1) for f in feats]
            v = ops.ms_aggregate(per, p)
            return chain.wrappers[0].whiten_rows(v.reshape(1, -1))

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

        a, b = tail_mdx().reshape(-1), tail_torch().reshape(-1)
        err = float((a - b).abs().max())
        t_tail, t_tail_torch = timeit(tail_mdx), timeit(tail_torch)
    return {"metric": "descriptors/sec, %s-GeM, 3 scales of 1024x768 + whitening, 1 GPU" % args.arch,
            "value": round(args.images / total, 2), "unit": "descriptors/s",
            "ms_per_image": round(1e3 * total / args.images, 3),
            "backbone_ms_per_image": round(t_backbone, 3), "backbone_ms_per_image_eager_launches": round(t_backbone_eager, 3),
            "tail_ms_per_image_mdx": round(t_tail, 4), "tail_ms_per_image_torch_ops": round(t_tail_torch, 4),
            "tail_max_abs_diff_vs_torch_ops": err, "dtype": "f32", "data": "synthetic",
            "hipgraph_replays": getattr(describe, "replays", 0)}


if __name__ == "__main__":
    main()
