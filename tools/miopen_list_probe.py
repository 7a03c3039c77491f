"""Steady-state time of the ResNet101 trunk (the product's `net.features`, fused bn/1x1 kernels included) for every input
shape of the 16-size extraction list x 3 scales at the batch extraction uses (8), under the MIOpen settings of THIS process
(VERDICT round 4, item 7).  One JSON line per (size, scale) + a summary line.

    MIOPEN_FIND_MODE=2 python tools/miopen_list_probe.py fast            # the package default (immediate mode, FAST find)
    MIOPEN_FIND_MODE=1 python tools/miopen_list_probe.py find benchmark   # cudnn.benchmark: MIOpen measures every solver once per shape
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench_extract import LIST_SHAPES
from mdir_amd.networks import init_network

tag = sys.argv[1] if len(sys.argv) > 1 else "fast"
torch.backends.cudnn.benchmark = "benchmark" in sys.argv[2:]
batch = 8
net = init_network({"architecture": "resnet101", "pooling": "gem", "whitening": False, "pretrained": False}).cuda().eval()
g = torch.Generator(device="cuda")
g.manual_seed(0)
total_first = total_steady = 0.0
with torch.no_grad():
    for (w, h) in LIST_SHAPES:
        for s in (1.0, 2 ** -0.5, 0.5):
            hs, ws = int(h * s), int(w * s)                 # F.interpolate(scale_factor): floor(in * s)
            x = torch.randn((batch, 3, hs, ws), generator=g, device="cuda")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            net.features(x)
            torch.cuda.synchronize()
            first = time.perf_counter() - t0
            net.features(x)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(4):
                net.features(x)
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 4 / batch
            total_first += first
            total_steady += ms
            print(json.dumps({"tag": tag, "w": w, "h": h, "scale": round(s, 4), "in": [hs, ws], "first_call_s": round(first, 3),
                              "steady_ms_per_image": round(ms, 4)}), flush=True)
print(json.dumps({"tag": tag, "summary": True, "find_mode": os.environ.get("MIOPEN_FIND_MODE"), "benchmark": torch.backends.cudnn.benchmark,
                  "sum_first_calls_s": round(total_first, 2), "sum_steady_ms_per_image_over_48_shapes": round(total_steady, 3)}), flush=True)
