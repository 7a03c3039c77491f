"""How much CPU does the process burn while it replays extraction graphs (no loader)?  process_time / wall = busy cores."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from mdir_amd import ops
from mdir_amd.graphs import ShapeGraphs
from mdir_amd.networks import init_network, extract_ms
dev = torch.device("cuda:0")
torch.manual_seed(3)
net = init_network({"architecture": "resnet101", "pooling": "gem", "whitening": False, "pretrained": False}).to(dev).eval()
ms = [1, 2 ** -0.5, 0.5]
for streams in ("1", "0"):
    os.environ["MDIR_AMD_SCALE_STREAMS"] = streams
    describe = ShapeGraphs(lambda x: extract_ms(net, x, ms, 3.0), warmup=1)
    x = torch.randn(4, 3, 768, 1024, device=dev)
    with torch.no_grad():
        for _ in range(3): describe(x)
        torch.cuda.synchronize()
        w0, c0 = time.perf_counter(), time.process_time()
        for _ in range(12): describe(x)
        w1, c1 = time.perf_counter(), time.process_time()      # host side only: launches enqueued
        torch.cuda.synchronize()
        w2, c2 = time.perf_counter(), time.process_time()
    print("scale streams %s: graphs %d; enqueue %.1f ms per replay (cpu %.1f ms); until done %.1f ms per replay (cpu %.1f ms): %.1f cores busy; threads %d"
          % (streams, len(describe.graphs), 1e3 * (w1 - w0) / 12, 1e3 * (c1 - c0) / 12, 1e3 * (w2 - w0) / 12, 1e3 * (c2 - c0) / 12, (c2 - c0) / (w2 - w0),
             len(os.listdir("/proc/self/task"))))
