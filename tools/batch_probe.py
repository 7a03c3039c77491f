"""Would batching equal-sized images through the trunk pay?  Graph-replayed ResNet101 trunk, 3 scales
on parallel streams, batch 1 / 2 / 4 of 1024x768: ms per image."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from mdir_amd.graphs import ShapeGraphs, parallel_map
from mdir_amd.networks import init_network

dev = torch.device("cuda:0")
torch.manual_seed(3)
arch = sys.argv[1] if len(sys.argv) > 1 else "resnet101"
net = init_network({"architecture": arch, "pooling": "gem", "whitening": False, "pretrained": False}).to(dev).eval()
scales = [1, 2 ** -0.5, 0.5]


def trunk(x):
    pyr = [x if s == 1 else F.interpolate(x, scale_factor=s, mode="bilinear", align_corners=False) for s in scales]
    return parallel_map(lambda t: net(t), pyr)


with torch.no_grad():
    for b in (1, 2, 4):
        x = torch.randn(b, 3, 768, 1024, device=dev)
        g = ShapeGraphs(trunk, warmup=1)
        g(x); g(x); g(x)
        torch.cuda.synchronize()
        a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            g(x)
        c.record(); torch.cuda.synchronize()
        print("%s batch %d: %.3f ms per image (replays %d)" % (arch, b, a.elapsed_time(c) / 10 / b, g.replays), flush=True)

# two graph instances of the batch-4 function replayed alternately on two streams
with torch.no_grad():
    x = torch.randn(4, 3, 768, 1024, device=dev)
    gs = [ShapeGraphs(trunk, warmup=1) for _ in range(2)]
    for g in gs:
        g(x); g(x); g(x)
    st = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(20):
        st[i % 2].wait_stream(torch.cuda.current_stream()) if i < 2 else None
        with torch.cuda.stream(st[i % 2]):
            gs[i % 2](x)
    for s_ in st:
        torch.cuda.current_stream().wait_stream(s_)
    c.record(); torch.cuda.synchronize()
    print("%s batch 4, two graphs in flight: %.3f ms per image" % (arch, a.elapsed_time(c) / 20 / 4), flush=True)
