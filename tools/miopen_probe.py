"""Which MIOpen kernels run the ResNet101 trunk on a NEW input shape (first call, later calls)?
   python tools/miopen_probe.py [H W] ; env: MIOPEN_* as given; BENCHMARK=1 -> cudnn.benchmark"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdir_amd.networks import init_network
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (683, 1024)
torch.backends.cudnn.benchmark = os.environ.get("BENCHMARK") == "1"
net = init_network({"architecture": "resnet101", "pooling": "gem", "whitening": False, "pretrained": False}).cuda().eval()
x = torch.randn(1, 3, h, w, device="cuda")
with torch.no_grad():
    for i in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        net.features(x)
        torch.cuda.synchronize(); print("call %d: %.1f ms" % (i, 1e3 * (time.perf_counter() - t0)), flush=True)
