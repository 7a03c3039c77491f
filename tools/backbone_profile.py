"""Kernel-time breakdown of the ResNet101 trunk at the 3 pyramid scales (run under rocprofv3 --stats)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from mdir_amd.networks import init_network

dev = torch.device("cuda:0")
torch.manual_seed(3)
net = init_network({"architecture": "resnet101", "pooling": "gem", "whitening": False, "pretrained": False}).to(dev).eval()
x = torch.randn(1, 3, 768, 1024, device=dev)
pyr = [x] + [F.interpolate(x, scale_factor=s, mode="bilinear", align_corners=False) for s in (2 ** -0.5, 0.5)]
import time
with torch.no_grad():
    for _ in range(3):
        for p in pyr:
            net.features(p)
    torch.cuda.synchronize()
    time.sleep(1.0)             # the summariser keeps only what follows the last >0.5 s gap
    for _ in range(5):
        for p in pyr:
            net.features(p)
torch.cuda.synchronize()
