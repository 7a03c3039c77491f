"""Runs mdx_conv1x1_bn_act on the two layer3 shapes (for rocprofv3 --pmc / --stats)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdir_amd import ops
dev = "cuda:0"
for cin, cout, res in ((1024, 256, False), (256, 1024, True)):
    x = torch.randn(4, cin, 48, 64, device=dev)
    wt = ops.conv1x1_transpose_weights(torch.randn(cout, cin, device=dev) / cin ** 0.5)
    mean, var = torch.randn(cout, device=dev) * 0.1, torch.rand(cout, device=dev) + 0.5
    idt = torch.randn(4, cout, 48, 64, device=dev) if res else None
    for _ in range(30):
        ops.conv1x1_bn_act(x, wt, mean, var, None, None, 1e-5, idt, True)
torch.cuda.synchronize()
