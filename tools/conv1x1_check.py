"""mdx_conv1x1_bn_act against F.conv2d + the module arithmetic in float64, over edge shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from mdir_amd import ops
dev = "cuda:0"
torch.manual_seed(1)
bad = 0
for (n, cin, cout, h, w, res, relu, bn) in [(1, 64, 64, 7, 9, False, True, True), (2, 256, 64, 33, 31, False, True, True),
                                            (4, 64, 256, 16, 16, True, True, True), (1, 1024, 256, 48, 64, False, True, True),
                                            (3, 256, 1024, 23, 17, True, False, True), (1, 512, 2048, 24, 32, True, True, True),
                                            (2, 16, 128, 5, 3, False, False, False), (1, 128, 512, 128, 96, True, True, True),
                                            (4, 256, 64, 256, 192, False, True, True)]:
    x = torch.randn(n, cin, h, w, device=dev)
    wt = torch.randn(cout, cin, 1, 1, device=dev) / cin ** 0.5
    mean, var = (torch.randn(cout, device=dev) * 0.1, torch.rand(cout, device=dev) + 0.5) if bn else (None, None)
    gamma, beta = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1
    idt = torch.randn(n, cout, h, w, device=dev) if res else None
    got = ops.conv1x1_bn_act(x, ops.conv1x1_transpose_weights(wt), mean, var, gamma, beta, 1e-5, idt, relu)
    y = F.conv2d(x.double(), wt.double())
    if bn:
        y = (y - mean.double().view(1, -1, 1, 1)) * (gamma.double() / torch.sqrt(var.double() + 1e-5)).view(1, -1, 1, 1) + beta.double().view(1, -1, 1, 1)
    else:
        y = y * gamma.double().view(1, -1, 1, 1) + beta.double().view(1, -1, 1, 1)
    if res:
        y = y + idt.double()
    if relu:
        y = y.clamp_min(0)
    err = float((got.double() - y).abs().max() / y.abs().max())
    print((n, cin, cout, h, w, res, relu, bn), "rel err %.2e" % err)
    bad += err > 2e-6
print("CONV-CHECK", "FAIL" if bad else "OK")
