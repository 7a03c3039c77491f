"""Is the exact similarity kernel bound by its schedule or by the chip's power envelope?  The same launch (N = 1 004 993,
Q = 70, D = 2048 fp32) on real-magnitude gaussian rows and on all-zero operands: the instruction stream is identical, only the
toggling in the matrix pipe (and with it power and the sustained clock) differs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdir_amd import ops

dev = "cuda:0"
N, Q, D = 1004993, 70, 2048
g = torch.Generator(device=dev); g.manual_seed(1)


def timed(ix, q, reps=20):
    out = torch.empty((Q, N), dtype=torch.float32, device=dev)
    for _ in range(3):
        ix.scores(q, "ND", out=out)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        ix.scores(q, "ND", out=out)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for name in ("gaussian unit rows", "all zero", "gaussian unit rows", "all zero"):
    if name.startswith("gauss"):
        x = torch.randn((N, D), generator=g, device=dev); x /= x.norm(dim=1, keepdim=True)
        q = torch.randn((Q, D), generator=g, device=dev); q /= q.norm(dim=1, keepdim=True)
    else:
        x = torch.zeros((N, D), device=dev); q = torch.zeros((Q, D), device=dev)
    ix = ops.DescriptorIndex(x, "ND")
    del x
    print("%-20s %.3f ms" % (name, timed(ix, q)), flush=True)
    del ix
    torch.cuda.empty_cache()
