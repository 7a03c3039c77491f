"""The exact similarity at the headline size on an index (build + multiply) and on the row-major matrix read in place."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdir_amd import ops

dev = "cuda:0"
N, Q, D = 1004993, 70, 2048
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.randn((N, D), generator=g, device=dev); x /= x.norm(dim=1, keepdim=True)
q = torch.randn((Q, D), generator=g, device=dev); q /= q.norm(dim=1, keepdim=True)
out = torch.empty((Q, N), device=dev)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


t_rm = timed(lambda: ops.scores_rowmajor(x, q, "ND", out=out))
ref = out.clone()
ix = ops.DescriptorIndex(x, "ND")
t_ix = timed(lambda: ix.scores(q, "ND", out=out))
assert torch.equal(ref, out)
del ix


def build_and_multiply():
    i = ops.DescriptorIndex(x, "ND")
    i.scores(q, "ND", out=out)
    i.close()


t_both = timed(build_and_multiply, reps=5)
print("row-major in place %.3f ms | resident index %.3f ms | index build + multiply %.3f ms (bit-identical scores)" % (t_rm, t_ix, t_both))
