"""Timing of the fp16-shard similarity (1M x 2048, Q=70) next to the fp32 one."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdir_amd import ops
n, nq, d = 1004993, 70, 2048
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
rows = torch.empty((n, d), device=dev)
for s in range(0, n, 65536):
    e = min(n, s + 65536); blk = torch.randn((e - s, d), generator=g, device=dev); rows[s:e] = blk / blk.norm(dim=1, keepdim=True)
q = (rows[torch.randperm(n, device=dev)[:nq]] + 0.05 * torch.randn((nq, d), generator=g, device=dev)); q /= q.norm(dim=1, keepdim=True); q = q.t().contiguous()
for st in ("f32", "f16"):
    ix = ops.DescriptorIndex(rows, "ND", storage=st)
    sc = torch.empty((nq, n), device=dev)
    for _ in range(3): ix.scores(q, "DN", out=sc)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): ix.scores(q, "DN", out=sc)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    nbytes = ix.device_bytes + 4 * nq * n
    print("%s shard: %.3f ms  %.1f TFLOP/s  %.2f TB/s of shard+scores bytes  (shard %.2f GB)" % (st, ms, 2 * nq * n * d / ms / 1e9, nbytes / ms / 1e9, ix.device_bytes / 1e9))
    ix.close()
