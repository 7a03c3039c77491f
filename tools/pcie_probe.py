#!/usr/bin/env python3
"""What the boundary would cost if it were handed HOST buffers (never the bench's `value`): the 70 query
vectors host -> device, the step, and the int64 ranking [70, N] (563 MB) device -> pinned host."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from mdir_amd import ops

n = bench.N_ROXFORD + bench.N_DISTRACTORS
dev = torch.device("cuda", 0)
rows = bench.gen_rows(0, n, dev)
qvecs, _ = bench.gen_queries(n, dev)
ix = ops.DescriptorIndex(rows, "ND")
del rows
q_host = qvecs.cpu().pin_memory()
sc = torch.empty((bench.NQ, n), dtype=torch.float32, device=dev)
rk = torch.empty((bench.NQ, n), dtype=torch.int64, device=dev)
ws = torch.empty(ops.rank_workspace_bytes(n, bench.NQ), dtype=torch.uint8, device=dev)
rk_host = torch.empty((bench.NQ, n), dtype=torch.int64).pin_memory()


def step(host):
    q = q_host.to(dev, non_blocking=True) if host else qvecs
    ix.scores(q, "DN", out=sc)
    ops.rank_full(sc, out=rk, workspace=ws)
    if host:
        rk_host.copy_(rk, non_blocking=True)


for host in (False, True):
    for _ in range(3):
        step(host)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step(host)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 10
    print("%s: %.3f ms per 70-query batch = %.0f queries/s" % ("host buffers (H2D queries + D2H int64 ranking)" if host else "device-resident", 1e3 * t, bench.NQ / t))
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5):
    rk_host.copy_(rk, non_blocking=True)
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / 5
print("D2H of the ranking alone: %.2f ms = %.1f GB/s (563 MB, pinned)" % (ms, rk.numel() * 8 / ms / 1e6))
