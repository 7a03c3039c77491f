"""Is a rank's step at G = 8 bound by the GPU or by the Python host path?  One rank's launches per step (the similarity of its
125 625-row shard in two equal chunks, the segment sort of its 9 queries over 16 peer blocks; no collective: one GPU) issued
back to back: wall clock per step against the HIP-event time of the same launches, and the host time of the enqueue alone."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mdir_amd import ops
from mdir_amd.sharded import shard_bounds

N, NQ, D, G = 1004993, 70, 2048, 8
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(0)
lo, hi = shard_bounds(N, G, 0)
n_local = hi - lo
rows = torch.randn((n_local, D), generator=g, device=dev)
rows /= rows.norm(dim=1, keepdim=True)
q = rows[:NQ].t().contiguous()
for chunks in (1, 2):
    cut = [(0, n_local)] if chunks == 1 else [(0, n_local // 2), (n_local // 2, n_local)]
    ixs = [ops.DescriptorIndex(rows[a:b].contiguous(), "ND") for a, b in cut]
    full = torch.randn((9, N), generator=g, device=dev) * 0.022
    widths = []
    for r in range(G):
        a, b = shard_bounds(N, G, r)
        widths += [b - a] if chunks == 1 else [(b - a) // 2, (b - a) - (b - a) // 2]
    blocks, o = [], 0
    for w in widths:
        blocks.append(full[:, o:o + w].contiguous())
        o += w

    def step():
        for ix in ixs:
            ix.scores(q, "DN")
        return ops.rank_full_segments(blocks)

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    K = 200
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record()
    for _ in range(K):
        step()
    b.record()
    t_host = (time.perf_counter() - t0) / K * 1e3          # the enqueue alone (nothing waits for the GPU)
    torch.cuda.synchronize()
    t_wall = (time.perf_counter() - t0) / K * 1e3
    print("chunks %d: GPU (HIP events over %d steps) %.3f ms/step, wall %.3f ms/step, host enqueue %.3f ms/step -> %s"
          % (chunks, K, a.elapsed_time(b) / K, t_wall, t_host, "host-bound" if t_host > 0.9 * t_wall else "GPU-bound"), flush=True)
    for ix in ixs:
        ix.close()
