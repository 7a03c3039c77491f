"""Headline kernels in two arrangements, for tools/overlap_trace.sh: `serial` (one stream) and `piped` (the ranking of batch k
on a second stream while the similarity of batch k+1 runs; two score / rank buffers) -- bench.py's pipelined_two_streams leg.
Prints ms per step of each; rankings of both arrangements must be identical.

    python tools/overlap_run.py [steps]          (MDX_SCORES_NSTAGE=2: the probe-only two-stage similarity kernel)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mdir_amd import ops

n, nq, d = 1004993, 70, 2048
K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(0)
rows = torch.empty((n, d), device=dev)
for s in range(0, n, 65536):
    e = min(n, s + 65536)
    blk = torch.randn((e - s, d), generator=g, device=dev)
    rows[s:e] = blk / blk.norm(dim=1, keepdim=True)
q = rows[torch.randperm(n, device=dev)[:nq]].t().contiguous()
ix = ops.DescriptorIndex(rows, "ND")
del rows
sc = [torch.empty((nq, n), device=dev) for _ in range(2)]
rk = [torch.empty((nq, n), dtype=torch.int64, device=dev) for _ in range(3)]
ws = [torch.empty(ops.rank_workspace_bytes(n, nq), dtype=torch.uint8, device=dev) for _ in range(2)]


def serial(steps):
    for _ in range(steps):
        ix.scores(q, "DN", out=sc[0])
        ops.rank_full(sc[0], out=rk[2], workspace=ws[0])


s_rank = torch.cuda.Stream(device=dev)


def piped(steps):
    cur = torch.cuda.current_stream(dev)
    done = [None, None]
    for k in range(steps):
        b = k & 1
        if done[b] is not None:
            cur.wait_event(done[b])
        ix.scores(q, "DN", out=sc[b])
        ready = torch.cuda.Event()
        ready.record(cur)
        s_rank.wait_event(ready)
        with torch.cuda.stream(s_rank):
            ops.rank_full(sc[b], out=rk[b], workspace=ws[b])
            done[b] = torch.cuda.Event()
            done[b].record(s_rank)
    cur.wait_stream(s_rank)


for name, fn in (("serial", serial), ("piped", piped), ("serial", serial), ("piped", piped)):
    fn(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(K)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print("%-7s %.4f ms/step  (MDX_SCORES_NSTAGE=%s)" % (name, 1e3 * t / K, os.environ.get("MDX_SCORES_NSTAGE", "3")), flush=True)
assert bool((rk[0] == rk[2]).all()) and bool((rk[1] == rk[2]).all())
print("rankings identical")
