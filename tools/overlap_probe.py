"""Does running step k's ranking concurrently with step k+1's similarity pay? (two HIP streams)"""
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
ix = ops.DescriptorIndex(rows, "ND")
sc = [torch.empty((nq, n), device=dev) for _ in range(2)]
rk = [torch.empty((nq, n), dtype=torch.int64, device=dev) for _ in range(2)]
ws = [torch.empty(ops.rank_workspace_bytes(n, nq), dtype=torch.uint8, device=dev) for _ in range(2)]
K = 20
def serial():
    for i in range(K):
        ix.scores(q, "DN", out=sc[0]); ops.rank_full(sc[0], out=rk[0], workspace=ws[0])
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
def piped():
    done_rank = [None, None]
    for i in range(K):
        b = i & 1
        with torch.cuda.stream(sA):
            if done_rank[b] is not None: sA.wait_event(done_rank[b])
            ix.scores(q, "DN", out=sc[b]); ev = torch.cuda.Event(); ev.record(sA)
        with torch.cuda.stream(sB):
            sB.wait_event(ev); ops.rank_full(sc[b], out=rk[b], workspace=ws[b]); f = torch.cuda.Event(); f.record(sB); done_rank[b] = f
    torch.cuda.current_stream().wait_stream(sA); torch.cuda.current_stream().wait_stream(sB)
import time
for name, fn in (("serial", serial), ("piped", piped), ("serial", serial), ("piped", piped)):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); t = time.perf_counter() - t0
    print("%-7s %.3f ms/step  %.0f q/s" % (name, 1e3 * t / K, nq * K / t))
a = rk[0][:, :3].cpu(); print(a[0].tolist())
