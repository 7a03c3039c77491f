"""Steps back to back on ONE stream against the ranking of step k overlapped with the similarity of step k+1 (two streams,
two score buffers).  Same kernels, same work per step."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdir_amd import ops

n, nq, d = 1004993, 70, 2048
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
rows = torch.empty((n, d), device=dev)
for s in range(0, n, 65536):
    e = min(n, s + 65536); blk = torch.randn((e - s, d), generator=g, device=dev); rows[s:e] = blk / blk.norm(dim=1, keepdim=True)
q = rows[torch.randperm(n, device=dev)[:nq]].t().contiguous()
ix = ops.DescriptorIndex(rows, "ND"); del rows
sc = [torch.empty((nq, n), device=dev) for _ in range(2)]
rk = [torch.empty((nq, n), dtype=torch.int64, device=dev) for _ in range(2)]
ws = [torch.empty(ops.rank_workspace_bytes(n, nq), dtype=torch.uint8, device=dev) for _ in range(2)]
K = 20

def serial():
    for k in range(K):
        ix.scores(q, "DN", out=sc[0]); ops.rank_full(sc[0], out=rk[0], workspace=ws[0])

s_rank = torch.cuda.Stream(device=dev)
def piped():
    cur = torch.cuda.current_stream(dev)
    done = [None, None]
    for k in range(K):
        b = k & 1
        if done[b] is not None:
            cur.wait_event(done[b])                    # the ranking that read sc[b] two steps ago has finished
        ix.scores(q, "DN", out=sc[b])
        ready = torch.cuda.Event(); ready.record(cur)
        s_rank.wait_event(ready)
        with torch.cuda.stream(s_rank):
            ops.rank_full(sc[b], out=rk[b], workspace=ws[b])
            done[b] = torch.cuda.Event(); done[b].record(s_rank)
    cur.wait_stream(s_rank)

for name, fn in (("one stream", serial), ("two streams", piped), ("one stream", serial), ("two streams", piped)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print("%-12s %.3f ms per step  %.0f queries/s" % (name, 1e3 * dt, nq / dt), flush=True)
ref = ops.rank_full(ix.scores(q, "DN"))
print("results equal:", bool((rk[0] == ref).all() and (rk[1] == ref).all()))
