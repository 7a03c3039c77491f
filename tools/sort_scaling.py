"""rank_full time per million elements as a function of the number of queries sorted at once
(does a working set inside the 256 MiB Infinity Cache sort faster per element?)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdir_amd import ops

n = 1004993
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
sc = torch.randn((70, n), generator=g, device=dev) * 0.022
rk = torch.empty((70, n), dtype=torch.int64, device=dev)
trash = torch.empty(1 << 29, dtype=torch.uint8, device=dev)
for nq in (5, 7, 10, 14, 18, 24, 35, 70):
    ws = torch.empty(ops.rank_workspace_bytes(n, nq), dtype=torch.uint8, device=dev)
    def run():
        for b in range(0, 70, nq):
            e = min(70, b + nq)
            if e - b == nq:
                ops.rank_full(sc[b:e], out=rk[b:e], workspace=ws)
    for _ in range(2): run()
    torch.cuda.synchronize()
    a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    a.record()
    for _ in range(reps): run()
    c.record(); torch.cuda.synchronize()
    done = (70 // nq) * nq
    ms = a.elapsed_time(c) / reps
    print("batch %2d queries: %d queries in %.3f ms -> %.2f us per M elements (x70 = %.3f ms)" % (nq, done, ms, ms * 1e3 / (done * n / 1e6), ms / done * 70), flush=True)
