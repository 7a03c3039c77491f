"""Latency of the ranking step at the reference's small configurations (no distractors)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdir_amd import ops


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(0)
    for name, n, nq, d in (("roxford5k", 4993, 70, 2048), ("rparis6k", 6322, 70, 2048), ("247tokyo1k", 1125, 1125, 512),
                           ("mining", 20000, 2000, 2048), ("oxford+100k", 104993, 70, 2048)):
        rows = torch.randn((n, d), generator=g, device=dev); rows /= rows.norm(dim=1, keepdim=True)
        q = rows[torch.randperm(n, device=dev)[:nq]].contiguous()
        ix = ops.DescriptorIndex(rows, "ND")
        sc = torch.empty((nq, n), dtype=torch.float32, device=dev)
        rk = torch.empty((nq, n), dtype=torch.int64, device=dev)
        ws = torch.empty(ops.rank_workspace_bytes(n, nq), dtype=torch.uint8, device=dev)
        t_sc = timed(lambda: ix.scores(q, "ND", out=sc))
        t_rk = timed(lambda: ops.rank_full(sc, out=rk, workspace=ws))
        t_tk = timed(lambda: ops.topk(sc, min(100, n)))
        print("%-12s n=%6d nq=%4d d=%4d  scores %7.1f us  rank_full %7.1f us  top100 %7.1f us" % (name, n, nq, d, t_sc, t_rk, t_tk), flush=True)
        ix.close()


main()
