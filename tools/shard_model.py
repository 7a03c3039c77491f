"""Per-rank compute of the sharded step at G = 1,2,4,8, measured on ONE GPU.

Rank 0's share of the work at each G: scores of 70 queries against its N/G-row shard, the
re-blocking copy after the all-to-all, and the full sort of its ceil(70/G) queries over all N
rows.  The collective itself cannot be measured here (one GPU per box); its cost is modelled as
bytes / (peer links x 153 GB/s x 0.7).  Prints a table; not a substitute for SCALE_rNN.json.
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdir_amd import ops
from mdir_amd.sharded import shard_bounds


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    n, nq, d = 1004993, 70, 2048
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(0)
    rows = torch.empty((n, d), dtype=torch.float32, device=dev)
    for s in range(0, n, 65536):
        e = min(n, s + 65536)
        rows[s:e] = torch.randn((e - s, d), generator=g, device=dev)
        rows[s:e] /= rows[s:e].norm(dim=1, keepdim=True)
    q = rows[torch.randperm(n, device=dev)[:nq]].contiguous()
    full = ops.DescriptorIndex(rows, "ND").scores(q, "ND")
    base = None
    print("%2s %10s %9s %9s %9s %9s %9s %8s" % ("G", "rows/rank", "scores", "reblock", "sort", "a2a(model)", "total", "speedup"))
    for G in ([int(a) for a in sys.argv[1:]] or (1, 2, 4, 8)):
        lo, hi = shard_bounds(n, G, 0)
        qlo, qhi = shard_bounds(nq, G, 0)
        ix = ops.DescriptorIndex(rows[lo:hi], "ND")
        sc = torch.empty((nq, hi - lo), dtype=torch.float32, device=dev)
        t_sc = timed(lambda: ix.scores(q, "ND", out=sc))
        mine = full[qlo:qhi].contiguous()
        blocks = [mine[:, shard_bounds(n, G, r)[0]:shard_bounds(n, G, r)[1]].contiguous() for r in range(G)]
        t_cat = timed(lambda: torch.cat(blocks, dim=1)) if G > 1 else 0.0
        rk = torch.empty((qhi - qlo, n), dtype=torch.int64, device=dev)
        ws = torch.empty(ops.rank_workspace_bytes(n, qhi - qlo), dtype=torch.uint8, device=dev)
        t_sort = timed(lambda: ops.rank_full(mine, out=rk, workspace=ws))
        sent = 4.0 * nq * (hi - lo) * (G - 1) / G
        t_a2a = 0.0 if G == 1 else 0.03 + sent / ((G - 1) * 153e9 * 0.7) * 1e3
        tot = t_sc + t_cat + t_sort + t_a2a
        base = base or tot
        print("%2d %10d %9.3f %9.3f %9.3f %9.3f %9.3f %8.2f" % (G, hi - lo, t_sc, t_cat, t_sort, t_a2a, tot, base / tot), flush=True)
        ix.close()


main()
