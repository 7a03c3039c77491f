"""Timing of the sort-free evaluation path (mdx_rank_of = gather + rank_count) at 1 M x 70."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mdir_amd import ops
n, nq = 1004993, 70
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
sc = torch.randn((nq, n), generator=g, device=dev) * 0.022
rng = np.random.default_rng(1)
top = torch.topk(sc, 200, dim=1).indices.cpu().numpy()          # labelled rows that rank near the top (the retrieval case: few rows precede them)
for per, kind in ((20, "top"), (200, "top"), (4, "random"), (20, "random"), (200, "random")):
    lists = [np.sort(top[q][rng.choice(200, per, replace=False)]) if kind == "top" else rng.choice(4993, per, replace=False) for q in range(nq)]
    pos, _, off = ops.rank_of(sc, lists)
    torch.cuda.synchronize()
    ids_t, off_t, _ = ops._csr(lists, dev)
    ref = ops.gather_scores(sc, ids_t, off_t)
    cnt = torch.zeros(ids_t.numel(), dtype=torch.int64, device=dev)
    ops.rank_count_(cnt, sc, 0, ref, ids_t, off_t)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): ops.rank_count_(cnt, sc, 0, ref, ids_t, off_t)
    b.record(); torch.cuda.synchronize()
    # check one query against a direct count
    q = 3; s = sc[q]; ids = torch.as_tensor(lists[q], device=dev)
    want = [(int((s > s[i]).sum()) + int(((s == s[i]) & (torch.arange(n, device=dev) < i)).sum())) for i in ids[:5]]
    print("refs/query %3d (%s): rank_count kernel %.3f ms; check %s" % (per, kind, a.elapsed_time(b) / 10, want == [int(x) for x in pos[off[q]:off[q] + len(want)]]), flush=True)
