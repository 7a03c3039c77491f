"""Exactness + timing of mdx_topk (radix-select path) against the oracle's full ranking."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mdir_amd import ops
from oracle import chain as OC
rng = np.random.default_rng(0)
for n, nq, k, kind in ((200000, 5, 100, "gauss"), (70000, 3, 1, "gauss"), (65536, 4, 1000, "ties"), (100000, 2, 50, "allequal"),
                       (300000, 3, 257, "concentrated"), (50000, 2, 10, "nan"), (20000, 6, 100, "gauss"), (1000, 3, 10, "gauss")):
    sc = (rng.standard_normal((nq, n)) * 0.022).astype(np.float32)
    if kind == "ties": sc = np.round(sc * 50) / 50
    if kind == "allequal": sc[:] = 0.125
    if kind == "concentrated": sc = (0.3 + rng.standard_normal((nq, n)) * 1e-4).astype(np.float32)
    if kind == "nan": sc[:, ::3] = np.nan
    want = OC.rank_full(sc)[:, :k]
    ids, vals = ops.topk(torch.from_numpy(sc).cuda(), k, id_offset=7)
    ids, vals = ids.cpu().numpy(), vals.cpu().numpy()
    assert (ids == want + 7).all(), (n, nq, k, kind, np.argwhere(ids != want + 7)[:5])
    assert np.array_equal(vals, np.take_along_axis(sc, want, axis=1), equal_nan=True), (n, nq, k, kind)
print("topk exact on all cases")
n, nq = 1004993, 70
sc = (torch.randn((nq, n), device="cuda") * 0.022)
ws = torch.empty(ops.rank_workspace_bytes(n, nq), dtype=torch.uint8, device="cuda")
for k in (10, 100, 1000, 10000):
    for _ in range(2): ops.topk(sc, k, workspace=ws)
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): ops.topk(sc, k, workspace=ws)
    b.record(); torch.cuda.synchronize()
    print("top-%d of 1M x 70: %.3f ms" % (k, a.elapsed_time(b) / 10))
