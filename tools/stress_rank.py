"""Soak test of the ranking kernels: random shapes and score distributions (continuous, heavy ties,
few distinct values, NaN / inf sprinkled in), rank_full / topk / rank_of against the C oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mdir_amd import ops
from oracle import chain as OC

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad, t0 = 0, time.time()
for it in range(iters):
    n = int(rng.integers(1, 30000)) if it % 4 else int(rng.integers(200000, 400000))
    nq = int(rng.integers(1, 24)) if n > 100000 else int(rng.integers(1, 90))
    kind = it % 5
    s = rng.standard_normal((nq, n)).astype(np.float32) * 0.02
    if kind == 1:
        s = np.round(s * 200) / 200                      # heavy ties
    elif kind == 2:
        s = rng.integers(0, 3, (nq, n)).astype(np.float32)      # three distinct values
    elif kind == 3:
        s[rng.random((nq, n)) < 0.01] = np.nan
        s[rng.random((nq, n)) < 0.01] = np.inf
        s[rng.random((nq, n)) < 0.01] = -0.0
    elif kind == 4:
        s[:] = 0.5                                        # all equal
    sd = torch.from_numpy(s).cuda()
    want = OC.rank_full(s)
    got = ops.rank_full(sd).cpu().numpy()
    if not np.array_equal(got, want):
        bad += 1
        print("rank_full MISMATCH it=%d n=%d nq=%d kind=%d" % (it, n, nq, kind))
    # the same scores as column blocks of random widths (the peer blocks of the multi-GPU exchange): tiles that end in the next block,
    # blocks narrower than a tile, empty blocks -- the ranking read in place equals the dense one
    cuts = np.sort(rng.choice(n + 1, size=int(min(n, rng.integers(1, 31))), replace=True))
    edges = [0] + [int(c) for c in cuts] + [n]
    blocks = [sd[:, a:b].contiguous() for a, b in zip(edges[:-1], edges[1:])]
    if len(blocks) <= 32 and not np.array_equal(ops.rank_full_segments(blocks).cpu().numpy(), want):
        bad += 1
        print("rank_full_segments MISMATCH it=%d n=%d nq=%d kind=%d widths=%s" % (it, n, nq, kind, [b.shape[1] for b in blocks]))
    k = int(min(n, rng.integers(1, 300))) if it % 2 else int(max(1, min(n, n // 256)))    # every other one: sampled-threshold regime
    ids, vals = ops.topk(sd, k)
    if not np.array_equal(ids.cpu().numpy(), want[:, :k]):
        bad += 1
        print("topk MISMATCH it=%d n=%d nq=%d k=%d kind=%d" % (it, n, nq, k, kind))
    lists = [rng.choice(n, size=int(min(n, rng.integers(0, 40))), replace=False) for _ in range(nq)]
    pos, _, off = ops.rank_of(sd, lists)
    pos = pos.cpu().numpy()
    inv = np.empty_like(want)
    np.put_along_axis(inv, want, np.broadcast_to(np.arange(n), want.shape), axis=1)
    for q in range(nq):
        if not np.array_equal(pos[off[q]:off[q + 1]], inv[q][lists[q]]):
            bad += 1
            print("rank_of MISMATCH it=%d n=%d nq=%d q=%d kind=%d" % (it, n, nq, q, kind))
            break
print("rank stress done: %d iterations, %d mismatches, %.1f s" % (iters, bad, time.time() - t0))
