"""Exactness of rank_full against the C oracle over distributions and column segments (written for the parked MSD route, tools/attic/rank_msd.h: MDX_SORT_MSD is only honoured by a library built with it)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mdir_amd import ops
from oracle import chain as OC
rng = np.random.default_rng(0)
bad = 0
for n, nq, kind in ((70000, 5, "gauss"), (65536, 3, "ties"), (300001, 2, "gauss"), (262144, 4, "three"), (131073, 3, "equal"),
                    (500000, 2, "nan"), (1004993, 3, "gauss"), (200000, 6, "ascending"), (200000, 2, "periodic")):
    s = (rng.standard_normal((nq, n)) * 0.02).astype(np.float32)
    if kind == "ties": s = np.round(s * 200) / 200
    if kind == "three": s = rng.integers(0, 3, (nq, n)).astype(np.float32)
    if kind == "equal": s[:] = 0.25
    if kind == "nan": s[rng.random((nq, n)) < 0.01] = np.nan; s[rng.random((nq, n)) < 0.01] = -0.0; s[rng.random((nq, n)) < 0.01] = np.inf
    if kind == "ascending": s = np.sort(s, axis=1)
    if kind == "periodic": s = np.sin(np.arange(n)[None, :] * (2 * np.pi / (n // 8192))).astype(np.float32) * np.ones((nq, 1), np.float32)
    got = ops.rank_full(torch.from_numpy(s).cuda(), id_offset=3).cpu().numpy()
    ok = np.array_equal(got, OC.rank_full(s) + 3)
    bad += not ok
    print(n, nq, kind, "ok" if ok else "MISMATCH", flush=True)
    if kind == "gauss" and n >= 300001:      # as column segments
        cuts = [0, n // 3 + 5, n // 3 + 5, 2 * n // 3 + 1, n]
        blocks = [torch.from_numpy(np.ascontiguousarray(s[:, a:b])).cuda() for a, b in zip(cuts[:-1], cuts[1:])]
        ok = np.array_equal(ops.rank_full_segments(blocks, id_offset=3).cpu().numpy(), got)
        bad += not ok
        print("   segments", "ok" if ok else "MISMATCH", flush=True)
print("msd check:", "all ok" if not bad else "%d MISMATCHES" % bad, "MDX_SORT_MSD=" + os.environ.get("MDX_SORT_MSD", ""))
