"""Quick exactness check of rank_full for the library given by MDIR_AMD_LIB (A/B builds)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mdir_amd import ops
from oracle import chain as OC
rng = np.random.default_rng(0)
for n, nq in ((70000, 5), (4993, 70), (8193, 3), (1, 1), (300001, 2)):
    sc = rng.standard_normal((nq, n)).astype(np.float32)
    if n > 100: sc[:, 10:40] = sc[:, 5:6]; sc[0, 50] = np.nan
    got = ops.rank_full(torch.from_numpy(sc).cuda()).cpu().numpy()
    assert (got == OC.rank_full(sc)).all(), (n, nq)
    ids, vals = ops.topk(torch.from_numpy(sc).cuda(), min(n, 100))
    assert (ids.cpu().numpy() == got[:, :min(n, 100)]).all()
print("rank ok", os.environ.get("MDIR_AMD_LIB"))
