import sys, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdir_amd import ops
from oracle import chain as OC
rng = np.random.default_rng(1)
for d in (512, 2048):
    db = rng.standard_normal((1125, d)).astype(np.float32); db /= np.linalg.norm(db, axis=1, keepdims=True)
    # non-negative descriptors too (GeM outputs are): no sign cancellation inside a row
    pos = np.abs(db); pos /= np.linalg.norm(pos, axis=1, keepdims=True)
    for name, m in (("signed", db), ("non-negative", pos.astype(np.float32))):
        ix = ops.DescriptorIndex(torch.from_numpy(m).cuda(), "ND")
        q = torch.from_numpy(m[:200].copy()).cuda()
        chain = OC.scores_chain(np.ascontiguousarray(m.T), np.ascontiguousarray(m[:200].T))
        exact = m[:200].astype(np.float64) @ m.astype(np.float64).T
        for mode in ("split3", "split2"):
            got = ix.scores(q, "ND", compute=mode).cpu().numpy()
            dg = np.abs(np.diag(got[:, :200]) - np.diag(chain[:, :200])).max()
            print(d, name, mode, "max|diff vs chain| all %.3g, on the diagonal (s=1) %.3g; vs float64 %.3g (chain %.3g)" % (
                np.abs(got - chain).max(), dg, np.abs(got - exact).max(), np.abs(chain - exact).max()))
